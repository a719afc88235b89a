// lipvq_mlp.h -- MFMA A-operand packing of a three-layer stack (shared by lipvq_mlp.hip and
// lipvq_fused.hip).  Layout notes: lipvq_mlp.hip.
#ifndef LIPVQ_MLP_H_
#define LIPVQ_MLP_H_
#include "lipvq_common.h"

__host__ __device__ static inline int feat_of_tile_row(int i) {
    return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1);
}

struct PackedLayout {
    int S0, S1, S2;      // k-steps per layer (k pairs)
    int T0, T1, T2;      // 32-feature output tiles per layer
    size_t oP0, oB0, oP1, oB1, oP2, oB2, total;   // offsets in floats
};

__host__ __device__ static inline PackedLayout packed_layout(int K0, int J0, int J1, int J2) {
    PackedLayout L;
    L.T0 = (J0 + 31) / 32; L.T1 = (J1 + 31) / 32; L.T2 = (J2 + 31) / 32;
    L.S0 = (K0 + 1) / 2; L.S1 = 16 * L.T0; L.S2 = 16 * L.T1;
    size_t o = 0;
    L.oP0 = o; o += (size_t)L.T0 * L.S0 * 64;
    L.oB0 = o; o += (size_t)L.T0 * 32;
    L.oP1 = o; o += (size_t)L.T1 * L.S1 * 64;
    L.oB1 = o; o += (size_t)L.T1 * 32;
    L.oP2 = o; o += (size_t)L.T2 * L.S2 * 64;
    L.oB2 = o; o += (size_t)L.T2 * 32;
    L.total = o;
    return L;
}

#if defined(__HIPCC__)
// A 32 x 32 accumulator tile of the transposed layers (lane = (batch row n, half h), register r = feature 32 t + 2 r + h) to and
// from row-major [rows][ld] memory with 16-byte accesses.  One v_permlane32_swap per register pair (low half's r >= 8 <-> high
// half's r < 8) leaves the low lane with features 32t .. 32t+15 and the high lane with 32t+16 .. 32t+31: four float4 per lane,
// 64 contiguous bytes, instead of sixteen 4-byte accesses scattered over the row (what the first mlp3 kernels did for every
// saved pre-activation and every gradient: 160 scattered wave-stores per 32-row tile).  `vec`: the caller checked that base is
// 16-byte aligned and ld a multiple of 4.  Features >= J are neither read nor written.
__device__ __forceinline__ void lq_tile_store16(float* __restrict__ base, int ld, int64_t row, bool valid, int t, int h,
                                                const f32x16& v, int J, bool vec) {
    const int f0 = 32 * t + 16 * h;
    if (vec && 32 * t + 32 <= J) {                      // (wave-uniform) a whole tile inside the row
        float lo8[8], hi8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[8 + j]), false, false);
            lo8[j] = __uint_as_float(sw[0]);
            hi8[j] = __uint_as_float(sw[1]);
        }
        if (valid) {
            float4* dst = reinterpret_cast<float4*>(base + (size_t)row * ld + f0);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = make_float4(lo8[2 * q], hi8[2 * q], lo8[2 * q + 1], hi8[2 * q + 1]);
        }
        return;
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + 2 * r + h;
            if (f < J) base[(size_t)row * ld + f] = v[r];
        }
    }
}

__device__ __forceinline__ f32x16 lq_tile_load16(const float* __restrict__ base, int ld, int64_t rowc, int t, int h, int J, bool vec) {
    f32x16 out;
    if (vec && 32 * t + 32 <= J) {
        const float4* src = reinterpret_cast<const float4*>(base + (size_t)rowc * ld + 32 * t + 16 * h);
        float L[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 x = src[q]; L[4 * q] = x.x; L[4 * q + 1] = x.y; L[4 * q + 2] = x.z; L[4 * q + 3] = x.w; }
        // lane (n, h) now holds features 32t + 16h + [0, 16): swap(even j, odd j) gives register j = feature 2j + h, 8 + j = 16 + 2j + h
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(L[2 * j]), __float_as_uint(L[2 * j + 1]), false, false);
            out[j] = __uint_as_float(sw[0]);
            out[8 + j] = __uint_as_float(sw[1]);
        }
        return out;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 32 * t + 2 * r + h;
        out[r] = (f < J) ? base[(size_t)rowc * ld + f] : 0.0f;
    }
    return out;
}
#endif

#endif
