// lipvq_nearest.hip -- nearest-code search (distance + first-minimum argmin + gather)
// ABI and reference citations: include/lipvq.h.  Arithmetic contract: lipvq_math.h.
#include "lipvq_common.h"

// ------------------------------------------------------------------------------------------
// nearest code: exact direct-difference distance, first-minimum argmin, gather
// ------------------------------------------------------------------------------------------
// One lane owns one latent row (its D floats live in registers); the workgroup streams the
// codebook through LDS in tiles and every lane reads the SAME code element (LDS broadcast).
// The distance is accumulated in the oracle's order (lq_sqdist8 / lq_sqdist32), so the result
// is bit-identical to torch's CPU kernels for D % 8 == 0.
template <int DCH, int DIST>
__global__ __launch_bounds__(256) void nearest_direct_kernel(
    const float* __restrict__ z, const float* __restrict__ cb, int64_t* __restrict__ idx,
    float* __restrict__ zq, unsigned long long* __restrict__ usage, float* __restrict__ best_out, int64_t N, int K,
    int KT) {
    constexpr int D = DCH * 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = slot < N;
    const int64_t row = valid ? slot : N - 1;

    float zr[D];
    {
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            float4 v = z4[i];
            zr[4 * i + 0] = v.x; zr[4 * i + 1] = v.y; zr[4 * i + 2] = v.z; zr[4 * i + 3] = v.w;
        }
    }
    float best_v = INFINITY;   // compared value (sqrt for DIST_NORM)
    float best_s = INFINITY;   // its square (DIST_NORM) -- a cheap necessary test before the sqrt
    int best_k = 0;

    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kt = (K - k0 < KT) ? (K - k0) : KT;
        __syncthreads();
        {
            const float4* src = reinterpret_cast<const float4*>(cb + (size_t)k0 * D);
            float4* dst = reinterpret_cast<float4*>(lds);
            const int n4 = kt * (D / 4);
            for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        for (int kk = 0; kk < kt; ++kk) {
            const float4* c4 = reinterpret_cast<const float4*>(lds + (size_t)kk * D);
            float s;
            if (DIST == LIPVQ_DIST_NORM) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
#pragma unroll
                for (int i = 0; i < DCH; ++i) {
                    const float4 lo = c4[2 * i], hi = c4[2 * i + 1];
                    const float d0 = zr[8 * i + 0] - lo.x, d1 = zr[8 * i + 1] - lo.y;
                    const float d2 = zr[8 * i + 2] - lo.z, d3 = zr[8 * i + 3] - lo.w;
                    const float d4 = zr[8 * i + 4] - hi.x, d5 = zr[8 * i + 5] - hi.y;
                    const float d6 = zr[8 * i + 6] - hi.z, d7 = zr[8 * i + 7] - hi.w;
                    a0 = lq_fma(d0, d0, a0); a1 = lq_fma(d1, d1, a1);
                    a2 = lq_fma(d2, d2, a2); a3 = lq_fma(d3, d3, a3);
                    a4 = lq_fma(d4, d4, a4); a5 = lq_fma(d5, d5, a5);
                    a6 = lq_fma(d6, d6, a6); a7 = lq_fma(d7, d7, a7);
                }
                s = ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7;
                if (s < best_s) {
                    const float v = lq_sqrt(s);
                    if (v < best_v) { best_v = v; best_s = s; best_k = k0 + kk; }
                }
            } else {
                float acc[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int l = 0; l < 8; ++l) acc[q][l] = 0.f;
#pragma unroll
                for (int i = 0; i < DCH; ++i) {
                    // chunks 0..(DCH/4*4 - 1) cycle through the 4 accumulators; left-overs go to accumulator 0
                    const int q = (i < (DCH / 4) * 4) ? (i & 3) : 0;
                    const float4 lo = c4[2 * i], hi = c4[2 * i + 1];
                    const float d0 = zr[8 * i + 0] - lo.x, d1 = zr[8 * i + 1] - lo.y;
                    const float d2 = zr[8 * i + 2] - lo.z, d3 = zr[8 * i + 3] - lo.w;
                    const float d4 = zr[8 * i + 4] - hi.x, d5 = zr[8 * i + 5] - hi.y;
                    const float d6 = zr[8 * i + 6] - hi.z, d7 = zr[8 * i + 7] - hi.w;
                    acc[q][0] = acc[q][0] + d0 * d0; acc[q][1] = acc[q][1] + d1 * d1;
                    acc[q][2] = acc[q][2] + d2 * d2; acc[q][3] = acc[q][3] + d3 * d3;
                    acc[q][4] = acc[q][4] + d4 * d4; acc[q][5] = acc[q][5] + d5 * d5;
                    acc[q][6] = acc[q][6] + d6 * d6; acc[q][7] = acc[q][7] + d7 * d7;
                }
                s = 0.f;
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const float v = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
                    s = (l == 0) ? v : s + v;
                }
                if (s < best_v) { best_v = s; best_k = k0 + kk; }
            }
        }
    }
    // (per-row atomics on one address serialise chip-wide when the codes collapse -- the reference's default initialisation
    // does that: 6.5 ms for a 524 288-row VQVAE batch; lq_usage_add combines a wave's duplicates first)
    if (usage) lq_usage_add(usage, best_k, valid);
    if (!valid) return;
    idx[row] = (int64_t)best_k;
    if (best_out) best_out[row] = best_v;
    if (zq) {
        const float4* src = reinterpret_cast<const float4*>(cb + (size_t)best_k * D);
        float4* dst = reinterpret_cast<float4*>(zq + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) dst[i] = src[i];
    }
}

// Any D (including D % 8 != 0): one lane per row, operands straight from global memory.
__global__ void nearest_generic_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                       int64_t* __restrict__ idx, float* __restrict__ zq,
                                       unsigned long long* __restrict__ usage,
                                       float* __restrict__ best_out, int64_t N, int K, int D, int dist) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = slot < N;
    const int64_t row = valid ? slot : N - 1;
    const float* zr = z + (size_t)row * D;
    float best_v = INFINITY;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
        const float* c = cb + (size_t)k * D;
        float v = (dist == LIPVQ_DIST_NORM) ? lq_sqrt(lq_sqdist8(zr, c, D)) : lq_sqdist32(zr, c, D);
        if (v < best_v) { best_v = v; best_k = k; }
    }
    if (usage) lq_usage_add(usage, best_k, valid);
    if (!valid) return;
    idx[row] = (int64_t)best_k;
    if (best_out) best_out[row] = best_v;
    if (zq)
        for (int d = 0; d < D; ++d) zq[(size_t)row * D + d] = cb[(size_t)best_k * D + d];
}

template <int DCH>
static int launch_nearest_direct(const float* z, const float* cb, int64_t* idx, float* zq,
                                 int64_t* usage, float* best, int64_t N, int K, int dist,
                                 hipStream_t st) {
    constexpr int D = DCH * 8;
    int KT = 8192 / D;                     // 32 KiB of LDS per codebook tile
    if (KT > K) KT = K;
    size_t lds = (size_t)KT * D * sizeof(float);
    unsigned blocks = (unsigned)((N + 255) / 256);
    if (dist == LIPVQ_DIST_NORM)
        hipLaunchKernelGGL((nearest_direct_kernel<DCH, LIPVQ_DIST_NORM>), dim3(blocks), dim3(256), lds, st,
                           z, cb, idx, zq, (unsigned long long*)usage, best, N, K, KT);
    else
        hipLaunchKernelGGL((nearest_direct_kernel<DCH, LIPVQ_DIST_SQSUM>), dim3(blocks), dim3(256), lds, st,
                           z, cb, idx, zq, (unsigned long long*)usage, best, N, K, KT);
    return check_launch("nearest_direct");
}

extern "C" int lipvq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq,
                                 int64_t* usage, float* best, int64_t N, int K, int D, int dist,
                                 void* stream) {
    if (N < 0 || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "nearest: bad sizes N=%lld K=%d D=%d", (long long)N, K, D);
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !idx) return fail(LIPVQ_EINVAL, "nearest: null pointer");
    if (dist != LIPVQ_DIST_NORM && dist != LIPVQ_DIST_SQSUM) return fail(LIPVQ_EINVAL, "nearest: unknown distance rule %d", dist);
    if (N > 2147483647LL * 64) return fail(LIPVQ_EUNSUPPORTED, "nearest: N too large");
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (((uintptr_t)z | (uintptr_t)codebook | (uintptr_t)zq) & 15) == 0;
    if (aligned) {
        switch (D) {
            case 32: return launch_nearest_direct<4>(z, codebook, idx, zq, usage, best, N, K, dist, st);
            case 64: return launch_nearest_direct<8>(z, codebook, idx, zq, usage, best, N, K, dist, st);
            case 128: return launch_nearest_direct<16>(z, codebook, idx, zq, usage, best, N, K, dist, st);
            case 208: return launch_nearest_direct<26>(z, codebook, idx, zq, usage, best, N, K, dist, st);
            default: break;
        }
    }
    hipLaunchKernelGGL(nearest_generic_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, z, codebook,
                       idx, zq, (unsigned long long*)usage, best, N, K, D, dist);
    return check_launch("nearest_generic");
}

