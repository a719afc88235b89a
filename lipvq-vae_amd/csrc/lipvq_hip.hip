// lipvq_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI of the LipVQ-VAE action tokenizer.
// ABI and reference citations: include/lipvq.h.  Arithmetic contract: lipvq_math.h.
//
// Kernel map
//   lipschitz_scale_kernel   one thread per latent unit (left-to-right row sum = oracle order)
//   mlp3_pack_kernel         nn.Linear weights -> MFMA A-operand order (see "MLP layout")
//   mlp3_kernel<T0,T1>       three Linear layers + activations, one wave per 32 rows, fp32 MFMA
//                            (v_mfma_f32_32x32x2_f32); activations never leave registers
//   nearest_direct_kernel    exact direct-difference distance + first-minimum argmin + gather
//   ste_kernel, mse_*        elementwise / two-pass deterministic reductions
//
// MLP layout.  The layers are evaluated TRANSPOSED, Y^T = W . X^T, with the weights as the MFMA
// A operand (32 output features x 2 k) and the activations as the B operand (2 k x 32 rows).
// The 32x32 result tile then has the batch row on the lane (col = lane & 31) and 16 output
// features in the lane's registers, which is exactly the B-operand shape of the next layer:
// register r of lane-half h feeds k-step r with k = 2r + h.  Choosing the feature <-> tile-row
// map  feature(i) = 2*((i & 3) + 4*(i >> 3)) + ((i >> 2) & 1)  makes that k order the natural
// 0,1,2,... order, so every output is ONE k-ordered fmaf chain starting from the bias -- the
// oracle's definition -- with no LDS round trip and no cross-lane traffic between layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lipvq.h"
#include "lipvq_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "%s: %s", what, hipGetErrorString(e));
    return LIPVQ_OK;
}

extern "C" int lipvq_abi_version(void) { return LIPVQ_ABI_VERSION; }
extern "C" const char* lipvq_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------
// v5:6-12  Lipschitz normalisation
// ------------------------------------------------------------------------------------------
__global__ void lipschitz_scale_kernel(const float* __restrict__ W, const float* __restrict__ ci,
                                       float* __restrict__ scale, float* __restrict__ Wn, int D,
                                       int H) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const float* w = W + (size_t)i * H;
    float s = 0.0f;
    for (int j = 0; j < H; ++j) s = s + lq_abs(w[j]);
    float sc = lq_softplus(ci[i]) / s;
    if (!(sc < 1.0f)) sc = 1.0f;
    if (scale) scale[i] = sc;
    if (Wn)
        for (int j = 0; j < H; ++j) Wn[(size_t)i * H + j] = w[j] * sc;
}

extern "C" int lipvq_lipschitz_scale_f32(const float* W, const float* ci, float* scale, float* Wn,
                                         int D, int H, void* stream) {
    if (!W || !ci || D <= 0 || H <= 0) return fail(LIPVQ_EINVAL, "lipschitz_scale: bad argument");
    hipLaunchKernelGGL(lipschitz_scale_kernel, dim3((D + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       W, ci, scale, Wn, D, H);
    return check_launch("lipschitz_scale");
}

// ------------------------------------------------------------------------------------------
// MLP: packing
// ------------------------------------------------------------------------------------------
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

// P[(t*S + s)*64 + lane] = W[32t + feat(lane & 31)][2s + (lane >> 5)]   (0 outside W)
__global__ void mlp3_pack_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                 float* __restrict__ P, float* __restrict__ B, int K, int J, int S,
                                 int T) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nP = (size_t)T * S * 64;
    if (gid < nP) {
        int lane = (int)(gid & 63);
        size_t ts = gid >> 6;
        int s = (int)(ts % S), t = (int)(ts / S);
        int f = 32 * t + feat_of_tile_row(lane & 31);
        int k = 2 * s + (lane >> 5);
        P[gid] = (f < J && k < K) ? W[(size_t)f * K + k] : 0.0f;
    }
    if (gid < (size_t)T * 32) B[gid] = ((int)gid < J) ? b[gid] : 0.0f;
}

extern "C" size_t lipvq_mlp3_packed_floats(int K0, int J0, int J1, int J2) {
    if (K0 <= 0 || J0 <= 0 || J1 <= 0 || J2 <= 0) return 0;
    return packed_layout(K0, J0, J1, J2).total;
}

extern "C" int lipvq_mlp3_pack_f32(const float* W0, const float* b0, const float* W1,
                                   const float* b1, const float* W2, const float* b2, float* packed,
                                   int K0, int J0, int J1, int J2, void* stream) {
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !packed) return fail(LIPVQ_EINVAL, "mlp3_pack: null pointer");
    if (K0 <= 0 || J2 <= 0 || J0 <= 0 || J1 <= 0 || (J0 & 31) || (J1 & 31) || J0 > 256 || J1 > 256)
        return fail(LIPVQ_EUNSUPPORTED, "mlp3_pack: hidden widths must be multiples of 32 in [32,256] (got %d,%d)", J0, J1);
    PackedLayout L = packed_layout(K0, J0, J1, J2);
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](const float* W, const float* b, size_t oP, size_t oB, int K, int J, int S, int T) {
        size_t n = (size_t)T * S * 64;
        hipLaunchKernelGGL(mlp3_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, b,
                           packed + oP, packed + oB, K, J, S, T);
    };
    launch(W0, b0, L.oP0, L.oB0, K0, J0, L.S0, L.T0);
    launch(W1, b1, L.oP1, L.oB1, J0, J1, L.S1, L.T1);
    launch(W2, b2, L.oP2, L.oB2, J1, J2, L.S2, L.T2);
    return check_launch("mlp3_pack");
}

// ------------------------------------------------------------------------------------------
// MLP: the fused three-layer kernel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case LIPVQ_ACT_GELU: return lq_gelu(v);
        case LIPVQ_ACT_SIGMOID: return lq_sigmoid(v);
        case LIPVQ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        default: return v;
    }
}

struct Mlp3Args {
    const float* x;
    const int64_t* gather_idx;
    const float* packed;
    float* y;
    float* pre0;
    float* pre1;
    float* pre2;
    int64_t N;
    int K0, J0, J1, J2;
    int act0, act1, act2;
};

template <int T0, int T1>
__global__ __launch_bounds__(256) void mlp3_kernel(Mlp3Args a) {
    const PackedLayout L = packed_layout(a.K0, a.J0, a.J1, a.J2);
    const float* __restrict__ P0 = a.packed + L.oP0;
    const float* __restrict__ B0 = a.packed + L.oB0;
    const float* __restrict__ P1 = a.packed + L.oP1;
    const float* __restrict__ B1 = a.packed + L.oB1;
    const float* __restrict__ P2 = a.packed + L.oP2;
    const float* __restrict__ B2 = a.packed + L.oB2;
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.N + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t row = tile * 32 + j;
        const bool valid = row < a.N;
        const int64_t rowc = valid ? row : a.N - 1;
        const float* __restrict__ xr =
            a.gather_idx ? a.x + (size_t)a.gather_idx[rowc] * a.K0 : a.x + (size_t)rowc * a.K0;

        // ---- layer 0: K0 -> 32*T0 ------------------------------------------------------
        f32x16 acc0[T0];
#pragma unroll
        for (int t = 0; t < T0; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[t][r] = B0[32 * t + 2 * r + h];
        for (int s = 0; s < L.S0; ++s) {
            const int k = 2 * s + h;
            const float bv = (k < a.K0) ? xr[k] : 0.0f;
#pragma unroll
            for (int t = 0; t < T0; ++t) {
                const float av = P0[((size_t)t * L.S0 + s) * 64 + lane];
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0[t], 0, 0, 0);
            }
        }
        if (a.pre0 && valid) {
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.pre0[(size_t)row * a.J0 + 32 * t + 2 * r + h] = acc0[t][r];
        }
#pragma unroll
        for (int t = 0; t < T0; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[t][r] = apply_act(acc0[t][r], a.act0);

        // ---- layer 1: 32*T0 -> 32*T1 ---------------------------------------------------
        f32x16 acc1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = B1[32 * t + 2 * r + h];
#pragma unroll
        for (int s = 0; s < 16 * T0; ++s) {
            const float bv = acc0[s / 16][s % 16];
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                const float av = P1[((size_t)t * (16 * T0) + s) * 64 + lane];
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc1[t], 0, 0, 0);
            }
        }
        if (a.pre1 && valid) {
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.pre1[(size_t)row * a.J1 + 32 * t + 2 * r + h] = acc1[t][r];
        }
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = apply_act(acc1[t][r], a.act1);

        // ---- layer 2: 32*T1 -> J2, one 32-feature output tile at a time -----------------
        for (int t2 = 0; t2 < L.T2; ++t2) {
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = B2[32 * t2 + 2 * r + h];
            const float* __restrict__ P2t = P2 + (size_t)t2 * (16 * T1) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 16 * T1; ++s) {
                const float bv = acc1[s / 16][s % 16];
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(P2t[(size_t)s * 64], bv, acc2, 0, 0, 0);
            }
            if (valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * t2 + 2 * r + h;
                    if (f < a.J2) {
                        if (a.pre2) a.pre2[(size_t)row * a.J2 + f] = acc2[r];
                        a.y[(size_t)row * a.J2 + f] = apply_act(acc2[r], a.act2);
                    }
                }
            }
        }
    }
}

typedef void (*mlp3_fn)(Mlp3Args);

static mlp3_fn mlp3_select(int T0, int T1) {
#define LQ_CASE(a_, b_) if (T0 == a_ && T1 == b_) return mlp3_kernel<a_, b_>;
    LQ_CASE(2, 4) LQ_CASE(4, 2)            // the reference's stacks: 64->128 and 128->64
    LQ_CASE(2, 1) LQ_CASE(2, 2) LQ_CASE(2, 3)   // other hidden_dim values (32, 64, 96)
    LQ_CASE(1, 1) LQ_CASE(1, 2)
#undef LQ_CASE
    return nullptr;
}

extern "C" int lipvq_mlp3_f32(const float* x, const int64_t* gather_idx, const float* packed, float* y,
                              float* pre0, float* pre1, float* pre2, int64_t N, int K0, int J0, int J1,
                              int J2, int act0, int act1, int act2, void* stream) {
    if (N < 0) return fail(LIPVQ_EINVAL, "mlp3: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!x || !packed || !y) return fail(LIPVQ_EINVAL, "mlp3: null pointer");
    if (K0 <= 0 || J2 <= 0 || (J0 & 31) || (J1 & 31) || J0 <= 0 || J1 <= 0)
        return fail(LIPVQ_EUNSUPPORTED, "mlp3: hidden widths must be positive multiples of 32 (got %d,%d)", J0, J1);
    mlp3_fn fn = mlp3_select(J0 / 32, J1 / 32);
    if (!fn) return fail(LIPVQ_EUNSUPPORTED, "mlp3: no kernel instance for hidden widths %d,%d", J0, J1);
    Mlp3Args a{x, gather_idx, packed, y, pre0, pre1, pre2, N, K0, J0, J1, J2, act0, act1, act2};
    int64_t ntiles = (N + 31) / 32;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;        // grid-stride beyond 8 blocks per CU
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mlp3");
}

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
    float* __restrict__ zq, unsigned long long* __restrict__ usage, float* __restrict__ best_out,
    const int* __restrict__ row_list, const int* __restrict__ row_count, int64_t N, int K, int KT) {
    constexpr int D = DCH * 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int64_t nrows = row_list ? (int64_t)(*row_count) : N;
    int64_t base = (int64_t)blockIdx.x * blockDim.x;
    if (base >= nrows) return;
    int64_t slot = base + threadIdx.x;
    const bool valid = slot < nrows;
    int64_t row = valid ? (row_list ? (int64_t)row_list[slot] : slot) : (row_list ? (int64_t)row_list[nrows - 1] : nrows - 1);

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
    if (!valid) return;
    idx[row] = (int64_t)best_k;
    if (best_out) best_out[row] = best_v;
    if (usage) atomicAdd(&usage[best_k], 1ull);
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
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const float* zr = z + (size_t)row * D;
    float best_v = INFINITY;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
        const float* c = cb + (size_t)k * D;
        float v = (dist == LIPVQ_DIST_NORM) ? lq_sqrt(lq_sqdist8(zr, c, D)) : lq_sqdist32(zr, c, D);
        if (v < best_v) { best_v = v; best_k = k; }
    }
    idx[row] = (int64_t)best_k;
    if (best_out) best_out[row] = best_v;
    if (usage) atomicAdd(&usage[best_k], 1ull);
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
                           z, cb, idx, zq, (unsigned long long*)usage, best, nullptr, nullptr, N, K, KT);
    else
        hipLaunchKernelGGL((nearest_direct_kernel<DCH, LIPVQ_DIST_SQSUM>), dim3(blocks), dim3(256), lds, st,
                           z, cb, idx, zq, (unsigned long long*)usage, best, nullptr, nullptr, N, K, KT);
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

// ------------------------------------------------------------------------------------------
// vq:74 straight-through value
// ------------------------------------------------------------------------------------------
__global__ void ste_kernel(const float* __restrict__ ze, const float* __restrict__ zq,
                           float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = ze[i] + (zq[i] - ze[i]);
}

extern "C" int lipvq_ste_f32(const float* ze, const float* zq, float* out, int64_t n_elem, void* stream) {
    if (n_elem < 0) return fail(LIPVQ_EINVAL, "ste: n < 0");
    if (n_elem == 0) return LIPVQ_OK;
    if (!ze || !zq || !out) return fail(LIPVQ_EINVAL, "ste: null pointer");
    int64_t blocks = (n_elem + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ste_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ze, zq, out, n_elem);
    return check_launch("ste");
}

// ------------------------------------------------------------------------------------------
// F.mse_loss pair: deterministic two-pass reduction in double
// ------------------------------------------------------------------------------------------
#define MSE_BLOCKS 512

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)a[i] - (double)b[i];
        acc += d * d;
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int64_t nx, int64_t nz,
                                                        float* __restrict__ out2) {
    __shared__ double sh[4];
    for (int which = 0; which < 2; ++which) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < MSE_BLOCKS; i += blockDim.x) acc += partial[which * MSE_BLOCKS + i];
        const double t = block_sum(acc, sh);
        if (threadIdx.x == 0) out2[which] = (float)(t / (double)(which == 0 ? nx : nz));
    }
}

extern "C" size_t lipvq_mse_workspace_bytes(void) { return 2 * MSE_BLOCKS * sizeof(double); }

extern "C" int lipvq_mse_pair_f32(const float* xr, const float* x, int64_t nx, const float* zq,
                                  const float* ze, int64_t nz, float* out2, void* workspace, void* stream) {
    if (!xr || !x || !zq || !ze || !out2 || !workspace || nx <= 0 || nz <= 0)
        return fail(LIPVQ_EINVAL, "mse_pair: bad argument");
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, st, xr, x, nx, part);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, st, zq, ze, nz, part + MSE_BLOCKS);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, part, nx, nz, out2);
    return check_launch("mse_pair");
}
