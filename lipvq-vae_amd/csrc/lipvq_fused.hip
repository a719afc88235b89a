// lipvq_fused.hip -- the BASELINE metric's path in ONE persistent launch:
//   x --encoder MLP + Lipschitz layer (fp32 MFMA, weights resident in LDS)--> z_e (registers only)
//     --centre, split to fp16 hi/lo (registers)--> MFMA screen against the prepared codebook (LDS-staged)
//     --certified rows: idx, z_q gather, usage;  uncertified rows: z_e row -> compact list for the exact kernel.
// Replaces reference backbone_lfqvae_v5.py:71-74 (encoder, to_latent, quantizer, z_latent) with the SAME
// results as lipvq_mlp3_f32 + lipvq_nearest_f32: phase A is mlp3_kernel's arithmetic (one k-ordered fmaf
// chain per output, lipvq_mlp.hip), phase B is screen_kernel's (lipvq_screen.hip).  z_e never touches HBM
// unless the caller asks for it (training) or a row is uncertified (<1 % of rows at BASELINE config 2).
//
// Why fused: the two stand-alone kernels are latency bound (rocprofv3 PMC, profiles/r01_b: MFMA pipe 13-18 %
// busy, half of all wave cycles parked in s_waitcnt/barriers): mlp3 streams its A operands from L2 at one
// wave per SIMD, and both pay 268 MB of z_e traffic.  Here the weights sit in LDS (one 16-byte read feeds
// four MFMAs), two waves share each SIMD so one wave's GELU/bookkeeping VALU work runs beside the other's
// MFMAs, and a workgroup persists over row blocks so the weights are loaded once.
#include "lipvq_mlp.h"
#include "lipvq_screen.h"

#define FUSED_WAVES 8
#define FUSED_THREADS (FUSED_WAVES * 64)

struct TokArgs {
    const float* x;              // [N][A]
    const float* packed;         // lipvq_mlp3_pack_f32 of (A -> 64 -> 128 -> D)
    const unsigned char* prep;   // lipvq_nearest_prepare_f32 of the codebook
    const float* cb;             // [K][D]
    int64_t* idx;                // [N]
    float* zq;                   // [N][D] or NULL
    unsigned long long* usage;   // [K] or NULL
    float* ze_out;               // [N][D] or NULL
    int* amb_count;              // workspace[0]
    int* amb_list;               // [N]
    float* amb_z;                // [N][D] compact z_e rows of uncertified rows
    int64_t N;
    int A, D, K;
    float gamma;
};

// T0 = 2 (64 features), T1 = 4 (128 features): the reference's encoder widths (v5:54-59).
template <int S>
__global__ __launch_bounds__(FUSED_THREADS) void tokenize_kernel(TokArgs a) {
    constexpr int T0 = 2, T1 = 4, T2 = S / 2;
    constexpr int S1 = 16 * T0, S2 = 16 * T1;               // k-steps (pairs) of layers 1 and 2
    constexpr int TILE_BYTES = S * 2048 + 128;
    constexpr int TC = (S <= 2) ? 8 : (S <= 4) ? 4 : 1;     // column tiles per LDS stage
    constexpr int STAGE_BYTES = TC * TILE_BYTES;
    constexpr int STAGE_VEC = STAGE_BYTES / 16;
    constexpr int VPT = (STAGE_VEC + FUSED_THREADS - 1) / FUSED_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const PackedLayout PL = packed_layout(a.A, 32 * T0, 32 * T1, 16 * S);
    const PrepLayout L = prep_layout(a.K, a.D);
    const int S0 = PL.S0;
    const int S0q = (S0 + 3) / 4;
    // LDS carve (floats unless noted)
    float* w_P0 = reinterpret_cast<float*>(lds);                       // [T0][S0q][64][4]
    float* w_B0 = w_P0 + T0 * S0q * 256;                               // [32*T0]
    float* w_P1 = w_B0 + 32 * T0;                                      // [T1][S1/4][64][4]
    float* w_B1 = w_P1 + T1 * (S1 / 4) * 256;
    float* w_P2 = w_B1 + 32 * T1;                                      // [T2][S2/4][64][4]
    float* w_B2 = w_P2 + T2 * (S2 / 4) * 256;
    float* w_mu = w_B2 + 32 * T2;                                      // [16*S]
    float* scr_all = w_mu + 16 * S;                                    // [FUSED_WAVES][96]
    unsigned char* stage0 = reinterpret_cast<unsigned char*>(scr_all + FUSED_WAVES * 96);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, h = lane >> 5;

    // ---- once per workgroup: weights (re-laid out 4 k-steps per 16-byte LDS read), biases, mu --------
    {
        const float* P0 = a.packed + PL.oP0;
        for (int i = tid; i < T0 * S0q * 256; i += FUSED_THREADS) {
            const int q = i & 3, l = (i >> 2) & 63, sq = (i >> 8) % S0q, t = (i >> 8) / S0q;
            const int s = 4 * sq + q;
            w_P0[i] = (s < S0) ? P0[((size_t)t * S0 + s) * 64 + l] : 0.0f;
        }
        const float* P1 = a.packed + PL.oP1;
        for (int i = tid; i < T1 * (S1 / 4) * 256; i += FUSED_THREADS) {
            const int q = i & 3, l = (i >> 2) & 63, sq = (i >> 8) % (S1 / 4), t = (i >> 8) / (S1 / 4);
            w_P1[i] = P1[((size_t)t * S1 + 4 * sq + q) * 64 + l];
        }
        const float* P2 = a.packed + PL.oP2;
        for (int i = tid; i < T2 * (S2 / 4) * 256; i += FUSED_THREADS) {
            const int q = i & 3, l = (i >> 2) & 63, sq = (i >> 8) % (S2 / 4), t = (i >> 8) / (S2 / 4);
            w_P2[i] = P2[((size_t)t * S2 + 4 * sq + q) * 64 + l];
        }
        for (int i = tid; i < 32 * T0; i += FUSED_THREADS) w_B0[i] = a.packed[PL.oB0 + i];
        for (int i = tid; i < 32 * T1; i += FUSED_THREADS) w_B1[i] = a.packed[PL.oB1 + i];
        for (int i = tid; i < 32 * T2; i += FUSED_THREADS) w_B2[i] = a.packed[PL.oB2 + i];
        const float* mu = reinterpret_cast<const float*>(a.prep + L.o_mu);
        for (int i = tid; i < 16 * S; i += FUSED_THREADS) w_mu[i] = mu[i];
    }
    __syncthreads();

    const unsigned* hdr = reinterpret_cast<const unsigned*>(a.prep);
    const unsigned char* tiles = a.prep + L.o_tiles;
    const int nstage = (L.ntiles + TC - 1) / TC;
    const int64_t nblk = (a.N + FUSED_WAVES * 32 - 1) / (FUSED_WAVES * 32);

    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = (blk * FUSED_WAVES + wave) * 32;
        const int64_t row = row0 + ln;
        const int64_t rowc = row < a.N ? row : a.N - 1;

        // ================= phase A: encoder + Lipschitz layer, fp32 MFMA ====================
        f32x16 h0[T0];
        {
            const float* xr = a.x + (size_t)rowc * a.A;
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h0[t][r] = w_B0[32 * t + 2 * r + h];
            for (int sq = 0; sq < S0q; ++sq) {
                float4 av[T0];
#pragma unroll
                for (int t = 0; t < T0; ++t) av[t] = *reinterpret_cast<const float4*>(w_P0 + ((t * S0q + sq) * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 2 * (4 * sq + q) + h;
                    const float bv = (k < a.A) ? xr[k] : 0.0f;
#pragma unroll
                    for (int t = 0; t < T0; ++t) {
                        const float aq = q == 0 ? av[t].x : q == 1 ? av[t].y : q == 2 ? av[t].z : av[t].w;
                        // padded k-steps (s >= S0) multiply zeros: acc + 0*0 = acc (oracle pads odd fan-in the same way)
                        h0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, bv, h0[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h0[t][r] = lq_gelu(h0[t][r]);
        }
        f32x16 h1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = w_B1[32 * t + 2 * r + h];
#pragma unroll
            for (int sq = 0; sq < S1 / 4; ++sq) {
                const float4 av = *reinterpret_cast<const float4*>(w_P1 + ((t * (S1 / 4) + sq) * 64 + lane) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, h0[(4 * sq + 0) / 16][(4 * sq + 0) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, h0[(4 * sq + 1) / 16][(4 * sq + 1) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, h0[(4 * sq + 2) / 16][(4 * sq + 2) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, h0[(4 * sq + 3) / 16][(4 * sq + 3) % 16], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[t][r] = lq_gelu(acc[r]);
        }
        f32x16 zf[T2];                   // z_e in fp32 (kept for uncertified rows / ze_out)
        f16x8 ah[S], al[S];
        float n2 = 0.0f;
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = w_B2[32 * t + 2 * r + h];
#pragma unroll
            for (int sq = 0; sq < S2 / 4; ++sq) {
                const float4 av = *reinterpret_cast<const float4*>(w_P2 + ((t * (S2 / 4) + sq) * 64 + lane) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, h1[(4 * sq + 0) / 16][(4 * sq + 0) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, h1[(4 * sq + 1) / 16][(4 * sq + 1) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, h1[(4 * sq + 2) / 16][(4 * sq + 2) % 16], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, h1[(4 * sq + 3) / 16][(4 * sq + 3) % 16], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float zv = lq_sigmoid(acc[r]);
                zf[t][r] = zv;
                // register r of tile t is feature 32t + 2r + h = screen slot (step 2t + (r >> 3), element r & 7)
                const float v = zv - w_mu[32 * t + 2 * r + h];
                const _Float16 vh = (_Float16)v;
                ah[2 * t + (r >> 3)][r & 7] = vh;
                al[2 * t + (r >> 3)][r & 7] = (_Float16)(v - (float)vh);
                n2 = lq_fma(v, v, n2);
            }
        }
        n2 += __shfl_xor(n2, 32, 64);
        if (a.ze_out && row < a.N) {
#pragma unroll
            for (int t = 0; t < T2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.ze_out[(size_t)row * a.D + 32 * t + 2 * r + h] = zf[t][r];
        }

        // ================= phase B: MFMA screen (same arithmetic as screen_kernel) ============
        float m1[16], m2[16];
        int k1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { m1[r] = INFINITY; m2[r] = INFINITY; k1[r] = 0; }
        uint4 pre[VPT];
        auto stage_load = [&](int st) {
            const uint4* src = reinterpret_cast<const uint4*>(tiles + (size_t)st * STAGE_BYTES);
            const size_t avail = ((size_t)L.ntiles * TILE_BYTES - (size_t)st * STAGE_BYTES) / 16;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int i = tid + v * FUSED_THREADS;
                pre[v] = (i < STAGE_VEC && (size_t)i < avail) ? src[i] : make_uint4(0, 0, 0, 0);
            }
        };
        auto stage_store = [&](int buf) {
            uint4* dst = reinterpret_cast<uint4*>(stage0 + (size_t)buf * STAGE_BYTES);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int i = tid + v * FUSED_THREADS;
                if (i < STAGE_VEC) dst[i] = pre[v];
            }
        };
        stage_load(0);
        __syncthreads();                 // previous block's readers of stage buffers / scratch are done
        stage_store(0);
        __syncthreads();
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage) stage_load(st + 1);
            const unsigned char* sb = stage0 + (size_t)(st & 1) * STAGE_BYTES;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                const int ct = st * TC + c;
                if (ct < L.ntiles) {
                    const unsigned char* tb = sb + (size_t)c * TILE_BYTES;
                    const float e2 = reinterpret_cast<const float*>(tb + S * 2048)[ln];
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = e2;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const f16x8 bh = *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 0) * 64 + lane) * 16);
                        const f16x8 bl = *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 1) * 64 + lane) * 16);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc, 0, 0, 0);
                    }
                    const int code = ct * 32 + ln;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r];
                        k1[r] = (v < m1[r]) ? code : k1[r];
                        m2[r] = __builtin_amdgcn_fmed3f(v, m1[r], m2[r]);
                        m1[r] = fminf(v, m1[r]);
                    }
                }
            }
            if (st + 1 < nstage) stage_store((st + 1) & 1);
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float om1 = __shfl_xor(m1[r], o, 64);
                const float om2 = __shfl_xor(m2[r], o, 64);
                const int ok1 = __shfl_xor(k1[r], o, 64);
                const float hi = fmaxf(m1[r], om1);
                m2[r] = fminf(fminf(m2[r], om2), hi);
                const bool take = (om1 < m1[r]) || (om1 == m1[r] && ok1 < k1[r]);
                k1[r] = take ? ok1 : k1[r];
                m1[r] = fminf(m1[r], om1);
            }
        }
        float* scr = scr_all + wave * 96;
        if (ln == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                scr[i] = m1[r];
                scr[32 + i] = m2[r];
                reinterpret_cast<int*>(scr)[64 + i] = k1[r];
            }
        }
        __syncthreads();
        int my_k = 0, my_slot = -1;
        bool certified = false;
        if (h == 0) {
            const float av = scr[ln], bv = scr[32 + ln];
            my_k = reinterpret_cast<const int*>(scr)[64 + ln];
            const float E2max = __uint_as_float(hdr[0]);
            const float Emax = lq_sqrt(__uint_as_float(hdr[1]));
            const float twoemax = __uint_as_float(hdr[2]);
            const float cross = 2.0f * lq_sqrt(n2) * Emax;
            const float eps = a.gamma * (E2max + cross) + 9.5367431640625e-07f * (n2 + E2max + cross);
            certified = (twoemax < 60000.0f) && (bv - av > 2.0f * eps) && (my_k < a.K);
            if (row < a.N) {
                if (certified) {
                    a.idx[row] = (int64_t)my_k;
                    if (a.usage) atomicAdd(&a.usage[my_k], 1ull);
                } else {
                    my_slot = atomicAdd(a.amb_count, 1);
                    a.amb_list[my_slot] = (int)row;
                }
            }
        }
        // uncertified rows: hand the exact fp32 z_e row to the exact kernel (both lane halves hold half the row)
        {
            const int slot = __shfl(my_slot, ln, 64);
            if (slot >= 0) {
#pragma unroll
                for (int t = 0; t < T2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a.amb_z[(size_t)slot * a.D + 32 * t + 2 * r + h] = zf[t][r];
            }
        }
        if (a.zq) {
            const int nvec = a.D / 4;
            for (int rr = 0; rr < 32; rr += 4) {
                const int src_lane = rr + (lane >> 4);
                const int kk = __shfl(my_k, src_lane, 64);
                const bool ok = __shfl((int)certified, src_lane, 64) != 0;
                const int64_t orow = row0 + src_lane;
                if (ok && orow < a.N) {
                    const float4* src = reinterpret_cast<const float4*>(a.cb + (size_t)kk * a.D);
                    float4* dst = reinterpret_cast<float4*>(a.zq + (size_t)orow * a.D);
                    for (int v = lane & 15; v < nvec; v += 16) dst[v] = src[v];
                }
            }
        }
    }
}

template <int S>
static size_t fused_lds_bytes(int A) {
    constexpr int T0 = 2, T1 = 4, T2 = S / 2;
    constexpr int TILE_BYTES = S * 2048 + 128;
    constexpr int TC = (S <= 2) ? 8 : (S <= 4) ? 4 : 1;
    const int S0q = ((A + 1) / 2 + 3) / 4;
    size_t fl = (size_t)T0 * S0q * 256 + 32 * T0 + (size_t)T1 * 8 * 256 + 32 * T1 + (size_t)T2 * 16 * 256 + 32 * T2 + 16 * S +
                FUSED_WAVES * 96;
    return fl * sizeof(float) + 2 * (size_t)TC * TILE_BYTES;
}

template <int S>
static int launch_tokenize(const TokArgs& a, hipStream_t st) {
    const size_t lds = fused_lds_bytes<S>(a.A);
    if (lds > 160 * 1024) return fail(LIPVQ_EUNSUPPORTED, "tokenize: %zu B of LDS needed", lds);
    auto kfn = tokenize_kernel<S>;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "tokenize: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    int64_t nblk = (a.N + FUSED_WAVES * 32 - 1) / (FUSED_WAVES * 32);
    int64_t blocks = nblk < 256 ? nblk : 256;            // one persistent workgroup per CU
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(FUSED_THREADS), lds, st, a);
    return check_launch("tokenize");
}

extern "C" int lipvq_tokenize_supported(int A, int J0, int J1, int D, int K) {
    return (A > 0 && A <= 64 && J0 == 64 && J1 == 128 && K > 0 && (D == 32 || D == 64 || D == 128)) ? 1 : 0;
}

extern "C" size_t lipvq_tokenize_workspace_bytes(int64_t N, int D) {
    if (N <= 0 || D <= 0) return 0;
    return 64 + (((sizeof(int) * (size_t)N) + 63) & ~(size_t)63) + sizeof(float) * (size_t)N * D;
}

// Fused encode + quantize (reference v5:71-74).  packed: lipvq_mlp3_pack_f32 of the encoder stack
// (A -> 64 -> 128 -> D with the Lipschitz-normalised W2, activations gelu, gelu, sigmoid); prep:
// lipvq_nearest_prepare_f32 of the codebook; workspace: lipvq_tokenize_workspace_bytes(N, D).
// Outputs exactly as lipvq_mlp3_f32 + lipvq_nearest_f32(LIPVQ_DIST_NORM): idx, zq (may be NULL),
// usage (may be NULL, accumulated), ze_out (may be NULL).  workspace[0] (int) = rows decided by the exact kernel.
extern "C" int lipvq_tokenize_f32(const float* x, const float* packed, const float* codebook, const void* prep,
                                  int64_t* idx, float* zq, int64_t* usage, float* ze_out, void* workspace, int64_t N,
                                  int A, int J0, int J1, int D, int K, void* stream) {
    if (N < 0) return fail(LIPVQ_EINVAL, "tokenize: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!x || !packed || !codebook || !prep || !idx || !workspace) return fail(LIPVQ_EINVAL, "tokenize: null pointer");
    if (!lipvq_tokenize_supported(A, J0, J1, D, K))
        return fail(LIPVQ_EUNSUPPORTED, "tokenize: unsupported shape A=%d J0=%d J1=%d D=%d K=%d", A, J0, J1, D, K);
    if (N > 2147483647LL) return fail(LIPVQ_EUNSUPPORTED, "tokenize: N too large");
    if ((((uintptr_t)codebook | (uintptr_t)zq) & 15) != 0) return fail(LIPVQ_EINVAL, "tokenize: codebook and zq must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    int* amb_count = (int*)ws;
    int* amb_list = (int*)(ws + 64);
    float* amb_z = (float*)(ws + 64 + (((sizeof(int) * (size_t)N) + 63) & ~(size_t)63));
    hipError_t e = hipMemsetAsync(amb_count, 0, 64, st);
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "tokenize: %s", hipGetErrorString(e));
    TokArgs a{x, packed, (const unsigned char*)prep, codebook, idx, zq, (unsigned long long*)usage, ze_out,
              amb_count, amb_list, amb_z, N, A, D, K, LIPVQ_SCREEN_GAMMA};
    int rc;
    switch (D) {
        case 32: rc = launch_tokenize<2>(a, st); break;
        case 64: rc = launch_tokenize<4>(a, st); break;
        default: rc = launch_tokenize<8>(a, st); break;
    }
    if (rc) return rc;
    return lipvq_launch_rows(amb_z, 1, codebook, idx, zq, usage, amb_list, amb_count, N, K, D, st);
}
