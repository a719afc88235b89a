// lipvq_embed.hip -- the step right after the tokenizer: the reference's
// ICLTransformer.input_embedding() + the interleave of (context_obs, context_actions, obs)
// (reference robomimic/models/obs_nets.py:2525-2543 and :2580-2596):
//     e = LayerNorm(Linear(z_latent) + time_embedding[t])          dropout is the caller's (identity in eval)
//     transformer_embeddings[b, 2t] = e(context_obs), [b, 2t+1] = e(context_actions), [b, 2T+t] = e(obs)
//
// MI355X design.  For the LipVQ tokenizer z_latent[n] IS codebook[idx[n]] (v5:47,84), so
//     Linear(z_latent)[n] = (codebook . W^T + b)[idx[n]]
// and the Linear over N rows collapses into a [K][E] table computed once per parameter update
// (linear_kernel, fp32 MFMA, k-ordered chain from the bias = the canonical Linear).  The per-action work
// is then a gather of one table row + the time embedding + LayerNorm, written straight into its interleaved
// slot of the [B][3T][E] transformer input: one launch (embed_rows_kernel), HBM-write bound
// (4 E bytes per action; the table and the time embeddings stay in L2), and z_latent itself never
// exists in HBM -- the tokenizer hands over 8-byte indices.  Rows that are not codebook rows (the
// observation streams, the VQVAE's straight-through value) take linear_kernel over their N rows and
// then the same embed_rows_kernel with idx = NULL.
//
// Canonical LayerNorm (shared with oracle/lipvq_oracle.c, bit for bit): a row is owned by 16 lanes, lane l holding the
// float4 groups q = l, l+16, ...;  lane sums run in that order and the 16 lane sums are combined by a butterfly with
// partners l^1, l^2, l^7, l^15 (four DPP adds);  mean = s*(1/E), var = sum((v-mean)^2)*(1/E) (fmaf chain, same order),
// rstd = 1/sqrt(var+eps), y = fmaf((v-mean)*rstd, w, b).
// ABI: include/lipvq.h.
#include <stdlib.h>

#include "lipvq_common.h"

// ---------------------------------------------------------------------------------------------------
// y[N][E] = x[N][Kin] . W[E][Kin]^T + b      (nn.Linear; obs_nets.py:2536 `embed_encoder`)
// WG = 4 waves; tile = 32 rows x 128 columns, one 32x32 MFMA tile per wave; K staged 32 at a time.
//   A operand: lane (m = lane & 31, kh = lane >> 5) = x[row0 + m][k0 + 2s + kh]
//   B operand: lane (n = lane & 31, kh)             = W[e0 + n][k0 + 2s + kh]
//   D[m][n]  : col n = lane & 31, row m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// ---------------------------------------------------------------------------------------------------
#define LIN_ROWS 32
#define LIN_KC 64

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     float* __restrict__ pre, int act, int64_t N, int Kin, int E) {
    __shared__ float xs[LIN_ROWS][LIN_KC + 1];
    __shared__ float ws[128][LIN_KC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * LIN_ROWS;
    const int e0 = blockIdx.y * 128;
    const int ecol = e0 + wave * 32 + li;
    const bool tile_ok = (e0 + wave * 32) < E;          // wave-uniform; a ragged last tile computes on zero rows of W
    f32x16 acc;
    const float b0 = (ecol < E && bias) ? bias[ecol] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b0;
    // Chunks of LIN_KC = 64 input features, the NEXT chunk's global loads in flight (registers) while the current one is
    // multiplied.  The first version (32-feature chunks, loads issued after the barrier that ended the previous chunk) paid one
    // L2 round trip per chunk: 30 us per call at N = 80, Kin = 208 -- 80 % of the default action branch's forward.
    constexpr int XV = LIN_ROWS * LIN_KC / 256, WV = 128 * LIN_KC / 256;       // floats per thread and chunk
    float xr[XV], wr[WV];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + 256 * i, r = f / LIN_KC, k = f % LIN_KC;
            const int64_t row = row0 + r;
            xr[i] = (row < N && k0 + k < Kin) ? x[(size_t)row * Kin + k0 + k] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int f = tid + 256 * i, r = f / LIN_KC, k = f % LIN_KC;
            const int e = e0 + r;
            wr[i] = (e < E && k0 + k < Kin) ? W[(size_t)e * Kin + k0 + k] : 0.0f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Kin; k0 += LIN_KC) {
#pragma unroll
        for (int i = 0; i < XV; ++i) { const int f = tid + 256 * i; xs[f / LIN_KC][f % LIN_KC] = xr[i]; }
#pragma unroll
        for (int i = 0; i < WV; ++i) { const int f = tid + 256 * i; ws[f / LIN_KC][f % LIN_KC] = wr[i]; }
        __syncthreads();
        if (k0 + LIN_KC < Kin) fetch(k0 + LIN_KC);          // uniform
        if (tile_ok) {
            const int kend = (Kin - k0 < LIN_KC) ? ((Kin - k0 + 1) >> 1) : (LIN_KC / 2);
            for (int s = 0; s < kend; ++s) {
                const float a = xs[li][2 * s + kh];
                const float b = ws[wave * 32 + li][2 * s + kh];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (ecol >= E) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < N) {
            if (pre) pre[(size_t)row * E + ecol] = acc[r];
            y[(size_t)row * E + ecol] = lq_act_apply(acc[r], act);
        }
    }
}

// Large-N variant of linear_kernel: 128 x 128 (or 256 x 64) tile per workgroup, each of the 4 waves owns 64 x 64 (2 x 2 MFMA tiles, so
// every LDS operand feeds two MFMAs), K staged 32 at a time with the next chunk's global loads in flight (registers)
// while the current chunk is multiplied.  Same k-ordered chains as linear_kernel: bit-identical results.
#define LINB_KC 32

// WR x WC waves, each owning a 64 x 64 block: (2, 2) = 128 x 128 tile; (4, 1) = 256 x 64 tile for narrow outputs (E <= 64)
template <int WR, int WC>
__global__ __launch_bounds__(256) void linear_big_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ pre, int act, int64_t N, int Kin, int E) {
    constexpr int TR = 64 * WR, TC = 64 * WC;
    __shared__ float xs[TR][LINB_KC + 1];
    __shared__ float ws[TC][LINB_KC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const int64_t row0 = (int64_t)blockIdx.x * TR;
    const int e0 = blockIdx.y * TC;
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ecol = e0 + (2 * wc + j) * 32 + li;
        const float b0 = (ecol < E && bias) ? bias[ecol] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b0;
    }
    // staging: thread -> (row r = tid >> 3 (+32 i), 4 consecutive k at 4 (tid & 7)): 128 B per row and instruction
    float4 xr[2 * WR], wrg[2 * WC];
    const int sr = tid >> 3, sk = 4 * (tid & 7);
    auto fetch = [&](int k0) {
        const bool kin = k0 + sk < Kin;                          // Kin is a multiple of 4 on this path
#pragma unroll
        for (int i = 0; i < 2 * WR; ++i) {
            const int64_t row = row0 + sr + 32 * i;
            xr[i] = (row < N && kin) ? *reinterpret_cast<const float4*>(x + (size_t)row * Kin + k0 + sk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2 * WC; ++i) {
            const int e = e0 + sr + 32 * i;
            wrg[i] = (e < E && kin) ? *reinterpret_cast<const float4*>(W + (size_t)e * Kin + k0 + sk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Kin; k0 += LINB_KC) {
#pragma unroll
        for (int i = 0; i < 2 * WR; ++i) {
            float* xd = &xs[sr + 32 * i][sk];
            xd[0] = xr[i].x; xd[1] = xr[i].y; xd[2] = xr[i].z; xd[3] = xr[i].w;
        }
#pragma unroll
        for (int i = 0; i < 2 * WC; ++i) {
            float* wd = &ws[sr + 32 * i][sk];
            wd[0] = wrg[i].x; wd[1] = wrg[i].y; wd[2] = wrg[i].z; wd[3] = wrg[i].w;
        }
        __syncthreads();
        if (k0 + LINB_KC < Kin) fetch(k0 + LINB_KC);            // in flight during the MFMAs below
        const int kend = (Kin - k0 < LINB_KC) ? ((Kin - k0 + 1) >> 1) : (LINB_KC / 2);
        for (int s2 = 0; s2 < kend; ++s2) {
            const int k = 2 * s2 + kh;
            const float a0 = xs[(2 * wr) * 32 + li][k], a1 = xs[(2 * wr + 1) * 32 + li][k];
            const float b0 = ws[(2 * wc) * 32 + li][k], b1 = ws[(2 * wc + 1) * 32 + li][k];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ecol = e0 + (2 * wc + j) * 32 + li;
            if (ecol >= E) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + (2 * wr + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < N) {
                    if (pre) pre[(size_t)row * E + ecol] = acc[i][j][r];
                    y[(size_t)row * E + ecol] = lq_act_apply(acc[i][j][r], act);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// out[slot(n)][:] = LayerNorm(src[idx ? idx[n] : n][:] + pos[n % T][:])
// 16 lanes own one row (4 rows per wavefront step): the per-row work that every lane repeats (the two reductions --
// four DPP adds each --, the reciprocal square root, address arithmetic) is shared by four rows per instruction, which
// is what keeps the VALU under the store stream.  NJ = float4 groups per lane (16 NJ >= E / 4).
// A workgroup takes 16*chunk consecutive rows per grid-stride step (step r: wave w, lane group g -> row n0 + 16 r + 4 w + g),
// so its stores sweep one contiguous span of the output; each wave fetches its indices for all steps with one vector
// load; ln_w / ln_b sit in LDS; the [T][E] time table is read through the vector L1 (20 KiB at the reference's T = 10).
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float lq_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// all-reduce over the 16 lanes of a DPP row; partner order l^1, l^2, l^7, l^15 (the oracle's row_allsum)
__device__ __forceinline__ float lq_row_allsum(float s) {
#ifdef LQ_EMB_NORED
    return s;               // ablation only (wrong results)
#endif
    s = s + lq_dpp<0xB1>(s);        // quad_perm [1,0,3,2]
    s = s + lq_dpp<0x4E>(s);        // quad_perm [2,3,0,1]
    s = s + lq_dpp<0x141>(s);       // row_half_mirror
    s = s + lq_dpp<0x140>(s);       // row_mirror
    return s;
}

// 64-lane butterfly (backward kernel only; no canonical order attached)
__device__ __forceinline__ float lq_wave_allsum(float s) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}

struct EmbedRowsArgs {
    const float* src;
    const int64_t* idx;
    const float* pos;
    const float* ln_w;
    const float* ln_b;
    float* out;
    float* stats;
    int64_t N, src_rows, out_bstride, out_tstride, out_offset;
    int T, E, chunk;
    float eps;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define EMB_CHUNK 16        // most steps per workgroup item (4 indices per step and wave: 64 lanes fetch 16 steps)

template <int NJ>
__global__ __launch_bounds__(256) void embed_rows_kernel(const EmbedRowsArgs a) {
    extern __shared__ float4 emb_lds[];                      // [ln_w: E4][ln_b: E4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l = lane & 15;
    const int E4 = a.E >> 2;
    float4* sw = emb_lds;
    float4* sb = sw + E4;
    for (int q = tid; q < E4; q += 256) {
        sw[q] = reinterpret_cast<const float4*>(a.ln_w)[q];
        sb[q] = reinterpret_cast<const float4*>(a.ln_b)[q];
    }
    __syncthreads();
    const float invE = 1.0f / (float)a.E;
    const float qnan = __builtin_nanf("");
    const int q16 = 16 / a.T, r16 = 16 - q16 * a.T;
    const int64_t per_item = 16 * (int64_t)a.chunk;
    for (int64_t n0 = (int64_t)blockIdx.x * per_item; n0 < a.N; n0 += (int64_t)gridDim.x * per_item) {
        // this wave's indices for all steps of the item in one vector load: lane (r = lane >> 2, g = lane & 3)
        int64_t kv = n0 + (lane >> 2) * 16 + wave * 4 + (lane & 3);
        if (a.idx && kv < a.N && (lane >> 2) < a.chunk) kv = a.idx[kv];
        int64_t n = n0 + wave * 4 + g;
        int64_t b = n / a.T;
        int t = (int)(n - b * a.T);
        for (int r = 0; r < a.chunk; ++r) {
            const bool live = n < a.N;
            const int64_t k = __shfl(kv, 4 * r + g, 64);
            const bool ok = k >= 0 && k < a.src_rows;          // a bad index poisons its row instead of faulting
            const float4* srow = reinterpret_cast<const float4*>(a.src + (size_t)((ok && live) ? k : 0) * a.E);
            const float4* prow = a.pos ? reinterpret_cast<const float4*>(a.pos + (size_t)(live ? t : 0) * a.E) : nullptr;
            float4 v[NJ];
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = l + 16 * j;
                if (q < E4) {
#ifdef LQ_EMB_NOLOAD
                    float4 c = make_float4((float)k, (float)q, 1.0f, 2.0f);      // ablation only (wrong results)
#else
                    float4 c = srow[q];
#endif
                    if (a.pos) {
                        const float4 p = prow[q];
                        c.x = c.x + p.x; c.y = c.y + p.y; c.z = c.z + p.z; c.w = c.w + p.w;
                    }
                    v[j] = c;
                    s = s + c.x; s = s + c.y; s = s + c.z; s = s + c.w;
                }
            }
            const float mean = lq_row_allsum(s) * invE;
            float ss = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (l + 16 * j < E4) {
                    float4 d = v[j];
                    d.x = d.x - mean; d.y = d.y - mean; d.z = d.z - mean; d.w = d.w - mean;
                    v[j] = d;
                    ss = lq_fma(d.x, d.x, ss); ss = lq_fma(d.y, d.y, ss);
                    ss = lq_fma(d.z, d.z, ss); ss = lq_fma(d.w, d.w, ss);
                }
            const float var = lq_row_allsum(ss) * invE;
            const float rstd = 1.0f / lq_sqrt(var + a.eps);
            float4* orow = reinterpret_cast<float4*>(a.out + (size_t)(b * a.out_bstride + t * a.out_tstride + a.out_offset));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = l + 16 * j;
                if (q < E4 && live) {
                    const float4 w = sw[q], bb = sb[q];
                    f32x4 o;
                    o.x = lq_fma(v[j].x * rstd, w.x, bb.x);
                    o.y = lq_fma(v[j].y * rstd, w.y, bb.y);
                    o.z = lq_fma(v[j].z * rstd, w.z, bb.z);
                    o.w = lq_fma(v[j].w * rstd, w.w, bb.w);
                    if (!ok) o = f32x4{qnan, qnan, qnan, qnan};
#ifdef LQ_EMB_NOSTORE
                    if (o.x == 12345.678f) *reinterpret_cast<f32x4*>(orow + q) = o;     // ablation only
#elif defined(LQ_EMB_NO_NT)
                    *reinterpret_cast<f32x4*>(orow + q) = o;
#else
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(orow + q));   // written once, read by a later launch
#endif
                }
            }
            if (a.stats && live && l == 0) {
                a.stats[2 * n] = ok ? mean : qnan;
                a.stats[2 * n + 1] = ok ? rstd : qnan;
            }
            n += 16;
            t += r16;
            b += q16;
            if (t >= a.T) { t -= a.T; ++b; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward of embed_rows_kernel.  gout is addressed like out.  Per row:
//   xhat = (src[k] + pos[t] - mean) * rstd,  gh = gout * w,
//   gv   = rstd * (gh - mean_e(gh) - xhat * mean_e(gh * xhat))
//   g_src[k] += gv (atomic; = gradient of the table / of the dense pre-LayerNorm rows),  g_pos[t] += gv,
//   g_lnw += gout * xhat,  g_lnb += gout   (kept in registers across the wave's rows, flushed once).
// ---------------------------------------------------------------------------------------------------
struct EmbedBwdArgs {
    const float* gout;
    const float* src;
    const int64_t* idx;
    const float* pos;
    const float* stats;
    const float* ln_w;
    float* g_src;
    float* g_pos;
    float* g_lnw;
    float* g_lnb;
    int64_t N, src_rows, out_bstride, out_tstride, out_offset;
    int T, E;
    float* gv;                 // large batches: the rows' gradients are written here (and scattered by a second pass), not added atomically
    int64_t* idx_clean;        // ... with the indices made safe for that pass (a row with a bad index: code 0, zero gradient)
};

// Large batches (lipvq_embed_rows_bwd_ws_f32).  The kernel above adds every row's E-float gradient to its table row AND to its
// time-embedding row with fp32 atomics: 2 N E atomics, the time-embedding ones onto T E addresses -- 8.7 ms at the metric's
// batch (N = 524 280, E = 512, T = 10), 55 ms when the codes collapse onto one table row.  Here a wave owns ONE time step t and
// walks batches b, b + W, ...: the time-embedding gradient stays in registers and costs one flush per wave; the row gradient is
// written once (gv [N][E]) and summed per code by the counting-sort scatter (lipvq_scatter.hip): no atomics on the table either.
template <int NJ>
__global__ __launch_bounds__(256) void embed_rows_bwd_gv_kernel(const EmbedBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int E4 = a.E >> 2;
    const float fE = (float)a.E;
    float4 w[NJ], aw[NJ], ab[NJ], ap[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        aw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ap[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[j] = q < E4 ? reinterpret_cast<const float4*>(a.ln_w)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __shared__ float4 red[3][4][64 * NJ];                      // the four waves' (ap, aw, ab) before ONE wave flushes them
    const int wave = threadIdx.x >> 6;
    const int64_t gpt = gridDim.x / a.T;                       // workgroups per time step (host: >= 1)
    if ((int64_t)blockIdx.x >= gpt * a.T) return;              // (whole workgroups: no barrier is skipped by part of one)
    const int t = (int)(blockIdx.x % a.T);                     // the four waves of a workgroup share the time step
    const int64_t wpt = gpt * 4;
    const int64_t B = (a.N + a.T - 1) / a.T;
    const float4* prow = a.pos ? reinterpret_cast<const float4*>(a.pos + (size_t)t * a.E) : nullptr;
    for (int64_t b = (int64_t)(blockIdx.x / a.T) * 4 + wave; b < B; b += wpt) {
        const int64_t n = b * a.T + t;
        if (n >= a.N) break;                                   // (wave-uniform)
        const int64_t k = a.idx ? a.idx[n] : n;
        const bool ok = k >= 0 && k < a.src_rows;
        float4* gvrow = a.gv ? reinterpret_cast<float4*>(a.gv + (size_t)n * a.E) : nullptr;
        if (a.idx_clean && lane == 0) a.idx_clean[n] = ok ? k : 0;
        if (!ok) {                                             // wave-uniform: a poisoned row contributes nothing
            if (gvrow) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (lane + 64 * j < E4) gvrow[lane + 64 * j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        const float mean = a.stats[2 * n], rstd = a.stats[2 * n + 1];
        const float4* srow = reinterpret_cast<const float4*>(a.src + (size_t)k * a.E);
        const float4* grow =
            reinterpret_cast<const float4*>(a.gout + (size_t)(b * a.out_bstride + t * a.out_tstride + a.out_offset));
        float4 xh[NJ], gh[NJ];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q < E4) {
                float4 c = srow[q];
                if (prow) {
                    const float4 p = prow[q];
                    c.x = c.x + p.x; c.y = c.y + p.y; c.z = c.z + p.z; c.w = c.w + p.w;
                }
                c.x = (c.x - mean) * rstd; c.y = (c.y - mean) * rstd;
                c.z = (c.z - mean) * rstd; c.w = (c.w - mean) * rstd;
                const float4 g = grow[q];
                aw[j].x += g.x * c.x; aw[j].y += g.y * c.y; aw[j].z += g.z * c.z; aw[j].w += g.w * c.w;
                ab[j].x += g.x; ab[j].y += g.y; ab[j].z += g.z; ab[j].w += g.w;
                float4 h;
                h.x = g.x * w[j].x; h.y = g.y * w[j].y; h.z = g.z * w[j].z; h.w = g.w * w[j].w;
                s1 += (h.x + h.y) + (h.z + h.w);
                s2 += (h.x * c.x + h.y * c.y) + (h.z * c.z + h.w * c.w);
                xh[j] = c;
                gh[j] = h;
            }
        }
        const float c1 = lq_wave_allsum(s1) / fE, c2 = lq_wave_allsum(s2) / fE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q < E4) {
                float4 gv;
                gv.x = rstd * (gh[j].x - c1 - xh[j].x * c2);
                gv.y = rstd * (gh[j].y - c1 - xh[j].y * c2);
                gv.z = rstd * (gh[j].z - c1 - xh[j].z * c2);
                gv.w = rstd * (gh[j].w - c1 - xh[j].w * c2);
                if (gvrow) gvrow[q] = gv;
                ap[j].x += gv.x; ap[j].y += gv.y; ap[j].z += gv.z; ap[j].w += gv.w;
            }
        }
    }
    // one flush per WORKGROUP (every flush is 3 E atomics onto 2 E + T E addresses: with one per wave of an 8 192-wave grid the
    // flushes cost more than the rows -- 1.15 ms for the whole launch)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        red[0][wave][lane + 64 * j] = ap[j];
        red[1][wave][lane + 64 * j] = aw[j];
        red[2][wave][lane + 64 * j] = ab[j];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < E4) {
            float4 sp = red[0][0][q], sw = red[1][0][q], sb = red[2][0][q];
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                const float4 p = red[0][u][q], ww = red[1][u][q], bb = red[2][u][q];
                sp.x += p.x; sp.y += p.y; sp.z += p.z; sp.w += p.w;
                sw.x += ww.x; sw.y += ww.y; sw.z += ww.z; sw.w += ww.w;
                sb.x += bb.x; sb.y += bb.y; sb.z += bb.z; sb.w += bb.w;
            }
            if (a.g_pos) {
                float* d = a.g_pos + (size_t)t * a.E + 4 * q;
                atomicAdd(d + 0, sp.x); atomicAdd(d + 1, sp.y); atomicAdd(d + 2, sp.z); atomicAdd(d + 3, sp.w);
            }
            if (a.g_lnw) {
                float* d = a.g_lnw + 4 * q;
                atomicAdd(d + 0, sw.x); atomicAdd(d + 1, sw.y); atomicAdd(d + 2, sw.z); atomicAdd(d + 3, sw.w);
            }
            if (a.g_lnb) {
                float* d = a.g_lnb + 4 * q;
                atomicAdd(d + 0, sb.x); atomicAdd(d + 1, sb.y); atomicAdd(d + 2, sb.z); atomicAdd(d + 3, sb.w);
            }
        }
    }
}

template <int NJ>
__global__ __launch_bounds__(256) void embed_rows_bwd_kernel(const EmbedBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int E4 = a.E >> 2;
    const float fE = (float)a.E;
    float4 w[NJ], aw[NJ], ab[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        aw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < E4) w[j] = reinterpret_cast<const float4*>(a.ln_w)[q];
    }
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave0; n < a.N; n += nwaves) {
        const int64_t k = a.idx ? a.idx[n] : n;
        if (k < 0 || k >= a.src_rows) continue;          // wave-uniform
        const int64_t b = n / a.T;
        const int t = (int)(n - b * a.T);
        const float mean = a.stats[2 * n], rstd = a.stats[2 * n + 1];
        const float4* srow = reinterpret_cast<const float4*>(a.src + (size_t)k * a.E);
        const float4* prow = a.pos ? reinterpret_cast<const float4*>(a.pos + (size_t)t * a.E) : nullptr;
        const float4* grow =
            reinterpret_cast<const float4*>(a.gout + (size_t)(b * a.out_bstride + t * a.out_tstride + a.out_offset));
        float4 xh[NJ], gh[NJ];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q < E4) {
                float4 c = srow[q];
                if (prow) {
                    const float4 p = prow[q];
                    c.x = c.x + p.x; c.y = c.y + p.y; c.z = c.z + p.z; c.w = c.w + p.w;
                }
                c.x = (c.x - mean) * rstd; c.y = (c.y - mean) * rstd;
                c.z = (c.z - mean) * rstd; c.w = (c.w - mean) * rstd;
                const float4 g = grow[q];
                aw[j].x += g.x * c.x; aw[j].y += g.y * c.y; aw[j].z += g.z * c.z; aw[j].w += g.w * c.w;
                ab[j].x += g.x; ab[j].y += g.y; ab[j].z += g.z; ab[j].w += g.w;
                float4 h;
                h.x = g.x * w[j].x; h.y = g.y * w[j].y; h.z = g.z * w[j].z; h.w = g.w * w[j].w;
                s1 += (h.x + h.y) + (h.z + h.w);
                s2 += (h.x * c.x + h.y * c.y) + (h.z * c.z + h.w * c.w);
                xh[j] = c;
                gh[j] = h;
            }
        }
        const float c1 = lq_wave_allsum(s1) / fE, c2 = lq_wave_allsum(s2) / fE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q < E4) {
                float4 gv;
                gv.x = rstd * (gh[j].x - c1 - xh[j].x * c2);
                gv.y = rstd * (gh[j].y - c1 - xh[j].y * c2);
                gv.z = rstd * (gh[j].z - c1 - xh[j].z * c2);
                gv.w = rstd * (gh[j].w - c1 - xh[j].w * c2);
                if (a.g_src) {
                    float* d = a.g_src + (size_t)k * a.E + 4 * q;
                    if (a.idx) {
                        atomicAdd(d + 0, gv.x); atomicAdd(d + 1, gv.y); atomicAdd(d + 2, gv.z); atomicAdd(d + 3, gv.w);
                    } else {                                  // one row, one writer
                        reinterpret_cast<float4*>(d)[0] = gv;
                    }
                }
                if (a.g_pos) {
                    float* d = a.g_pos + (size_t)t * a.E + 4 * q;
                    atomicAdd(d + 0, gv.x); atomicAdd(d + 1, gv.y); atomicAdd(d + 2, gv.z); atomicAdd(d + 3, gv.w);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < E4) {
            if (a.g_lnw) {
                float* d = a.g_lnw + 4 * q;
                atomicAdd(d + 0, aw[j].x); atomicAdd(d + 1, aw[j].y); atomicAdd(d + 2, aw[j].z); atomicAdd(d + 3, aw[j].w);
            }
            if (a.g_lnb) {
                float* d = a.g_lnb + 4 * q;
                atomicAdd(d + 0, ab[j].x); atomicAdd(d + 1, ab[j].y); atomicAdd(d + 2, ab[j].z); atomicAdd(d + 3, ab[j].w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
static int embed_grid(int64_t rows_per_wg_step, int64_t N) {
    int64_t g = (N + rows_per_wg_step - 1) / rows_per_wg_step;
#ifndef LQ_EMB_GRID
#define LQ_EMB_GRID 2048
#endif
    if (g > LQ_EMB_GRID) g = LQ_EMB_GRID;              // 8 workgroups per CU, grid-stride beyond
    if (g < 1) g = 1;
    return (int)g;
}

// small batches (a training step) spread 16 rows per workgroup; large ones up to 16 * EMB_CHUNK consecutive rows
static int embed_chunk(int64_t N) {
    const int64_t c = N / (16 * 2048);
    return c < 1 ? 1 : (c > EMB_CHUNK ? EMB_CHUNK : (int)c);
}

static int embed_check(const char* what, int64_t N, int T, int E, int64_t src_rows, int64_t bstride, int64_t tstride,
                       int64_t offset) {
    if (N < 0 || T <= 0 || src_rows <= 0) return fail(LIPVQ_EINVAL, "%s: bad sizes N=%lld T=%d src_rows=%lld", what,
                                                      (long long)N, T, (long long)src_rows);
    if (E <= 0 || (E & 3) || E > 1024) return fail(LIPVQ_EUNSUPPORTED, "%s: E=%d (need a multiple of 4, <= 1024)", what, E);
    if ((bstride & 3) || (tstride & 3) || (offset & 3) || bstride < 0 || tstride < 0 || offset < 0)
        return fail(LIPVQ_EINVAL, "%s: output strides/offset must be non-negative multiples of 4 floats", what);
    return LIPVQ_OK;
}

extern "C" {

int lipvq_linear_act_f32(const float* x, const float* W, const float* b, float* y, float* pre, int64_t N, int Kin,
                         int E, int act, void* stream) {
    if (!x || !W || !y) return fail(LIPVQ_EINVAL, "lipvq_linear_act_f32: null pointer");
    if (N < 0 || Kin <= 0 || E <= 0) return fail(LIPVQ_EINVAL, "lipvq_linear_act_f32: bad sizes");
    if (act < LIPVQ_ACT_NONE || act > LIPVQ_ACT_RELU) return fail(LIPVQ_EINVAL, "lipvq_linear_act_f32: bad activation %d", act);
    if (N == 0) return LIPVQ_OK;
    const int64_t gx = (N + LIN_ROWS - 1) / LIN_ROWS;
    if (gx > 0x7fffffffLL) return fail(LIPVQ_EUNSUPPORTED, "lipvq_linear_act_f32: N too large");
#ifndef LQ_LIN_BIG_MIN
#define LQ_LIN_BIG_MIN 512          // workgroups of the 128 x 128 tiling needed before it pays (2 per CU)
#endif
    const bool narrow = E <= 64;                                  // 256 x 64 tiles instead of 128 x 128
    const int TR = narrow ? 256 : 128, TC = narrow ? 64 : 128;
    const int64_t big_wgs = ((N + TR - 1) / TR) * ((E + TC - 1) / TC);
    if (big_wgs >= LQ_LIN_BIG_MIN && (Kin & 3) == 0) {
        dim3 grid((unsigned)((N + TR - 1) / TR), (unsigned)((E + TC - 1) / TC));
        if (narrow)
            hipLaunchKernelGGL((linear_big_kernel<4, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, W, b, y, pre, act, N, Kin, E);
        else
            hipLaunchKernelGGL((linear_big_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, W, b, y, pre, act, N, Kin, E);
        return check_launch("linear_big_kernel");
    }
    dim3 grid((unsigned)gx, (unsigned)((E + 127) / 128));
    hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, W, b, y, pre, act, N, Kin, E);
    return check_launch("linear_kernel");
}

int lipvq_linear_f32(const float* x, const float* W, const float* b, float* y, int64_t N, int Kin, int E,
                     void* stream) {
    return lipvq_linear_act_f32(x, W, b, y, nullptr, N, Kin, E, LIPVQ_ACT_NONE, stream);
}

int lipvq_embed_rows_f32(const float* src, const int64_t* idx, const float* pos, const float* ln_w, const float* ln_b,
                         float eps, float* out, float* stats, int64_t N, int T, int E, int64_t src_rows,
                         int64_t out_batch_stride, int64_t out_t_stride, int64_t out_offset, void* stream) {
    if (!src || !ln_w || !ln_b || !out) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_f32: null pointer");
    const int rc = embed_check("lipvq_embed_rows_f32", N, T, E, src_rows, out_batch_stride, out_t_stride, out_offset);
    if (rc) return rc;
    if (!idx && src_rows < N) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_f32: src has fewer rows than N");
    if (N == 0) return LIPVQ_OK;
    const int chunk = embed_chunk(N);
    EmbedRowsArgs a{src, idx, pos, ln_w, ln_b, out, stats, N, src_rows, out_batch_stride, out_t_stride, out_offset, T, E,
                    chunk, eps};
    const dim3 grid(embed_grid(16 * chunk, N)), block(256);
    const size_t lds = (size_t)2 * E * sizeof(float);       // ln_w, ln_b
    hipStream_t st = (hipStream_t)stream;
    const int nj = (E / 4 + 15) / 16;
    if (nj <= 2) hipLaunchKernelGGL(embed_rows_kernel<2>, grid, block, lds, st, a);
    else if (nj <= 4) hipLaunchKernelGGL(embed_rows_kernel<4>, grid, block, lds, st, a);
    else if (nj <= 8) hipLaunchKernelGGL(embed_rows_kernel<8>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(embed_rows_kernel<16>, grid, block, lds, st, a);
    return check_launch("embed_rows_kernel");
}

int lipvq_embed_rows_bwd_f32(const float* gout, const float* src, const int64_t* idx, const float* pos,
                             const float* stats, const float* ln_w, float* g_src, float* g_pos, float* g_lnw,
                             float* g_lnb, int64_t N, int T, int E, int64_t src_rows, int64_t out_batch_stride,
                             int64_t out_t_stride, int64_t out_offset, void* stream) {
    if (!gout || !src || !stats || !ln_w) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_bwd_f32: null pointer");
    const int rc = embed_check("lipvq_embed_rows_bwd_f32", N, T, E, src_rows, out_batch_stride, out_t_stride, out_offset);
    if (rc) return rc;
    if (!idx && src_rows < N) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_bwd_f32: src has fewer rows than N");
    if (N == 0) return LIPVQ_OK;
    EmbedBwdArgs a{gout, src, idx, pos, stats, ln_w, g_src, g_pos, g_lnw, g_lnb, N, src_rows,
                   out_batch_stride, out_t_stride, out_offset, T, E, nullptr, nullptr};
    int g = embed_grid(4, N);
    if (g > 512) g = 512;                 // fewer waves = fewer final flushes of the LayerNorm gradients
    const dim3 grid(g), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch ((E + 255) / 256) {
        case 1: hipLaunchKernelGGL(embed_rows_bwd_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(embed_rows_bwd_kernel<2>, grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL(embed_rows_bwd_kernel<3>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(embed_rows_bwd_kernel<4>, grid, block, 0, st, a); break;
    }
    return check_launch("embed_rows_bwd_kernel");
}

// Large batches of indexed rows: no atomics on the table or the time embedding (embed_rows_bwd_gv_kernel + the counting-sort
// scatter).  workspace: gv [N][E] floats, the sanitised indices [N], then the scatter's own workspace.
int lipvq_embed_rows_bwd_ws_supported(int64_t N, int T, int E, int64_t src_rows) {
    return N >= 32768 && T >= 1 && T <= 1024 && E >= 4 && (E & 3) == 0 && E <= 1024 && src_rows <= 16384 &&
           lipvq_scatter_add_sorted_supported(N, (int)src_rows, E);
}

size_t lipvq_embed_rows_bwd_workspace_bytes(int64_t N, int T, int E, int64_t src_rows) {
    if (!lipvq_embed_rows_bwd_ws_supported(N, T, E, src_rows)) return 0;
    const size_t gv = ((size_t)N * E * sizeof(float) + 255) & ~(size_t)255, ic = ((size_t)N * sizeof(int64_t) + 255) & ~(size_t)255;
    return gv + ic + lipvq_scatter_add_sorted_workspace_bytes(N, (int)src_rows, E);
}

int lipvq_embed_rows_bwd_ws_f32(const float* gout, const float* src, const int64_t* idx, const float* pos, const float* stats,
                                const float* ln_w, float* g_src, float* g_pos, float* g_lnw, float* g_lnb, void* workspace,
                                int64_t N, int T, int E, int64_t src_rows, int64_t out_batch_stride, int64_t out_t_stride,
                                int64_t out_offset, void* stream) {
    if (!gout || !src || !stats || !ln_w || (idx && !workspace)) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_bwd_ws_f32: null pointer");
    const int rc = embed_check("lipvq_embed_rows_bwd_ws_f32", N, T, E, src_rows, out_batch_stride, out_t_stride, out_offset);
    if (rc) return rc;
    if (!idx && src_rows < N) return fail(LIPVQ_EINVAL, "lipvq_embed_rows_bwd_ws_f32: src has fewer rows than N");
    const bool okay = idx ? lipvq_embed_rows_bwd_ws_supported(N, T, E, src_rows) != 0 : (N >= 32768 && T <= 1024);
    if (!okay)
        return fail(LIPVQ_EUNSUPPORTED, "lipvq_embed_rows_bwd_ws_f32: N=%lld T=%d E=%d src_rows=%lld outside the supported range",
                    (long long)N, T, E, (long long)src_rows);
    const size_t gvb = ((size_t)N * E * sizeof(float) + 255) & ~(size_t)255, icb = ((size_t)N * sizeof(int64_t) + 255) & ~(size_t)255;
    // dense rows (idx == NULL): row n of g_src has one writer -- the row gradients ARE g_src, nothing to scatter
    float* gv = !g_src ? nullptr : (idx ? (float*)workspace : g_src);
    int64_t* idx_clean = (g_src && idx) ? (int64_t*)((char*)workspace + gvb) : nullptr;
    EmbedBwdArgs a{gout, src, idx, pos, stats, ln_w, g_src, g_pos, g_lnw, g_lnb, N, src_rows,
                   out_batch_stride, out_t_stride, out_offset, T, E, gv, idx_clean};
    static int grid_knob = -1;                                 // LIPVQ_EMBED_BWD_GRID: measurement knob
    if (grid_knob < 0) { const char* e = lq_knob("LIPVQ_EMBED_BWD_GRID"); grid_knob = e ? atoi(e) : 1024; }
    const dim3 grid(grid_knob < T ? T : grid_knob), block(256);   // >= one workgroup per time step (T <= 1024)
    hipStream_t st = (hipStream_t)stream;
    switch ((E + 255) / 256) {
        case 1: hipLaunchKernelGGL(embed_rows_bwd_gv_kernel<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(embed_rows_bwd_gv_kernel<2>, grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL(embed_rows_bwd_gv_kernel<3>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(embed_rows_bwd_gv_kernel<4>, grid, block, 0, st, a); break;
    }
    if (int e = check_launch("embed_rows_bwd_gv_kernel")) return e;
    if (!g_src || !idx) return LIPVQ_OK;
    return lipvq_scatter_add_sorted_f32(gv, idx_clean, g_src, (char*)workspace + gvb + icb, N, (int)src_rows, E, 0, stream);
}

}  // extern "C"
