// lipvq_screen.hip -- nearest-code search, fast path: MFMA screening + exact re-scoring.
// Replaces the same reference lines as lipvq_nearest.hip (v5:37-48) with the SAME results.
//
// Idea.  argmin_k |z - e_k| only needs the exact (torch-order, fp32) distance for codes that can
// possibly win.  A cheap approximation  d~(n,k) = |e'_k|^2 - 2 z'_n . e'_k  (centred operands
// z' = z - mu, e' = e - mu; the row constant |z'|^2 is dropped) is computed on the matrix cores:
// every operand is scaled by an exact power of two (block floating point, see lipvq_screen.h), split
// into two fp16 pieces (x = hi + lo, 22 significant bits) and three products hi*hi + lo*hi + hi*lo
// are accumulated in fp32 by v_mfma_f32_32x32x16_f16.  Each lane
// tracks, per row, the smallest and second smallest d~ it has seen and the code of the smallest.
// If, after all codes, the second smallest exceeds the smallest by more than W_n = 2 eps_n (eps_n
// bounds |d~ - d| for the row, see "error bound"), the approximate argmin IS the exact argmin in
// real arithmetic with a margin that also covers the rounding of the reference's own fp32
// distance, hence it is the reference's index.  Otherwise the row is appended to a list and the
// exact kernel (nearest_rows_kernel, or nearest_rows_encode_kernel when z_e was never stored: torch's
// 8-accumulator order, sqrt comparison, first-minimum rule) decides it.  No row is ever decided by
// the approximation alone unless it is certified.
//
// Error bound.  |d~ - d| <= eps_n = gamma * (E2max + 2 |z'_n| Emax), with E2max = max_k |e'_k|^2,
// Emax = max_k |e'_k|.  Analytically the fp16 split leaves 3 * 2^-22 * sum|z'e'| per dot product;
// hence at most 2^-20.4 of (E2max + 2 |z'| Emax); the fp32 accumulation inside the MFMA and the rounding of
// |e'|^2 add a few 2^-24 of the partial sums' magnitude.  Measured over 2*10^7 pairs of all supported widths and
// operand magnitudes from 1e-6 to 1e6 (scripts/measure_bound.py, measure_bound_scales.py): 2^-22.  gamma
// (LIPVQ_SCREEN_GAMMA) is 2^-18: 5x the analytic split bound, 16x the largest error observed;
// tests/test_gpu_screen.py (test_error_bound_holds, test_any_magnitude) assert a >= 4x margin on every run.
#include <stdlib.h>
#include <string.h>

#include "lipvq_screen.h"

extern "C" size_t lipvq_nearest_prep_bytes(int K, int D) {
    if (K <= 0 || D <= 0) return 0;
    return prep_layout(K, D).total;
}

// column means: 64 columns x 16 row groups per workgroup, double accumulation in a fixed order (deterministic).  (With 4 row
// groups a thread walked K/4 strided rows one after the other: 103 us at K = 1024 -- a quarter of a screen launch, paid by every
// training step at large batches and by the first tokenize call after each codebook update.)
#define PREP_MEAN_GROUPS 16
__global__ __launch_bounds__(64 * PREP_MEAN_GROUPS) void prep_mean_kernel(const float* __restrict__ cb, float* __restrict__ mu,
                                                                          unsigned* __restrict__ hdr, int K, int D, int Dpad) {
    __shared__ double part[PREP_MEAN_GROUPS][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + c;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;              // four independent chains: four loads in flight per thread
    if (d < D) {
        int k = g;
        for (; k + 3 * PREP_MEAN_GROUPS < K; k += 4 * PREP_MEAN_GROUPS) {
            s0 += (double)cb[(size_t)k * D + d];
            s1 += (double)cb[(size_t)(k + PREP_MEAN_GROUPS) * D + d];
            s2 += (double)cb[(size_t)(k + 2 * PREP_MEAN_GROUPS) * D + d];
            s3 += (double)cb[(size_t)(k + 3 * PREP_MEAN_GROUPS) * D + d];
        }
        for (; k < K; k += PREP_MEAN_GROUPS) s0 += (double)cb[(size_t)k * D + d];
    }
    part[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && d < Dpad) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < PREP_MEAN_GROUPS; ++q) t += part[q][c];
        const float m = (d < D) ? (float)(t / (double)K) : 0.0f;
        mu[d] = m;
        atomicMax(&hdr[4], __float_as_uint(fabsf(m)));          // max |mu|: the fused kernel's launch-wide scale (lipvq_fused.hip)
    }
}

// one thread per (code, step, half): 8 centred, -2-scaled elements -> fp16 hi/lo fragments
__global__ void prep_scale_kernel(unsigned* __restrict__ hdr) {           // after prep_e2_kernel: se from max |-2e'|
    hdr[3] = (unsigned)lq_scale_exp(__uint_as_float(hdr[2]));
}

__global__ void prep_frag_kernel(const float* __restrict__ cb, const float* __restrict__ mu, const unsigned* __restrict__ hdr,
                                 unsigned char* __restrict__ tiles, int K, int D, PrepLayout L) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)L.ntiles * 32 * L.S * 2;
    if (gid >= n) return;
    const int h = (int)(gid & 1);
    const int s = (int)((gid >> 1) % L.S);
    const int k = (int)(gid / (2 * (size_t)L.S));
    const float fe = lq_pow2f((int)hdr[3]);                      // 2^se, written by prep_scale_kernel
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 16 * s + 2 * j + h;
        float v = 0.0f;
        if (k < K && d < D) v = (-2.0f * (cb[(size_t)k * D + d] - mu[d])) * fe;
        const _Float16 vh = (_Float16)v;
        hi[j] = vh;
        lo[j] = (_Float16)(v - (float)vh);
    }
    unsigned char* t = tiles + (size_t)(k >> 5) * L.tile_bytes;
    const int c = k & 31;
    // Pad codes (k >= K: the tail of the last real tile and the tiles that round the count up to whole LDS stages) must lose
    // against every real code.  prep_e2_kernel gave them |e'|^2 = +inf -- but the PACK bookkeeping writes the tile index into the
    // low mantissa bits of every value (lq_track_one), which turns +inf into a NaN, and v_med3_f32 with a NaN operand returns the
    // MINIMUM of the other two: the lane's second minimum became its minimum and every row whose best code sat in a lane that
    // also holds a pad code went to the exact kernel (all rows at K = 37 or 128, three quarters at K = 1000; results were
    // right, only slow -- found in round 3 by the first test that asserts "something was certified" at K = 37).  So pad codes get
    // a huge FINITE |e'|^2: 2^(100 - se), i.e. 2^(100 + sz) in a row's units, above anything a real code can reach (< 2^40)
    // and below overflow for every row scale the kernels use; a row so small that it did overflow is merely not certified.
    if (k >= K && s == 0 && h == 0) {
        int ex = 100 - (int)hdr[3];
        ex = ex > 120 ? 120 : ex;
        reinterpret_cast<float*>(t + (size_t)L.S * 2048)[c] = lq_pow2f(ex);
    }
    *reinterpret_cast<f16x8*>(t + (((size_t)s * 2 + 0) * 64 + h * 32 + c) * 16) = hi;   // lane = h*32 + c
    *reinterpret_cast<f16x8*>(t + (((size_t)s * 2 + 1) * 64 + h * 32 + c) * 16) = lo;
    // the one-product screen's copy: hi fragments only, then the same |e'|^2 and |e'| (written by prep_e2_kernel, earlier in the
    // stream; the pad codes' value from just above)
    unsigned char* th = tiles + (L.o_tiles_hi - L.o_tiles) + (size_t)(k >> 5) * L.tile_bytes_hi;
    *reinterpret_cast<f16x8*>(th + ((size_t)s * 64 + h * 32 + c) * 16) = hi;
    if (s == 0 && h == 0) {
        const float* src = reinterpret_cast<const float*>(t + (size_t)L.S * 2048);
        float* dst = reinterpret_cast<float*>(th + (size_t)L.S * 1024);
        dst[c] = src[c];
        dst[32 + c] = src[32 + c];
    }
}

// one thread per code, after the scale is known: the one-product screen's codebook-side numbers.  With E = (-2 e') 2^se as
// prep_frag_kernel forms it (the same fp32 expression) and Eh its fp16 rounding:  en = |E| / 2^se (rounded up) into the tile,
// rho = max_k |E - Eh| / |E| into hdr[5].
__global__ void prep_res_kernel(const float* __restrict__ cb, const float* __restrict__ mu, unsigned char* __restrict__ tiles,
                                unsigned* __restrict__ hdr, int K, int D, PrepLayout L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float fe = lq_pow2f((int)hdr[3]);
    double c2 = 0.0, d2 = 0.0;
    for (int d = 0; d < D; ++d) {
        const float v = (-2.0f * (cb[(size_t)k * D + d] - mu[d])) * fe;
        const float r = v - (float)(_Float16)v;                       // exact
        c2 += (double)r * (double)r;
        d2 += (double)v * (double)v;
    }
    float* enp = reinterpret_cast<float*>(tiles + (size_t)(k >> 5) * L.tile_bytes + (size_t)L.S * 2048) + 32 + (k & 31);
    *enp = (float)(sqrt(d2) / (double)fe * (1.0 + 1e-6));
    if (d2 > 0.0) atomicMax(&hdr[5], __float_as_uint((float)(sqrt(c2 / d2) * (1.0 + 1e-6))));      // non-negative floats order like their bits
}

// one thread per code: |e'|^2 (double), statistics for the error bound
__global__ void prep_e2_kernel(const float* __restrict__ cb, const float* __restrict__ mu,
                               unsigned char* __restrict__ tiles, unsigned* __restrict__ hdr, int K, int D,
                               PrepLayout L) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L.ntiles * 32) return;
    float* e2p = reinterpret_cast<float*>(tiles + (size_t)(k >> 5) * L.tile_bytes + (size_t)L.S * 2048) + (k & 31);
    float* enp = e2p + 32;                             // |e'| (rounded up): the one-product screen's per-code error scale
    if (k >= K) { *e2p = INFINITY; *enp = 0.0f; return; }
    double s = 0.0;
    float mx = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float v = cb[(size_t)k * D + d] - mu[d];
        s += (double)v * (double)v;
        mx = fmaxf(mx, fabsf(2.0f * v));
    }
    const float e2 = (float)s;
    *e2p = e2;
    // |-2 e'| (rounded up): what the one-product bound multiplies by.  prep_res_kernel overwrites it with the value of the
    // REPRESENTED (scaled, then unscaled) operand once the scale is known; until then this already is an upper-bound-grade
    // value, so the certification never depends on the order of the preparation launches
    *enp = (float)(2.0 * sqrt(s) * (1.0 + 1e-6));
    atomicMax(&hdr[0], __float_as_uint(e2));          // non-negative floats order like their bit patterns
    atomicMax(&hdr[1], __float_as_uint(e2));          // Emax^2 (same quantity; kept separate for clarity)
    atomicMax(&hdr[2], __float_as_uint(mx));
}

extern "C" int lipvq_nearest_prepare_f32(const float* codebook, void* prep, int K, int D, void* stream) {
    if (!codebook || !prep || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "nearest_prepare: bad argument");
    PrepLayout L = prep_layout(K, D);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = (unsigned char*)prep;
    hipError_t e = hipMemsetAsync(base, 0, 64, st);
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "nearest_prepare: %s", hipGetErrorString(e));
    float* mu = (float*)(base + L.o_mu);
    hipLaunchKernelGGL(prep_mean_kernel, dim3((L.Dpad + 63) / 64), dim3(64 * PREP_MEAN_GROUPS), 0, st, codebook, mu, (unsigned*)base, K, D, L.Dpad);
    hipLaunchKernelGGL(prep_e2_kernel, dim3((L.ntiles * 32 + 255) / 256), dim3(256), 0, st, codebook, mu, base + L.o_tiles,
                       (unsigned*)base, K, D, L);
    hipLaunchKernelGGL(prep_scale_kernel, dim3(1), dim3(1), 0, st, (unsigned*)base);
    hipLaunchKernelGGL(prep_res_kernel, dim3((K + 255) / 256), dim3(256), 0, st, codebook, mu, base + L.o_tiles, (unsigned*)base, K, D, L);
    size_t n = (size_t)L.ntiles * 32 * L.S * 2;
    hipLaunchKernelGGL(prep_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codebook, mu,
                       (const unsigned*)base, base + L.o_tiles, K, D, L);
    return check_launch("nearest_prepare");
}

// ------------------------------------------------------------------------------------------
// screening kernel: 8 waves x 32 rows per workgroup (main loop: lq_screen_core, lipvq_screen.h)
// ------------------------------------------------------------------------------------------
template <int S, bool DBG, bool COARSE = false>
__global__ __launch_bounds__(SCREEN_WAVES * 64) void screen_kernel(
    const float* __restrict__ z, const unsigned char* __restrict__ prep, const float* __restrict__ cb,
    int64_t* __restrict__ idx, float* __restrict__ zq, unsigned long long* __restrict__ usage,
    int* __restrict__ amb_list, int* __restrict__ amb_count, float* __restrict__ dbg, int64_t N, int K, int D,
    float gamma) {
    using SC = StandaloneScreen<S>;
    using C = ScreenCfg<S, SC::TC, COARSE>;
    // PACK bookkeeping: where the three-product screen needs it (S <= 4), and ALWAYS for the one-product screen -- its margin
    // (2^-9 of the cross term) dwarfs the 2^(TB-23) the packed tile index perturbs a value by
    constexpr bool PK = SC::PACK || COARSE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const PrepLayout L = prep_layout(K, D);
    const unsigned* hdr = reinterpret_cast<const unsigned*>(prep);
    const float* mu = reinterpret_cast<const float*>(prep + L.o_mu);
    const unsigned char* tiles = prep + (COARSE ? L.o_tiles_hi : L.o_tiles);       // (the one-product screen stages hi-only tiles)
    const size_t tile_bytes = COARSE ? L.tile_bytes_hi : L.tile_bytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 31, h = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * SCREEN_WAVES + wave) * 32;
    const int64_t row = row0 + ln;
    const int64_t rowc = row < N ? row : N - 1;
    lq_ws_begin(amb_count);

    // this wave's 32 rows -> centred, row-scaled fp16 hi/lo A fragments (slot (h, j) of step s = feature 16s + 2j + h)
    f16x8 ah[S], al[S];
    float n2 = 0.0f, fown, a2lo = 0.0f, fzr = 1.0f;
    {
        const float* zr = z + (size_t)rowc * D;
        float vv[S][8];
        float amax = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * s + 2 * j + h;
                const float v = (d < D) ? zr[d] - mu[d] : 0.0f;
                vv[s][j] = v;
                n2 = lq_fma(v, v, n2);
                amax = fmaxf(amax, lq_abs(v));
            }
        n2 += __shfl_xor(n2, 32, 64);                     // |z'|^2 of row `ln`, in both halves
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int sz = lq_scale_exp(amax);
        const float fz = lq_pow2f(sz);
        fown = lq_pow2f(sz + (int)hdr[3]);                // units of this row's MFMA results: 2^(sz+se), |sz+se| <= 120
        fzr = fz;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = vv[s][j] * fz;
                const _Float16 vh = (_Float16)v;
                const float r = v - (float)vh;                         // exact: the one-product screen's row-side residual
                ah[s][j] = vh;
                al[s][j] = (_Float16)r;
                if constexpr (COARSE) a2lo = lq_fma(r, r, a2lo);
            }
        if constexpr (COARSE) a2lo += __shfl_xor(a2lo, 32, 64);
    }
    float frow[16];
    lq_row_factors(fown, lane, frow);
    // COARSE: the per-row scale of the one-product error bound (lq_coarse_zn), in the register layout of frow
    const float zn = COARSE ? lq_coarse_zn(a2lo, n2, fzr, fown, __uint_as_float(hdr[5])) : 0.0f;
    float znr[16];
    lq_row_factors(zn, lane, znr);

    float m1[16], m2[16];
    int k1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { m1[r] = INFINITY; m2[r] = INFINITY; k1[r] = 0; }

    if (DBG) {
        // test hook: plain loop straight from global memory, dumping every approximate distance
        for (int ct = 0; ct < L.ntiles; ++ct) {
            const unsigned char* tb = tiles + (size_t)ct * C::TILE_BYTES;
            const float e2 = reinterpret_cast<const float*>(tb + C::FRAG_BYTES)[ln];
            const float en = reinterpret_cast<const float*>(tb + C::FRAG_BYTES + 128)[ln];
            f32x16 acc;                                   // same arithmetic as lq_screen_core: chain from zero, |e'|^2 f added last
#pragma unroll                                            // (COARSE: chain seeded with |e'|^2 f - w, the booked lower bound)
            for (int r = 0; r < 16; ++r) acc[r] = COARSE ? lq_fma(-znr[r], en, e2 * frow[r]) : 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const f16x8 bh = COARSE ? *reinterpret_cast<const f16x8*>(tb + ((size_t)s * 64 + lane) * 16)
                                        : *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 0) * 64 + lane) * 16);
                const f16x8 bl = COARSE ? bh : *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 1) * 64 + lane) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc, 0, 0, 0);
                if constexpr (!COARSE) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc, 0, 0, 0);
                }
            }
            const int code = ct * 32 + ln;
            if (dbg && code < L.Kpad) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    // back to unscaled units; COARSE: the bound term is taken out again -- the test wants d~ itself
                    if (rr < N) dbg[(size_t)rr * L.Kpad + code] = (COARSE ? lq_fma(znr[r], en, acc[r]) : lq_fma(e2, frow[r], acc[r])) / frow[r];
                }
            }
            lq_track_part<0, 16, false, COARSE>(acc, e2, en, frow, znr, code, 0xffffffffu, m1, m2, k1);
        }
    } else {
        lq_screen_core_rg<S, SCREEN_WAVES * 64, SC::TC, SC::NB, PK, 1, COARSE>(
            reinterpret_cast<const f16x8 (&)[1][S]>(ah), reinterpret_cast<const f16x8 (&)[1][S]>(al), tiles, L.ntiles, lds, tid, frow,
            reinterpret_cast<const float (&)[1][16]>(znr), reinterpret_cast<float (&)[1][16]>(m1), reinterpret_cast<float (&)[1][16]>(m2),
            reinterpret_cast<int (&)[1][16]>(k1));
    }
    int my_k;
    const float pack_eps = (PK && !DBG) ? lq_pow2f(lq_pack_bits(L.ntiles) - 23) : 0.0f;
    const unsigned keep_mask = (PK && !DBG) ? ~((1u << lq_pack_bits(L.ntiles)) - 1u) : 0xffffffffu;
    unsigned char* scratch = lds + (size_t)wave * LQ_DECIDE_BYTES;
    LqDecision dec;
    bool certified;
    // (the debug hook runs with the caller's gamma, which need not bound anything: its uncertified rows get no short list)
    if (PK && !DBG) {
        certified = lq_screen_decide<true, COARSE>(m1, m2, k1, scratch, hdr, n2, fown, gamma, K, D, lane, my_k, dec, pack_eps, keep_mask,
                                                   zn, tiles, tile_bytes, C::FRAG_BYTES);
        lq_screen_emit<true>(dec, certified, true, my_k, row, row < N, amb_count, amb_list, N, K, lane, keep_mask, scratch);
    } else {
        certified = lq_screen_decide<false, COARSE>(m1, m2, k1, scratch, hdr, n2, fown, gamma, K, D, lane, my_k, dec, 0.0f, 0xffffffffu,
                                                    zn, tiles, tile_bytes, C::FRAG_BYTES);
        lq_screen_emit<false>(dec, certified, !DBG, my_k, row, row < N, amb_count, amb_list, N, K, lane, keep_mask, scratch);
    }
    if (h == 0 && row < N && certified) idx[row] = (int64_t)my_k;
    if (usage) lq_usage_add(usage, my_k, h == 0 && row < N && certified);
    if (zq) lq_screen_gather(cb, zq, my_k, certified, row0, N, D, lane);
    lq_ws_publish(amb_count);                        // the grid's last workgroup publishes the number of listed rows (lipvq_screen.h)
}

// Exact scan of codes [kb, ke) for one row held in registers: torch's 8-accumulator order, sqrt comparison, first minimum.
// Every 32 dimensions the partial sums are combined in the final order; since each accumulator only grows (squares are
// non-negative and fp32 addition is monotone) and the combination is monotone in every accumulator, a partial value that
// already reaches the best square so far proves the full square does too, and the rest of that code's row is not
// fetched.  Same results bit for bit; the scan is bound by re-reading the codebook (1.15 ms for 3 317 rows at K = 8192,
// D = 128 without the early exit).
// The plain VQVAE's distance (vq:57-60, `(z_e.unsqueeze(1) - E).pow(2).sum(-1)`) of one row in registers against one code:
// lq_sqdist32's order for D = 8 DCH < 512 (rounded squares; 8-vector i -> accumulator i mod 4 while whole groups of four remain,
// left-overs -> accumulator 0; accumulators 1..3 added to 0; lanes left to right) -- the same code as nearest_direct_kernel.
template <int DCH>
__device__ __forceinline__ float lq_sq32_row(const float (&zr)[DCH * 8], const float4* __restrict__ c4) {
    static_assert(DCH < 64, "the cascade of torch's sum starts at D = 512");
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[q][l] = 0.f;
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
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
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float v = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
        s = (l == 0) ? v : s + v;
    }
    return s;
}

template <int DCH, bool SEEDED = false, int DIST = LIPVQ_DIST_NORM>
__device__ __forceinline__ void lq_exact_scan(const float (&zr)[DCH * 8], const float* __restrict__ cb, int kb, int ke,
                                              float& best_v, float& best_s, int& best_k, float s_prune = INFINITY) {
    constexpr int D = DCH * 8;
    if constexpr (DIST == LIPVQ_DIST_SQSUM) {
        // the compared value IS the sum (no root, so no two sums share a value they do not have): first minimum over ascending k
        for (int k = kb; k < ke; ++k) {
            const float v = lq_sq32_row<DCH>(zr, reinterpret_cast<const float4*>(cb + (size_t)k * D));
            if (v < best_v) { best_v = v; best_s = v; best_k = k; }
        }
        return;
    }
#ifdef LQ_SCAN_G
    constexpr int G = LQ_SCAN_G;
#else
    // chunks of 8 dimensions between two early-exit tests.  Measured (same-run builds): every 32 dimensions pays at D = 128,
    // K = 8192 (cfg3 launch 4.74 -> 4.10 ms) and is neutral at D = 64; at D = 208 -- the training-step route, where every row
    // of a small batch is scanned and uniform-random data prunes nothing -- six tests per code cost 18 %, so none there.
    // SEEDED (rows listed by the screen, which hands over its best candidate): the candidate's exact square bounds the scan
    // from the first chunk on, so the test pays at every width.  (Tried and dropped: loading code k+1's first 32 dimensions
    // while code k is tested -- 64 more registers beside the 128 of the row: 345 -> 531 us at K = 8192, D = 128.)
    constexpr int G = (DCH > 16 && !SEEDED) ? DCH : 4;
#endif
    for (int k = kb; k < ke; ++k) {
        const float4* c4 = reinterpret_cast<const float4*>(cb + (size_t)k * D);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        bool dead = false;
#pragma unroll
        for (int g = 0; g < DCH; g += G) {
#pragma unroll
            for (int i = g; i < (g + G < DCH ? g + G : DCH); ++i) {
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
#ifndef LQ_NO_EARLY_EXIT
            if (g + G < DCH) {
                const float part = ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7;
                if (part >= best_s || part > s_prune) { dead = true; break; }
            }
#endif
        }
        if (dead) continue;
        const float s = ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7;
        if (s < best_s) {
            const float v = lq_sqrt(s);
            if (v < best_v) { best_v = v; best_s = s; best_k = k; }
        }
    }
}

// The screen's best candidate of a listed row bounds the exact scan: with s* the candidate's exact square (same 8-accumulator
// order), no code whose square exceeds s* (1 + 2^-21) can reach the candidate's square ROOT -- one ulp of the root spans at most
// 2^-22 of the square, so the slack also covers squares that differ from s* but share its root (the first-index rule then still
// sees them).  The bound only prunes; the result is what the unbounded scan finds.
template <int DCH>
__device__ __forceinline__ float lq_seed_bound(const float (&zr)[DCH * 8], const float* __restrict__ cb, int seed) {
    const float4* c4 = reinterpret_cast<const float4*>(cb + (size_t)seed * (DCH * 8));
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
    const float s = ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7;
    const float b = lq_fma(s, 4.76837158203125e-07f, s) + 1.1754944e-38f;      // s (1 + 2^-21), never below s itself
    return (b == b) ? b : INFINITY;                                            // a NaN square prunes nothing
}

// One thread's share of a listed row's exact decision (thread = (row r of the workgroup, slice sl of SL)), by what
// lq_screen_emit (lipvq_screen.h) wrote for the row:
//  * short lists: slice j scores candidate j exactly -- typically two codes instead of K;
//  * lane masks: the codes congruent to a flagged lane mod 32, tiles of 32 codes dealt to the slices;
//  * nothing: codes [sl per, (sl+1) per) are scanned; the screen's best candidate, when there is one, bounds both kinds of scan.
// Returns the slice's (root, code); the caller takes the smallest root, the LOWER CODE among equal roots (torch.argmin's
// first-minimum rule, v5:46).
template <int DCH, int SL, int DIST = LIPVQ_DIST_NORM>
__device__ __forceinline__ void lq_rows_search(const float (&zr)[DCH * 8], const float* __restrict__ cb, int K, int sl, int cslot,
                                               const int* __restrict__ seed_list, const int* __restrict__ cand_list,
                                               size_t cand_cap, float& best_v, int& best_k) {
    constexpr int D = DCH * 8;
    const int per = (K + SL - 1) / SL;
    const int kb = sl * per, ke = (kb + per < K) ? kb + per : K;
    float best_s = INFINITY;
    best_v = INFINITY;
    best_k = kb < K ? kb : K - 1;            // always a valid code, even if every distance is NaN (torch.argmin
                                             // of an all-NaN row is unspecified; an out-of-range index is not an option)
    int n0 = -1, n1 = -1;
    const int* cl = nullptr;
    if (cand_list && (size_t)cslot < cand_cap) {
        cl = cand_list + (size_t)cslot * 16;
        n0 = cl[0]; n1 = cl[8];
    }
    if (n0 >= 0 && n1 >= 0 && n0 + n1 >= 1 && n0 <= LQ_CAND_MAX && n1 <= LQ_CAND_MAX) {
        const int j = sl < n0 + n1 ? sl : 0;                      // idle slices repeat candidate 0
        const int code = j < n0 ? cl[2 + j] : cl[10 + (j - n0)];
        if (code >= 0 && code < K) {                              // (lq_screen_emit lists valid codes only)
            const float4* c4 = reinterpret_cast<const float4*>(cb + (size_t)code * D);
            if constexpr (DIST == LIPVQ_DIST_SQSUM) {
                const float v = lq_sq32_row<DCH>(zr, c4);
                best_k = code;
                best_v = (v == v) ? v : INFINITY;
                return;
            }
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
            const float v = lq_sqrt(((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7);
            best_k = code;
            best_v = (v == v) ? v : INFINITY;                     // a NaN root never wins; the code stays valid
            return;
        }
    }
    float prune = INFINITY;
    if (seed_list) {                         // rows listed by the screen come with its best candidate
        int seed = seed_list[cslot];
        seed = (seed >= 0 && seed < K) ? seed : 0;
        if constexpr (DIST == LIPVQ_DIST_NORM) prune = lq_seed_bound<DCH>(zr, cb, seed);      // (the sum rule scans without early exits)
    }
    if (n0 != -1 && n1 != -1 && cl) {
        // lane masks (at least one part said -2; a part with a short list contributes the lanes of its mask all the same)
        const unsigned lanes = ((unsigned)cl[1] & 0xffffu) | (((unsigned)cl[9] & 0xffffu) << 16);
        const int ntile = (K + 31) / 32;
        for (int t = sl; t < ntile; t += SL) {                    // ascending codes within a slice: first minimum kept
            unsigned m = lanes;
            while (m) {
                const int l = __builtin_ctz(m);
                m &= m - 1;
                const int k = 32 * t + l;
                if (k < K) lq_exact_scan<DCH, true, DIST>(zr, cb, k, k + 1, best_v, best_s, best_k, prune);
            }
        }
        return;
    }
    if (seed_list) lq_exact_scan<DCH, true, DIST>(zr, cb, kb, ke, best_v, best_s, best_k, prune);
    else lq_exact_scan<DCH, false, DIST>(zr, cb, kb, ke, best_v, best_s, best_k);
}

// (root, code) of a row over its SL slices: smallest root, among equal roots the lower code (first-minimum rule).  Thread
// layout tid = sl * 4 + r: the 16 slices a wave holds for row r sit in lanes r, r+4, ..: four xor-shuffles, then the four
// waves' results through LDS.  (The first version let one thread walk all 64 slices: 17 k cycles of dependent LDS reads,
// a third of the workgroup's critical path once the search had shrunk to two candidates.)
__device__ __forceinline__ void lq_rows_reduce(float& bv, int& bk, float (*s_v)[64], int (*s_k)[64], int r, int tid) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int ok = __shfl_xor(bk, off, 64);
        if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
    }
    if ((tid & 63) < 4) { s_v[r][tid >> 6] = bv; s_k[r][tid >> 6] = bk; }
    __syncthreads();
    if (tid < 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ov = s_v[r][q];
            const int ok = s_k[r][q];
            if (q == 0 || ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// exact decision for the listed rows: 4 rows x 64 code slices per workgroup
// ------------------------------------------------------------------------------------------
template <int DCH, int DIST = LIPVQ_DIST_NORM>
__global__ __launch_bounds__(256) void nearest_rows_kernel(
    const float* __restrict__ z, const float* __restrict__ cb, int64_t* __restrict__ idx, float* __restrict__ zq,
    unsigned long long* __restrict__ usage, const int* __restrict__ row_list, const int* __restrict__ row_count,
    int K, int z_by_slot, int count_direct, const int* __restrict__ seed_list, const int* __restrict__ cand_list, size_t cand_cap,
    const int* __restrict__ slot_list) {
    constexpr int D = DCH * 8;
    constexpr int RB = 4, SL = 64;               // rows per workgroup, code slices per row
    __shared__ float s_v[RB][SL];
    __shared__ int s_k[RB][SL];
    const int count = row_count ? *row_count : count_direct;     // row_list == NULL: every row 0..count_direct-1
    const int r = threadIdx.x & (RB - 1), sl = threadIdx.x / RB;
  for (int base = blockIdx.x * RB; base < count; base += gridDim.x * RB) {
    const int slot = base + r;
    const bool valid = slot < count;
    // slot_list (round 3): the slots nearest_lists_kernel left over (lane masks, full scans); `count` is then their number
    const int cslot = slot_list ? slot_list[valid ? slot : count - 1] : (valid ? slot : count - 1);
    const int64_t row = row_list ? row_list[cslot] : cslot;
    float zr[D];
    {
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)(z_by_slot ? (int64_t)cslot : row) * D);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            const float4 v = z4[i];
            zr[4 * i + 0] = v.x; zr[4 * i + 1] = v.y; zr[4 * i + 2] = v.z; zr[4 * i + 3] = v.w;
        }
    }
    float best_v;
    int best_k;
    lq_rows_search<DCH, SL, DIST>(zr, cb, K, sl, cslot, seed_list, cand_list, cand_cap, best_v, best_k);
    lq_rows_reduce(best_v, best_k, s_v, s_k, r, threadIdx.x);
    if (sl == 0) {
        if (valid) idx[row] = (int64_t)best_k;
        s_k[r][0] = best_k;
    }
    if (usage && threadIdx.x < 64) lq_usage_add(usage, (sl == 0 && valid) ? s_k[r][0] : 0, sl == 0 && valid);
    __syncthreads();
    if (zq && valid) {
        const int bk = s_k[r][0];
        const float4* src = reinterpret_cast<const float4*>(cb + (size_t)bk * D);
        float4* dst = reinterpret_cast<float4*>(zq + (size_t)row * D);
        for (int v = sl; v < D / 4; v += SL) dst[v] = src[v];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Listed rows with SHORT candidate lists, at rate (round 3).  nearest_rows_kernel above gives a row to 64 threads that each
// load the whole row and one candidate -- right for the fraction of a percent of rows the three-product screen leaves, 64-fold
// redundant for the 10-40 % the one-product screen leaves (cfg3: 103 k rows, 926 us).  Here ONE WAVE owns a row per iteration and
// its eight 8-lane groups score eight candidates at once: lane j of a group keeps torch's accumulator j (features j, j + 8, ...:
// the k-ordered fma chain of lq_sqdist8; for the sum rule the four accumulators of lane column j, lq_sqdist32), the eight
// partials are added in lane order, roots compared, lower code among equal values.  Lane-mask rows (4 % of the one-product
// screen's rows; popcount(mask) x K/32 candidates) take the same loop while that is at most 128 candidates; longer scans and rows
// with no list at all (n = -1: a meaningless screen) are appended to slot2_list for the scanning kernel.
// ------------------------------------------------------------------------------------------
template <int DCH, int DIST>
__global__ __launch_bounds__(256) void nearest_lists_kernel(
    const float* __restrict__ z, const float* __restrict__ cb, int64_t* __restrict__ idx, float* __restrict__ zq,
    unsigned long long* __restrict__ usage, const int* __restrict__ row_list, const int* __restrict__ row_count,
    int K, int z_by_slot, const int* __restrict__ cand_list, size_t cand_cap, int* __restrict__ slot2_list,
    int* __restrict__ slot2_count, int all_here) {
    constexpr int D = DCH * 8;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int count = *row_count;
    for (int slot = wid; slot < count; slot += nw) {
        const int* cl = (size_t)slot < cand_cap ? cand_list + (size_t)slot * 16 : nullptr;
        const int64_t row = row_list[slot];
        const float* zr = z + (size_t)(z_by_slot ? (int64_t)slot : row) * D;
        const int best_k = lq_lists_row<DCH, DIST>([&](int f) { return zr[f]; }, cb, K, cl, lane, all_here != 0);
        if (best_k < 0) {                                                    // a long scan: the scanning kernel's
            if (lane == 0) slot2_list[atomicAdd(slot2_count, 1)] = slot;
            continue;
        }
        if (lane == 0) {
            idx[row] = (int64_t)best_k;
            if (usage) atomicAdd(&usage[best_k], 1ull);
        }
        if (zq) {
            const float4* src = reinterpret_cast<const float4*>(cb + (size_t)best_k * D);
            float4* dst = reinterpret_cast<float4*>(zq + (size_t)row * D);
            for (int v4 = lane; v4 < D / 4; v4 += 64) dst[v4] = src[v4];
        }
    }
}

// Any latent width (no compile-time D, rows not necessarily 16-byte aligned): the same decision for a listed row -- short lists,
// lane masks, or a full scan -- with both operands read straight from global memory (L2) through lq_sqdist8 / lq_sqdist32, i.e.
// torch's remainder handling included (lipvq_math.h).  The route of widths outside {32, 64, 128, 208}: a few thousand rows at most.
template <int DIST>
__global__ __launch_bounds__(256) void nearest_rows_any_kernel(
    const float* __restrict__ z, const float* __restrict__ cb, int64_t* __restrict__ idx, float* __restrict__ zq,
    unsigned long long* __restrict__ usage, const int* __restrict__ row_list, const int* __restrict__ row_count,
    int K, int D, int z_by_slot, int count_direct, const int* __restrict__ cand_list, size_t cand_cap) {
    constexpr int RB = 4, SL = 64;
    __shared__ float s_v[RB][SL];
    __shared__ int s_k[RB][SL];
    const int count = row_count ? *row_count : count_direct;
    const int r = threadIdx.x & (RB - 1), sl = threadIdx.x / RB;
    auto score = [&](const float* zr, int k) {
        const float* c = cb + (size_t)k * D;
        const float v = (DIST == LIPVQ_DIST_NORM) ? lq_sqrt(lq_sqdist8(zr, c, D)) : lq_sqdist32(zr, c, D);
        return (v == v) ? v : INFINITY;                        // a NaN never wins; the code stays valid
    };
    for (int base = blockIdx.x * RB; base < count; base += gridDim.x * RB) {
        const int slot = base + r;
        const bool valid = slot < count;
        const int cslot = valid ? slot : count - 1;
        const int64_t row = row_list ? row_list[cslot] : cslot;
        const float* zr = z + (size_t)(z_by_slot ? (int64_t)cslot : row) * D;
        const int per = (K + SL - 1) / SL;
        const int kb = sl * per, ke = (kb + per < K) ? kb + per : K;
        float best_v = INFINITY;
        int best_k = kb < K ? kb : K - 1;
        int n0 = -1, n1 = -1;
        const int* cl = nullptr;
        if (cand_list && (size_t)cslot < cand_cap) {
            cl = cand_list + (size_t)cslot * 16;
            n0 = cl[0]; n1 = cl[8];
        }
        if (n0 >= 0 && n1 >= 0 && n0 + n1 >= 1 && n0 <= LQ_CAND_MAX && n1 <= LQ_CAND_MAX) {
            const int j = sl < n0 + n1 ? sl : 0;
            int code = j < n0 ? cl[2 + j] : cl[10 + (j - n0)];
            code = (code >= 0 && code < K) ? code : 0;
            best_k = code;
            best_v = score(zr, code);
        } else if (n0 != -1 && n1 != -1 && cl) {
            const unsigned lanes = ((unsigned)cl[1] & 0xffffu) | (((unsigned)cl[9] & 0xffffu) << 16);
            const int ntile = (K + 31) / 32;
            for (int t = sl; t < ntile; t += SL) {                // ascending codes within a slice: first minimum kept
                unsigned m = lanes;
                while (m) {
                    const int l = __builtin_ctz(m);
                    m &= m - 1;
                    const int k = 32 * t + l;
                    if (k < K) {
                        const float v = score(zr, k);
                        if (v < best_v) { best_v = v; best_k = k; }
                    }
                }
            }
        } else {
            for (int k = kb; k < ke; ++k) {
                const float v = score(zr, k);
                if (v < best_v) { best_v = v; best_k = k; }
            }
        }
        lq_rows_reduce(best_v, best_k, s_v, s_k, r, threadIdx.x);
        if (sl == 0) {
            if (valid) idx[row] = (int64_t)best_k;
            s_k[r][0] = best_k;
        }
        if (usage && threadIdx.x < 64) lq_usage_add(usage, (sl == 0 && valid) ? s_k[r][0] : 0, sl == 0 && valid);
        __syncthreads();
        if (zq && valid) {
            const int bk = s_k[r][0];
            for (int d = sl; d < D; d += SL) zq[(size_t)row * D + d] = cb[(size_t)bk * D + d];
        }
        __syncthreads();
    }
}

// One row per workgroup, 256 code slices: the training-step route (every row of a batch of a few hundred rows is decided
// exactly, lipvq_nearest_rows_f32).  With 4 rows x 64 slices a thread walked K/64 codes one after the other -- 16 dependent
// row fetches at K = 1024: 45 us for 80 rows, as long as both MLP launches of the step together; here it walks K/256.
template <int DCH, int DIST = LIPVQ_DIST_NORM>
__global__ __launch_bounds__(256) void nearest_rows1_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t* __restrict__ idx, float* __restrict__ zq,
                                                            unsigned long long* __restrict__ usage, int count, int K) {
    constexpr int D = DCH * 8;
    constexpr int SL = 256;
    __shared__ float s_v[4];
    __shared__ int s_k[4];
    const int tid = threadIdx.x;
    for (int row = blockIdx.x; row < count; row += gridDim.x) {
        float zr[D];
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            const float4 v = z4[i];
            zr[4 * i + 0] = v.x; zr[4 * i + 1] = v.y; zr[4 * i + 2] = v.z; zr[4 * i + 3] = v.w;
        }
        const int per = (K + SL - 1) / SL;
        const int kb = tid * per, ke = (kb + per < K) ? kb + per : K;
        float bv = INFINITY, bs = INFINITY;
        int bk = kb < K ? kb : K - 1;
        lq_exact_scan<DCH, false, DIST>(zr, cb, kb, ke, bv, bs, bk);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int ok = __shfl_xor(bk, off, 64);
            if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_k[tid >> 6] = bk; }
        __syncthreads();
        bv = s_v[0]; bk = s_k[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (s_v[q] < bv || (s_v[q] == bv && s_k[q] < bk)) { bv = s_v[q]; bk = s_k[q]; }
        if (tid == 0) {
            idx[row] = (int64_t)bk;
            if (usage) atomicAdd(&usage[bk], 1ull);
        }
        if (zq && tid < D / 4)
            reinterpret_cast<float4*>(zq + (size_t)row * D)[tid] = reinterpret_cast<const float4*>(cb + (size_t)bk * D)[tid];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Small batches (the reference's training step and its rollouts: B*T = 1 ... a few hundred rows), every row decided exactly,
// spread over the whole chip (round 3).  nearest_rows1_kernel gives a row to ONE workgroup, whose 256 threads each walk their
// own code rows: every wave-level load touches 64 different lines, the CU's address unit serialises them, and the row's
// 852 KB of codebook (K = 1024, D = 208) pass through one CU's L1 -- 30 us for 80 rows, the largest node of the graphed step.
// Here a workgroup owns 4 rows x 64 codes: the 64 code rows are staged ONCE into LDS with coalesced 16-byte loads (row stride
// D + 1 floats: the column reads of the scoring loop hit 32 different banks), wave w scores row w against them -- lane = code,
// z read as LDS broadcasts, lq_sqdist8 / lq_sqdist32: torch's orders -- and the grid is (row groups) x (code groups):
// 20 x 16 = 320 workgroups at N = 80, K = 1024.  A row's code-group minima meet in one 64-bit atomic per row; the workgroup that
// arrives last at the row group's counter reads the winners, writes idx / z_q, counts the usage and puts keys and counter back to
// zero (the workspace is zero between launches).  Any width that is a multiple of 4 and fits the LDS image.
// ------------------------------------------------------------------------------------------
#define NSM_ROWS 4
#define NSM_CODES 64

static inline size_t nsm_lds_bytes(int D) { return ((size_t)NSM_CODES * (D + 1) + (size_t)NSM_ROWS * D) * sizeof(float); }
static inline int nsm_code_groups(int K) { return (K + NSM_CODES - 1) / NSM_CODES; }
#define NSM_MAX_ROWS 4096
// the counters take a FIXED 4 KB at the head of the workspace (one int per row group of the largest batch): a buffer that served one
// shape serves any other -- no call's partial minima ever lie where another call's counters do
static inline size_t nsm_counter_bytes(int64_t) { return (size_t)(NSM_MAX_ROWS / NSM_ROWS) * sizeof(int); }

// DT: the width at compile time (the scoring loop unrolls: its LDS reads are requested in batches instead of one wait per pair --
// with a runtime width the kernel was latency-bound, 40 us for 80 rows), or 0 for any other width
template <int DIST, int DT>
__global__ __launch_bounds__(256) void nearest_small_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t* __restrict__ idx, float* __restrict__ zq,
                                                            unsigned long long* __restrict__ usage, int count, int K, int D_rt, int CG,
                                                            int* __restrict__ counters, unsigned long long* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) float nsm_lds[];
    __shared__ int s_last;
    const int D = DT ? DT : D_rt;
    const int LD = D + 1;
    float* s_z = nsm_lds;                                     // [NSM_ROWS][D], 16-byte aligned rows (D % 4 == 0)
    float* s_cb = nsm_lds + NSM_ROWS * D;                     // [NSM_CODES][D + 1]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rg = blockIdx.x / CG, cg = blockIdx.x - rg * CG;
    const int D4 = D >> 2;
    // consecutive threads: consecutive 16-byte pieces of consecutive code rows.  With the width known every piece is REQUESTED before
    // the first one is written (a loop of load -> LDS write paid one L2 round trip per pass: 13 at D = 208, most of the kernel)
    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < NSM_ROWS * D4) {
        const int r = tid / D4, q = tid - r * D4;
        int row = rg * NSM_ROWS + r;
        row = row < count ? row : count - 1;
        zv = reinterpret_cast<const float4*>(z + (size_t)row * D)[q];
    }
    if constexpr (DT != 0) {
        constexpr int PIECES = NSM_CODES * (DT / 4), NIT = (PIECES + 255) / 256;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (DT / 4), q = i - c * (DT / 4);
            const int k = cg * NSM_CODES + c;
            v[it] = (i < PIECES && k < K) ? reinterpret_cast<const float4*>(cb + (size_t)k * DT)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (DT / 4), q = i - c * (DT / 4);
            if (i < PIECES) {
                float* dst = s_cb + c * LD + 4 * q;
                dst[0] = v[it].x; dst[1] = v[it].y; dst[2] = v[it].z; dst[3] = v[it].w;
            }
        }
    } else {
        for (int i = tid; i < NSM_CODES * D4; i += 256) {
            const int c = i / D4, q = i - c * D4;
            const int k = cg * NSM_CODES + c;
            const float4 v = k < K ? reinterpret_cast<const float4*>(cb + (size_t)k * D)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            float* dst = s_cb + c * LD + 4 * q;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    if (tid < NSM_ROWS * D4) reinterpret_cast<float4*>(s_z)[tid] = zv;
    __syncthreads();
    const int row = rg * NSM_ROWS + w;
    const bool valid = row < count;
    int bk = cg * NSM_CODES + lane;
    float bv = INFINITY;
    if (bk < K) {
        const float* zr = s_z + w * D;
        const float* c = s_cb + lane * LD;
        float v;
        if constexpr (DIST == LIPVQ_DIST_NORM && DT != 0) {
            // lq_sqdist8's order for a multiple of 8 (accumulator j takes the dimensions j mod 8, then ((((((a0+a1)+a2)+a3)+a4)+a5)+a6)+a7),
            // written out so that it unrolls over the whole width
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
#pragma unroll
            for (int i = 0; i < DT; i += 8) {
                const float4 zl = *reinterpret_cast<const float4*>(zr + i), zh = *reinterpret_cast<const float4*>(zr + i + 4);
                const float d0 = zl.x - c[i + 0], d1 = zl.y - c[i + 1], d2 = zl.z - c[i + 2], d3 = zl.w - c[i + 3];
                const float d4 = zh.x - c[i + 4], d5 = zh.y - c[i + 5], d6 = zh.z - c[i + 6], d7 = zh.w - c[i + 7];
                a0 = lq_fma(d0, d0, a0); a1 = lq_fma(d1, d1, a1); a2 = lq_fma(d2, d2, a2); a3 = lq_fma(d3, d3, a3);
                a4 = lq_fma(d4, d4, a4); a5 = lq_fma(d5, d5, a5); a6 = lq_fma(d6, d6, a6); a7 = lq_fma(d7, d7, a7);
            }
            v = lq_sqrt(((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7);
        } else {
            v = (DIST == LIPVQ_DIST_NORM) ? lq_sqrt(lq_sqdist8(zr, c, D)) : lq_sqdist32(zr, c, D);
        }
        if (v < INFINITY) bv = v;                             // (a NaN or an overflowed distance never wins: the other kernels' `v < best`)
    } else {
        bk = K - 1;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {                  // smallest value, among equal values the lower code
        const float ov = __shfl_xor(bv, off, 64);
        const int ok = __shfl_xor(bk, off, 64);
        if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
    }
    if (CG > 1) {
        // (value, code) as one 64-bit key whose order is the decision rule -- smaller value first, among equal values the lower code
        // (values are non-negative floats: their bit patterns order like the numbers) -- kept COMPLEMENTED, so that "nothing yet" is
        // the zero the workspace rests at.  Device-scope atomics only, no fence: a release/acquire pair at agent scope writes back and
        // invalidates the XCD's L2 on this chip -- with __threadfence() around plain stores this kernel took 35 us at 80 rows and
        // 225 us at 500.  The wave waits for its atomic's return before the barrier, so the key is in place before the count.
        if (lane == 0 && valid) {
            const unsigned long long key = ~(((unsigned long long)__float_as_uint(bv) << 32) | (unsigned int)bk);
            const unsigned long long old = atomicMax(&keys[row], key);
            asm volatile("" ::"v"((unsigned int)old));
        }
        __syncthreads();
        if (tid == 0) {
            const int old = atomicAdd(&counters[rg], 1);
            s_last = old == CG - 1;
        }
        __syncthreads();
        if (!s_last) return;                                  // (workgroup-uniform)
        if (tid == 0) atomicExch(&counters[rg], 0);           // zero between launches
        unsigned long long fin = 0ull;
        if (lane == 0 && valid) fin = ~atomicExch(&keys[row], 0ull);
        bk = __builtin_amdgcn_readfirstlane((int)(unsigned int)(fin & 0xffffffffull));
    }
    if (!valid) return;                                       // (wave-uniform)
    if (lane == 0) {
        idx[row] = (int64_t)bk;
        if (usage) atomicAdd(&usage[bk], 1ull);
    }
    if (zq)
        for (int q = lane; q < D4; q += 64)
            reinterpret_cast<float4*>(zq + (size_t)row * D)[q] = reinterpret_cast<const float4*>(cb + (size_t)bk * D)[q];
}

extern "C" int lipvq_nearest_small_supported(int64_t N, int K, int D) {
    return N >= 1 && N <= NSM_MAX_ROWS && K >= 1 && K <= 65536 && D >= 4 && (D & 3) == 0 && nsm_lds_bytes(D) <= 64 * 1024 ? 1 : 0;
}

extern "C" size_t lipvq_nearest_small_workspace_bytes(int64_t N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return nsm_counter_bytes(N) + (size_t)N * sizeof(unsigned long long);
}

// workspace: lipvq_nearest_small_workspace_bytes(N, K) bytes (4096 of counters + one 64-bit key per row), ZERO on entry; the call
// leaves it zero, so that one zero-filled buffer serves every later call on the same stream, whatever its shape.
extern "C" int lipvq_nearest_small_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage, void* workspace,
                                       int64_t N, int K, int D, int dist, void* stream) {
    if (N < 0 || K <= 0) return fail(LIPVQ_EINVAL, "nearest_small: bad sizes");
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !idx || !workspace) return fail(LIPVQ_EINVAL, "nearest_small: null pointer");
    if (dist != LIPVQ_DIST_NORM && dist != LIPVQ_DIST_SQSUM) return fail(LIPVQ_EINVAL, "nearest_small: unknown distance rule %d", dist);
    if (!lipvq_nearest_small_supported(N, K, D))
        return fail(LIPVQ_EUNSUPPORTED, "nearest_small: N=%lld K=%d D=%d (lipvq_nearest_small_supported)", (long long)N, K, D);
    if ((((uintptr_t)z | (uintptr_t)codebook | (uintptr_t)zq | (uintptr_t)workspace) & 15) != 0)
        return fail(LIPVQ_EINVAL, "nearest_small: z, codebook, zq and workspace must be 16-byte aligned");
    const int CG = nsm_code_groups(K);
    const int RGn = (int)((N + NSM_ROWS - 1) / NSM_ROWS);
    int* counters = (int*)workspace;
    unsigned long long* keys = (unsigned long long*)((unsigned char*)workspace + nsm_counter_bytes(N));
    const size_t lds = nsm_lds_bytes(D);
    auto go = [&](auto kfn) {
        hipLaunchKernelGGL(kfn, dim3((unsigned)(RGn * CG)), dim3(256), lds, (hipStream_t)stream, z, codebook, idx, zq,
                           (unsigned long long*)usage, (int)N, K, D, CG, counters, keys);
    };
#define LQ_NSM(DT_) do { if (dist == LIPVQ_DIST_NORM) go(nearest_small_kernel<LIPVQ_DIST_NORM, DT_>); else go(nearest_small_kernel<LIPVQ_DIST_SQSUM, DT_>); } while (0)
    switch (D) {
        case 32: LQ_NSM(32); break;
        case 64: LQ_NSM(64); break;
        case 128: LQ_NSM(128); break;
        case 208: LQ_NSM(208); break;
        default: LQ_NSM(0); break;
    }
#undef LQ_NSM
    return check_launch("nearest_small");
}

// ------------------------------------------------------------------------------------------
// exact decision for listed rows WITHOUT a stored z_e: the workgroup first recomputes z_e of its 4 rows from x
// (encoder + Lipschitz layer as plain fp32 fmaf chains in natural k order, bias first, odd fan-in padded with one
// zero term -- the canonical arithmetic, hence the same bits the MFMA path produced), then runs the search above.
// This is what lets the fused tokenize launch skip the 134 MB z_e write when the caller does not want z_e.
// Encoder widths are the reference's (A -> 64 -> 128 -> D).
// ------------------------------------------------------------------------------------------
struct RawEncoder { const float *W0, *b0, *W1, *b1, *W2, *b2; };

template <int DCH>
__global__ __launch_bounds__(256) void nearest_rows_encode_kernel(
    const float* __restrict__ x, RawEncoder w, int A, const float* __restrict__ cb, int64_t* __restrict__ idx,
    float* __restrict__ zq, unsigned long long* __restrict__ usage, const int* __restrict__ row_list,
    const int* __restrict__ row_count, int K, const int* __restrict__ seed_list, const int* __restrict__ cand_list, size_t cand_cap) {
    constexpr int D = DCH * 8;
    constexpr int RB = 4, SL = 64;
    __shared__ float s_x[RB][64];
    __shared__ __attribute__((aligned(16))) float s_h0[RB][64];
    __shared__ __attribute__((aligned(16))) float s_h1[RB][128];
    __shared__ __attribute__((aligned(16))) float s_z[RB][D];
    __shared__ float s_v[RB][SL];
    __shared__ int s_k[RB][SL];
    const int count = *row_count;
    const int tid = threadIdx.x;
#ifdef LQ_ROWS_STAMPS        /* diagnostic build: one workgroup prints its cycle stamps */
    long long rst_prev = __builtin_amdgcn_s_memtime(), rst_t[8];
    int rst_n = 0;
#define LQ_RSTAMP(name) do { const long long t_ = __builtin_amdgcn_s_memtime(); if (rst_n < 8) rst_t[rst_n++] = t_ - rst_prev; rst_prev = t_; \
        if (rst_n == 7 && blockIdx.x == 7 && tid == 0) printf("x %lld | L0 %lld | L1 %lld | L2 %lld | search %lld | reduce %lld | zq %lld\n", rst_t[0], rst_t[1], rst_t[2], rst_t[3], rst_t[4], rst_t[5], rst_t[6]); } while (0)
#else
#define LQ_RSTAMP(name) do { } while (0)
#endif
  for (int base = blockIdx.x * RB; base < count; base += gridDim.x * RB) {
    // ---- encoder for the 4 rows -------------------------------------------------------------------------------
    // thread j owns output j of layers 1 and 2 for all four rows; its weight rows are fetched whole (16-byte loads) and FIRST:
    // they do not depend on x, so their L2 round trip overlaps the row-list -> x chain.  (The first version walked each row with
    // dependent 4-byte loads, 2 (r, j) pairs per thread: ~30 us of round trips that were the whole launch once the search
    // shrank to two candidates per row.)
    const bool al16 = ((((uintptr_t)w.W1) | ((uintptr_t)w.W2)) & 15) == 0;          // uniform
    auto load4 = [&](const float* p) {
        if (al16) return *reinterpret_cast<const float4*>(p);
        return make_float4(p[0], p[1], p[2], p[3]);
    };
    float4 wv[16], wa[16], wb[16];                 // W1 row (64), the two halves of the W2 row (128)
    if (tid < 128) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wv[i] = load4(w.W1 + (size_t)tid * 64 + 4 * i);
    }
    if (tid < D) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wa[i] = load4(w.W2 + (size_t)tid * 128 + 4 * i);
#pragma unroll
        for (int i = 0; i < 16; ++i) wb[i] = load4(w.W2 + (size_t)tid * 128 + 64 + 4 * i);
    }
    for (int o = tid; o < RB * A; o += 256) {
        const int r = o / A, k = o - r * A;
        const int sl_ = base + r < count ? base + r : count - 1;
        s_x[r][k] = x[(size_t)row_list[sl_] * A + k];
    }
    __syncthreads();
    LQ_RSTAMP("x staged");
    {
        const int r = tid >> 6, j = tid & 63;
        float acc = w.b0[j];
        for (int k = 0; k < A; ++k) acc = lq_fma(s_x[r][k], w.W0[(size_t)j * A + k], acc);
        if (A & 1) acc = lq_fma(0.0f, 0.0f, acc);
        s_h0[r][j] = lq_gelu(acc);
    }
    __syncthreads();
    LQ_RSTAMP("layer0");
    if (tid < 128) {
        const int j = tid;
        float acc[RB];
        const float bj = w.b1[j];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = bj;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float4 hv = *reinterpret_cast<const float4*>(&s_h0[r][4 * i]);       // (broadcast read)
                acc[r] = lq_fma(hv.x, wv[i].x, acc[r]);
                acc[r] = lq_fma(hv.y, wv[i].y, acc[r]);
                acc[r] = lq_fma(hv.z, wv[i].z, acc[r]);
                acc[r] = lq_fma(hv.w, wv[i].w, acc[r]);
            }
#pragma unroll
        for (int r = 0; r < RB; ++r) s_h1[r][j] = lq_gelu(acc[r]);
    }
    __syncthreads();
    LQ_RSTAMP("layer1");
    if (tid < D) {
        const int j = tid;
        float acc[RB];
        const float bj = w.b2[j];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = bj;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float4 hv = *reinterpret_cast<const float4*>(&s_h1[r][4 * i]);
                acc[r] = lq_fma(hv.x, wa[i].x, acc[r]);
                acc[r] = lq_fma(hv.y, wa[i].y, acc[r]);
                acc[r] = lq_fma(hv.z, wa[i].z, acc[r]);
                acc[r] = lq_fma(hv.w, wa[i].w, acc[r]);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float4 hv = *reinterpret_cast<const float4*>(&s_h1[r][64 + 4 * i]);
                acc[r] = lq_fma(hv.x, wb[i].x, acc[r]);
                acc[r] = lq_fma(hv.y, wb[i].y, acc[r]);
                acc[r] = lq_fma(hv.z, wb[i].z, acc[r]);
                acc[r] = lq_fma(hv.w, wb[i].w, acc[r]);
            }
#pragma unroll
        for (int r = 0; r < RB; ++r) s_z[r][j] = lq_sigmoid(acc[r]);
    }
    __syncthreads();
    LQ_RSTAMP("layer2");
    // ---- exact search (same as nearest_rows_kernel, z from LDS) ---------------------------------------------------
    const int r = tid & (RB - 1), sl = tid / RB;
    const int slot = base + r;
    const bool valid = slot < count;
    const int64_t row = row_list[valid ? slot : count - 1];
    float zr[D];
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
        const float4 v = reinterpret_cast<const float4*>(&s_z[r][0])[i];
        zr[4 * i + 0] = v.x; zr[4 * i + 1] = v.y; zr[4 * i + 2] = v.z; zr[4 * i + 3] = v.w;
    }
    float best_v;
    int best_k;
    lq_rows_search<DCH, SL>(zr, cb, K, sl, valid ? slot : count - 1, seed_list, cand_list, cand_cap, best_v, best_k);
    LQ_RSTAMP("search");
    lq_rows_reduce(best_v, best_k, s_v, s_k, r, tid);
    if (sl == 0) {
        if (valid) idx[row] = (int64_t)best_k;
        s_k[r][0] = best_k;
    }
    if (usage && tid < 64) lq_usage_add(usage, (sl == 0 && valid) ? s_k[r][0] : 0, sl == 0 && valid);
    __syncthreads();
    LQ_RSTAMP("reduce+idx");
    if (zq && valid) {
        const int bk = s_k[r][0];
        const float4* src = reinterpret_cast<const float4*>(cb + (size_t)bk * D);
        float4* dst = reinterpret_cast<float4*>(zq + (size_t)row * D);
        for (int v = sl; v < D / 4; v += SL) dst[v] = src[v];
    }
    __syncthreads();
    LQ_RSTAMP("zq");
  }
}

int lipvq_launch_rows_encode(const float* x, const float* const* raw6, int A, const float* cb, int64_t* idx, float* zq,
                             int64_t* usage, const int* amb_list, const int* amb_count, int64_t N, int K, int D,
                             hipStream_t st) {
    const int* amb_seed = amb_list + lq_list_ints(N);
    RawEncoder w{raw6[0], raw6[1], raw6[2], raw6[3], raw6[4], raw6[5]};
    int64_t blocks = (N + 3) / 4;
    static int grid_cap = -1;                   // LIPVQ_ROWS_GRID: measurement knob
    if (grid_cap < 0) { const char* e = lq_knob("LIPVQ_ROWS_GRID"); grid_cap = e ? atoi(e) : 1024; }      // (the count lives on the device; the grid strides)
    if (blocks > grid_cap) blocks = grid_cap;
    auto go = [&](auto kfn) {
        hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), 0, st, x, w, A, cb, idx, zq,
                           (unsigned long long*)usage, amb_list, lq_ws_listed(amb_count), K, amb_seed, amb_list + 2 * lq_list_ints(N),
                           lq_cand_cap(N));
    };
    switch (D) {
        case 32: go(nearest_rows_encode_kernel<4>); break;
        case 64: go(nearest_rows_encode_kernel<8>); break;
        case 128: go(nearest_rows_encode_kernel<16>); break;
        case 208: go(nearest_rows_encode_kernel<26>); break;
        default: return fail(LIPVQ_EUNSUPPORTED, "nearest_rows_encode: D=%d has no instance", D);
    }
    return check_launch("nearest_rows_encode");
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern "C" size_t lipvq_nearest_workspace_bytes(int64_t N) {
    if (N <= 0) return 0;
    return 64 + lq_lists_bytes(N);    // [0] uncertified-row counter, then the row list, the best-candidate list and the short lists
}

// One-product ("coarse") or three-product screen.  LIPVQ_SCREEN_MODE=coarse|fine (read per launch; measurement knob --
// identical results): the default is the shape's measured winner (lq_screen_coarse_default).
int lq_screen_coarse(int S, int K) {
    const char* e = lq_knob("LIPVQ_SCREEN_MODE");
    if (e && !strcmp(e, "coarse")) return 1;
    if (e && !strcmp(e, "fine")) return 0;
    return lq_screen_coarse_default(S, K);
}

// 1 if the screened routes (lipvq_nearest_screened_f32, lipvq_vq_nearest_screened_f32, lipvq_tokenize_f32) would run the
// one-product screen for this shape right now (callers that budget the exact stage -- the host-side screen monitor, the bench's
// roofline -- ask; results do not depend on it)
extern "C" int lipvq_screen_is_coarse(int K, int D) {
    return (K > 0 && lq_screen_S(D)) ? lq_screen_coarse(lq_screen_S(D), K) : 0;
}

template <int S>
static int launch_screen(const float* z, const unsigned char* prep, const float* cb, int64_t* idx, float* zq,
                         int64_t* usage, int* amb_list, int* amb_count, float* dbg, int64_t N, int K, int D,
                         float gamma, hipStream_t st) {
    using SC = StandaloneScreen<S>;
    size_t lds = lq_ring_bytes<S, SC::TC, SC::NB>();
    if (lds < (size_t)SCREEN_WAVES * LQ_DECIDE_BYTES) lds = (size_t)SCREEN_WAVES * LQ_DECIDE_BYTES;      // per-wave transpose slices reuse the stages
    const int64_t rows_per_block = SCREEN_WAVES * 32;
    unsigned blocks = (unsigned)((N + rows_per_block - 1) / rows_per_block);
    // the debug hook takes the arithmetic from the sign of its gamma (negative: the one-product chain, bound factor |gamma|)
    const bool coarse = dbg ? (gamma < 0.0f) : (lq_screen_coarse(S, K) != 0);
    if (gamma < 0.0f) gamma = -gamma;
    auto kfn = dbg ? (coarse ? screen_kernel<S, true, true> : screen_kernel<S, true, false>)
                   : (coarse ? screen_kernel<S, false, true> : screen_kernel<S, false, false>);
    static LqLdsReserve reserved[4];            // per instantiation and kernel flavour: per-device, thread-safe (lipvq_common.h)
    if (lds > 64 * 1024)
        if (int rc = lipvq_reserve_lds(reserved[(dbg ? 1 : 0) + (coarse ? 2 : 0)], (const void*)kfn, lds, "screen")) return rc;
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(SCREEN_WAVES * 64), lds, st, z, prep, cb, idx, zq,
                       (unsigned long long*)usage, amb_list, amb_count, dbg, N, K, D, gamma);
    return check_launch("screen");
}

template <int DCH, int DIST = LIPVQ_DIST_NORM>
static int launch_rows_t(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                         const int* amb_list, const int* amb_count, int64_t N, int K, hipStream_t st) {
    if (!amb_list && N <= 4096 && K >= 512) {                // every row of a small batch: one row per workgroup
        hipLaunchKernelGGL((nearest_rows1_kernel<DCH, DIST>), dim3((unsigned)N), dim3(256), 0, st, z, cb, idx, zq,
                           (unsigned long long*)usage, (int)N, K);
        return check_launch("nearest_rows1");
    }
    // the count lives on the device: a bounded grid strides over however many rows were listed
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    // counts on the device (lipvq_screen.h, workspace header): the screening launch PUBLISHED the number of listed rows before it
    // ended (lq_ws_publish); the list kernel appends what it leaves to the slot-2 list behind the header's slot-2 counter
    const int* listed = amb_list ? lq_ws_listed(amb_count) : nullptr;
    if (amb_list) {
        // rows with short candidate lists (95 % and more of the listed rows): one wave per row, eight candidates at a time;
        // what it leaves (lane masks, no list) goes through slot2 to the scanning kernel
        int* slot2_list = const_cast<int*>(amb_list) + lq_slot2_offset_ints(N);
        int* slot2_count = const_cast<int*>(amb_count) + LQ_WS_SLOT2;       // zeroed by the screening launch's first workgroup
        int64_t lb = (N + 3) / 4;
        if (lb > 2048) lb = 2048;
        const int all_here = K <= LQ_LISTS_ALL_K ? 1 : 0;
        hipLaunchKernelGGL((nearest_lists_kernel<DCH, DIST>), dim3((unsigned)lb), dim3(256), 0, st, z, cb, idx, zq,
                           (unsigned long long*)usage, amb_list, listed, K, z_by_slot, amb_list + 2 * lq_list_ints(N),
                           lq_cand_cap(N), slot2_list, slot2_count, all_here);
        if (int rc = check_launch("nearest_lists")) return rc;
        if (all_here) return LIPVQ_OK;                                       // nothing was left to the scanning kernel
        hipLaunchKernelGGL((nearest_rows_kernel<DCH, DIST>), dim3((unsigned)blocks), dim3(256), 0, st, z, cb, idx, zq,
                           (unsigned long long*)usage, amb_list, slot2_count, K, z_by_slot, 0, amb_list + lq_list_ints(N),
                           amb_list + 2 * lq_list_ints(N), lq_cand_cap(N), slot2_list);
        return check_launch("nearest_rows");
    }
    hipLaunchKernelGGL((nearest_rows_kernel<DCH, DIST>), dim3((unsigned)blocks), dim3(256), 0, st, z, cb, idx, zq,
                       (unsigned long long*)usage, nullptr, nullptr, K, z_by_slot, (int)N, nullptr, nullptr, (size_t)0, nullptr);
    return check_launch("nearest_rows");
}

template <int DIST>
static int launch_rows_any(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                           const int* amb_list, const int* amb_count, int64_t N, int K, int D, hipStream_t st) {
    if (D <= 0) return fail(LIPVQ_EINVAL, "nearest_rows: D=%d", D);
    int64_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((nearest_rows_any_kernel<DIST>), dim3((unsigned)blocks), dim3(256), 0, st, z, cb, idx, zq,
                       (unsigned long long*)usage, amb_list, amb_list ? lq_ws_listed(amb_count) : nullptr, K, D, z_by_slot,
                       amb_list ? 0 : (int)N, amb_list ? amb_list + 2 * lq_list_ints(N) : nullptr,
                       amb_list ? lq_cand_cap(N) : (size_t)0);
    return check_launch("nearest_rows_any");
}

int lipvq_launch_rows(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                      const int* amb_list, const int* amb_count, int64_t N, int K, int D, hipStream_t st, int dist) {
    if (dist == LIPVQ_DIST_SQSUM) {
        switch (D) {
            case 32: return launch_rows_t<4, LIPVQ_DIST_SQSUM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
            case 64: return launch_rows_t<8, LIPVQ_DIST_SQSUM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
            case 128: return launch_rows_t<16, LIPVQ_DIST_SQSUM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
            case 208: return launch_rows_t<26, LIPVQ_DIST_SQSUM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
            default: return launch_rows_any<LIPVQ_DIST_SQSUM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, D, st);
        }
    }
    switch (D) {
        case 32: return launch_rows_t<4>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
        case 64: return launch_rows_t<8>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
        case 128: return launch_rows_t<16>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
        case 208: return launch_rows_t<26>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, st);
        default: return launch_rows_any<LIPVQ_DIST_NORM>(z, z_by_slot, cb, idx, zq, usage, amb_list, amb_count, N, K, D, st);
    }
}

static int screened_impl(const float* z, const float* cb, const void* prep, int64_t* idx, float* zq,
                         int64_t* usage, void* workspace, float* dbg, int64_t N, int K, int D, float gamma,
                         hipStream_t st, int dist = LIPVQ_DIST_NORM) {
    int* amb_count = (int*)workspace + LQ_WS_LIVE;          // the header's live counters (lipvq_screen.h); [0] = the reported count
    int* amb_list = (int*)((unsigned char*)workspace + 64);
    // (this entry point takes a workspace in any state, hence the fill; the fused launches keep theirs clean: lq_ws_finish)
    hipError_t e = hipMemsetAsync(workspace, 0, 64, st);
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "nearest_screened: %s", hipGetErrorString(e));
    const unsigned char* p = (const unsigned char*)prep;
    int rc;
    switch (lq_screen_S(D)) {                   // widths between the instances run the next larger one on zero-padded columns
        case 2: rc = launch_screen<2>(z, p, cb, idx, zq, usage, amb_list, amb_count, dbg, N, K, D, gamma, st); break;
        case 4: rc = launch_screen<4>(z, p, cb, idx, zq, usage, amb_list, amb_count, dbg, N, K, D, gamma, st); break;
        case 8: rc = launch_screen<8>(z, p, cb, idx, zq, usage, amb_list, amb_count, dbg, N, K, D, gamma, st); break;
        case 13: rc = launch_screen<13>(z, p, cb, idx, zq, usage, amb_list, amb_count, dbg, N, K, D, gamma, st); break;
        default: return fail(LIPVQ_EUNSUPPORTED, "nearest_screened: D=%d has no screening instance (1 ... 208)", D);
    }
    if (rc) return rc;
    return lipvq_launch_rows(z, 0, cb, idx, zq, usage, amb_list, amb_count, N, K, D, st, dist);
}

extern "C" int lipvq_nearest_screened_supported(int K, int D) {
    return (K > 0 && lq_screen_S(D) != 0) ? 1 : 0;            // any width 1 ... 208
}

// Exact decision of EVERY row by the re-scoring kernel (4 rows x 64 code slices per workgroup): no codebook
// preparation, no screening.  The cheapest route for training-step batches, where the codebook changes every step
// (the preparation alone costs ~110 us) and a screen launch has a ~75 us floor: 13 us at N = 80, D = 208, K = 1024.
extern "C" int lipvq_nearest_rows_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage,
                                      int64_t N, int K, int D, void* stream) {
    if (N < 0 || K <= 0) return fail(LIPVQ_EINVAL, "nearest_rows: bad sizes");
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !idx) return fail(LIPVQ_EINVAL, "nearest_rows: null pointer");
    if (N > 0x7fffffffLL) return fail(LIPVQ_EUNSUPPORTED, "nearest_rows: N too large");
    return lipvq_launch_rows(z, 0, codebook, idx, zq, usage, nullptr, nullptr, N, K, D, (hipStream_t)stream);
}

// Same contract as lipvq_nearest_f32(.., LIPVQ_DIST_NORM) -- identical idx / zq / usage -- through the
// screening fast path.  prep: lipvq_nearest_prepare_f32 of THIS codebook; workspace:
// lipvq_nearest_workspace_bytes(N).  After the call workspace[0] (int) holds the number of rows
// that needed the exact kernel.
extern "C" int lipvq_nearest_screened_f32(const float* z, const float* codebook, const void* prep, int64_t* idx,
                                          float* zq, int64_t* usage, void* workspace, int64_t N, int K, int D,
                                          void* stream) {
    if (N < 0 || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "nearest_screened: bad sizes");
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !prep || !idx || !workspace) return fail(LIPVQ_EINVAL, "nearest_screened: null pointer");
    if (N > 2147483647LL) return fail(LIPVQ_EUNSUPPORTED, "nearest_screened: N too large");
    if (!(D & 3) && (((uintptr_t)z | (uintptr_t)codebook | (uintptr_t)zq) & 15) != 0)       // (the kernels take float4 paths iff D % 4 == 0)
        return fail(LIPVQ_EINVAL, "nearest_screened: z, codebook and zq must be 16-byte aligned");
    return screened_impl(z, codebook, prep, idx, zq, usage, workspace, nullptr, N, K, D, LIPVQ_SCREEN_GAMMA,
                         (hipStream_t)stream);
}

// The plain VQVAE's rule (vq:57-63) through the same two routes (include/lipvq.h).
extern "C" int lipvq_vq_nearest_screened_f32(const float* z, const float* codebook, const void* prep, int64_t* idx,
                                             float* zq, int64_t* usage, void* workspace, int64_t N, int K, int D,
                                             void* stream) {
    if (N < 0 || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "vq_nearest_screened: bad sizes");
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !prep || !idx || !workspace) return fail(LIPVQ_EINVAL, "vq_nearest_screened: null pointer");
    if (N > 2147483647LL) return fail(LIPVQ_EUNSUPPORTED, "vq_nearest_screened: N too large");
    if (!(D & 3) && (((uintptr_t)z | (uintptr_t)codebook | (uintptr_t)zq) & 15) != 0)
        return fail(LIPVQ_EINVAL, "vq_nearest_screened: z, codebook and zq must be 16-byte aligned");
    return screened_impl(z, codebook, prep, idx, zq, usage, workspace, nullptr, N, K, D, LIPVQ_SCREEN_GAMMA,
                         (hipStream_t)stream, LIPVQ_DIST_SQSUM);
}

extern "C" int lipvq_vq_nearest_rows_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage,
                                         int64_t N, int K, int D, void* stream) {
    if (N < 0 || K <= 0) return fail(LIPVQ_EINVAL, "vq_nearest_rows: bad sizes");
    if (N == 0) return LIPVQ_OK;
    if (!z || !codebook || !idx) return fail(LIPVQ_EINVAL, "vq_nearest_rows: null pointer");
    if (N > 0x7fffffffLL) return fail(LIPVQ_EUNSUPPORTED, "vq_nearest_rows: N too large");
    return lipvq_launch_rows(z, 0, codebook, idx, zq, usage, nullptr, nullptr, N, K, D, (hipStream_t)stream, LIPVQ_DIST_SQSUM);
}

// Test hook: also writes the approximate distances d~ [N][Kpad] and uses the caller's gamma
// (gamma = 0 sends every row to the exact kernel; a huge gamma certifies nothing either).
extern "C" int lipvq_screen_debug_f32(const float* z, const float* codebook, const void* prep, int64_t* idx,
                                      float* zq, int64_t* usage, void* workspace, float* dtilde, float gamma,
                                      int64_t N, int K, int D, void* stream) {
    if (!z || !codebook || !prep || !idx || !workspace || N <= 0) return fail(LIPVQ_EINVAL, "screen_debug: bad argument");
    if (!lipvq_nearest_screened_supported(K, D)) return fail(LIPVQ_EUNSUPPORTED, "screen_debug: unsupported D");
    return screened_impl(z, codebook, prep, idx, zq, usage, workspace, dtilde, N, K, D, gamma, (hipStream_t)stream);
}
