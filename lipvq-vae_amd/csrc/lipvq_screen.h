// lipvq_screen.h -- layout of the prepared codebook and constants shared by the screening kernels
// (lipvq_screen.hip, lipvq_fused.hip).  Design notes: lipvq_screen.hip.
#ifndef LIPVQ_SCREEN_H_
#define LIPVQ_SCREEN_H_
#include <hip/hip_fp16.h>

#include "lipvq_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LIPVQ_SCREEN_GAMMA 3.814697265625e-06f   /* 2^-18 */
#ifndef SCREEN_WAVES
#define SCREEN_WAVES 8
#endif

struct PrepLayout {
    int S, Dpad, Kpad, ntiles;
    size_t o_hdr, o_mu, o_tiles, tile_bytes, total;
};

__host__ __device__ static inline PrepLayout prep_layout(int K, int D) {
    PrepLayout L;
    L.S = (D + 15) / 16;
    L.Dpad = L.S * 16;
    L.Kpad = ((K + 31) / 32) * 32;
    L.ntiles = ((L.Kpad / 32 + 7) / 8) * 8;   // whole LDS stages: pad tiles carry e2 = +inf, zero fragments
    L.o_hdr = 0;                       // 16 words: [0] E2max bits, [1] Emax^2 bits, [2] max|2e'| bits, [3] se (int)
    L.o_mu = 64;
    L.o_tiles = L.o_mu + sizeof(float) * (size_t)L.Dpad;
    L.o_tiles = (L.o_tiles + 255) & ~(size_t)255;
    L.tile_bytes = (size_t)L.S * 2048 + 128;    // S steps x {hi,lo} x 32 codes x 2 halves x 16 B, then 32 x e2
    L.total = L.o_tiles + (size_t)L.ntiles * L.tile_bytes;
    return L;
}


// ------------------------------------------------------------------------------------------
// The screening main loop, shared by screen_kernel (lipvq_screen.hip) and tokenize_kernel
// (lipvq_fused.hip).  NT threads stream the prepared codebook through two LDS stage buffers of TC
// column tiles; each wave multiplies its 32 rows (fp16 hi/lo A fragments ah/al) against every
// tile and keeps, per accumulator register (= row) and lane (= code mod 32), the smallest d~, its
// code, and the second smallest d~.
//
// Measured (same-box A/B builds, scripts/ablate.sh; LQ_ABL_* macros below are those timing-only builds):
//  * the bare LDS-read + MFMA chain takes the same ~0.17 ms (cfg2) at 1, 2 or 4 waves per SIMD, with or without
//    reading B fragments a tile ahead: it is the matrix pipe at ~1.2 PF/s executed (MFMA-dense clocks);
//  * variants that did NOT pay: book-keeping tile t-1 between the MFMAs of tile t in the same wave (VALU issue
//    delays the dependent chain; the SIMD's other wave already fills those slots), deferring half the waves by
//    one tile (helped with register-staged copies, hurt once staging became DMA), sched_group_barrier, 3-4
//    waves per SIMD in the fused kernel (spills).
// ------------------------------------------------------------------------------------------
// Block floating point for the fp16 split (tests/test_gpu_screen.py::test_any_magnitude):  x = hi + lo carries 22
// significant bits only while lo is a NORMAL fp16 number, i.e. |x| >= 2^-3.  So operands are multiplied by exact
// powers of two before the split: the codebook by 2^se (global, max |-2e'| lands in [2^13, 2^14)), every latent row
// by its own 2^sz (row max in [2^13, 2^14)).  Elements below 2^-17 of their row/codebook maximum then lose bits,
// which is 2^-39 of that maximum -- negligible.  The MFMA result is in units of 2^(sz+se); since the argmin only
// compares values of ONE row, the bookkeeping simply runs in those units (|e'|^2 is scaled on accumulator
// initialisation) and the certification threshold is scaled the same way.
__device__ __forceinline__ int lq_scale_exp(float maxabs) {
    // 2^k with maxabs * 2^k in [2^13, 2^14); clamped so that the factor stays an ordinary float
    const int e = (int)((__float_as_uint(maxabs) >> 23) & 0xff) - 127;       // floor(log2) for normal numbers
    int k = 13 - e;
    k = k < -60 ? -60 : k;
    k = k > 60 ? 60 : k;
    return (maxabs > 0.0f && maxabs < INFINITY) ? k : 0;
}
__device__ __forceinline__ float lq_pow2f(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }   // k in [-126, 127]

#ifdef LQ_OPT_TC
constexpr int screen_default_tc(int S) { return (S <= 4) ? LQ_OPT_TC : (S <= 8) ? 2 : 1; }
#else
constexpr int screen_default_tc(int S) { return (S <= 2) ? 8 : (S <= 4) ? 4 : (S <= 8) ? 2 : 1; }
#endif

template <int S, int TC_ = screen_default_tc(S)>
struct ScreenCfg {
    static constexpr int TILE_BYTES = S * 2048 + 128;
    static constexpr int TC = TC_;                                              // tiles per stage (divides 8)
    static constexpr int STAGE_BYTES = TC * TILE_BYTES;
    static constexpr int STAGE_VEC = STAGE_BYTES / 16;
};

__device__ __forceinline__ void lq_track(const f32x16& acc, int code, float (&m1)[16], float (&m2)[16], int (&k1)[16]) {
#ifdef LQ_ABL_NOTRACK
    m1[0] = fminf(m1[0], acc[0] + acc[5] + acc[10] + acc[15]); return;   // keeps the MFMAs alive
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        const bool lt = v < m1[r];              // one compare feeds both selects (fminf would cost two
        k1[r] = lt ? code : k1[r];              // NaN-canonicalising v_max as well)
        m2[r] = __builtin_amdgcn_fmed3f(v, m1[r], m2[r]);
        m1[r] = lt ? v : m1[r];
    }
}

template <int S, int NT, int TC_ = screen_default_tc(S)>
__device__ __forceinline__ void lq_screen_core(const f16x8 (&ah)[S], const f16x8 (&al)[S],
                                               const unsigned char* __restrict__ tiles, int ntiles,
                                               unsigned char* stage0, int tid, const float (&frow)[16],
                                               float (&m1)[16], float (&m2)[16], int (&k1)[16]) {
    using C = ScreenCfg<S, TC_>;
    constexpr int VPT = (C::STAGE_VEC + NT - 1) / NT;
    const int lane = tid & 63, ln = lane & 31;
#ifdef LQ_ABL_NOLOOP
    const int nstage = 0;            // ablation build only (scripts/ablate.sh)
#else
    const int nstage = ntiles / C::TC;
#endif
    // Stage copies go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write
    // issue; ablation: the register-staged copy cost 84 us of a 440 us launch).  One wave-instruction
    // moves 64 x 16 B to a wave-uniform LDS base + lane*16, which is exactly this linear copy.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto stage_dma = [&](int st, int buf) {
        const unsigned char* src = tiles + (size_t)st * C::STAGE_BYTES;
        unsigned char* dst = stage0 + (size_t)buf * C::STAGE_BYTES;
        const int wbase = tid & ~63;                                     // first thread of this wave
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int i = tid + v * NT;
            if (i < C::STAGE_VEC)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)i * 16),
                                                 (lds_ptr_t)(dst + (size_t)(wbase + v * NT) * 16), 16, 0, 0);
        }
    };
    __syncthreads();                              // earlier readers of the stage buffers are done
    stage_dma(0, 0);
    __syncthreads();                              // (drains the DMA: hipcc waits vmcnt(0) before the barrier)
    for (int st = 0; st < nstage; ++st) {
#ifndef LQ_ABL_NOSTAGE
        if (st + 1 < nstage) stage_dma(st + 1, (st + 1) & 1);     // its last readers passed the previous barrier
#endif
#ifdef LQ_ABL_NOSTAGE
        const unsigned char* sb = stage0;
#else
        const unsigned char* sb = stage0 + (size_t)(st & 1) * C::STAGE_BYTES;
#endif
#pragma unroll
        for (int c = 0; c < C::TC; ++c) {
            const unsigned char* tb = sb + (size_t)c * C::TILE_BYTES;
            const int code = (st * C::TC + c) * 32 + ln;
            const float e2 = reinterpret_cast<const float*>(tb + S * 2048)[ln];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = e2 * frow[r];      // |e'|^2 in the units of row (r, h)
#pragma unroll
            for (int s = 0; s < S; ++s) {
#ifdef LQ_ABL_NOLDSB
                const f16x8 bh = ah[(s + 1) % S], bl = al[(s + 1) % S];      // ablation only: no LDS fragment reads (wrong results)
#else
                const f16x8 bh = *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 0) * 64 + lane) * 16);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + 1) * 64 + lane) * 16);
#endif
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc, 0, 0, 0);
            }
            lq_track(acc, code, m1, m2, k1);
        }
#ifndef LQ_ABL_NOBARRIER
        __syncthreads();
#endif
    }
}

// frow[r] = factor of row (r, h) = the row this lane's accumulator register r belongs to, fetched from the lane that
// owns that row (row i lives in lanes i and i + 32, both hold fown)
__device__ __forceinline__ void lq_row_factors(float fown, int lane, float (&frow)[16]) {
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) frow[r] = __shfl(fown, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
}

// After the last tile: per row, the global (smallest, its code, second smallest) over the 32 lanes of the
// half-wave that holds the row -- and the certification decision.
//
// The 16 x 3 per-lane values are TRANSPOSED through LDS (each wave uses a 4 KiB slice of the stage buffers,
// which are idle between the loop's last barrier and the next block's first DMA): value of (row i, lane l) goes
// to [i][l]; then lane L (row L & 31, part L >> 5) reads the 16 entries [row][16*part ..] with four 16-byte
// reads, reduces them in registers, and one xor-32 shuffle joins the two parts.  ~170 instructions per
// 32-row tile instead of ~720 for a 5-step shuffle butterfly over 16 registers x 3 values (ablation: the
// butterfly + its LDS hand-off cost 46 us of a 350 us launch), and no workgroup barrier.
// Returns certified (valid in every lane, duplicated across the halves); my_k = the row's code.
__device__ __forceinline__ bool lq_screen_decide(const float (&m1)[16], const float (&m2)[16], const int (&k1)[16],
                                                 unsigned char* wave_lds /* 4 KiB, this wave only */,
                                                 const unsigned* hdr, float n2, float fown, float gamma, int K, int D,
                                                 int lane, int& my_k) {
    const int ln = lane & 31, h = lane >> 5;
    float* tv = reinterpret_cast<float*>(wave_lds);           // [32 rows][32 lanes], reused by the three passes
    // ---- pass 1: m1 -> best value, its position among my 16 entries, second smallest m1 ----------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) tv[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + ln] = m1[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    float best = INFINITY, second = INFINITY;
    int pos = 0;
    {
        const float4* pv = reinterpret_cast<const float4*>(tv + ln * 32 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v4 = pv[q];
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool take = vv[e] < best;       // equal minima need no tie-break: second == best then, the
                second = __builtin_amdgcn_fmed3f(vv[e], best, second);   // row is not certified and the exact
                pos = take ? 4 * q + e : pos;                             // kernel decides it
                best = take ? vv[e] : best;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                          // pass-1 reads are issued before pass 2 overwrites
    // ---- pass 2: k1 -> the code at that position ------------------------------------------------------------
    int* tk = reinterpret_cast<int*>(wave_lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) tk[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + ln] = k1[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int bk = tk[ln * 32 + 16 * h + pos];
    __builtin_amdgcn_wave_barrier();
    // ---- pass 3: m2 -----------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) tv[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + ln] = m2[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    {
        const float4* pv = reinterpret_cast<const float4*>(tv + ln * 32 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v4 = pv[q];
            second = fminf(second, fminf(fminf(v4.x, v4.y), fminf(v4.z, v4.w)));
        }
    }
    // ---- join the two 16-lane parts of the row ---------------------------------------------------------------
    {
        const float ob = __shfl_xor(best, 32, 64);
        const float os = __shfl_xor(second, 32, 64);
        const int ok = __shfl_xor(bk, 32, 64);
        second = fminf(fminf(second, os), fmaxf(best, ob));
        const bool take = ob < best;
        bk = take ? ok : bk;
        best = take ? ob : best;
    }
    __builtin_amdgcn_wave_barrier();
    my_k = bk;
    const float E2max = __uint_as_float(hdr[0]);
    const float Emax = lq_sqrt(__uint_as_float(hdr[1]));
    const float twoemax = __uint_as_float(hdr[2]);
    const float cross = 2.0f * lq_sqrt(n2) * Emax;
    // Certification margin, in this row's units (fown = 2^(sz+se); best/second are d~ * fown).  With S_k = |z - e_k|^2 in
    // real arithmetic and d~_k = S_k - |z'|^2 +- eps_s:
    //  (1) screening error: |d~ - d| <= eps_s = gamma (E2max + 2 |z'| Emax)                       (lipvq_screen.hip, "Error bound")
    //  (2) the reference's own fp32 arithmetic (v5:41-46, torch.norm's 8-accumulator sum, then sqrt, then argmin): every
    //      term of the sum is non-negative, so |fl(S) - S| <= c u S with u = 2^-24 and c = D/8 + 9 (one rounding per
    //      difference, squared: 2u; D/8 fma roundings per accumulator; 7 adds); the correctly rounded roots of two sums
    //      differ strictly once fl(S_k) / fl(S_k1) > ((1+u)/(1-u))^2.  Together: the reference decides k1 against k with a
    //      STRICT inequality (so no first-index tie rule can interfere) whenever  S_k - S_k1 > 2 (c + 2) u S_k1.
    //  Hence "second - best > 2 eps_s + 2 (D/8 + 12) u s1" certifies k1, where s1 >= S_k1 is bounded from the screen's own
    //  numbers: S_k1 <= best + |z'|^2 + eps_s, plus the rounding of n2 itself ((D + 2) u n2).  The bound is relative to the
    //  WINNER's distance (not to the largest distance the row can see), so it is rigorous for every D and negligible for
    //  rows close to a code.
    const int Dpad16 = ((D + 15) / 16) * 16;
    const float u24 = 5.9604644775390625e-08f;                                  // 2^-24
    const float eps_s = gamma * (E2max + cross) * fown;
    const float n2s = n2 * fown;
    const float s1 = fmaxf(0.0f, best + n2s) + eps_s + (float)(Dpad16 + 2) * u24 * n2s;
    const float thr = 2.0f * eps_s + 2.125f * (float)(Dpad16 / 8 + 12) * u24 * s1;
    // non-finite inputs make the comparison false
    bool certified = (twoemax < INFINITY) && (second - best > thr) && (bk >= 0) && (bk < K);
#ifdef LQ_ABL_CERT_ALL
    certified = true; my_k = (my_k >= 0 && my_k < K) ? my_k : (lane * 7) % K;
#endif
    return certified;
}

// usage[k] += 1 for every lane with `active`, with duplicates inside the wave combined first: when many rows map
// to few codes (the reference's default initialisation maps EVERY row to one code) per-row atomics on one address
// serialise chip-wide (measured: 1.7 ms instead of 0.57 ms for a 524 288-row batch with 32 codes).  Up to four
// leader rounds (each: the first active lane's code, a ballot of the lanes that share it, ONE atomic of the count)
// then plain atomics for whatever is left (the typical well-spread case pays four cheap rounds).
__device__ __forceinline__ void lq_usage_add(unsigned long long* __restrict__ usage, int k, bool active) {
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const unsigned long long act = __ballot(active);
        if (act == 0ull) return;                                   // wave-uniform
        const int leader = __ffsll((long long)act) - 1;
        const int kl = __shfl(k, leader, 64);
        const unsigned long long same = __ballot(active && k == kl);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&usage[kl], (unsigned long long)__popcll(same));
        active = active && k != kl;
    }
    if (active) atomicAdd(&usage[k], 1ull);
}

// z_q rows of certified rows: 16 lanes copy one codebook row (16 B each), 4 rows per pass
__device__ __forceinline__ void lq_screen_gather(const float* __restrict__ cb, float* __restrict__ zq, int my_k,
                                                 bool certified, int64_t row0, int64_t N, int D, int lane) {
#ifdef LQ_ABL_NOGATHER
    return;
#endif
    const int nvec = D / 4;
    for (int rr = 0; rr < 32; rr += 4) {
        const int src_lane = rr + (lane >> 4);
        const int kk = __shfl(my_k, src_lane, 64);
        const bool ok = __shfl((int)certified, src_lane, 64) != 0;
        const int64_t orow = row0 + src_lane;
        if (ok && orow < N) {
            const float4* src = reinterpret_cast<const float4*>(cb + (size_t)kk * D);
            float4* dst = reinterpret_cast<float4*>(zq + (size_t)orow * D);
            for (int v = lane & 15; v < nvec; v += 16) dst[v] = src[v];
        }
    }
}

// exact decision for listed rows (lipvq_screen.hip); z_by_slot: z is a compact [count][D] buffer
int lipvq_launch_rows(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                      const int* amb_list, const int* amb_count, int64_t N, int K, int D, hipStream_t st);
// the same for rows whose z_e was never stored: recomputed from x with the raw (unpacked) encoder weights
// raw6 = {W0, b0, W1, b1, W2 (Lipschitz-normalised), b2}
int lipvq_launch_rows_encode(const float* x, const float* const* raw6, int A, const float* cb, int64_t* idx, float* zq,
                             int64_t* usage, const int* amb_list, const int* amb_count, int64_t N, int K, int D,
                             hipStream_t st);
#endif
