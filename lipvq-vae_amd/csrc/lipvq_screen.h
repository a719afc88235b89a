// lipvq_screen.h -- layout of the prepared codebook and constants shared by the screening kernels
// (lipvq_screen.hip, lipvq_fused.hip).  Design notes: lipvq_screen.hip.
#ifndef LIPVQ_SCREEN_H_
#define LIPVQ_SCREEN_H_
#include <hip/hip_fp16.h>

#include <type_traits>

#include "lipvq_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LIPVQ_SCREEN_GAMMA 3.814697265625e-06f   /* 2^-18 */
#ifndef SCREEN_WAVES
#define SCREEN_WAVES 8
#endif

struct PrepLayout {
    int S, Dpad, Kpad, ntiles;
    size_t o_hdr, o_mu, o_tiles, tile_bytes, total;
    size_t o_tiles_hi, tile_bytes_hi;      // the one-product screen's tiles: hi fragments only (half the bytes to stage)
};

// k-steps (of 16 columns) of the screening instance that serves a D-column latent: the instances are 2, 4, 8 and 13 k-steps
// (D = 32, 64, 128, 208); any other width up to 208 runs the next larger one on zero-padded columns (zeros add nothing to a
// dot product, so the screened values and their error bound are those of the unpadded row); 0 = no instance.
__host__ __device__ static inline int lq_screen_S(int D) {
    const int s = (D + 15) / 16;
    return D <= 0 ? 0 : s <= 2 ? 2 : s <= 4 ? 4 : s <= 8 ? 8 : s <= 13 ? 13 : 0;
}

__host__ __device__ static inline PrepLayout prep_layout(int K, int D) {
    PrepLayout L;
    L.S = lq_screen_S(D) ? lq_screen_S(D) : (D + 15) / 16;
    L.Dpad = L.S * 16;
    L.Kpad = ((K + 31) / 32) * 32;
    L.ntiles = ((L.Kpad / 32 + 7) / 8) * 8;   // whole LDS stages: pad tiles carry e2 = +inf, zero fragments
    L.o_hdr = 0;                       // 16 words: [0] E2max bits, [1] Emax^2 bits, [2] max|2e'| bits, [3] se (int), [4] max|mu| bits,
                                       // [5] max_k |E_k - hi(E_k)| / |E_k| bits (the one-product screen's codebook-side residual ratio)
    L.o_mu = 64;
    L.o_tiles = L.o_mu + sizeof(float) * (size_t)L.Dpad;
    L.o_tiles = (L.o_tiles + 255) & ~(size_t)255;
    L.tile_bytes = (size_t)L.S * 2048 + 256;    // S steps x {hi,lo} x 32 codes x 2 halves x 16 B, then 32 x |e'|^2, then 32 x |e'| (rounded up)
    L.o_tiles_hi = (L.o_tiles + (size_t)L.ntiles * L.tile_bytes + 255) & ~(size_t)255;
    L.tile_bytes_hi = (size_t)L.S * 1024 + 256;    // S steps x 32 codes x 2 halves x 16 B of hi fragments, then 32 x |e'|^2, 32 x |e'|
    L.total = L.o_tiles_hi + (size_t)L.ntiles * L.tile_bytes_hi;
    return L;
}


// ------------------------------------------------------------------------------------------
// Block floating point for the fp16 split (tests/test_gpu_screen.py::test_any_magnitude):  x = hi + lo carries 22
// significant bits only while lo is a NORMAL fp16 number, i.e. |x| >= 2^-3.  So operands are multiplied by exact
// powers of two before the split: the codebook by 2^se (global, max |-2e'| lands in [2^13, 2^14)), every latent row
// by its own 2^sz (row max in [2^13, 2^14)).  Elements below 2^-17 of their row/codebook maximum then lose bits,
// which is 2^-39 of that maximum -- negligible.  The MFMA result is in units of 2^(sz+se); since the argmin only
// compares values of ONE row, the bookkeeping simply runs in those units (|e'|^2 is scaled when the bookkeeping adds
// it) and the certification threshold is scaled the same way.
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

template <int S, int TC_ = screen_default_tc(S), bool COARSE = false>
struct ScreenCfg {
    static constexpr int FRAG_BYTES = COARSE ? S * 1024 : S * 2048;            // one tile's fragments (hi only / hi and lo per k-step)
    static constexpr int TILE_BYTES = FRAG_BYTES + 256;
    static constexpr int TC = TC_;                                              // tiles per stage (divides 8)
    static constexpr int STAGE_BYTES = TC * TILE_BYTES;
    static constexpr int STAGE_VEC = STAGE_BYTES / 16;
};

// PACK bookkeeping (tile index in the low mantissa bits, lq_track_one) where the bookkeeping is the bottleneck: few MFMAs per
// tile (S <= 4: 12 or fewer MFMAs against 80 bookkeeping instructions); wider latents hide it under 24+ MFMAs
#ifndef LQ_PACK_FOR
#ifndef LQ_PACK_MAX_S
#define LQ_PACK_MAX_S 4
#endif
#define LQ_PACK_FOR(S) ((S) <= LQ_PACK_MAX_S)
#endif
// the stand-alone screen kernel: at most 80 KiB of stage ring, so that two workgroups share a CU
template <int S>
struct StandaloneScreen {
    static constexpr int TC = (S <= 2) ? 4 : (S <= 4) ? 2 : 1;
    static constexpr int NB = (S <= 8) ? 4 : 3;
    static constexpr bool PACK = LQ_PACK_FOR(S);
};

// Bookkeeping of registers [lo, hi) of a finished 32 x 32 tile of d~ - |e'|^2 f (the chain starts from zero; the |e'|^2
// term of the lane's code is added here, one fma per element, off the MFMA chain's critical path): per row (register)
// and lane (code mod 32) the smallest value, its code and the second smallest.
// One element of the bookkeeping.  PACK: the tile index is written into the low bits of the value (v_and_or_b32), so the
// smallest value carries its own code and no index array is kept: med3 + min on the packed floats (3 instructions + the fma
// instead of 4 + the fma, and 16 registers less).  The perturbation, below 2^(TB-23) of the value's own magnitude, is part of
// the certification margin (lq_screen_decide, pack_eps).  id = the code (unpacked) or the tile index (packed).
// LQ_ABL_NOE2 (ablation build, wrong results): the |e'|^2 f term is dropped -- what a bookkeeping of 3 instead of 4 vector
// instructions per element (the term folded into the chain's C operand at no cost) could gain AT MOST
#ifdef LQ_ABL_NOE2
#define LQ_E2_TERM(e2, f, a) (a)
#else
#define LQ_E2_TERM(e2, f, a) lq_fma(e2, f, a)
#endif
template <bool PACK>
__device__ __forceinline__ void lq_track_one(float v, int id, unsigned keep_mask, float& m1, float& m2, int& k1) {
    if constexpr (PACK) {
        // one instruction each (hipcc splits the and/or when both constants sit in scalar registers, and wraps fminf in two
        // NaN-canonicalising v_max): the tile index is kept in a vector register by the caller
        float key;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(v), "s"(keep_mask), "v"(id));
        m2 = __builtin_amdgcn_fmed3f(key, m1, m2);
        asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(key), "v"(m1));
    } else {
        const bool lt = v < m1;                 // one compare feeds both selects (fminf would cost two
        k1 = lt ? id : k1;                      // NaN-canonicalising v_max as well)
        m2 = __builtin_amdgcn_fmed3f(v, m1, m2);
        m1 = lt ? v : m1;
    }
}

// Bookkeeping of registers [LO, HI) of a finished 32 x 32 tile of d~ - |e'|^2 f (the chain starts from zero; the |e'|^2
// term of the lane's code is added here, one fma per element, off the MFMA chain's critical path): per row (register)
// and lane (code mod 32) the smallest value, its code and the second smallest.
// COARSE (round 3): the chain holds ONE product per k-step, hi x hi -- a third of the matrix work -- and the value that is booked is
// a LOWER BOUND of the code's distance.  With Z = the row's scaled centred latent, Zh its fp16 rounding, dZ = Z - Zh (EXACT in
// fp32: the split computes it), and E_k, Eh_k, dE_k the same for the code's scaled (-2 e'):
//     Z.E_k - Zh.Eh_k = dZ.Eh_k + Zh.dE_k + dZ.dE_k,   |.| <= |dZ| (|E_k| + |dE_k|) + (|Z| + |dZ|) |dE_k| + |dZ| |dE_k|   (Cauchy-Schwarz)
//                                                         <= |E_k| (|dZ| + rho (|Z| + 3 |dZ|)),   rho = max_k |dE_k| / |E_k|
// -- the MEASURED residual norms of this row and this codebook (about 0.3 x 2^-11 of the operand norms: 3-4x tighter than the
// a-priori half-ulp bound, and rigorous whatever the operands do in fp16: denormals, zeros).  So  w(n, k) = znr_n en_k  with
// en_k = |E_k| / 2^se = |-2 e'_k| as represented (prep_res_kernel, rounded up) and znr_n = (|dZ| + rho (|Z| + 3 |dZ|)) 2^se (1 + 2^-10)
// (lq_coarse_zn);  L = d~ - w <= d  for every code, and  L + 2 w >= d  for the winner (lq_screen_decide).
// The chain is SEEDED with |e'|^2 f - w (one fma per element when the tile starts) instead of starting from
// zero and adding the terms afterwards (two): the accumulator is the booked value.
// a2lo = sum of the squared split residuals (scaled units), n2 = |z'|^2 (unscaled), fz = 2^sz, fown = 2^(sz + se)
__device__ __forceinline__ float lq_coarse_zn(float a2lo, float n2, float fz, float fown, float rho) {
    const float A = lq_sqrt(a2lo), B = lq_sqrt(n2) * fz;
    const float t = lq_fma(rho, lq_fma(3.0f, A, B), A);
    return t * (fown / fz) * 1.0009765625f;                 // (1 + 2^-10): the norms' own fp32 rounding (D + 2 roundings each), generously
}
template <int LO, int HI, bool PACK = false, bool COARSE = false, bool SEEDED = COARSE>
__device__ __forceinline__ void lq_track_part(const f32x16& acc, float e2, float en, const float (&frow)[16], const float (&znr)[16],
                                              int id, unsigned keep_mask, float (&m1)[16], float (&m2)[16], int (&k1)[16]) {
#ifdef LQ_ABL_NOTRACK
    if (LO == 0) m1[0] = fminf(m1[0], acc[0] + acc[5] + acc[10] + acc[15]);   // keeps the MFMAs alive
    return;
#endif
#pragma unroll
    for (int r = LO; r < HI; ++r) {
        // COARSE / SEEDED: the chain started from |e'|^2 f (- w) (lq_screen_core_rg seeds it), the accumulator IS the booked value
        const float v = SEEDED ? acc[r] : LQ_E2_TERM(e2, frow[r], acc[r]);
        lq_track_one<PACK>(v, id, keep_mask, m1[r], m2[r], k1[r]);
    }
}

// the pending tile's registers that are booked behind MFMA j of the NM MFMAs of the running tile (NM = 3 S, or S for the
// one-product chain): [16 j / NM, 16 (j + 1) / NM)   (j a compile-time constant after unrolling)
template <int S, bool PACK, bool COARSE = false, bool SEEDED = COARSE>
__device__ __forceinline__ void lq_track_after_mfma(int j, const f32x16& acc, float e2, float en, const float (&frow)[16],
                                                    const float (&znr)[16], int id, unsigned keep_mask, float (&m1)[16],
                                                    float (&m2)[16], int (&k1)[16]) {
#ifdef LQ_ABL_NOTRACK
    if (j == 0) m1[0] = fminf(m1[0], acc[0] + acc[5] + acc[10] + acc[15]);
    return;
#endif
    constexpr int NM = COARSE ? S : 3 * S;
    const int lo = (16 * j) / NM, hi = (16 * (j + 1)) / NM;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (r >= lo && r < hi) {
            const float v = SEEDED ? acc[r] : LQ_E2_TERM(e2, frow[r], acc[r]);
            lq_track_one<PACK>(v, id, keep_mask, m1[r], m2[r], k1[r]);
        }
}

// ------------------------------------------------------------------------------------------
// The screening main loop, shared by screen_kernel (lipvq_screen.hip) and tokenize_kernel (lipvq_fused.hip).
// NT threads stream the prepared codebook through a ring of NB LDS stage buffers of TC column tiles each; every wave
// multiplies its 32 rows (fp16 hi/lo A fragments ah/al) against every tile and keeps, per accumulator register (= row)
// and lane (= code mod 32), the smallest d~, its code, and the second smallest d~.
//
// What in-kernel cycle stamps showed about the first version of this loop (scripts/stamps.py, round 2; 256 tiles per wave
// at BASELINE config 2: 98 k cycles of MFMA issue, 380-520 k cycles measured): hipcc read each k-step's two B fragments
// right before the three MFMAs that use them and waited for the LDS round trip four times per tile; the 80 bookkeeping
// instructions of a tile ran after its chain; and a quarter of the loop was spent waiting, at the per-stage barrier, for a
// stage copy issued only one stage earlier.  Hence:
//  * B fragments (and |e'|^2) are read TWO k-steps ahead of the MFMAs that use them (ring of three fragment pairs);
//  * the bookkeeping of tile t-1 (16/S registers per k-step) is issued between the MFMAs of tile t -- a wave's own vector
//    instructions do run beside its fp16 MFMAs (scripts/probe/probe_pipes2.hip: four per MFMA are free; beside an fp32
//    MFMA none are) -- the two chains alternate between two named accumulators, so nothing is copied;
//  * a chain starts from C = 0 and |e'|^2 f is added by the bookkeeping (no accumulator initialisation on the chain);
//  * stage copies run NB - 1 stages ahead (LDS-DMA stays in flight across barriers: raw s_barrier and COUNTED vmcnt, never
//    __syncthreads, which drains it); the wait for stage st+1, the barrier and the issue of stage st+NB-1 sit in the MIDDLE
//    of stage st, so that the last k-steps of a stage can already read the next stage's first fragments.
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void lq_wait_vmcnt() {            // all but the N youngest vector-memory operations of this wave are done
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS the ring needs: NB stage buffers + 1 KiB that absorbs the copies issued for stages past the end
template <int S, int TC_, int NB>
constexpr size_t lq_ring_bytes() { return (size_t)NB * ScreenCfg<S, TC_>::STAGE_BYTES + 1024; }

// number of low mantissa bits that hold the tile index in PACK mode, and the relative perturbation that costs
__host__ __device__ static inline int lq_pack_bits(int ntiles) { int b = 1; while ((1 << b) < ntiles) ++b; return b; }

// RG row groups per wave (round 3): the wave multiplies RG x 32 rows against every tile, so that one pair of B-fragment reads,
// one |e'|^2 read, one stage hand-over (wait, barrier, DMA issue) serve RG x 3 MFMAs per k-step instead of 3, and consecutive
// MFMAs go to different accumulators.  ah/al/m1/m2/k1 carry the group as their leading dimension.
// DEFER (round 4): the caller's last vector-memory stores before the screen (the fused launch's z_e rows of its last layer-2 tile:
// NDEF store instructions per wave when `have_def`) are issued by `deferred()` BEHIND the prologue's stage copies instead of in
// front of them.  vmcnt retires in order: issued in front, their write acknowledgements (microseconds) stood between every wave
// and "stage 0 has landed" -- 3.5 % of the cfg2 launch (profiles/r04_g_ze_store_placement.txt).  Behind the copies, the prologue's
// wait and the first hand-over simply leave NDEF more operations outstanding.
struct LqNoDeferred { __device__ __forceinline__ void operator()() const {} };
// SEED (round 4; three-product chains whose rows share ONE scale -- the fused launch's sigmoid latents): the chain of a tile starts
// from |e'|^2 f, the same value in all sixteen registers (eight v_pk_mov_b32), instead of from zero with the term added per element
// when the tile is booked (sixteen fmas): the accumulator is the booked value, as in the one-product chain.  MEASURED SLOWER (cfg2
// 0.3832 -> 0.3861 ms, same box: the chain's first MFMA now waits for its C operand where it took an inline zero, and the fmas it
// saves sat off the critical path) -- kept behind LQ_SEED_E2 as a record, off by default.  Screening values move
// by roundings of the partial sums (covered by gamma: lipvq_screen.hip, "Error bound"); indices are whatever the exact arithmetic says.
template <int S, int NT, int TC_ = screen_default_tc(S), int NB = 4, bool PACK = false, int RG = 1, bool COARSE = false, int NDEF = 0,
          typename DEFERRED = LqNoDeferred, bool SEED = false>
__device__ __forceinline__ void lq_screen_core_rg(const f16x8 (&ah)[RG][S], const f16x8 (&al)[RG][S],
                                                  const unsigned char* __restrict__ tiles, int ntiles,
                                                  unsigned char* stage0, int tid, const float (&frow)[16],
                                                  const float (&znr)[RG][16],
                                                  float (&m1)[RG][16], float (&m2)[RG][16], int (&k1)[RG][16],
                                                  bool have_def = false, DEFERRED deferred = DEFERRED()) {
    using C = ScreenCfg<S, TC_, COARSE>;
    // NB = 2 is NOT a ring this loop can run: with one stage in flight (PD = 1) stage st+1 is only ISSUED at the hand-over in the
    // middle of stage st, and nothing waits for it before the read-ahead crosses into it.  (An experiment build with 2 x 4 tiles
    // returned the right answers 20x slower -- rows screened against bytes still in flight fail their certificate and go to the
    // exact stage -- which is luck, not a guarantee: profiles/r03_z_barrier_and_ring_ab.txt.)
    static_assert(NB >= 3 && NB <= 4, "ring of 3 or 4 stage buffers");
    constexpr int NW = NT / 64;
    constexpr int CHUNKS = (C::STAGE_BYTES + 1023) / 1024;          // 1 KiB = one wave-instruction of the LDS-DMA
    constexpr int CPW = (CHUNKS + NW - 1) / NW;                       // DMA instructions per wave and stage: the SAME for every wave
    constexpr int PD = NB - 1;                                        // stages in flight ahead of the one being read
    constexpr int NSTEP = C::TC * S;                                  // k-steps per stage
    constexpr int MID = (C::TC >= 2) ? (C::TC / 2) * S : S / 2;       // the k-step in front of which the mid-stage hand-over sits
#ifndef LQ_FRAG_RING_S4
#define LQ_FRAG_RING_S4 2      /* measured: 4 slots (reads three k-steps ahead) change nothing */
#endif
#ifndef LQ_FRAG_RING_S8
#define LQ_FRAG_RING_S8 2      /* experiment knob: 4 = reads three k-steps ahead at S = 8 */
#endif
    constexpr int FR = (S == 4 && NSTEP % LQ_FRAG_RING_S4 == 0) ? LQ_FRAG_RING_S4
                     : (S == 8 && NSTEP % LQ_FRAG_RING_S8 == 0) ? LQ_FRAG_RING_S8 : 2;      // fragment ring slots; reads run FR - 1 k-steps ahead
    constexpr int FD = FR - 1;
    constexpr int PERIOD = ((C::TC & 1) || (NSTEP % FR)) ? 2 : 1;     // stages per loop trip: an even number of tiles, whole fragment rings
    static_assert(FR == 2 || NSTEP % FR == 0, "a wider fragment ring needs whole rings per stage");
    static_assert(C::STAGE_BYTES >= 1024 && C::STAGE_BYTES % 16 == 0, "stage copies are whole KiB pieces");
    static_assert(NSTEP - FD >= MID, "next-stage fragments are read after the hand-over");
    static_assert(CPW * PD < 64, "vmcnt immediate");
    const int lane = tid & 63, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: DMA addresses stay scalar base + lane offset
#ifdef LQ_ABL_NOLOOP
    const int nstage = 0;            // ablation build only
#else
    const int nstage = ntiles / C::TC;                              // ntiles is a multiple of 8: nstage % PERIOD == 0
#endif
    unsigned char* dummy = stage0 + (size_t)NB * C::STAGE_BYTES;    // 1 KiB nobody reads
    // Stage copies go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write issue).  One
    // wave-instruction moves 64 x 16 B to a wave-uniform LDS base + lane*16.  Wave w copies KiB pieces w, w + NW, ...; the
    // last piece of a stage is anchored at the stage's END (it overlaps its predecessor: same bytes), a wave whose piece
    // index runs past the stage repeats the last piece, and a stage past the end of the codebook is "copied" as pieces of
    // the first stage into the dummy KiB: EVERY wave issues exactly CPW full, unmasked instructions per call, so the loop
    // body has no branch and the counted waits below mean the same thing in every wave at every stage.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto stage_dma = [&](int st, int buf) {
#ifndef LQ_ABL_NOSTAGE
        const bool real = st < nstage;
        const unsigned char* src = tiles + (real ? (size_t)st * C::STAGE_BYTES : (size_t)0);
        unsigned char* dst = stage0 + (size_t)buf * C::STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            int chunk = wave + j * NW;
            chunk = chunk < CHUNKS ? chunk : CHUNKS - 1;
            int off = chunk * 1024;
            off = off + 1024 <= C::STAGE_BYTES ? off : C::STAGE_BYTES - 1024;
            unsigned char* d = real ? dst + off : dummy;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off + lane * 16), (lds_ptr_t)d, 16, 0, 0);
        }
#endif
    };
    auto frag = [&](const unsigned char* tb, int s, int hl) {
        return COARSE ? *reinterpret_cast<const f16x8*>(tb + ((size_t)s * 64 + lane) * 16)          // hi-only tiles
                      : *reinterpret_cast<const f16x8*>(tb + (((size_t)s * 2 + hl) * 64 + lane) * 16);
    };
    lq_wg_barrier();                              // earlier users of the stage buffers (previous row block) are done
    // prologue: stages 0 .. PD-1 in flight, stage 0 landed
#pragma unroll
    for (int p = 0; p < PD; ++p) stage_dma(p, p);
    static_assert((PD - 1) * CPW + NDEF < 64, "vmcnt immediate");
    if (NDEF > 0 && have_def) {                   // (launch-uniform)
        deferred();
#ifndef LQ_ABL_NOSTAGE
        lq_wait_vmcnt<(PD - 1) * CPW + NDEF>();
#endif
    } else {
#ifndef LQ_ABL_NOSTAGE
        lq_wait_vmcnt<(PD - 1) * CPW>();
#endif
    }
    lq_wg_barrier();

    // two named accumulators (per row group): the chain of tile i runs into one while the other (tile i-1) is booked
    f32x16 accA[RG], accB[RG];
#pragma unroll
    for (int g_ = 0; g_ < RG; ++g_)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[g_][r] = INFINITY;      // "no previous tile": INFINITY never beats anything
    float e2A = 0.0f, e2B = 0.0f;
    int codeA = 0, codeB = 0;                     // the code of the lane in the tile (unpacked) or the tile index (PACK)
    const unsigned keep_mask = PACK ? ~((1u << lq_pack_bits(ntiles)) - 1u) : 0xffffffffu;
    // fragment ring: slot ((g + so) % FR) holds k-step g of the current stage (g = NSTEP: k-step 0 of the next stage); so = 0,
    // or the stage's position in the loop trip when NSTEP is odd
    f16x8 fh[FR], fl[FR];
    float e2q[2] = {0.0f, 0.0f};                  // |e'|^2 of the tile in accumulator A / B
    float enq[2] = {0.0f, 0.0f};                  // COARSE: |e'| of the same tiles
    float enA = 0.0f, enB = 0.0f;
#ifndef LQ_ABL_NOLDSB
#pragma unroll
    for (int g = 0; g < FD; ++g) {                // k-steps 0 .. FD-1 of stage 0 (FD <= S: all in tile 0)
        fh[g] = frag(stage0, g, 0);
        if constexpr (!COARSE) fl[g] = frag(stage0, g, 1);
    }
    e2q[0] = reinterpret_cast<const float*>(stage0 + C::FRAG_BYTES)[ln];
    if constexpr (COARSE) enq[0] = reinterpret_cast<const float*>(stage0 + C::FRAG_BYTES + 128)[ln];
    static_assert(FD <= S, "prologue reads stay in tile 0");
#endif
    int buf = 0;                                  // ring position of stage st
    auto do_stage = [&](auto POS, int st) {
        constexpr int par = (decltype(POS)::value * C::TC) & 1;    // 0: the stage's first tile runs into accA, 1: into accB
        constexpr int so = (decltype(POS)::value * NSTEP) % FR;    // fragment slot of the stage's k-step 0
        const unsigned char* sb = stage0 + (size_t)buf * C::STAGE_BYTES;
        int nbuf = buf + 1; nbuf = nbuf == NB ? 0 : nbuf;
        const unsigned char* nsb = stage0 + (size_t)nbuf * C::STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < NSTEP; ++g) {
            const int c = g / S, s = g % S;
            if (g == MID) {
                // ---- mid-stage hand-over: stage st+1 has landed everywhere (the PD-2 younger stages may still fly); everyone has
                // left stage st-1, whose buffer is the one stage st+PD goes to
#ifndef LQ_ABL_NOSTAGE
                // (the first hand-over of a pass with deferred stores: they are younger than stage 1's copy and need not be done)
                if (NDEF > 0 && have_def && st == 0) lq_wait_vmcnt<(PD >= 2 ? (PD - 2) * CPW : 0) + (PD >= 2 ? NDEF : 0)>();
                else lq_wait_vmcnt<(PD >= 2 ? (PD - 2) * CPW : 0)>();
#endif
#ifndef LQ_ABL_NOBARRIER
                lq_wg_barrier();
#endif
                int b = buf + PD; b = b >= NB ? b - NB : b;
                stage_dma(st + PD, b);
            }
            f32x16 (&acc)[RG] = (((c + par) & 1) == 0) ? accA : accB;
            const f32x16 (&prev)[RG] = (((c + par) & 1) == 0) ? accB : accA;
            const float e2_prev = (((c + par) & 1) == 0) ? e2B : e2A;
            const float en_prev = (((c + par) & 1) == 0) ? enB : enA;
            const int code_prev = (((c + par) & 1) == 0) ? codeB : codeA;
            if (s == 0) {
                const float e2c = e2q[(c + par) & 1];
                const float enc = enq[(c + par) & 1];
                if constexpr (SEED && !COARSE) {
                    // sixteen copies of one value as register PAIRS: v_pk_mov_b32 moves two dwords per instruction (hipcc writes
                    // fifteen v_mov_b32 -- as many vector instructions as the fmas this arrangement removes)
                    typedef float lq_f2 __attribute__((ext_vector_type(2)));
                    const float e2f = e2c * frow[0];
                    lq_f2 pr;
                    pr.x = e2f;
                    pr.y = e2f;
#pragma unroll
                    for (int g_ = 0; g_ < RG; ++g_) {
                        acc[g_][0] = pr.x;
                        acc[g_][1] = pr.y;
#pragma unroll
                        for (int q = 1; q < 8; ++q) {
                            lq_f2 d;
                            asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(d) : "v"(pr));
                            acc[g_][2 * q] = d.x;
                            acc[g_][2 * q + 1] = d.y;
                        }
                    }
                } else {
#pragma unroll
                    for (int g_ = 0; g_ < RG; ++g_)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g_][r] = COARSE ? lq_fma(-znr[g_][r], enc, e2c * frow[r]) : 0.0f;
                }
                int code = PACK ? (st * C::TC + c) : (st * C::TC + c) * 32 + ln;
                if constexpr (PACK) asm volatile("" : "+v"(code));      // the tile index lives in a vector register (lq_track_one)
                if (((c + par) & 1) == 0) { e2A = e2c; enA = enc; codeA = code; } else { e2B = e2c; enB = enc; codeB = code; }
            }
            const f16x8 bh = fh[(g + so) % FR];
            f16x8 bl = bh;
            if constexpr (!COARSE) bl = fl[(g + so) % FR];
            // k-step g + FD of this stage, or (g + FD >= NSTEP > MID: the hand-over has passed) of the next stage's first tile.
            // Unconditional: behind the last stage it reads bytes of the ring that nobody uses (no branch in the loop body).
            // (LQ_FRAG_BEFORE_MFMA: measurement knob, the placement until round 3 -- in front of the k-step's first MFMA.)
            auto read_ahead = [&]() {
#ifndef LQ_ABL_NOLDSB
                const int g1 = g + FD;
                const unsigned char* base = (g1 < NSTEP) ? sb : nsb;
                const int gg = (g1 < NSTEP) ? g1 : g1 - NSTEP;
                const unsigned char* tb = base + (size_t)(gg / S) * C::TILE_BYTES;
                fh[(g1 + so) % FR] = frag(tb, gg % S, 0);
                if constexpr (!COARSE) fl[(g1 + so) % FR] = frag(tb, gg % S, 1);
                if (gg % S == 0) {
                    e2q[(g1 / S + par) & 1] = reinterpret_cast<const float*>(tb + C::FRAG_BYTES)[ln];   // g1 / S >= TC: next stage
                    if constexpr (COARSE) enq[(g1 / S + par) & 1] = reinterpret_cast<const float*>(tb + C::FRAG_BYTES + 128)[ln];
                }
#endif
            };
#ifdef LQ_FRAG_BEFORE_MFMA
            read_ahead();
#endif
            // pinned order: first MFMA, the read-ahead, then (its share of the pending tile's bookkeeping, MFMA) x 2 -- left alone, hipcc
            // lumps the bookkeeping behind the chain, where nothing hides it.  The read-ahead sits BEHIND the k-step's first MFMA
            // (round 3): hipcc guards that MFMA's fragment with `s_waitcnt lgkmcnt(0)`, not a counted wait, so reads issued in front
            // of it were waited for at once -- a whole LDS round trip exposed per k-step (cfg2 -3.6 %, cfg3 -3 %, same box; a 4-slot
            // ring on top changes nothing more: profiles/r03_z_frag_read_placement_ab.txt)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g_ = 0; g_ < RG; ++g_) {
                acc[g_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g_][s], bh, acc[g_], 0, 0, 0);
#ifndef LQ_FRAG_BEFORE_MFMA
                if (g_ == 0) { __builtin_amdgcn_sched_barrier(0); read_ahead(); __builtin_amdgcn_sched_barrier(0); }
#endif
                lq_track_after_mfma<S, PACK, COARSE, COARSE || SEED>(COARSE ? s : 3 * s + 0, prev[g_], e2_prev, en_prev, frow, znr[g_], code_prev, keep_mask,
                                                     m1[g_], m2[g_], k1[g_]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!COARSE) {
#pragma unroll
                for (int g_ = 0; g_ < RG; ++g_) {
                    acc[g_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g_][s], bh, acc[g_], 0, 0, 0);
                    lq_track_after_mfma<S, PACK, false, SEED>(3 * s + 1, prev[g_], e2_prev, en_prev, frow, znr[g_], code_prev, keep_mask,
                                                        m1[g_], m2[g_], k1[g_]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g_ = 0; g_ < RG; ++g_) {
                    acc[g_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g_][s], bl, acc[g_], 0, 0, 0);
                    lq_track_after_mfma<S, PACK, false, SEED>(3 * s + 2, prev[g_], e2_prev, en_prev, frow, znr[g_], code_prev, keep_mask,
                                                        m1[g_], m2[g_], k1[g_]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        buf = nbuf;
    };
    for (int st = 0; st < nstage; st += PERIOD) {
        do_stage(std::integral_constant<int, 0>{}, st);
        if constexpr (PERIOD == 2) do_stage(std::integral_constant<int, 1>{}, st + 1);
    }
    // the last tile's chain: ntiles is even, so it ran into accB
#pragma unroll
    for (int g_ = 0; g_ < RG; ++g_)
        lq_track_part<0, 16, PACK, COARSE, COARSE || SEED>(accB[g_], e2B, enB, frow, znr[g_], codeB, keep_mask, m1[g_], m2[g_], k1[g_]);
    // the copies issued for stages past the end go to the dummy KiB, but they count: drain them, then every wave has left
    // the stage buffers (the callers reuse them as per-wave scratch: lq_screen_decide)
#ifndef LQ_ABL_NOSTAGE
    lq_wait_vmcnt<0>();
#endif
    lq_wg_barrier();
}

// one row group: the original interface (screen_kernel, tokenize_kernel's 32-row instances)
template <int S, int NT, int TC_ = screen_default_tc(S), int NB = 4, bool PACK = false>
__device__ __forceinline__ void lq_screen_core(const f16x8 (&ah)[S], const f16x8 (&al)[S],
                                               const unsigned char* __restrict__ tiles, int ntiles,
                                               unsigned char* stage0, int tid, const float (&frow)[16],
                                               float (&m1)[16], float (&m2)[16], int (&k1)[16]) {
    lq_screen_core_rg<S, NT, TC_, NB, PACK, 1, false>(reinterpret_cast<const f16x8 (&)[1][S]>(ah), reinterpret_cast<const f16x8 (&)[1][S]>(al),
                                                      tiles, ntiles, stage0, tid, frow, reinterpret_cast<const float (&)[1][16]>(frow),
                                                      reinterpret_cast<float (&)[1][16]>(m1), reinterpret_cast<float (&)[1][16]>(m2),
                                                      reinterpret_cast<int (&)[1][16]>(k1));
}

// frow[r] = factor of row (r, h) = the row this lane's accumulator register r belongs to, fetched from the lane that
// owns that row (row i lives in lanes i and i + 32, both hold fown)
__device__ __forceinline__ void lq_row_factors(float fown, int lane, float (&frow)[16]) {
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) frow[r] = __shfl(fown, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
}

// LDS scratch per wave of lq_screen_decide: 32 rows of 32 values, row stride 36 floats (16-byte aligned rows whose float4 reads
// fall on distinct bank groups: a 32-float stride made every 16-byte read of the transposed image an 8-way bank conflict)
#define LQ_DECIDE_STRIDE 36
#define LQ_DECIDE_BYTES (32 * LQ_DECIDE_STRIDE * 4)

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
// What lq_screen_decide leaves behind for lq_screen_emit: the 16 per-lane minima of this lane's part of the row (lane l of the
// half-wave held the codes = l mod 32; part h of row i sees lanes 16h .. 16h+15), the smallest per-lane SECOND minimum of the
// part, the row's best value and the two coefficients of the margin  thr(v) = t0 + p (|best| + |v|).
struct LqDecision {
    float vv[16];
    float m2min, best, t0, p;
    bool screen_ok;              // the screen's numbers mean something (finite codebook bound, best code in range)
};

// COARSE (one-product screen): the booked values are lower bounds L = d~ - w (lq_track_part); `zn` is this lane's row's
// error scale (lq_coarse_zn) and `tiles` / `tile_bytes` / `frag_bytes` locate the winner's |e'| (the hi-only tiles) -- its upper bound is L + 2 w.
template <bool PACK = false, bool COARSE = false>
__device__ __forceinline__ bool lq_screen_decide(const float (&m1)[16], const float (&m2)[16], const int (&k1)[16],
                                                 unsigned char* wave_lds /* LQ_DECIDE_BYTES, this wave only */,
                                                 const unsigned* hdr, float n2, float fown, float gamma, int K, int D,
                                                 int lane, int& my_k, LqDecision& dec, float pack_eps = 0.0f,
                                                 unsigned keep_mask = 0xffffffffu, float zn = 0.0f,
                                                 const unsigned char* tiles = nullptr, size_t tile_bytes = 0, int frag_bytes = 0) {
    constexpr int TS = LQ_DECIDE_STRIDE;
    const int ln = lane & 31, h = lane >> 5;
    float* tv = reinterpret_cast<float*>(wave_lds);           // [32 rows][TS], reused by the passes
    // ---- pass 1: m1 -> best value, its position among my 16 entries, second smallest m1 ----------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) tv[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + ln] = m1[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    float best = INFINITY, second = INFINITY;
    int pos = 0;
    {
        const float4* pv = reinterpret_cast<const float4*>(tv + ln * TS + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v4 = pv[q];
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dec.vv[4 * q + e] = vv[e];
                const bool take = vv[e] < best;       // equal minima need no tie-break: second == best then, the
                second = __builtin_amdgcn_fmed3f(vv[e], best, second);   // row is not certified and the exact
                pos = take ? 4 * q + e : pos;                             // kernel decides it
                best = take ? vv[e] : best;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                          // pass-1 reads are issued before the next pass overwrites
    // ---- pass 2: m2 -----------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) tv[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + ln] = m2[r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float m2min = INFINITY;
    {
        const float4* pv = reinterpret_cast<const float4*>(tv + ln * TS + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v4 = pv[q];
            m2min = fminf(m2min, fminf(fminf(v4.x, v4.y), fminf(v4.z, v4.w)));
        }
    }
    second = fminf(second, m2min);
    __builtin_amdgcn_wave_barrier();
    int bk;
    if constexpr (PACK) {
        // the smallest value carries its tile index; its lane (= code mod 32) is the column it was read from
        bk = (int)(__float_as_uint(best) & ~keep_mask) * 32 + 16 * h + pos;
    } else {
        // ---- pass 3: k1 -> the code at that position; the image STAYS in the scratch for lq_screen_emit ----------
        int* tk = reinterpret_cast<int*>(wave_lds);
#pragma unroll
        for (int r = 0; r < 16; ++r) tk[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + ln] = k1[r];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        bk = tk[ln * TS + 16 * h + pos];
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
    //  A width that is not a multiple of 8 adds the reference's remainder handling to (2): up to 7 more additions, four of them
    //  of separately rounded products (lq_sqdist8; the sum rule adds its scalar tail likewise): 11 more roundings.
    const int Dpad16 = ((D + 15) / 16) * 16 + ((D & 7) ? 88 : 0);             // (enters only as Dpad16 / 8 + 12 and Dpad16 + 2 below)
    const float u24 = 5.9604644775390625e-08f;                                  // 2^-24
    const float eps_s = gamma * (E2max + cross) * fown;
    const float n2s = n2 * fown;
    //  (0) COARSE: `best` is the winner's LOWER bound; its distance is at most best + 2 w(n, k1) (lq_track_part), every other code's
    //      at least its own booked value: the same inequality with the winner's side raised by w2 = 2 w.
    float w2 = 0.0f;
    if constexpr (COARSE) {
        const int bkc = (bk >= 0 && bk < K) ? bk : 0;
        const float en = reinterpret_cast<const float*>(tiles + (size_t)(bkc >> 5) * tile_bytes + (size_t)frag_bytes + 128)[bkc & 31];
        w2 = 2.0f * zn * en;
        w2 = lq_fma(w2, 9.5367431640625e-07f, w2);          // (1 + 2^-20): the product's own rounding, generously
    }
    const float s1 = fmaxf(0.0f, best + w2 + n2s) + eps_s + (float)(Dpad16 + 2) * u24 * n2s;
    //  (3) PACK bookkeeping (lq_track_one): best and second carry the tile index in their low bits, a perturbation below
    //      pack_eps = 2^(TB-23) of each value's own magnitude, and so does every other code's value that `second` bounds.
    //  The same inequality, applied to ANY code's value v instead of `second`, says that code loses strictly to k1 in the
    //  reference's arithmetic: thr(v) = t0 + p (|best| + |v|)  (lq_screen_emit lists the codes it does not exclude).
    dec.t0 = w2 + 2.0f * eps_s + 2.125f * (float)(Dpad16 / 8 + 12) * u24 * s1;
    dec.p = 2.125f * pack_eps;
    dec.best = best;
    dec.m2min = m2min;
    dec.screen_ok = (twoemax < INFINITY) && (bk >= 0) && (bk < K);
    const float thr = dec.t0 + dec.p * (lq_abs(best) + lq_abs(second));
    // non-finite inputs make the comparison false
    bool certified = dec.screen_ok && (second - best > thr);
#ifdef LQ_ABL_CERT_ALL
    certified = true; my_k = (my_k >= 0 && my_k < K) ? my_k : (lane * 7) % K;
#endif
    return certified;
}

// Rows the screen could not certify (plus rows whose screen is meaningless: fp16 overflow of the codebook, `lists_ok` false):
// one slot of the workspace lists each -- the row, the screen's best candidate (bounds a full exact scan) and, new in round 2,
// WHICH codes can still win.  By (1)-(3) above every code whose screened value v exceeds best + thr(v) loses strictly to k1
// in the reference's own arithmetic, so the reference's argmin is the first exact minimum among the others.  Per lane of the
// half-wave (= codes congruent to that lane mod 32) the screen kept the smallest value (its code known) and the second
// smallest (code unknown).  Per part h of the row (lanes 16h .. 16h+15), at [8h] of the slot's 16 ints: {n, mask, c0 .. c5}
//   n >= 0 : the part's candidates are exactly the n listed codes (the lanes' minima within the margin); a typical
//            uncertified row is a near-tie of two codes: two candidates instead of K;
//   n = -2 : some lane's SECOND minimum may be within the margin too (or more than LQ_CAND_MAX lanes are): every code of the
//            lanes in `mask` (bit l = lane 16h + l) has to be scored -- |mask| K/32 codes.  The mask holds every lane whose
//            minimum is <= vmax, the largest value the margin admits: v - best <= t0 + p (|best| + |v|) implies
//            v <= (best + t0 + p |best|) / (1 - p), and a lane with ANY admissible code has its minimum below that;
//   n = -1 : no information (scan all K codes).
// `mask` is always valid unless n = -1; the exact kernels use the lane masks of both parts as soon as one part says -2.
#define LQ_CAND_MAX 6
__host__ __device__ static inline size_t lq_list_ints(int64_t N);
// slots that get a list (later slots: full scan).  Every row can have one since round 3: the one-product screen leaves 10-40 % of
// the rows to the exact kernel -- each with its two or three candidates -- where the three-product screen left a fraction of a percent
__host__ __device__ static inline size_t lq_cand_cap(int64_t N) { return (size_t)N + 64; }

// Workspace header of the screened routes (16 ints), round 4.  PUBLISHED by the screening launch before it ends (what the
// kernels behind it and the host read): [0] rows the screen left to an exact decision (listed + decided in the launch's tail),
// [1] listed rows (slots).  LIVE counters at [LQ_WS_LIVE ...], zero between calls:
//   +0/+1 one 64-bit word: low half = listed rows (slot reservations add to it), high half = workgroups of the screening launch
//         that have arrived at its end.  The workgroup whose arrival completes the grid gets both halves back from its own
//         atomic: it publishes [0] and [1] and zeroes the word -- every other workgroup's reservations returned before that
//         workgroup arrived, so the low half is final;
//   +2    slots the list kernel left to the scanning kernel -- zeroed by the NEXT screening launch's first workgroup (nothing
//         reads or writes it between the scanning kernel of one call and the list kernel of the next);
// No fill launch per call (it was a 4.7 us launch of its own, a twelfth of a 65 536-row shard's time), no finishing pass in the
// kernels behind the launch (their grids are thousands of workgroups: that many arrivals on one word serialise for tens of
// microseconds).  lipvq_tokenize_workspace_init zeroes a fresh workspace once.
#define LQ_WS_LIVE 8
#define LQ_WS_SLOT2 2
// host side: the published number of listed rows, given the pointer to the live counters
static inline const int* lq_ws_listed(const int* live) { return live - LQ_WS_LIVE + 1; }
#if defined(__HIPCC__)
__device__ __forceinline__ void lq_ws_begin(int* __restrict__ live) {
    if (blockIdx.x == 0 && threadIdx.x == 0) live[LQ_WS_SLOT2] = 0;
}
// every thread of every workgroup of the screening launch calls this once, last thing
__device__ __forceinline__ void lq_ws_publish(int* __restrict__ live) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's reservations and list stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(live), 1ull << 32);
        if ((unsigned)(old >> 32) == gridDim.x - 1u) {                   // the last workgroup of the grid
            const int listed = (int)(unsigned)old;
            live[-LQ_WS_LIVE] = listed;
            live[-LQ_WS_LIVE + 1] = listed;
            *reinterpret_cast<unsigned long long*>(live) = 0ull;
        }
    }
}
// a wave reserves n slots of the row list: the base slot (the low half of the live word)
__device__ __forceinline__ int lq_ws_reserve(int* __restrict__ live, int n) {
    return (int)(unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(live), (unsigned long long)(unsigned)n);
}
#endif

// What a listed half-row hands over: head = n / -2 / -1 (see above), the lane mask, up to LQ_CAND_MAX codes.
template <bool PACK>
__device__ __forceinline__ void lq_emit_candidates(const LqDecision& dec, bool lists_ok, int K, int lane, unsigned keep_mask,
                                                   const unsigned char* wave_lds, int& head, unsigned& mask_out,
                                                   int (&codes)[LQ_CAND_MAX]) {
    const int ln = lane & 31, h = lane >> 5;
    const float ab = lq_abs(dec.best);
    // vmax, rounded up generously (the two roundings of the quotient are far below the 2^-20 slack)
    const float vmax0 = (dec.best + dec.t0 + dec.p * ab) / (1.0f - dec.p);
    const float vmax = vmax0 + lq_abs(vmax0) * 9.5367431640625e-07f + 1.1754944e-38f;
    int n = 0;
    unsigned mask = 0u;
    bool nothing = !lists_ok || !dec.screen_ok || !(vmax == vmax);
    const bool second_in = !(dec.m2min - dec.best > dec.t0 + dec.p * (ab + lq_abs(dec.m2min)));
#pragma unroll
    for (int q = 0; q < LQ_CAND_MAX; ++q) codes[q] = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = dec.vv[e];
        if (!(v > vmax)) mask |= 1u << e;                               // (a NaN stays in)
        if (!(v - dec.best > dec.t0 + dec.p * (ab + lq_abs(v)))) {      // not excluded (a NaN excludes nothing)
            int code;
            if constexpr (PACK) code = (int)(__float_as_uint(v) & ~keep_mask) * 32 + 16 * h + e;
            else code = reinterpret_cast<const int*>(wave_lds)[ln * LQ_DECIDE_STRIDE + 16 * h + e];
            nothing = nothing || code < 0 || code >= K;
#pragma unroll
            for (int q = 0; q < LQ_CAND_MAX; ++q)
                if (q == n) codes[q] = code;                             // (static indexing: the array stays in registers)
            ++n;
        }
    }
    head = nothing ? -1 : ((second_in || n > LQ_CAND_MAX) ? -2 : n);
    mask_out = mask;
}

// the slot of a listed row: the row and the screen's best candidate (half 0), the part's eight ints (both halves)
__device__ __forceinline__ void lq_emit_store(bool need, int slot /* valid in both halves */, int head, unsigned mask,
                                              const int (&codes)[LQ_CAND_MAX], int my_k, int64_t row, int* __restrict__ amb_list,
                                              int64_t N, int lane) {
    const int h = lane >> 5;
    if (h == 0 && need) {
        amb_list[slot] = (int)row;
        amb_list[lq_list_ints(N) + slot] = my_k;                        // the screen's best candidate: bounds the exact scan
    }
    if (!need || (size_t)slot >= lq_cand_cap(N)) return;
    int* out = amb_list + 2 * lq_list_ints(N) + (size_t)slot * 16 + 8 * h;
    // the part's eight ints {n, mask, c0 .. c5} as TWO 16-byte stores (the slot is 64-byte aligned, the part 32): eight scattered
    // dword stores per listed half-row kept the wave's vmcnt busy well into the next block (its z_q copy waits for vmcnt(0))
    static_assert(LQ_CAND_MAX == 6, "a part is {n, mask, six codes}");
    typedef int lq_i4 __attribute__((ext_vector_type(4)));
    lq_i4* out4 = reinterpret_cast<lq_i4*>(out);
    out4[0] = lq_i4{head, (int)mask, codes[0], codes[1]};
    out4[1] = lq_i4{codes[2], codes[3], codes[4], codes[5]};
}

template <bool PACK>
__device__ __forceinline__ void lq_screen_emit(const LqDecision& dec, bool certified, bool lists_ok, int my_k, int64_t row,
                                               bool row_valid, int* __restrict__ amb_count, int* __restrict__ amb_list, int64_t N,
                                               int K, int lane, unsigned keep_mask, const unsigned char* wave_lds) {
    const bool need = row_valid && !certified;                          // the same in both halves of the row
    if (__builtin_amdgcn_ballot_w64(need) == 0ull) return;              // wave-uniform: most waves have nothing to list
    const int ln = lane & 31, h = lane >> 5;
    // ONE atomic per wave for all its listed rows (the one-product screen lists several rows per wave and block; a slot per row by
    // its own atomicAdd serialised them on one address): the first listed lane reserves popcount slots.  It is issued FIRST and its
    // result used LAST: the candidates are worked out into registers while the reservation travels to L2 and back.
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(h == 0 && need);            // (bits 0 .. 31 only)
    const int leader = __builtin_ctzll(lm);
    int base = 0;
    if (lane == leader) base = lq_ws_reserve(amb_count, __builtin_popcountll(lm));
    int head;
    unsigned mask;
    int codes[LQ_CAND_MAX];
    lq_emit_candidates<PACK>(dec, lists_ok, K, lane, keep_mask, wave_lds, head, mask, codes);
    // ---- now the slot ----
    base = __shfl(base, leader, 64);
    int slot = base + __builtin_popcountll(lm & ((1ull << lane) - 1ull));
    slot = __shfl(slot, ln, 64);                                        // the row's other half learns the slot
    lq_emit_store(need, slot, head, mask, codes, my_k, row, amb_list, N, lane);
}

#ifndef LQ_LISTS_ALL_K
#define LQ_LISTS_ALL_K 2048       /* codebooks up to this size: the list kernel decides every listed row (no scanning-kernel launch) */
#endif
// One listed row decided by ONE WAVE (round 3; the body of nearest_lists_kernel, lipvq_screen.hip -- and, round 4, of the fused
// launch's in-place decisions, lq_screen_decide_inplace below): its eight 8-lane groups score eight candidates at once; lane j of a
// group keeps torch's accumulator j (features j, j + 8, ...: the k-ordered fma chain of lq_sqdist8; for the sum rule the four
// accumulators of lane column j, lq_sqdist32), the eight partials are added in lane order, roots compared, lower code among equal
// values.  cl = the row's 16 ints {n0, mask0, c0..c5, n1, mask1, c0..c5} as lq_emit_store lays them out (global memory or LDS), or
// nullptr (no list at all).  zload(f) = feature f of the row.  Returns the code in every lane, or -1: a long scan left to the
// scanning kernel (never with all_here).
template <int DCH, int DIST, typename ZLOAD>
__device__ __forceinline__ int lq_lists_row(ZLOAD zload, const float* __restrict__ cb, int K, const int* cl, int lane, bool all_here) {
    constexpr int D = DCH * 8;
    const int g = lane >> 3, j = lane & 7;
    int n0 = -1, n1 = -1;
    if (cl) { n0 = cl[0]; n1 = cl[8]; }
    const bool shortlist = n0 >= 0 && n1 >= 0 && n0 + n1 >= 1 && n0 <= LQ_CAND_MAX && n1 <= LQ_CAND_MAX;   // wave-uniform
    // lane masks (some lane's second minimum may be within the margin, or a part listed more than LQ_CAND_MAX codes): every
    // code congruent to a flagged lane mod 32 is a candidate -- popcount(mask) x K/32 of them, eight at a time like a short list
    const bool lanescan = !shortlist && n0 != -1 && n1 != -1 && (n0 == -2 || n1 == -2 || n0 > LQ_CAND_MAX || n1 > LQ_CAND_MAX);
    const unsigned lmask = lanescan ? (((unsigned)cl[1] & 0xffffu) | (((unsigned)cl[9] & 0xffffu) << 16)) : 0u;
    // ... as long as that is a few rounds of eight: a long scan would hold this wave for hundreds of dependent rounds while
    // the scanning kernel spreads a row over 64 slices (measured at K = 8192: 4.6 k such rows, 256 codes per flagged lane:
    // 290 us here against 101 us there)
    bool scan_here = lanescan && lmask != 0u && __builtin_popcount(lmask) * ((K + 31) / 32) <= 128;
    // all_here (round 4; codebooks of <= LQ_LISTS_ALL_K codes): the caller is the call's last word on the row -- whatever it came
    // with (a long lane scan, no list at all: every lane flagged) is scanned by this wave, eight codes a round.  Such rows are a
    // few per ten thousand, and a launch of the scanning kernel for them (usually for nothing: the count lives on the device)
    // cost every call 4.5 us.
    unsigned lm_eff = lmask;
    if (!shortlist && !scan_here) {
        if (!all_here) return -1;
        scan_here = true;
        lm_eff = (lanescan && lmask != 0u) ? lmask : 0xffffffffu;
    }
    const bool scanning = !shortlist;                                    // (wave-uniform; shortlist rows read their list)
    const int P = scanning ? __builtin_popcount(lm_eff) : 0;
    const int nc = scanning ? P * ((K + 31) / 32) : n0 + n1;
    float zv[DCH];
#pragma unroll
    for (int i = 0; i < DCH; ++i) zv[i] = zload(8 * i + j);
    float best_v = INFINITY;
    int best_k = 0x7fffffff;
    for (int c0 = 0; c0 < nc; c0 += 8) {
        const int ci = c0 + g;
        bool live = ci < nc;
        int code;
        if (scanning) {
            const int t = ci / P, w = ci - t * P;                          // the w-th flagged lane of tile t
            unsigned m = lm_eff;
            for (int q = 0; q < w; ++q) m &= m - 1;
            code = 32 * t + __builtin_ctz(m | 0x80000000u);
            live = live && code < K;
        } else {
            code = live ? (ci < n0 ? cl[2 + ci] : cl[10 + (ci - n0)]) : cl[n0 > 0 ? 2 : 10];
        }
        code = (code >= 0 && code < K) ? code : 0;                       // (lq_screen_emit lists valid codes only)
        const float* c = cb + (size_t)code * D;
        float s;
        if constexpr (DIST == LIPVQ_DIST_NORM) {
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < DCH; ++i) { const float d = zv[i] - c[8 * i + j]; a = lq_fma(d, d, a); }
            s = __shfl(a, 8 * g, 64);
#pragma unroll
            for (int l = 1; l < 8; ++l) s = s + __shfl(a, 8 * g + l, 64);
            s = lq_sqrt(s);
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < DCH; ++i) {
                const int q = (i < (DCH / 4) * 4) ? (i & 3) : 0;
                const float d = zv[i] - c[8 * i + j];
                acc[q] = acc[q] + d * d;
            }
            const float vl = ((acc[0] + acc[1]) + acc[2]) + acc[3];
            s = __shfl(vl, 8 * g, 64);
#pragma unroll
            for (int l = 1; l < 8; ++l) s = s + __shfl(vl, 8 * g + l, 64);
        }
        const float v = (live && s == s) ? s : INFINITY;                 // a NaN never wins
        const int kk = live ? code : 0x7fffffff;
        if (v < best_v || (v == best_v && kk < best_k)) { best_v = v; best_k = kk; }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best_v, off, 64);
        const int ok = __shfl_xor(best_k, off, 64);
        if (ov < best_v || (ov == best_v && ok < best_k)) { best_v = ov; best_k = ok; }
    }
    if (best_k < 0 || best_k >= K) {                                     // every value NaN: any valid code of the list
        const int fb = scanning ? __builtin_ctz(lm_eff) : cl[n0 > 0 ? 2 : 10];
        best_k = (fb >= 0 && fb < K) ? fb : 0;
    }
    return best_k;
}

// Round 4: the rows the screen does not certify are decided IN PLACE by the wave that screened them (codebooks of at most
// LQ_LISTS_ALL_K codes under the three-product screen: a few rows per thousand, mostly two candidates).  The parts lq_emit_candidates
// works out go to the wave's scratch in the list kernel's slot format instead of the workspace, and lq_lists_row -- the list
// kernel's own body: same candidates, same accumulator order, same tie rule -- reads them there; the row's z_e comes back from the
// rows this wave stored in phase A (those stores have retired: every hand-over of the screen loop waited behind them; agent-scope
// loads, since this CU's vector cache may hold nothing older of these lines but need not be asked).  The row then continues as a
// certified one (index, usage, z_q gather), so the call has no list kernel behind it: one launch less per call (6 us of a 65 us
// launch at 65 536 rows).  The number of rows decided this way is still counted into the live word and published as before
// (LLFQVAE_V4.last_exact_rows, the screen monitor).
template <bool PACK, int DCH, int DIST>
__device__ __forceinline__ bool lq_screen_decide_inplace(const LqDecision& dec, bool certified, bool lists_ok, int& my_k, int64_t row0,
                                                         bool row_valid, int* __restrict__ amb_count, const float* __restrict__ z,
                                                         const float* __restrict__ cb, int K, int lane, unsigned keep_mask,
                                                         unsigned char* wave_lds) {
    const bool need = row_valid && !certified;                          // the same in both halves of the row
    if (__builtin_amdgcn_ballot_w64(need) == 0ull) return certified;    // wave-uniform: most waves have nothing to decide
    const int ln = lane & 31, h = lane >> 5;
    unsigned long long lm = __builtin_amdgcn_ballot_w64(h == 0 && need);                  // (bits 0 .. 31 only)
    if (lane == __builtin_ctzll(lm)) (void)lq_ws_reserve(amb_count, __builtin_popcountll(lm));   // counted, nothing reserved
    int head;
    unsigned mask;
    int codes[LQ_CAND_MAX];
    lq_emit_candidates<PACK>(dec, lists_ok, K, lane, keep_mask, wave_lds, head, mask, codes);
    __builtin_amdgcn_wave_barrier();                                    // the decide image has been read (unpacked codes)
    int* parts = reinterpret_cast<int*>(wave_lds);                      // [32 rows][16 ints]: 2 KiB of the wave's LQ_DECIDE_BYTES
    static_assert(LQ_CAND_MAX == 6 && 32 * 16 * 4 <= LQ_DECIDE_BYTES, "a part is {n, mask, six codes}");
    if (need) {
        typedef int lq_i4 __attribute__((ext_vector_type(4)));
        lq_i4* out4 = reinterpret_cast<lq_i4*>(parts + ln * 16 + 8 * h);
        out4[0] = lq_i4{head, (int)mask, codes[0], codes[1]};
        out4[1] = lq_i4{codes[2], codes[3], codes[4], codes[5]};
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    constexpr int D = DCH * 8;
    while (lm) {                                                        // wave-uniform
        const int r = __builtin_ctzll(lm);
        lm &= lm - 1ull;
        const float* zr = z + (size_t)(row0 + r) * D;
        const int bk = lq_lists_row<DCH, DIST>([&](int f) { return __hip_atomic_load(zr + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
                                               cb, K, parts + r * 16, lane, true);
        if (ln == r) { my_k = bk; certified = true; }
    }
    __builtin_amdgcn_wave_barrier();                                    // the parts have been read before the scratch's next use
    return certified;
}

// z_q rows of certified rows: 16 lanes copy one codebook row (16 B each), 4 rows per pass
// z_q rows of certified rows: 16 lanes copy 64 floats of one codebook row (16 B each per pass), 4 rows per pass of the wave, 8
// passes = the wave's 32 rows; a "trip" covers floats [64 trip, 64 trip + 64) of the rows.
//  * lq_screen_gather: loads of all eight passes of a trip, then the stores (the first version waited for each pass's load
//    before storing: eight L2 round trips).
//  * lq_gather_dma / lq_gather_flush (fused kernel): the copy goes THROUGH LDS -- LDS-DMA with per-lane source addresses (the
//    gather) into a per-wave staging area, later LDS -> registers -> global stores -- so that no registers hold the rows while
//    the encoder's first layer runs in between (held in registers, hipcc spilled them to scratch and waited for every load).
struct LqGatherTrip { float4 val[8]; };
__device__ __forceinline__ void lq_gather_load(LqGatherTrip& g, const float* __restrict__ cb, int my_k, bool row_ok, int64_t row0,
                                               int64_t N, int D, int lane, int trip) {
    const int nvec = D / 4;
    const int v = 16 * trip + (lane & 15);
    const int vc = v < nvec ? v : nvec - 1;                     // always a valid address
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int src_lane = 4 * p + (lane >> 4);
        const int kk = __shfl(my_k, src_lane, 64);
        const bool ok = (__shfl((int)row_ok, src_lane, 64) != 0) && (row0 + src_lane < N);
        g.val[p] = reinterpret_cast<const float4*>(cb + (size_t)(ok ? kk : 0) * D)[vc];     // certified rows have 0 <= kk < K
    }
}
__device__ __forceinline__ void lq_gather_store(const LqGatherTrip& g, float* __restrict__ zq, bool row_ok, int64_t row0, int64_t N,
                                                int D, int lane, int trip) {
    const int nvec = D / 4;
    const int v = 16 * trip + (lane & 15);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int src_lane = 4 * p + (lane >> 4);
        const bool ok = (__shfl((int)row_ok, src_lane, 64) != 0) && (row0 + src_lane < N) && (v < nvec);
        if (ok) reinterpret_cast<float4*>(zq + (size_t)(row0 + src_lane) * D)[v] = g.val[p];
    }
}
// passes [p0, p0 + NP) of one trip: codebook bytes -> this wave's LDS staging area (NP KiB), no registers
template <int NP>
__device__ __forceinline__ void lq_gather_dma(unsigned char* wave_stage, const float* __restrict__ cb, int my_k, bool row_ok,
                                              int64_t row0, int64_t N, int D, int lane, int trip, int p0) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int nvec = D / 4;
    const int v = 16 * trip + (lane & 15);
    const int vc = v < nvec ? v : nvec - 1;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int src_lane = 4 * (p0 + p) + (lane >> 4);
        const int kk = __shfl(my_k, src_lane, 64);
        const bool ok = (__shfl((int)row_ok, src_lane, 64) != 0) && (row0 + src_lane < N);
        const float4* src = reinterpret_cast<const float4*>(cb + (size_t)(ok ? kk : 0) * D) + vc;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(wave_stage + p * 1024), 16, 0, 0);
    }
}
// ... and from the staging area to z_q (call after the copies have landed: vmcnt)
template <int NP>
__device__ __forceinline__ void lq_gather_flush(const unsigned char* wave_stage, float* __restrict__ zq, bool row_ok, int64_t row0,
                                                int64_t N, int D, int lane, int trip, int p0) {
    const int nvec = D / 4;
    const int v = 16 * trip + (lane & 15);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int src_lane = 4 * (p0 + p) + (lane >> 4);
        const bool ok = (__shfl((int)row_ok, src_lane, 64) != 0) && (row0 + src_lane < N) && (v < nvec);
        const float4 val = *reinterpret_cast<const float4*>(wave_stage + p * 1024 + lane * 16);
        // (nontemporal: an output stream of N x D floats that this launch never reads back; see the z_e stores in lipvq_fused.hip)
        typedef float lq_f4v __attribute__((ext_vector_type(4)));
        if (ok) __builtin_nontemporal_store((lq_f4v){val.x, val.y, val.z, val.w}, reinterpret_cast<lq_f4v*>(zq + (size_t)(row0 + src_lane) * D) + v);
    }
}
__device__ __forceinline__ void lq_screen_gather(const float* __restrict__ cb, float* __restrict__ zq, int my_k,
                                                 bool certified, int64_t row0, int64_t N, int D, int lane) {
#ifdef LQ_ABL_NOGATHER
    return;
#endif
    if (D & 3) {
        // rows that are not whole 16-byte vectors (any-width route): 16 lanes copy one row element by element, 4 rows per pass
        for (int p = 0; p < 8; ++p) {
            const int src_lane = 4 * p + (lane >> 4);
            const int kk = __shfl(my_k, src_lane, 64);
            const bool ok = (__shfl((int)certified, src_lane, 64) != 0) && (row0 + src_lane < N);
            if (ok)
                for (int d = lane & 15; d < D; d += 16) zq[(size_t)(row0 + src_lane) * D + d] = cb[(size_t)kk * D + d];
        }
        return;
    }
    for (int trip = 0; 64 * trip < D; ++trip) {
        LqGatherTrip g;
        lq_gather_load(g, cb, my_k, certified, row0, N, D, lane, trip);
        lq_gather_store(g, zq, certified, row0, N, D, lane, trip);
    }
}

// Which screen a shape runs by default (LIPVQ_SCREEN_MODE=coarse|fine overrides per launch: lipvq_screen.hip).  Measured on one box
// (profiles/r03_y_coarse_sweep_last_build.txt: 524 288 rows, whole call incl. the exact stage, one-product / three-product time):
//   D =  64: K = 1024 0.99, 4096 0.78, 8192 0.68      D = 128: K = 1024 0.87, 2048 0.76, 4096 0.65, 8192 0.57 (BASELINE config 3)
//   D = 208: K = 1024 0.91 (the reference's own widths), 4096 0.64, 8192 0.55
// The one-product screen trades two thirds of the matrix work for an exact stage over 5-10 % of the rows: it pays where the
// screen is most of the launch -- wide latents from the reference's default codebook size on, narrow ones against large codebooks
// (at D = 64, K = 1024 -- the metric's shape -- the two are level and the three-product screen stays).
static inline int lq_screen_coarse_default(int S, int K) { return (K >= 4096 || (S >= 8 && K >= 1024)) ? 1 : 0; }
int lq_screen_coarse(int S, int K);

// workspace of the screened routes: [64 B: counter] [row list: lq_list_ints(N) ints] [best-candidate list: the same]
// [short lists: 16 ints x lq_cand_cap(N)]
__host__ __device__ static inline size_t lq_list_ints(int64_t N) { return ((size_t)N + 15) & ~(size_t)15; }
// ... [slots the list kernel left to the scanning kernel: lq_list_ints(N) ints; their count is the header's second int]
__host__ __device__ static inline size_t lq_lists_bytes(int64_t N) { return sizeof(int) * (3 * lq_list_ints(N) + 16 * lq_cand_cap(N)); }
__host__ __device__ static inline size_t lq_slot2_offset_ints(int64_t N) { return 2 * lq_list_ints(N) + 16 * lq_cand_cap(N); }

// exact decision for listed rows (lipvq_screen.hip); z_by_slot: z is a compact [count][D] buffer
// dist: LIPVQ_DIST_NORM (v5:43-46: torch.norm's order, roots compared) or LIPVQ_DIST_SQSUM (vq:57-63: pow(2).sum's order)
int lipvq_launch_rows(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                      const int* amb_list, const int* amb_count, int64_t N, int K, int D, hipStream_t st,
                      int dist = LIPVQ_DIST_NORM);
// the same for rows whose z_e was never stored: recomputed from x with the raw (unpacked) encoder weights
// raw6 = {W0, b0, W1, b1, W2 (Lipschitz-normalised), b2}
int lipvq_launch_rows_encode(const float* x, const float* const* raw6, int A, const float* cb, int64_t* idx, float* zq,
                             int64_t* usage, const int* amb_list, const int* amb_count, int64_t N, int K, int D,
                             hipStream_t st);
#endif
