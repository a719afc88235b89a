// lipvq_fused.hip -- the BASELINE metric's path in ONE persistent launch:
//   x --encoder MLP + Lipschitz layer (fp32 MFMA, weights resident in LDS)--> z_e (registers only)
//     --centre, split to fp16 hi/lo (registers)--> MFMA screen against the prepared codebook (LDS-staged)
//     --certified rows: idx, z_q gather, usage;  uncertified rows: z_e row -> compact list for the exact kernel.
// Replaces reference backbone_lfqvae_v5.py:71-74 (encoder, to_latent, quantizer, z_latent) with the SAME
// results as lipvq_mlp3_f32 + lipvq_nearest_f32: phase A is mlp3_kernel's arithmetic (one k-ordered fmaf
// chain per output, lipvq_mlp.hip), phase B is screen_kernel's (lipvq_screen.hip).  z_e never touches HBM
// as an operand of the screen; it is written (coalesced 16-byte stores) only when the caller asks for it.  The
// <1 % of rows the screen cannot certify are re-encoded from x by the exact kernel (plain fmaf chains: same bits).
//
// Why fused: the two stand-alone kernels are latency bound (rocprofv3 PMC, profiles/r01_b: MFMA pipe 13-18 %
// busy, half of all wave cycles parked in s_waitcnt/barriers): mlp3 streams its A operands from L2 at one
// wave per SIMD, and both pay 268 MB of z_e traffic.  Here the weights sit in LDS (one 16-byte read feeds
// four MFMAs), two waves share each SIMD so one wave's GELU/bookkeeping VALU work runs beside the other's
// MFMAs, and a workgroup persists over row blocks so the weights are loaded once.
#include <string.h>
#include <atomic>

#include "lipvq_mlp.h"
#include "lipvq_screen.h"

#ifdef LQ_ABL_NOGELU            // ablation builds only (scripts/ablate.sh): wrong results, timing only
#define FUSED_GELU(v) (v)
#else
#define FUSED_GELU(v) lq_gelu(v)
#endif
#ifdef LQ_ABL_NOSIGMOID
#define FUSED_SIGMOID(v) (v)
#else
#define FUSED_SIGMOID(v) lq_sigmoid(v)
#endif

#ifndef FUSED_WAVES
#define FUSED_WAVES 8
#endif
#define FUSED_THREADS (FUSED_WAVES * 64)
// LDS budget: the D = 128 instance holds 105 KB of weights, so its codebook stages are one tile deep
#ifndef FUSED_HIST_MAX
#define FUSED_HIST_MAX 2048
#endif
// stage ring of the fused kernel (LDS left beside the encoder weights): S <= 4: four buffers of two/four tiles; S = 8: the
// weights take 98 KB, three buffers of one tile
// The Lipschitz layer's weights are STREAMED through the stage ring (instead of living in LDS) from this many k-steps on:
// 13 (D = 208) has no choice -- 112 KB of A operands; for 8 (D = 128) it is a trade measured in round 3: 64 KB of LDS go from
// a phase that is a twentieth of the work at K = 8192 to the codebook ring of the phase that is the rest.
#ifndef LQ_STREAM2_MIN_S
#define LQ_STREAM2_MIN_S 9
#endif
#ifndef LQ_RING8_TC
#define LQ_RING8_TC 1
#endif
#ifndef LQ_RING8_NB
#define LQ_RING8_NB 3
#endif
#ifdef LQ_EXP_RING_TC             /* experiment builds: the S = 4 instance's ring shape from the command line */
constexpr int fused_ring_tc(int S) { return S == 4 ? LQ_EXP_RING_TC : (S <= 2) ? 4 : (S <= 4) ? 2 : (S == 8) ? LQ_RING8_TC : 1; }
constexpr int fused_ring_nb(int S) { return S == 4 ? LQ_EXP_RING_NB : (S <= 4) ? 4 : (S == 8) ? LQ_RING8_NB : 3; }
#else
constexpr int fused_ring_tc(int S) { return (S <= 2) ? 4 : (S <= 4) ? 2 : (S == 8) ? LQ_RING8_TC : 1; }
#ifndef LQ_RING13_NB
#define LQ_RING13_NB 3
#endif
constexpr int fused_ring_nb(int S) { return (S <= 4) ? 4 : (S == 8) ? LQ_RING8_NB : LQ_RING13_NB; }
#endif
#ifndef LQ_COARSE_TC_WIDE
#define LQ_COARSE_TC_WIDE 2       /* tiles per stage of the one-product S = 8 instance (1 = as the three-product ring) */
#endif
// S = 8 (cfg3): 1.610 -> 1.555 ms, same box.  Not S = 13: with two tiles per stage that instance runs 10 ms instead of 0.9 (its
// 26-k-step stage body no longer fits the registers it has left beside 104 of row fragments) -- profiles/r04_f_coarse_ring_ab.txt
constexpr int fused_ring_tc_coarse(int S) { return S == 8 ? LQ_COARSE_TC_WIDE * fused_ring_tc(S) : fused_ring_tc(S); }
#ifndef LQ_EXP_WGS_PER_CU
#define LQ_EXP_WGS_PER_CU 1
#endif

// lq_gelu_poly2 (two elements per instruction: v_pk_fma_f32), cut into four stages so that one stage can follow each MFMA of a
// 4-MFMA group (same operations in the same order as lq_gelu_poly: bit-identical values)
struct Gelu2 {
    lq_v2f x, t, u, s;
    __device__ __forceinline__ void stage0(float a, float b) {
        x = (lq_v2f){a, b};
        t = x * x;
        u = lq_fma2(t, lq_bc2(0.11111111111111111111f), lq_bc2(-1.0f));
        s = lq_fma2(lq_bc2(0.00012666420661844313f), u, lq_bc2(-0.00043783686123788357f));
        s = lq_fma2(s, u, lq_bc2(0.0008924771682359278f));
    }
    __device__ __forceinline__ void stage1() {
        s = lq_fma2(s, u, lq_bc2(-0.002175821689888835f));
        s = lq_fma2(s, u, lq_bc2(0.005515238270163536f));
        s = lq_fma2(s, u, lq_bc2(-0.01217574905604124f));
        s = lq_fma2(s, u, lq_bc2(0.02415713667869568f));
    }
    __device__ __forceinline__ void stage2() {
        s = lq_fma2(s, u, lq_bc2(-0.043842192739248276f));
        s = lq_fma2(s, u, lq_bc2(0.07253222167491913f));
        s = lq_fma2(s, u, lq_bc2(-0.11009667813777924f));
        s = lq_fma2(s, u, lq_bc2(0.15749694406986237f));
    }
    __device__ __forceinline__ void stage3(float& o0, float& o1) {
        s = lq_fma2(s, u, lq_bc2(-0.2287982553243637f));
        s = lq_fma2(s, u, lq_bc2(0.4701318144798279f));
        const lq_v2f o = lq_fma2(t * lq_bc2(0.35355339059327376220f), s, lq_bc2(0.5f) * x);
        o0 = o.x; o1 = o.y;
    }
};

// the hidden layers' activation in the staged form of Gelu2: the ReLU instance needs no stages
template <bool RELU>
struct ActStage : Gelu2 {};
template <>
struct ActStage<true> {
    float a_, b_;
    __device__ __forceinline__ void stage0(float a, float b) { a_ = a; b_ = b; }
    __device__ __forceinline__ void stage1() {}
    __device__ __forceinline__ void stage2() {}
    __device__ __forceinline__ void stage3(float& o0, float& o1) { o0 = a_ > 0.0f ? a_ : 0.0f; o1 = b_ > 0.0f ? b_ : 0.0f; }
};

template <int S, bool FAST>
__host__ __device__ static size_t fused_lds_bytes_nohist(int A) {
    constexpr int T0 = 2, T1 = 4, T2 = (S + 1) / 2;
    const int S0q = ((A + 1) / 2 + 3) / 4;
    const int S0h = (A + 15) / 16;
    size_t fl = FAST ? (size_t)T0 * S0h * 256 + (size_t)T1 * (2 * T0) * 256 + (size_t)T2 * (2 * T1) * 256
                     : (size_t)T0 * S0q * 256 + (size_t)T1 * 8 * 256 + (S >= LQ_STREAM2_MIN_S ? (size_t)0 : (size_t)T2 * 16 * 256);
    fl += 32 * T0 + 32 * T1 + 32 * T2 + 16 * S;
    // (one size for both screens' instances: the larger ring)
    constexpr size_t ring_fine = (size_t)fused_ring_nb(S) * ScreenCfg<S, fused_ring_tc(S), false>::STAGE_BYTES + 1024;
    constexpr size_t ring_coarse = (size_t)fused_ring_nb(S) * ScreenCfg<S, fused_ring_tc_coarse(S), true>::STAGE_BYTES + 1024;
    const size_t ring = ((ring_fine > ring_coarse ? ring_fine : ring_coarse) + 63) & ~(size_t)63;
    return fl * sizeof(float) + ring;
}
// the per-workgroup usage histogram: codebooks of up to FUSED_HIST_MAX codes, where the LDS has room for it
template <int S, bool FAST>
__host__ __device__ static bool fused_hist_fits(int A, int K) {
    return K <= FUSED_HIST_MAX && fused_lds_bytes_nohist<S, FAST>(A) + (size_t)K * 4 <= (size_t)160 * 1024;
}
template <int S, bool FAST>
static size_t fused_lds_bytes(int A, int K) {
    return fused_lds_bytes_nohist<S, FAST>(A) + (fused_hist_fits<S, FAST>(A, K) ? (size_t)K * 4 : 0);
}

struct TokArgs {
    const float* x;              // [N][A]
    const float* packed;         // lipvq_mlp3_pack_f32 of (A -> 64 -> 128 -> D)
    const unsigned char* packed16;  // fast mode: lipvq_mlp3_pack_f16_f32 of the same stack (NULL in parity mode)
    const unsigned char* prep;   // lipvq_nearest_prepare_f32 of the codebook
    const float* cb;             // [K][D]
    int64_t* idx;                // [N]
    float* zq;                   // [N][D] or NULL
    unsigned long long* usage;   // [K] or NULL
    float* ze_out;               // [N][D] or NULL
    int* amb_count;              // workspace[0]
    int* amb_list;               // [N]
    const float* w2q;            // layer-2 weights re-laid out for streaming (S > 8 only; lives in the workspace)
    float* pre0;                 // TRAIN instances: the three pre-activations [N][64], [N][128], [N][D] the backward needs
    float* pre1;
    float* pre2;
    int64_t N;
    int A, D, K;
    float gamma;
    int coarse;                  // host side only: the one-product screen instance runs (ze_out is then never NULL)
    int inplace;                 // uncertified rows are decided by the wave that screened them (needs ze_out; lipvq_screen.h)
    int defer_ze;                // S <= 4: the last tile's z_e stores go behind the screen's first stage copies   } launch-uniform schedule choices,
    int nt_ze;                   // z_e rows are stored nontemporal                                                } same results: lq_schedule()
    int ze_ring;                 // ze_out is a RING: a wave's 32 rows of every row block go to the same 32 rows (in-place launches whose
                                 // caller wants no z_e: lq_ze_ring) -- the rows live in L2 until the wave has decided them
};

#ifndef LQ_PROLOGUE_DMA_MIN_S
#define LQ_PROLOGUE_DMA_MIN_S 99  /* measurement knob: instances whose layer-1/2 weights travel by LDS-DMA under the first layer 0 (none: see the prologue) */
#endif
#ifdef LQ_CT_SCHEDULE             /* measurement builds: the two schedule choices as compile-time constants (defer_ze | nt_ze << 1) */
#define LQ_DEFER_FLAG(a) ((LQ_CT_SCHEDULE & 1) != 0)
#define LQ_NT_FLAG(a) ((LQ_CT_SCHEDULE & 2) != 0)
#else
#define LQ_DEFER_FLAG(a) ((a).defer_ze != 0)
#define LQ_NT_FLAG(a) ((a).nt_ze != 0)
#endif

// T0 = 2 (64 features), T1 = 4 (128 features): the reference's encoder widths (v5:54-59).
// FAST: the encoder's three GEMMs as fp16 MFMAs (v_mfma_f32_32x32x16_f16, fp32 accumulation) -- the "fast" mode
// of SURVEY section 7 / BASELINE config 2's half-precision encoder: NOT bit-identical to the oracle (a fraction of a
// percent of the indices differ, all between near-equidistant codes); everything after z_e is the parity path.
// TRAIN: also stores the pre-activations of the three layers (16-byte stores, lq_tile_store16) -- the forward half of a training
// step in this one launch instead of mlp3_wg_kernel (with saved pre-activations) + the stand-alone screen + its z_e round trip.
// RG, WAVES (round 3): a wave owns RG groups of 32 rows per iteration (phase A runs once per group, phase B multiplies all of
// them against every codebook fragment it reads: lq_screen_core_rg), WAVES waves per workgroup.  (8, RG = 1) is the round-2
// kernel; (8, 2) halves the LDS fragment reads, |e'|^2 reads and stage hand-overs per MFMA at two waves per SIMD; (4, 2) is
// ONE wave per SIMD with the whole 512-register file (the D = 128 instance: its A fragments alone are 64 registers per group).
// COARSE (round 3): phase B runs the one-product screen (lq_screen_core_rg<.., COARSE>): a third of the matrix work, lower-bound
// bookkeeping, 10-40 % of the rows left to the exact kernel with their two or three candidates -- which then needs z_e, so the
// host always passes a z_e buffer in this mode (a.ze_out).
// VQ (round 3): the plain VQVAE's encoder in the same launch (reference backbone.py:14-21: three Linear + ReLU, the last one on the
// latent itself) -- ReLU instead of GELU / GELU / sigmoid, no Lipschitz scale (the packed weights are the plain ones).  A ReLU
// latent is unbounded, so the launch-wide fp16 scale of the sigmoid instance does not exist: the row keeps its centred latent
// in fp32 until all tiles are finished and is then split with its OWN power of two, as the stand-alone screen does.
template <int S, bool FAST, bool TRAIN, int RG, int WAVES, bool COARSE = false, bool VQ = false>
__device__ __forceinline__ void tokenize_body(const TokArgs& a) {
    static_assert(!(FAST && TRAIN), "training uses the parity arithmetic");
    static_assert(!VQ || (RG == 1 && !FAST), "the ReLU instance: one row group, parity arithmetic");
    static_assert(RG == 1 || RG == 2, "one or two row groups per wave");
    constexpr int THREADS = WAVES * 64;
    // the one-product screen stages hi-only tiles (half the bytes): twice the tiles per stage in the same LDS, i.e. half the
    // stage hand-overs (vmcnt wait + workgroup barrier + DMA issue) per tile where a stage was ONE tile (S = 8)
    constexpr int TCF = COARSE ? fused_ring_tc_coarse(S) : fused_ring_tc(S), NBF = fused_ring_nb(S);
    using RingCfg = ScreenCfg<S, TCF, COARSE>;
    constexpr size_t SLAB_STRIDE = ScreenCfg<S, fused_ring_tc(S), false>::STAGE_BYTES;   // the streamed layer-2 slabs' three buffers (either screen)
    constexpr int T0 = 2, T1 = 4, T2 = (S + 1) / 2;         // S odd (D = 208): the last 32-feature tile is half used
    constexpr int S1 = 16 * T0, S2 = 16 * T1;               // k-steps (pairs) of layers 1 and 2
    // STREAM2: the Lipschitz layer's weights (T2 x 16 KB of fp32 MFMA A operands: 112 KB at D = 208) do not fit beside the stage
    // ring, so they are streamed, one 16 KB output-tile slab at a time, through that ring -- which is idle during the encoder
    // phase -- by the same LDS-DMA + counted-vmcnt mechanism as the codebook (all eight waves work on the same tile).
    constexpr bool STREAM2 = !FAST && S >= LQ_STREAM2_MIN_S;
    constexpr bool PROLOGUE_DMA = !FAST && S >= LQ_PROLOGUE_DMA_MIN_S;    // layers 1 / 2 copied by LDS-DMA under the first layer 0 (prologue)
    static_assert(!FAST || (S % 2 == 0 && S <= 8), "fast mode: D in {32, 64, 128}");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef LQ_STAMPS
    const long long st_top = __builtin_amdgcn_s_memtime();            // kernel entry: the prologue is [st_top, st_begin)
    const long long st_top_rt = (long long)__builtin_amdgcn_s_memrealtime();      // (100 MHz, the same clock on every CU)
#endif

    const PackedLayout PL = packed_layout(a.A, 32 * T0, 32 * T1, 16 * S);
    const PrepLayout L = prep_layout(a.K, a.D);
    const int S0 = PL.S0;
    const int S0q = (S0 + 3) / 4;
    // LDS carve (floats unless noted).  FAST: the three weight blocks are fp16 MFMA fragments instead:
    // [T][k-steps of 16][64 lanes][8 halfs]
    const int S0h = (a.A + 15) / 16;                                   // fp16 k-steps of layer 0
    float* w_P0 = reinterpret_cast<float*>(lds);                       // [T0][S0q][64][4]
    float* w_B0 = w_P0 + (FAST ? T0 * S0h * 256 : T0 * S0q * 256);     // [32*T0]
    float* w_P1 = w_B0 + 32 * T0;                                      // [T1][S1/4][64][4]
    float* w_B1 = w_P1 + (FAST ? T1 * (2 * T0) * 256 : T1 * (S1 / 4) * 256);
    float* w_P2 = w_B1 + 32 * T1;                                      // [T2][S2/4][64][4]
    float* w_B2 = w_P2 + (FAST ? T2 * (2 * T1) * 256 : STREAM2 ? 0 : T2 * (S2 / 4) * 256);
    float* w_mu = w_B2 + 32 * T2;                                      // [16*S]
    unsigned char* stage0 = reinterpret_cast<unsigned char*>(w_mu + 16 * S);   // 2 stage buffers (also the
                                                                                // per-wave transpose slices of the decision)
    // per-workgroup usage histogram (K <= FUSED_HIST_MAX): LDS atomics per row, ONE global atomic per non-empty
    // bin at the end of this persistent workgroup -- skewed code distributions would otherwise serialise on a
    // few global addresses (see lq_usage_add)
    constexpr size_t RING_OWN = (size_t)NBF * RingCfg::STAGE_BYTES + 1024, RING_FINE = lq_ring_bytes<S, fused_ring_tc(S), NBF>();
    constexpr size_t STAGES_BYTES = RING_OWN > RING_FINE ? RING_OWN : RING_FINE;      // the ring + its dummy KiB (one LDS carve for both screens)
    static_assert(STAGES_BYTES >= (size_t)WAVES * LQ_DECIDE_BYTES, "the decision's per-wave transposes live in the stage ring");
    // per-wave private slice of the (idle) ring: the decision's transposes (LQ_DECIDE_BYTES) and the z_q copy's staging (GP KiB)
    constexpr size_t WSLICE = (STAGES_BYTES >= (size_t)WAVES * 8192) ? 8192 : LQ_DECIDE_BYTES;
    static_assert(LQ_DECIDE_BYTES >= 4096 && WSLICE >= LQ_DECIDE_BYTES, "a slice holds the transposes and four staged KiB");
    unsigned* hist = reinterpret_cast<unsigned*>(stage0 + ((STAGES_BYTES + 63) & ~(size_t)63));
    const bool use_hist = a.usage && fused_hist_fits<S, FAST>(a.A, a.K);         // (the same rule sizes the launch's LDS)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (scalar row/address arithmetic)
    const int ln = lane & 31, h = lane >> 5;

    // (round 4: the launch-wide scalars and the first row block's inputs are requested BEFORE the weight copy -- their scalar /
    // HBM round trips used to start behind the prologue's barrier, in front of the first layer-0 MFMA)
    const unsigned* hdr = reinterpret_cast<const unsigned*>(a.prep);
    const unsigned char* tiles = a.prep + (COARSE ? L.o_tiles_hi : L.o_tiles);     // (the one-product screen stages hi-only tiles)
    const size_t tile_bytes = COARSE ? L.tile_bytes_hi : L.tile_bytes;
    // Scale of the fp16 split.  The stand-alone screen scales every latent row by its own power of two (block floating point,
    // lipvq_screen.h: it must cope with latents of any magnitude).  Here z_e is a sigmoid output, so |z_e - mu| <= 1 + max|mu|
    // =: Bz for EVERY row, and ONE power of two fz (Bz fz in [2^13, 2^14)) serves the whole launch: the split happens tile by
    // tile as the encoder finishes (no fp32 copy of z_e, no second pass, one row factor instead of sixteen).  What a per-row
    // scale bought -- "lo" pieces staying normal fp16 numbers for rows of small magnitude -- is replaced by a rule: a row with
    // |z'|^2 below tiny2 (its elements' 2^-25 fz^-1 absolute split error would no longer be 2^-20 |z'| Emax / sqrt(D)) is
    // never certified and goes to the exact kernel.  No such row exists unless z_e collapses onto the codebook mean.
    const int sz = lq_scale_exp(1.0f + __uint_as_float(hdr[4]));
    const float fz = lq_pow2f(sz);
    const float fown = lq_pow2f(sz + (int)hdr[3]);
    const float tiny2 = (float)(16 * S) * lq_pow2f(-10 - 2 * sz);
    const int64_t nblk = (a.N + WAVES * RG * 32 - 1) / (WAVES * RG * 32);
    // this lane's inputs of the NEXT row block (k = 2 q + h, q < XPF), fetched a whole screening phase ahead: the first version
    // loaded each x value right in front of the MFMA that consumed it (four serialised HBM round trips per block)
    constexpr int XPF = 8;                                   // fan-in up to 16 is prefetched; wider inputs load at block start
    const bool x_pref = !FAST && a.A <= 2 * XPF;
    float xqg[RG][XPF];
#ifndef LQ_LANE_W_MIN_S
#define LQ_LANE_W_MIN_S 8        /* measurement knob: the instances that re-form lane-dependent addresses per use (99: none); at S = 4 nothing spills */
#endif
    auto load_x = [&](int64_t blk_, const int g_, float (&xq)[XPF]) {
        int64_t r_ = ((blk_ * WAVES + wave) * RG + g_) * 32 + ln;
        r_ = r_ < a.N ? r_ : a.N - 1;
        const int64_t last = a.N * a.A - 1;
        // (S >= 8: the lane half is opaque here -- hipcc had hoisted the XPF partial addresses `x + 4 k` out of the block loop and
        // spilled them; each reload's `vmcnt(0)` sat behind the x load issued just before it: XPF global loads in series per block)
        int h_x = h;
        if constexpr (S >= LQ_LANE_W_MIN_S) asm volatile("" : "+v"(h_x));
#pragma unroll
        for (int q = 0; q < XPF; ++q) {
            const int k = 2 * q + h_x;
            int64_t i_ = r_ * a.A + k;
            i_ = i_ < last ? i_ : last;                      // always a valid address; masked below (no divergent branch)
            const float v = a.x[i_];
            xq[q] = (k < a.A) ? v : 0.0f;
        }
    };
    if (x_pref) {
#pragma unroll
        for (int g_ = 0; g_ < RG; ++g_) load_x(blockIdx.x, g_, xqg[g_]);
    }
    lq_ws_begin(a.amb_count);
    // ---- once per workgroup: weights (re-laid out 4 k-steps per 16-byte LDS read), biases, mu --------
    {
        // the small vectors first (one element per thread each, clamped index: unconditional loads, all in flight together with
        // the weight loads below; round 4 -- each used to be its own load / wait / store round trip)
        static_assert(32 * T1 <= THREADS && 32 * T2 <= THREADS && 16 * S <= THREADS, "one element per thread");
        const float* mu = reinterpret_cast<const float*>(a.prep + L.o_mu);
        auto bias_src = [&](size_t o, int n) { const int i = tid < n ? tid : n - 1; return a.packed[o + 32 * (i >> 5) + 2 * (i & 15) + ((i >> 4) & 1)]; };
        const float vb0 = bias_src(PL.oB0, 32 * T0), vb1 = bias_src(PL.oB1, 32 * T1), vb2 = bias_src(PL.oB2, 32 * T2);
        const float vmu = mu[tid < 16 * S ? tid : 16 * S - 1];
        if (FAST) {
            // packed16 = [P0h | P1h | P2h], each already [t][s][lane][8 halfs] = 4 floats per (t, s, lane)
            const float* src = reinterpret_cast<const float*>(a.packed16);
            const int n0 = T0 * S0h * 256, n1 = T1 * (2 * T0) * 256, n2 = T2 * (2 * T1) * 256;
            // 16-byte pieces, four loads in flight per thread and round (the element loop was one L2 round trip per float)
            auto copy16 = [&](float* dst, const float* from, int nfl) {
                const float4* s4 = reinterpret_cast<const float4*>(from);
                float4* d4 = reinterpret_cast<float4*>(dst);
                const int nv = nfl / 4;
                for (int v0 = tid; v0 < nv; v0 += 4 * THREADS) {
                    float4 r[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int v = v0 + q * THREADS; r[q] = s4[v < nv ? v : nv - 1]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int v = v0 + q * THREADS; if (v < nv) d4[v] = r[q]; }
                }
            };
            copy16(w_P0, src, n0);
            copy16(w_P1, src + n0, n1);
            copy16(w_P2, src + n0 + n1, n2);
        } else {
            // Round 4: every load of the prologue is issued before the first LDS store.  The first version copied element by
            // element (`w[i] = P[..]`: hipcc emits load, s_waitcnt vmcnt(0), ds_write_b32 per iteration -- 16 + 16 dependent L2
            // round trips per thread, then five more for biases, mu and the histogram): 22 k cycles = 11 us of a 65 536-row
            // shard's 62 us launch (profiles/r04_a_stamps_shards.txt).  The packed stack is [t][s][64 lanes]; the LDS image is
            // [t][s / 4][64 lanes][4 k-steps], so one image entry (16 bytes) = four loads 64 floats apart, whole 256-byte
            // lines per wave-instruction.
            constexpr int NV1 = T1 * (S1 / 4) * 64, NV2 = STREAM2 ? 0 : T2 * (S2 / 4) * 64;      // 16-byte entries
            constexpr int R1 = (NV1 + THREADS - 1) / THREADS, R2 = (NV2 + THREADS - 1) / THREADS;
            const float* P0 = a.packed + PL.oP0;
            const float* P1 = a.packed + PL.oP1;
            const float* P2 = a.packed + PL.oP2;
            // (round 4, measured, NOT the default: S >= LQ_PROLOGUE_DMA_MIN_S) layers 1 and 2 by LDS-DMA, issued BEHIND the loads of
            // layer 0 / biases / mu / x (vmcnt retires in order: what layer 0 needs is older and is waited for alone) and waited for
            // only where layer 1 first reads them -- the first row block's layer 0 runs under the copy.  One wave-instruction moves 16
            // image entries: lane l brings component l & 3 of entry 16 c + (l >> 2).  Same box, against the register-staged copy in
            // front of one barrier (profiles/r04_l_prologue_dma_ab.txt): cfg2 0.3884 -> 0.3922 ms (0.0592 -> 0.0607 at 65 536 rows: the
            // dword-granular copy is slower than 16-byte register stores and layer 0 too short to hide it); icrt within the noise.
            constexpr int RR1 = PROLOGUE_DMA ? 1 : R1, RR2 = PROLOGUE_DMA ? 1 : (R2 > 0 ? R2 : 1);
            float4 r1[RR1], r2[RR2];
            if constexpr (!PROLOGUE_DMA) {
#pragma unroll
                for (int it = 0; it < R1; ++it) {
                    int v = tid + it * THREADS;
                    v = v < NV1 ? v : NV1 - 1;
                    const float* src = P1 + ((size_t)(v >> 6) * 4) * 64 + (v & 63);
                    r1[it] = make_float4(src[0], src[64], src[128], src[192]);
                }
#pragma unroll
                for (int it = 0; it < R2; ++it) {
                    int v = tid + it * THREADS;
                    v = v < NV2 ? v : NV2 - 1;
                    const float* src = P2 + ((size_t)(v >> 6) * 4) * 64 + (v & 63);
                    r2[it] = make_float4(src[0], src[64], src[128], src[192]);
                }
            }
            // layer 0 (fan-in A: S0 k-steps, padded to whole groups of four with zeros), entry v = (t * S0q + sq) * 64 + l
            const int NV0 = T0 * S0q * 64;
            auto p0_entry = [&](int v0, float (&e)[4]) {
                const int l = v0 & 63, sq = (v0 >> 6) % S0q, t = (v0 >> 6) / S0q;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int s = 4 * sq + q;
                    const float v = P0[((size_t)t * S0 + (s < S0 ? s : S0 - 1)) * 64 + l];          // (always a valid address)
                    e[q] = (s < S0) ? v : 0.0f;
                }
            };
            float e0[4];
            p0_entry(tid < NV0 ? tid : NV0 - 1, e0);                      // (one entry per thread up to fan-in 16: loads first ...)
            if constexpr (PROLOGUE_DMA) {
                typedef __attribute__((address_space(3))) void* lds_ptr_t;
                typedef const __attribute__((address_space(1))) void* glb_ptr_t;
                auto dma_image = [&](float* dstw, const float* P, int NV) {
                    for (int c = wave; c < NV / 16; c += WAVES) {
                        const int v = 16 * c + (lane >> 2);
                        const float* src = P + ((size_t)(v >> 6) * 4 + (lane & 3)) * 64 + (v & 63);
                        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dstw + 64 * c), 4, 0, 0);
                    }
                };
                static_assert(NV1 % 16 == 0 && NV2 % 16 == 0, "whole wave-instructions");
                dma_image(w_P1, P1, NV1);                                 // ... then the copies (younger than every load above)
                if (NV2 > 0) dma_image(w_P2, P2, NV2);
            }
            if (tid < NV0) *reinterpret_cast<float4*>(w_P0 + (size_t)tid * 4) = make_float4(e0[0], e0[1], e0[2], e0[3]);
            for (int v0 = tid + THREADS; v0 < NV0; v0 += THREADS) {       // (fan-in > 16 only)
                float e[4];
                p0_entry(v0, e);
                *reinterpret_cast<float4*>(w_P0 + (size_t)v0 * 4) = make_float4(e[0], e[1], e[2], e[3]);
            }
            if constexpr (!PROLOGUE_DMA) {
#pragma unroll
                for (int it = 0; it < R1; ++it) {
                    const int v = tid + it * THREADS;
                    if (v < NV1) *reinterpret_cast<float4*>(w_P1 + (size_t)v * 4) = r1[it];
                }
#pragma unroll
                for (int it = 0; it < R2; ++it) {
                    const int v = tid + it * THREADS;
                    if (v < NV2) *reinterpret_cast<float4*>(w_P2 + (size_t)v * 4) = r2[it];
                }
            }
        }
        // biases re-laid out [t][h][r] = b[32 t + 2 r + h]: the 16 values of a lane's accumulator tile are 64 contiguous bytes
        // (four 16-byte LDS reads, broadcast within the half-wave) instead of sixteen 4-byte reads
        if (tid < 32 * T0) w_B0[tid] = vb0;
        if (tid < 32 * T1) w_B1[tid] = vb1;
        if (tid < 32 * T2) w_B2[tid] = vb2;
        if (tid < 16 * S) w_mu[tid] = vmu;
        if (use_hist)
            for (int i = tid; i < a.K; i += THREADS) hist[i] = 0u;
    }
    if constexpr (PROLOGUE_DMA) lq_wg_barrier();      // (LDS stores only: the layer-1/2 copies stay in flight -- waited for behind the first layer 0)
    else __syncthreads();

#ifdef LQ_EXP_STAGGER             /* experiment: the workgroup in the SIMDs' odd wave slots starts LQ_EXP_STAGGER cycles late */
    if (__builtin_amdgcn_s_getreg(6148) & 1) {
        const long long t0_ = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0_ < LQ_EXP_STAGGER) __builtin_amdgcn_s_sleep(8);
    }
#endif

#ifdef LQ_STAMPS
    // diagnostic build only (scripts/stamps.py): per-wave cycles per segment, accumulated over the row blocks and written to
    // the unused upper half of the row list; no output value depends on them
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    const long long st_begin = __builtin_amdgcn_s_memtime();
#define LQ_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define LQ_STAMP(i) do { } while (0)
#endif
    // the z_q copy of a row block is deferred to the start of the NEXT block (parity kernel) and routed through LDS (lq_gather_dma):
    // its first round overlaps layer 0, further rounds (wider latents) follow
#ifdef LQ_NO_DEFER_GATHER
    constexpr bool DEFER_GATHER = false;
#else
    constexpr bool DEFER_GATHER = !FAST;
#endif
    int pend_kg[RG];
    bool pend_okg[RG];
    int64_t pend_row0g[RG];
    bool have_pend = false;
#pragma unroll
    for (int g_ = 0; g_ < RG; ++g_) { pend_kg[g_] = 0; pend_okg[g_] = false; pend_row0g[g_] = 0; }
    // z_e row store of one finished 32-feature tile.  Lane (n, h) holds features 32t + 2r + h (r = 0..15) of row n: the even ones in
    // the low half-wave, the odd ones in the high half.  One v_permlane32_swap per register pair (low half's r >= 8 <-> high
    // half's r < 8) leaves the low lane with features 32t .. 32t+15 and the high lane with 32t+16 .. 32t+31, i.e. four 16-byte
    // stores of consecutive floats per lane (64 contiguous bytes per lane) instead of sixteen 4-byte stores scattered over the row.
    int64_t ze_shift = 0;          // ring mode: rows of row block blk are stored ze_shift rows lower (set per block)
    auto store_ze_tile = [&](const int64_t row, const int t, const f32x16& acc) {
            {
                float lo8[8], hi8[8];                 // after the swaps: lo8[j] = feature base + 2j, hi8[j] = base + 2j + 1
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // vdst = low half's register r = 8 + j (goes up), src = high half's register r = j (comes down)
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j]), __float_as_uint(acc[8 + j]), false, false);
                    // sw[0]: lanes 0-31 keep acc[j] (own even features 2j), lanes 32-63 receive low half's acc[8+j]
                    // sw[1]: lanes 0-31 receive high half's acc[j] (odd features 2j+1), lanes 32-63 keep acc[8+j]
                    lo8[j] = __uint_as_float(sw[0]);
                    hi8[j] = __uint_as_float(sw[1]);
                }
                // low lanes : lo8[j] = feat 2j (own, even), hi8[j] = feat 2j+1 (from the high lane)        -> base 32t
                // high lanes: lo8[j] = feat 16+2j (from the low lane, even), hi8[j] = feat 16+2j+1 (own) -> base 32t+16
                if (row < a.N && 2 * t + h < S) {                // (the high lanes' 16 features of a half-used last tile do not exist)
                    float4* dst = reinterpret_cast<float4*>(a.ze_out + (size_t)(row - ze_shift) * a.D + 32 * t + 16 * h);
                    // nontemporal (round 4): written once, read back for a fraction of a percent of the rows -- the stream should not
                    // displace the codebook tiles and weights every workgroup re-reads from L2 (with the z_q rows the same way:
                    // cfg2 -2.4 %, icrt -2.9 %, same box, profiles/r04_e_nt_stores_ab.txt)
                    typedef float lq_f4v __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (S > 4 || TRAIN || LQ_NT_FLAG(a)) __builtin_nontemporal_store((lq_f4v){lo8[2 * q], hi8[2 * q], lo8[2 * q + 1], hi8[2 * q + 1]}, reinterpret_cast<lq_f4v*>(dst) + q);
                        else reinterpret_cast<lq_f4v*>(dst)[q] = (lq_f4v){lo8[2 * q], hi8[2 * q], lo8[2 * q + 1], hi8[2 * q + 1]};   // (launch-uniform: TokArgs; S > 4 and the training instances: always nontemporal -- nothing to choose there, and the second copy of the stores cost icrt 1.4 %)
                }
            }
    };
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
#ifdef LQ_STAMPS
        st_prev = __builtin_amdgcn_s_memtime();
#endif
        ze_shift = a.ze_ring ? (blk - (int64_t)blockIdx.x) * (WAVES * RG * 32) : 0;
        f16x8 ahg[RG][S], alg[RG][S];
        // z_e of each group's LAST layer-2 tile, stored behind the screen's first copies -- in the instances with registers to spare
        // (S <= 4: cfg2 0.4006 -> 0.3944 ms; at S = 13 the sixteen registers held across the end of layer 2 cost more than the
        // wait they remove: icrt 0.8945 -> 0.9149, profiles/r04_g_ze_store_placement.txt)
#ifndef LQ_DEFER_ZE_MAX_S
#define LQ_DEFER_ZE_MAX_S 4
#endif
        constexpr bool DEFER_ZE = S <= LQ_DEFER_ZE_MAX_S && !TRAIN;
        f32x16 zdefg[RG];
        float n2g[RG], a2g[RG], fzg[RG], fowng[RG];
      // ---- phase A of row group GC: the round-2 block body, on this group's rows / fragments / pending z_q copy ----
      auto encode_group = [&](auto GC) {
        constexpr int g = decltype(GC)::value;
        // the lane index the LDS weight reads are addressed with, opaque per row block in the instances that spill (S >= 8): as loop
        // invariants hipcc kept the lanes' LDS addresses in registers from the kernel's first lines and spilled them -- a reload is a
        // vector-memory load, and `s_waitcnt vmcnt(0)` in front of its use also waits for every store and copy in flight
        int lane_w = lane;
        if constexpr (S >= LQ_LANE_W_MIN_S) asm volatile("" : "+v"(lane_w));
        const int h_w = lane_w >> 5;                         // (the centring vector's LDS reads: not hoisted out of the block loop either)
        const int64_t row0 = ((blk * WAVES + wave) * RG + g) * 32;
        const int64_t row = row0 + ln;
        const int64_t rowc = row < a.N ? row : a.N - 1;
        f16x8 (&ah)[S] = ahg[g];
        f16x8 (&al)[S] = alg[g];
        const float (&xq)[XPF] = xqg[g];
        const int pend_k = pend_kg[g];
        const bool pend_ok = pend_okg[g];
        const int64_t pend_row0 = pend_row0g[g];
        float n2 = 0.0f, a2lo = 0.0f, amax = 0.0f;
        float zt[VQ ? T2 : 1][16];                           // VQ: the centred latent in fp32 until the row's scale is known
        // sigmoid, centring, row statistics and the optional z_e store of one finished 32-feature tile
        auto finish_tile = [&](const int t, f32x16& acc) {
            if constexpr (TRAIN) lq_tile_store16(a.pre2, a.D, row, row < a.N, t, h, acc, a.D, true);
          if constexpr (VQ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (2 * t + (r >> 3) >= S) { zt[t][r] = 0.0f; continue; }
                const float zv = acc[r] > 0.0f ? acc[r] : 0.0f;      // the canonical ReLU (lq_act_apply)
                acc[r] = zv;
                const float v = zv - w_mu[32 * t + 2 * r + h_w];
                zt[t][r] = v;
                n2 = lq_fma(v, v, n2);
                amax = fmaxf(amax, lq_abs(v));
            }
          } else {
#if !defined(LQ_ABL_NOSIGMOID) && !defined(LQ_SCALAR_SIGMOID)
            if constexpr (!FAST) {
                // the canonical sigmoid, two elements per instruction; a tile that holds a NaN or an infinity (wave-uniform test,
                // practically never true) takes the one-element form, whose clamps propagate NaN
                lq_v2f chk = lq_bc2(0.0f);
#pragma unroll
                for (int r = 0; r < 16; r += 2) chk = lq_nonfinite_acc2((lq_v2f){acc[r], acc[r + 1]}, chk);
                if (__builtin_amdgcn_ballot_w64(!(chk.x == 0.0f && chk.y == 0.0f)) != 0ull) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = lq_sigmoid(acc[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        if (2 * t + (r >> 3) >= S) continue;
                        const lq_v2f z2 = lq_sigmoid2_finite((lq_v2f){acc[r], acc[r + 1]});
                        acc[r] = z2.x; acc[r + 1] = z2.y;
                    }
                }
            }
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (2 * t + (r >> 3) >= S) continue;             // S odd: features past D in the last tile (zero weights) are not z_e
                // fast mode: hardware exp2 / rcp (1 ulp each) instead of the canonical exp polynomial + IEEE division
#if !defined(LQ_ABL_NOSIGMOID) && !defined(LQ_SCALAR_SIGMOID)
                const float zv = FAST ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[r] * -1.44269504088896341f)) : acc[r];
#else
                const float zv = FAST ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[r] * -1.44269504088896341f))
                                      : FUSED_SIGMOID(acc[r]);
#endif
                acc[r] = zv;
                const float v = zv - w_mu[32 * t + 2 * r + h_w];
                n2 = lq_fma(v, v, n2);
                // register r of tile t is feature 32t + 2r + h = screen slot (step 2t + (r >> 3), element r & 7): scaled by the
                // kernel-wide power of two fz and split into fp16 hi + lo right here (nothing of z_e is kept in fp32)
                const float vs = v * fz;
                const _Float16 vh = (_Float16)vs;
                const float rs = vs - (float)vh;                       // exact: the one-product screen's row-side residual
                ah[2 * t + (r >> 3)][r & 7] = vh;
                al[2 * t + (r >> 3)][r & 7] = (_Float16)rs;
                if constexpr (COARSE) a2lo = lq_fma(rs, rs, a2lo);
            }
          }
            // z_e rows for the exact stage (store_ze_tile).  The LAST tile's are not stored here: its sixteen values wait in zdefg and
            // are stored behind the screen's first stage copies (lq_screen_core_rg, DEFER) -- in front of them their write
            // acknowledgements stood between every wave and "stage 0 has landed" (vmcnt retires in order)
#ifndef LQ_ABL_NOZESTORE
            if (a.ze_out) {
                if (DEFER_ZE && LQ_DEFER_FLAG(a) && t == T2 - 1) zdefg[g] = acc;
                else store_ze_tile(row, t, acc);
            }
#endif
        };

        if constexpr (!FAST) {
            // ================= phase A: encoder + Lipschitz layer, fp32 MFMA ====================
            // An fp32 MFMA blocks its own wave's vector instructions for its whole duration (scripts/probe/probe_pipes2.hip:
            // 64 + 8 + 4.4 n cycles for an MFMA followed by n independent v_fma), so GELU / sigmoid work cannot hide under
            // the chain; what CAN be hidden is latency: the LDS reads of the weights (16 bytes per lane feed four MFMAs) and
            // of the biases are issued one group ahead of the MFMAs that use them (sched_barrier keeps hipcc from sinking them
            // back in front of their use, where it waits for each LDS round trip), and x comes from the prefetch above.
            auto gelu_fixup = [&](f32x16& out, const f32x16& pre) {
                // largest |pre| of the tile by v_max3 (half an instruction per element instead of a multiply and a compare);
                // a NaN is skipped by max3, but the polynomial has already turned it into a NaN, as the tail would
                float mx = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(mx) : "v"(mx), "v"(pre[r]), "v"(pre[r + 1]));
                if (__builtin_amdgcn_ballot_w64(!(mx * mx < 18.0f)) != 0ull) {          // wave-uniform, practically never taken
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!(pre[r] * pre[r] < 18.0f)) out[r] = lq_gelu_tail(pre[r]);
                }
            };
            auto bias16 = [&](const float* wb, int t) {                   // the lane's 16 bias values of tile t (see the LDS fill)
                f32x16 b;
                const float4* p4 = reinterpret_cast<const float4*>(wb + (2 * t + h) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 v = p4[q]; b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
                return b;
            };
            f32x16 h0[T0];
            // the previous block's z_q rows: round 0 of the copy travels codebook -> LDS staging while layer 0 runs
            constexpr int NTRIP = (16 * S + 63) / 64;                    // 64 floats of every row per gather trip
            constexpr int GP = (STAGES_BYTES >= (size_t)WAVES * 8192) ? 8 : 4;      // passes (KiB per wave) per round
            constexpr int NROUND = NTRIP * (8 / GP);
            // every wave's staging area and its decision scratch are ONE private slice of the ring (stride WSLICE): a wave may start
            // the copy while the others are still deciding the previous block -- no workgroup barrier in front of it (round 4; the
            // staging areas used to overlap other waves' scratch: every block began by waiting for its slowest wave)
            unsigned char* gstage = stage0 + (size_t)wave * WSLICE;
            const bool gnow = DEFER_GATHER && have_pend && a.zq;         // wave-uniform
            if (gnow) {
#ifdef LQ_GATHER_BARRIER         /* measurement knob: the barrier as it was */
                lq_wg_barrier();
#endif
                lq_gather_dma<GP>(gstage, a.cb, pend_k, pend_ok, pend_row0, a.N, a.D, lane, 0, 0);
            }
            {
#pragma unroll
                for (int t = 0; t < T0; ++t) h0[t] = bias16(w_B0, t);
                if (x_pref) {
#pragma unroll
                    for (int sq = 0; sq < XPF / 4; ++sq) {
                        if (sq < S0q) {                                    // wave-uniform
                            float4 av[T0];
#pragma unroll
                            for (int t = 0; t < T0; ++t) av[t] = *reinterpret_cast<const float4*>(w_P0 + ((t * S0q + sq) * 64 + lane_w) * 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float bv = xq[4 * sq + q];
#pragma unroll
                                for (int t = 0; t < T0; ++t) {
                                    const float aq = q == 0 ? av[t].x : q == 1 ? av[t].y : q == 2 ? av[t].z : av[t].w;
                                    // padded k-steps (s >= S0) multiply zeros: acc + 0*0 = acc (oracle pads odd fan-in the same way)
                                    h0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, bv, h0[t], 0, 0, 0);
                                }
                            }
                        }
                    }
                } else {
                    const float* xr = a.x + (size_t)rowc * a.A;
                    for (int sq = 0; sq < S0q; ++sq) {
                        float4 av[T0];
                        float bvv[4];
#pragma unroll
                        for (int t = 0; t < T0; ++t) av[t] = *reinterpret_cast<const float4*>(w_P0 + ((t * S0q + sq) * 64 + lane_w) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int k = 2 * (4 * sq + q) + h;
                            bvv[q] = (k < a.A) ? xr[k] : 0.0f;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int t = 0; t < T0; ++t) {
                                const float aq = q == 0 ? av[t].x : q == 1 ? av[t].y : q == 2 ? av[t].z : av[t].w;
                                h0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, bvv[q], h0[t], 0, 0, 0);
                            }
                    }
                }
                // GELU: the straight-line polynomial on every element, the rare |x| >= sqrt(18) ones fixed up behind a wave-uniform
                // branch (lq_gelu's per-element branch cost a divergent-branch sequence per value)
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const f32x16 pre = h0[t];
                    if constexpr (TRAIN) lq_tile_store16(a.pre0, 32 * T0, row, row < a.N, t, h, pre, 32 * T0, true);
                    if constexpr (VQ) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) h0[t][r] = pre[r] > 0.0f ? pre[r] : 0.0f;
                    } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
#ifndef LQ_ABL_NOGELU
                        const lq_v2f g2 = lq_gelu_poly2((lq_v2f){pre[r], pre[r + 1]});
                        h0[t][r] = g2.x; h0[t][r + 1] = g2.y;
#endif
                    }
#ifndef LQ_ABL_NOGELU
                    gelu_fixup(h0[t], pre);
#endif
                    }
                }
            }
            if (gnow) {
#pragma unroll
                for (int rd = 0; rd < NROUND; ++rd) {                    // round rd = (trip rd / (8 / GP), passes GP (rd % (8 / GP)) ...)
                    lq_wait_vmcnt<0>();
                    lq_gather_flush<GP>(gstage, a.zq, pend_ok, pend_row0, a.N, a.D, lane, rd / (8 / GP), GP * (rd % (8 / GP)));
                    if (rd + 1 < NROUND) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // my staging reads are done before it is refilled
                        lq_gather_dma<GP>(gstage, a.cb, pend_k, pend_ok, pend_row0, a.N, a.D, lane, (rd + 1) / (8 / GP),
                                          GP * ((rd + 1) % (8 / GP)));
                    }
                }
            }
            if (PROLOGUE_DMA && g == 0 && blk == (int64_t)blockIdx.x) {   // the workgroup's first row block: layers 1 / 2 have landed
                lq_wait_vmcnt<0>();
                lq_wg_barrier();
            }
            LQ_STAMP(0);
            // ---- layers 1 and 2: one stream of 16-byte weight reads (T1 S1/4 of layer 1, then T2 S2/4 of layer 2), each read
            // issued one 4-MFMA group ahead of its use.  The GELU of tile t-1 is written between the MFMAs of tile t (two
            // elements per group): it cannot overlap the MFMAs (see above), but it keeps the polynomial out of a lumped
            // 16 x 17-instruction block behind each chain.
            f32x16 h1[T1];
            f32x16 pend;                      // pre-activations of the previous tile, GELU pending
            // LUMPED (round 4): a tile's GELU as ONE block behind its MFMA chain instead of staged between the MFMAs of the next
            // tile's chain (rounds 2-3).  scripts/probe/probe_roles.hip: fp32 MFMAs and fp32 vector work share the SIMD's lanes -- a
            // pair of tiles on a SIMD's two waves takes the SUM of both (5 340 cycles) however the instructions are arranged -- but
            // the interleaved stream loses another 9 % to issue arbitration between the two waves (5 843).  Same values either way:
            // cfg2 0.4095 -> 0.3997 ms per launch, same box (profiles/r04_b_lumped_gelu_ab.txt).  Not where every output tile of
            // layer 2 waits at a workgroup barrier for its streamed weights (S = 13: 0.934 -> 1.00 ms lumped) -- there the staging stays.
#ifdef LQ_NO_LUMPED              /* measurement knob: the staged arrangement everywhere */
            constexpr bool LUMPED = false;
#else
            constexpr bool LUMPED = !VQ && !STREAM2;
#endif
            constexpr int G1 = S1 / 4, G2 = S2 / 4;             // groups per tile
            auto wread = [&](int gidx) {                        // group gidx of the stream (compile-time after unrolling)
                if (STREAM2 && gidx >= T1 * G1) gidx = T1 * G1 - 1;             // streamed layer 2: its groups are read from the slab ring
                return (gidx < T1 * G1) ? *reinterpret_cast<const float4*>(w_P1 + (gidx * 64 + lane_w) * 4)
                                        : *reinterpret_cast<const float4*>(w_P2 + ((gidx - T1 * G1) * 64 + lane_w) * 4);
            };
            auto slab_dma = [&](int t_, int buf_) {             // one 16 KB output-tile slab of layer 2 into a ring buffer: 2 KiB per wave
#ifdef LQ_SLAB_DMA_BUILTIN
                typedef __attribute__((address_space(3))) void* lds_ptr_e;
                typedef const __attribute__((address_space(1))) void* glb_ptr_e;
#endif
                const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w2q) + (size_t)t_ * (G2 * 1024);
                unsigned char* dst = stage0 + (size_t)buf_ * SLAB_STRIDE;
                // the lane offset is made opaque HERE: the 14 source addresses of a row block's slabs do not change from block to
                // block, hipcc hoisted them out of the block loop as 64-bit VGPR pairs and (S = 13: 190 spilled registers) spilled
                // eleven of them -- each DMA then sat behind `scratch_load; s_waitcnt vmcnt(0)`, and vmcnt(0) also waits for every
                // slab copy in flight: the ring ran one memory round trip per KiB.  Two adds per copy instead.
                unsigned lane16 = (unsigned)lane * 16u;
#ifndef LQ_SLAB_DMA_HOISTED       /* measurement knob: the addresses as they were */
                asm volatile("" : "+v"(lane16));
#endif
#pragma unroll
                for (int j = 0; j < G2 * 1024 / 1024 / WAVES; ++j) {
                    const int off = (wave + j * WAVES) * 1024;
#ifdef LQ_SLAB_DMA_BUILTIN
                    __builtin_amdgcn_global_load_lds((glb_ptr_e)(src + off + lane16), (lds_ptr_e)(dst + off), 16, 0, 0);
#else
                    // issued as inline asm: behind the builtin hipcc guards the tile's first LDS read of the slab ring with
                    // `s_waitcnt vmcnt(0)` (a DS read may alias an LDS-DMA write) -- the copy of slab t+2, issued two instructions
                    // earlier, had to LAND before tile t could start: a memory round trip per tile.  Which read needs which copy
                    // is this loop's own protocol (counted lq_wait_vmcnt + barrier above).
                    const unsigned long long gaddr = (unsigned long long)(uintptr_t)(src + off) + lane16;
                    const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dst + off));
                    unsigned m0_keep;                                   // (m0 is handed back: hipcc does not accept it as a clobber)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(m0_keep) : "v"(gaddr), "s"(ldsaddr) : "memory");
#endif
                }
            };
            if constexpr (STREAM2) {
                // everyone is past the previous row block's decision scratch (it lives in ring buffers 0 and 1), then the first two
                // slabs start: they have the whole of layer 1 to land
                lq_wg_barrier();
                slab_dma(0, 0);
                slab_dma(1, 1);
            }
            float4 wn = wread(0);
            f32x16 bnext = bias16(w_B1, 0);
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                f32x16 acc = bnext;
#pragma unroll
                for (int sq = 0; sq < G1; ++sq) {
                    const float4 av = wn;
                    wn = wread(t * G1 + sq + 1);                 // the next group's weights (layer 2's first group after the last)
                    if (sq == G1 - 1 && (t + 1 < T1 || !STREAM2)) bnext = (t + 1 < T1) ? bias16(w_B1, t + 1) : bias16(w_B2, 0);
                    __builtin_amdgcn_sched_barrier(0x6);         // reads stay in front of this group's MFMAs (VALU/SALU may move)
                    ActStage<VQ> g;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, h0[(4 * sq + 0) / 16][(4 * sq + 0) % 16], acc, 0, 0, 0);
                    if (t > 0 && !LUMPED) g.stage0(pend[2 * sq], pend[2 * sq + 1]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, h0[(4 * sq + 1) / 16][(4 * sq + 1) % 16], acc, 0, 0, 0);
                    if (t > 0 && !LUMPED) g.stage1();
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, h0[(4 * sq + 2) / 16][(4 * sq + 2) % 16], acc, 0, 0, 0);
                    if (t > 0 && !LUMPED) g.stage2();
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, h0[(4 * sq + 3) / 16][(4 * sq + 3) % 16], acc, 0, 0, 0);
                    if (t > 0 && !LUMPED) {
#ifndef LQ_ABL_NOGELU
                        float o0, o1;
                        g.stage3(o0, o1);
                        h1[t - 1][2 * sq] = o0; h1[t - 1][2 * sq + 1] = o1;
#else
                        h1[t - 1][2 * sq] = pend[2 * sq]; h1[t - 1][2 * sq + 1] = pend[2 * sq + 1];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0x6);
                }
#ifndef LQ_ABL_NOGELU
                if constexpr (!VQ) { if (t > 0 && !LUMPED) gelu_fixup(h1[t - 1], pend); }
#endif
                if constexpr (LUMPED) {
#ifndef LQ_ABL_NOGELU
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const lq_v2f g2 = lq_gelu_poly2((lq_v2f){acc[r], acc[r + 1]});
                        h1[t][r] = g2.x; h1[t][r + 1] = g2.y;
                    }
                    gelu_fixup(h1[t], acc);
#else
                    h1[t] = acc;
#endif
                }
                pend = acc;
                if constexpr (TRAIN) lq_tile_store16(a.pre1, 32 * T1, row, row < a.N, t, h, acc, 32 * T1, true);
            }
            LQ_STAMP(1);
            // the last tile's GELU runs inside the first layer-2 chain: steps 0 .. 47 of that chain only read h1[0..2]
            constexpr int SLAB_BYTES = G2 * 1024;                          // one output tile's A operands: 16 KB
            constexpr int SLAB_CPW = SLAB_BYTES / 1024 / WAVES;
            static_assert(!STREAM2 || (SLAB_BYTES % (1024 * WAVES) == 0 && SLAB_BYTES <= SLAB_STRIDE && 3 * SLAB_STRIDE <= STAGES_BYTES),
                          "slabs ride in the stage ring");
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const float* wslab = w_P2;
                if constexpr (STREAM2) {
                    // slab t has landed in every wave's part (slab t+1 may still fly); everyone has left tile t-1, whose buffer
                    // slab t+2 goes to.  (Slabs 0 and 1 were issued in front of layer 1.)
                    // (round 4, not kept: counting the previous tile's four z_e stores out of this wait -- they are younger than slab t --
                    // made icrt slower, 0.886 -> 0.908 ms: profiles/r04_g_ze_store_placement.txt)
                    if (t + 1 < T2) lq_wait_vmcnt<SLAB_CPW>(); else lq_wait_vmcnt<0>();
                    lq_wg_barrier();
                    if (t + 2 < T2) slab_dma(t + 2, (t + 2) % 3);
                    // the lane's LDS address is formed per tile from an opaque lane offset: as a loop invariant hipcc kept it in a
                    // register from the kernel's first lines and (S = 13) spilled it -- its reload in front of the tile's reads is a
                    // vector-memory load, and waiting for it (`vmcnt(0)`) waits for the slab copy just issued as well
                    unsigned lane4 = (unsigned)lane * 4u;
                    asm volatile("" : "+v"(lane4));
                    wslab = reinterpret_cast<const float*>(stage0 + (size_t)(t % 3) * SLAB_STRIDE) + lane4;
                    wn = *reinterpret_cast<const float4*>(wslab);                     // the tile's first group (read after the barrier)
                    bnext = bias16(w_B2, t);                                            // (no bias prefetch here: registers)
                }
                f32x16 acc = bnext;
#pragma unroll
                for (int sq = 0; sq < G2; ++sq) {
                    if (t == 0 && sq == 3 * (S2 / 16)) {
                        // h1[T1-1] is needed from here on (k-steps 48..63 of a 128-wide layer): finish its GELU
#ifndef LQ_ABL_NOGELU
                        if constexpr (!VQ && !LUMPED) gelu_fixup(h1[T1 - 1], pend);
#endif
                    }
                    const float4 av = wn;
                    if constexpr (STREAM2) {
                        if (sq + 1 < G2) wn = *reinterpret_cast<const float4*>(wslab + (sq + 1) * 256);
                    } else {
                        if (t * G2 + sq + 1 < T2 * G2) wn = wread(T1 * G1 + t * G2 + sq + 1);
                    }
                    if (!STREAM2 && sq == G2 - 1 && t + 1 < T2) bnext = bias16(w_B2, t + 1);
                    __builtin_amdgcn_sched_barrier(0x6);
                    const bool pg = (t == 0 && sq < 8 && !LUMPED);   // the 16 pending GELUs ride on groups 0..7 (< 12: they only read h1[0..2])
                    ActStage<VQ> g;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, h1[(4 * sq + 0) / 16][(4 * sq + 0) % 16], acc, 0, 0, 0);
                    if (pg) g.stage0(pend[(2 * sq) & 15], pend[(2 * sq + 1) & 15]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, h1[(4 * sq + 1) / 16][(4 * sq + 1) % 16], acc, 0, 0, 0);
                    if (pg) g.stage1();
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, h1[(4 * sq + 2) / 16][(4 * sq + 2) % 16], acc, 0, 0, 0);
                    if (pg) g.stage2();
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, h1[(4 * sq + 3) / 16][(4 * sq + 3) % 16], acc, 0, 0, 0);
                    if (pg) {
#ifndef LQ_ABL_NOGELU
                        float o0, o1;
                        g.stage3(o0, o1);
                        h1[T1 - 1][(2 * sq) & 15] = o0; h1[T1 - 1][(2 * sq + 1) & 15] = o1;
#else
                        h1[T1 - 1][(2 * sq) & 15] = pend[(2 * sq) & 15]; h1[T1 - 1][(2 * sq + 1) & 15] = pend[(2 * sq + 1) & 15];
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0x6);
                }
                finish_tile(t, acc);
            }
        } else {
            // ---- fast mode: fp16 operands, fp32 accumulation.  k index of (step s, lane half h, element j) = 16 s + 2 j + h,
            // i.e. the B operand of step s is registers 8 (s & 1) .. +7 of tile s >> 1 of the previous layer as they stand.
            const _Float16* wh0 = reinterpret_cast<const _Float16*>(w_P0);
            const _Float16* wh1 = reinterpret_cast<const _Float16*>(w_P1);
            const _Float16* wh2 = reinterpret_cast<const _Float16*>(w_P2);
            const float* xr = a.x + (size_t)rowc * a.A;
            f32x16 h0[T0];
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h0[t][r] = w_B0[(2 * t + h) * 16 + r];
            for (int s2 = 0; s2 < S0h; ++s2) {
                f16x8 bx;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s2 + 2 * j + h;
                    bx[j] = (_Float16)((k < a.A) ? xr[k] : 0.0f);
                }
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const f16x8 av = *reinterpret_cast<const f16x8*>(wh0 + ((size_t)(t * S0h + s2) * 64 + lane) * 8);
                    h0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bx, h0[t], 0, 0, 0);
                }
            }
            f16x8 b0[2 * T0];
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) b0[2 * t + (r >> 3)][r & 7] = (_Float16)FUSED_GELU(h0[t][r]);
            f16x8 b1[2 * T1];
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = w_B1[(2 * t + h) * 16 + r];
#pragma unroll
                for (int s2 = 0; s2 < 2 * T0; ++s2) {
                    const f16x8 av = *reinterpret_cast<const f16x8*>(wh1 + ((size_t)(t * (2 * T0) + s2) * 64 + lane) * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0[s2], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) b1[2 * t + (r >> 3)][r & 7] = (_Float16)FUSED_GELU(acc[r]);
            }
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = w_B2[(2 * t + h) * 16 + r];
#pragma unroll
                for (int s2 = 0; s2 < 2 * T1; ++s2) {
                    const f16x8 av = *reinterpret_cast<const f16x8*>(wh2 + ((size_t)(t * (2 * T1) + s2) * 64 + lane) * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1[s2], acc, 0, 0, 0);
                }
                finish_tile(t, acc);
            }
        }
        LQ_STAMP(2);
        n2 += __shfl_xor(n2, 32, 64);
        n2g[g] = n2;
        fzg[g] = fz;
        fowng[g] = fown;
        if constexpr (VQ) {
            // the row's own scale (block floating point, lipvq_screen.h): its largest |z'| lands in [2^13, 2^14)
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const int szr = lq_scale_exp(amax);
            const float fzr = lq_pow2f(szr);
            fzg[g] = fzr;
            fowng[g] = lq_pow2f(szr + (int)hdr[3]);
#pragma unroll
            for (int t = 0; t < T2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (2 * t + (r >> 3) >= S) continue;
                    const float vs = zt[t][r] * fzr;
                    const _Float16 vh = (_Float16)vs;
                    const float rs = vs - (float)vh;
                    ah[2 * t + (r >> 3)][r & 7] = vh;
                    al[2 * t + (r >> 3)][r & 7] = (_Float16)rs;
                    if constexpr (COARSE) a2lo = lq_fma(rs, rs, a2lo);
                }
        }
        if constexpr (COARSE) a2lo += __shfl_xor(a2lo, 32, 64);
        a2g[g] = a2lo;
      };
        encode_group(std::integral_constant<int, 0>{});
        if constexpr (RG > 1) encode_group(std::integral_constant<int, 1>{});
        float frow[16];                                       // one scale for every row here (see fz): a single register
        if constexpr (VQ) lq_row_factors(fowng[0], lane, frow);  // (the ReLU instance: every row its own)
        else {
#pragma unroll
            for (int r = 0; r < 16; ++r) frow[r] = fown;
        }
        // ================= phase B: MFMA screen (lq_screen_core_rg, lipvq_screen.h) ==============
        float m1g[RG][16], m2g[RG][16];
        int k1g[RG][16];
#pragma unroll
        for (int g_ = 0; g_ < RG; ++g_)
#pragma unroll
            for (int r = 0; r < 16; ++r) { m1g[g_][r] = INFINITY; m2g[g_][r] = INFINITY; k1g[g_][r] = 0; }
        LQ_STAMP(3);
        if (x_pref) {                                         // next row block's inputs: a whole phase ahead
#pragma unroll
            for (int g_ = 0; g_ < RG; ++g_) load_x(blk + gridDim.x < nblk ? blk + gridDim.x : blk, g_, xqg[g_]);
        }
        // (S = 13: 104 registers of A fragments leave no room for an index array; COARSE: the packed index's perturbation is far
        // below the one-product margin at any S)
        constexpr bool PACKF = LQ_PACK_FOR(S) || S > 8 || COARSE;
        float zng[RG], znrg[RG][16];                       // COARSE: the rows' error scale (lq_track_part), in frow's register layout
#pragma unroll
        for (int g_ = 0; g_ < RG; ++g_) {
            zng[g_] = COARSE ? lq_coarse_zn(a2g[g_], n2g[g_], fzg[g_], fowng[g_], __uint_as_float(hdr[5])) : 0.0f;
            if constexpr (COARSE) lq_row_factors(zng[g_], lane, znrg[g_]);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) znrg[g_][r] = 0.0f;
            }
        }
#ifdef LQ_ABL_NOZESTORE
        const bool have_def = false;
#else
        const bool have_def = DEFER_ZE && LQ_DEFER_FLAG(a) && a.ze_out != nullptr;
#endif
        auto deferred_ze = [&]() {
#pragma unroll
            for (int g_ = 0; g_ < RG; ++g_) store_ze_tile(((blk * WAVES + wave) * RG + g_) * 32 + ln, T2 - 1, zdefg[g_]);
        };
#ifndef LQ_SEED_E2
#define LQ_SEED_E2 0              /* 1: measurement knob, the three-product chain seeded with |e'|^2 f (lq_screen_core_rg, SEED): 7 vector
                                     instructions less per tile and 0.8 % SLOWER (profiles/r04_m_seed_e2_ab.txt) -- not the default */
#endif
        constexpr bool SEED_E2 = LQ_SEED_E2 && !COARSE && !VQ;                // (VQ: every row its own scale -- sixteen products either way)
        lq_screen_core_rg<S, THREADS, TCF, NBF, PACKF, RG, COARSE, 4 * RG, decltype(deferred_ze), SEED_E2>(ahg, alg, tiles, L.ntiles, stage0, tid, frow, znrg, m1g, m2g, k1g, have_def, deferred_ze);
        LQ_STAMP(4);
        const unsigned keep_mask = PACKF ? ~((1u << lq_pack_bits(L.ntiles)) - 1u) : 0xffffffffu;
        unsigned char* scratch = stage0 + (size_t)wave * WSLICE;
#pragma unroll
        for (int g_ = 0; g_ < RG; ++g_) {
            const int64_t row0 = ((blk * WAVES + wave) * RG + g_) * 32;
            const int64_t row = row0 + ln;
            int my_k;
            LqDecision dec;
            bool certified = lq_screen_decide<PACKF, COARSE>(m1g[g_], m2g[g_], k1g[g_], scratch, hdr, n2g[g_], fowng[g_], a.gamma, a.K, a.D, lane,
                                                             my_k, dec, PACKF ? lq_pow2f(lq_pack_bits(L.ntiles) - 23) : 0.0f, keep_mask,
                                                             zng[g_], tiles, tile_bytes, ScreenCfg<S, TCF, COARSE>::FRAG_BYTES);
            const bool row_sane = VQ || n2g[g_] >= tiny2;             // (see fz above; such a row's screen values bound nothing)
            certified = certified && row_sane;
            // in place (round 4; small codebooks under the three-product screen): this wave decides its uncertified rows itself
            // (lq_screen_decide_inplace) and they go on as certified ones -- no list kernel behind the launch
            if (!COARSE && a.inplace)
                certified = lq_screen_decide_inplace<PACKF, 2 * S, VQ ? LIPVQ_DIST_SQSUM : LIPVQ_DIST_NORM>(
                    dec, certified, row_sane, my_k, row0 - ze_shift, row < a.N, a.amb_count, a.ze_out, a.cb, a.K, lane, keep_mask, scratch);
            else
                lq_screen_emit<PACKF>(dec, certified, row_sane, my_k, row, row < a.N, a.amb_count, a.amb_list, a.N, a.K, lane, keep_mask, scratch);
            if (h == 0 && row < a.N && certified) {
                a.idx[row] = (int64_t)my_k;
                if (use_hist) atomicAdd(&hist[my_k], 1u);              // LDS atomic
            }
            if (a.usage && !use_hist) lq_usage_add(a.usage, my_k, h == 0 && row < a.N && certified);
            if (DEFER_GATHER) { pend_kg[g_] = my_k; pend_okg[g_] = certified; pend_row0g[g_] = row0; }
            else if (a.zq) lq_screen_gather(a.cb, a.zq, my_k, certified, row0, a.N, a.D, lane);
            if (RG > 1 && g_ + 1 < RG) __builtin_amdgcn_wave_barrier();     // the next group's transposes reuse this wave's scratch
        }
        if (DEFER_GATHER) have_pend = true;
        LQ_STAMP(5);
        LQ_STAMP(6);
    }
    if (DEFER_GATHER && have_pend && a.zq) {                                                                          // the last block's
#pragma unroll
        for (int g_ = 0; g_ < RG; ++g_) lq_screen_gather(a.cb, a.zq, pend_kg[g_], pend_okg[g_], pend_row0g[g_], a.N, a.D, lane);
    }
#ifdef LQ_STAMPS
    if (lane == 0) {
        long long* dbg = reinterpret_cast<long long*>(a.amb_list + (a.N / 2 & ~1)) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 7; ++i) dbg[i] = st_acc[i];
        dbg[7] = 0;
        dbg[8] = __builtin_amdgcn_s_memtime() - st_begin;
        dbg[9] = st_begin;
        dbg[10] = st_begin - st_top;
        dbg[11] = (long long)__builtin_amdgcn_s_memrealtime();          // 100 MHz wall clock at this wave's end
        dbg[12] = st_top_rt;
    }
#endif
    // the grid's last workgroup publishes the number of listed rows and zeroes the live counters (lipvq_screen.h); its barrier is
    // the one the histogram flush needs, and the arrival's round trip flies beside the flush's atomics
    lq_ws_publish(a.amb_count);
    if (use_hist) {
        for (int i = tid; i < a.K; i += THREADS) {
            const unsigned c = hist[i];
            if (c) atomicAdd(&a.usage[i], (unsigned long long)c);
        }
    }
}

// two waves per SIMD (8 waves, 256 registers each) / one wave per SIMD (4 waves, the whole 512-register file each)
template <int S, bool FAST, bool TRAIN = false, int RG = 1, bool COARSE = false, bool VQ = false>
__global__ __launch_bounds__(FUSED_THREADS) void tokenize_kernel(TokArgs a) {
    tokenize_body<S, FAST, TRAIN, RG, FUSED_WAVES, COARSE, VQ>(a);
}
template <int S, bool FAST, bool TRAIN, int RG, bool COARSE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tokenize_kernel_w4(TokArgs a) {
    tokenize_body<S, FAST, TRAIN, RG, 4, COARSE>(a);
}

// Which (waves per workgroup, row groups per wave) instance runs.  Measured on one box (profiles/r03_d_tokenize_shapes_ab.txt):
//   cfg2 (S = 4): w8rg1 0.468 ms, w8rg2 0.480 (99 spilled registers), w4rg2 0.526, w4rg1 0.549
//   cfg3 (S = 8): w8rg1 2.98 ms,  w4rg2 3.05 (no spills, half the LDS reads per MFMA -- and nothing gained), w4rg1 3.53
//   icrt (S = 13): w8rg1 1.19 ms, w4rg1 1.27
// so the round-2 shape stays the default everywhere; the others remain as instances the parity tests run
// (LIPVQ_TOK_SHAPE=w8rg1|w8rg2|w4rg2|w4rg1: measurement knob; results identical).
// Schedule choices of the fused launch with IDENTICAL results whose better setting depends on the DEVICE (round 4,
// profiles/r04_i_clock_ab.txt): MI355X devices hold different clocks under the same kernel (MI355X_MICROARCH.md, DVFS give-back items
// 3-5).  With the last tile's z_e stores deferred and nontemporal, cfg2's launch takes 0.387 ms on a device that keeps 2.22 GHz under
// it (0.404 without the deferral) -- and 0.437 ms on a device that answers the denser issue stream with 1.97 GHz (0.403 without:
// 2.12 GHz).  Defaults: the settings that are never bad (no deferral, nontemporal); lipvq_tokenize_tune_f32 measures the four
// combinations on the caller's device and shape and keeps the winner for that device; the options tok_defer_ze / tok_nt_ze override.
#ifndef LQ_DEFAULT_DEFER_ZE
#define LQ_DEFAULT_DEFER_ZE 0
#endif
struct LqSchedule { int defer_ze, nt_ze; };
static std::atomic<int> g_tuned[64];                 // per device: 0 = not tuned, else 1 + (defer_ze | nt_ze << 1)
static LqSchedule lq_schedule(int ring = 0) {
    LqSchedule sc{LQ_DEFAULT_DEFER_ZE, ring ? 0 : 1};      // (ring rows are meant to stay in L2: plain stores unless tuned / told otherwise)
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        const int t = g_tuned[dev].load(std::memory_order_relaxed);
        if (t) { sc.defer_ze = (t - 1) & 1; sc.nt_ze = ((t - 1) >> 1) & 1; }
    }
    if (const char* e = lq_knob("LIPVQ_TOK_DEFER_ZE")) sc.defer_ze = e[0] != '0';
    if (const char* e = lq_knob("LIPVQ_TOK_NT_ZE")) sc.nt_ze = e[0] != '0';
    return sc;
}

// In-place decisions (lq_screen_decide_inplace): three-product screen, codebooks the list kernel would finish alone, z_e rows stored.
// Same box, cfg2, in place against the list kernel behind the launch (profiles/r04_j_inplace_ab.txt): 65 536 rows 0.0637 -> 0.0585 ms,
// 131 072 rows 0.1096 -> 0.1054, 262 144 rows 0.2069 -> 0.1993, 524 288 rows 0.3910 -> 0.3863.  (With the last z_e tile's stores
// deferred -- the schedule of the round's first builds -- the full batch measured level, 0.3929 -> 0.3936, and a size rule kept
// it on the list kernel; without the deferral the wave that stops ~2 us for a row no longer costs its workgroup the 6 us saved.)
// (LIPVQ_TOK_INPLACE=0 / 1: measurement knob -- never / whenever possible; LQ_INPLACE_MAX_ROWS: compile-time size limit; results identical)
#ifndef LQ_INPLACE_MAX_ROWS
#define LQ_INPLACE_MAX_ROWS 2147483647
#endif
static int lq_inplace(bool have_ze, int coarse, int K, int64_t N) {
    const bool can = have_ze && !coarse && K <= LQ_LISTS_ALL_K;
    if (const char* e = lq_knob("LIPVQ_TOK_INPLACE")) {
        if (e[0] == '0') return 0;
        if (e[0] == '1') return can ? 1 : 0;
    }
    return (can && N <= LQ_INPLACE_MAX_ROWS) ? 1 : 0;
}
// Ring mode of the z_e scratch (round 4, late): where the launch decides its uncertified rows in place and the caller wants no z_e,
// nothing reads a row's z_e after its wave has decided the row block -- so every block's 32 rows of a wave go to the SAME 32 rows of
// the scratch (2 048 waves x 32 rows x D floats: 16 MB at D = 64) instead of streaming over N x D floats.  Measured, same box, three
// alternating passes (profiles/r04_n_ze_ring_ab.txt): cfg2 0.3848 -> 0.3829 ms.  The stores still leave the L2 (WRITE_SIZE 275 -> 250
// MB per launch, HBM traffic 298 -> 270 MB: this L2 writes the rows through whether or not they are overwritten 50 us later), so the
// gain is the read side of the in-place decisions and 118 MB of address range less, not the 134 MB of stores hoped for.
// (LIPVQ_TOK_ZE_RING=0: measurement knob, the full scratch.)
static int lq_ze_ring(bool caller_wants_ze, int inplace) {
    if (const char* e = lq_knob("LIPVQ_TOK_ZE_RING")) if (e[0] == '0') return 0;
    return (!caller_wants_ze && inplace) ? 1 : 0;
}
struct TokShape { int waves, rg; };
static TokShape tok_shape_env() {              // read per launch (a getenv: nanoseconds), so that a test can switch shapes in-process
    const char* e = lq_knob("LIPVQ_TOK_SHAPE");
    TokShape t{0, 0};
    if (e && !strcmp(e, "w8rg1")) t = {8, 1};
    if (e && !strcmp(e, "w8rg2")) t = {8, 2};
    if (e && !strcmp(e, "w4rg2")) t = {4, 2};
    if (e && !strcmp(e, "w4rg1")) t = {4, 1};
    return t;
}
template <int S, bool FAST, bool TRAIN>
static TokShape tok_shape(int64_t N) {
    constexpr bool HAS_RG2 = !FAST && !TRAIN && S <= 8;      // the instances that exist (launch_tokenize)
    constexpr bool HAS_W4 = !FAST && !TRAIN;
    TokShape t = tok_shape_env();
    // Size rule (round 4): a launch of at most 32 768 rows is at most one 32-row block per SIMD of the chip -- as 4-wave workgroups
    // (one wave per SIMD, 256 CUs) instead of 8-wave ones (two per SIMD on half the CUs): same box, 32 768 rows, w8rg1 -> w4rg1:
    // icrt 0.137 -> 0.099 ms, cfg3 0.205 -> 0.184, cfg2 0.0528 -> 0.0509 (profiles/r04_k_small_launch_shape_ab.txt).  An explicit
    // tok_shape is honoured at any batch size.
    if (t.waves == 0) t = (HAS_W4 && N <= 4 * 32 * 256) ? TokShape{4, 1} : TokShape{8, 1};
    if ((t.rg == 2 && !HAS_RG2) || (t.waves == 4 && !HAS_W4)) t = {8, 1};
    return t;
}

template <typename KFN>
static int launch_tokenize_as(KFN kfn, LqLdsReserve& reserved, const TokArgs& a, size_t lds, int waves, int rg, hipStream_t st) {
    if (int rc = lipvq_reserve_lds(reserved, (const void*)kfn, lds, "tokenize")) return rc;
    const int64_t unit = (int64_t)waves * rg * 32;
    const int64_t nblk = (a.N + unit - 1) / unit;
    int64_t cap = 256 * LQ_EXP_WGS_PER_CU;                                                        // one persistent workgroup per CU
    // (LIPVQ_TOK_GRID: measurement knob -- e.g. 252 leaves four CUs to a collective's kernel, scripts/dev/rccl_contention.py)
    if (const char* e = lq_knob("LIPVQ_TOK_GRID")) { const int64_t g_ = atoll(e); if (g_ > 0) cap = g_; }
    const int64_t blocks = nblk < cap ? nblk : cap;
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(waves * 64), lds, st, a);
    return check_launch("tokenize");
}

template <int S, bool FAST, bool TRAIN = false>
static int launch_tokenize(const TokArgs& a, hipStream_t st) {
    const size_t lds = fused_lds_bytes<S, FAST>(a.A, a.K);
    if (lds > 160 * 1024) return fail(LIPVQ_EUNSUPPORTED, "tokenize: %zu B of LDS needed", lds);
    const TokShape sh = tok_shape<S, FAST, TRAIN>(a.N);
    static LqLdsReserve reserved[6];            // per instantiation and shape: per-device, thread-safe (lipvq_common.h)
    if constexpr (!FAST && !TRAIN && S >= 8) {  // (one wave per SIMD with the whole register file: the instances that spill at two)
        if (a.coarse && sh.waves == 4 && sh.rg == 1)
            return launch_tokenize_as(tokenize_kernel_w4<S, FAST, TRAIN, 1, true>, reserved[5], a, lds, 4, 1, st);
    }
    if constexpr (!FAST) {                      // (parity and training instances; the training forward writes z_e anyway)
        if (a.coarse) return launch_tokenize_as(tokenize_kernel<S, FAST, TRAIN, 1, true>, reserved[4], a, lds, 8, 1, st);
    }
    if constexpr (!FAST && !TRAIN) {
        if (sh.waves == 4 && sh.rg == 1) return launch_tokenize_as(tokenize_kernel_w4<S, FAST, TRAIN, 1>, reserved[3], a, lds, 4, 1, st);
    }
    if constexpr (!FAST && !TRAIN && S <= 8) {
        if (sh.waves == 8 && sh.rg == 2) return launch_tokenize_as(tokenize_kernel<S, FAST, TRAIN, 2>, reserved[1], a, lds, 8, 2, st);
        if (sh.waves == 4 && sh.rg == 2) return launch_tokenize_as(tokenize_kernel_w4<S, FAST, TRAIN, 2>, reserved[2], a, lds, 4, 2, st);
    }
    return launch_tokenize_as(tokenize_kernel<S, FAST, TRAIN, 1>, reserved[0], a, lds, 8, 1, st);
}

// the plain VQVAE's instances (ReLU encoder, per-row scales): both screens
template <int S>
static int launch_tokenize_vq(const TokArgs& a, hipStream_t st) {
    const size_t lds = fused_lds_bytes<S, false>(a.A, a.K);
    if (lds > 160 * 1024) return fail(LIPVQ_EUNSUPPORTED, "vq_tokenize: %zu B of LDS needed", lds);
    static LqLdsReserve reserved[4];
    if (a.pre0) {                                              // the training forward: the three pre-activations are stored too
        if (a.coarse) return launch_tokenize_as(tokenize_kernel<S, false, true, 1, true, true>, reserved[3], a, lds, 8, 1, st);
        return launch_tokenize_as(tokenize_kernel<S, false, true, 1, false, true>, reserved[2], a, lds, 8, 1, st);
    }
    if (a.coarse) return launch_tokenize_as(tokenize_kernel<S, false, false, 1, true, true>, reserved[1], a, lds, 8, 1, st);
    return launch_tokenize_as(tokenize_kernel<S, false, false, 1, false, true>, reserved[0], a, lds, 8, 1, st);
}

// fp16 MFMA fragments of the encoder stack for the fast mode: [layer][tile t][step s][lane][8 halfs] with
// element (t, s, lane, j) = W[32 t + feat(lane & 31)][16 s + 2 j + (lane >> 5)]  (0 outside the matrix)
__global__ void mlp3_pack_f16_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                                     _Float16* __restrict__ out, int A, int J0, int J1, int D) {
    const int S0h = (A + 15) / 16, S1h = J0 / 16, S2h = J1 / 16;
    const int T0 = J0 / 32, T1 = J1 / 32, T2 = (D + 31) / 32;
    const size_t n0 = (size_t)T0 * S0h * 512, n1 = (size_t)T1 * S1h * 512, n2 = (size_t)T2 * S2h * 512;
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n0 + n1 + n2) return;
    const float* W;
    int K, J, Sh;
    size_t e = g;
    if (e < n0) { W = W0; K = A; J = J0; Sh = S0h; }
    else if (e < n0 + n1) { e -= n0; W = W1; K = J0; J = J1; Sh = S1h; }
    else { e -= n0 + n1; W = W2; K = J1; J = D; Sh = S2h; }
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const size_t ts = e >> 9;
    const int s = (int)(ts % Sh), t = (int)(ts / Sh);
    const int f = 32 * t + feat_of_tile_row(lane & 31);
    const int k = 16 * s + 2 * j + (lane >> 5);
    out[g] = (_Float16)((f < J && k < K) ? W[(size_t)f * K + k] : 0.0f);
}

extern "C" size_t lipvq_mlp3_packed_f16_bytes(int A, int J0, int J1, int D) {
    if (A <= 0 || J0 <= 0 || J1 <= 0 || D <= 0 || (J0 & 31) || (J1 & 31)) return 0;
    const size_t S0h = (A + 15) / 16, T2 = (D + 31) / 32;
    return ((size_t)(J0 / 32) * S0h + (size_t)(J1 / 32) * (J0 / 16) + T2 * (J1 / 16)) * 512 * sizeof(_Float16);
}

extern "C" int lipvq_mlp3_pack_f16_f32(const float* W0, const float* W1, const float* W2, void* packed16, int A, int J0,
                                       int J1, int D, void* stream) {
    if (!W0 || !W1 || !W2 || !packed16) return fail(LIPVQ_EINVAL, "mlp3_pack_f16: null pointer");
    const size_t bytes = lipvq_mlp3_packed_f16_bytes(A, J0, J1, D);
    if (!bytes) return fail(LIPVQ_EUNSUPPORTED, "mlp3_pack_f16: unsupported shape A=%d J0=%d J1=%d D=%d", A, J0, J1, D);
    const size_t n = bytes / sizeof(_Float16);
    hipLaunchKernelGGL(mlp3_pack_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W0, W1, W2,
                       (_Float16*)packed16, A, J0, J1, D);
    return check_launch("mlp3_pack_f16");
}

extern "C" int lipvq_tokenize_supported(int A, int J0, int J1, int D, int K) {
    return (A > 0 && A <= 64 && J0 == 64 && J1 == 128 && K > 0 && (D == 32 || D == 64 || D == 128 || D == 208)) ? 1 : 0;
}

// the fast mode has no D = 208 instance (its fp16 weights would fit, but the mode exists for BASELINE config 2's shapes)
extern "C" int lipvq_tokenize_fast_supported(int A, int J0, int J1, int D, int K) {
    return (lipvq_tokenize_supported(A, J0, J1, D, K) && D != 208) ? 1 : 0;
}

// layer-2 weights of the streamed instance (D = 208): [t][16 groups][64 lanes][4 k-steps], i.e. the LDS image of one output
// tile's A operands as contiguous 16 KB slabs, so that the kernel can copy them with the LDS-DMA
static size_t w2q_floats(int D) { return (D + 15) / 16 >= LQ_STREAM2_MIN_S ? (size_t)((D + 31) / 32) * 16 * 256 : 0; }

__global__ void w2q_pack_kernel(const float* __restrict__ P2, float* __restrict__ out, int T2, int S2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T2 * (S2 / 4) * 256) return;
    const int q = (int)(i & 3), l = (int)((i >> 2) & 63), sq = (int)((i >> 8) % (S2 / 4)), t = (int)((i >> 8) / (S2 / 4));
    out[i] = P2[((size_t)t * S2 + 4 * sq + q) * 64 + l];
}

// Zero the header of a workspace of lipvq_tokenize_workspace_bytes / lipvq_nearest_workspace_bytes: ONCE, before its first use by
// lipvq_tokenize_f32 / _fast / _train / lipvq_vq_tokenize*_f32.  Those calls leave the header's counters at zero themselves.
extern "C" int lipvq_tokenize_workspace_init(void* workspace, void* stream) {
    if (!workspace) return fail(LIPVQ_EINVAL, "tokenize_workspace_init: null pointer");
    hipError_t e = hipMemsetAsync(workspace, 0, 64, (hipStream_t)stream);
    if (e != hipSuccess) return fail(LIPVQ_EHIP, "tokenize_workspace_init: %s", hipGetErrorString(e));
    return LIPVQ_OK;
}

extern "C" size_t lipvq_tokenize_workspace_bytes(int64_t N, int D) {
    if (N <= 0 || D <= 0) return 0;
    // uncertified-row counter, row list, best-candidate list, short lists, then (D = 208) the streamed layer-2 weights, then a z_e
    // scratch [N][D] for the one-product screen's exact stage (used when the caller passes no ze_out; always part of the size, so
    // that the mode may change between calls)
    return 64 + lq_lists_bytes(N) + sizeof(float) * w2q_floats(D) + 256 + sizeof(float) * (size_t)N * (size_t)D;
}

// Fused encode + quantize (reference v5:71-74).  packed: lipvq_mlp3_pack_f32 of the encoder stack
// (A -> 64 -> 128 -> D with the Lipschitz-normalised W2, activations gelu, gelu, sigmoid); raw6: the same six
// tensors unpacked, {W0, b0, W1, b1, W2, b2} (device pointers; the array itself is host memory); prep:
// lipvq_nearest_prepare_f32 of the codebook; workspace: lipvq_tokenize_workspace_bytes(N, D).
// Outputs exactly as lipvq_mlp3_f32 + lipvq_nearest_f32(LIPVQ_DIST_NORM): idx, zq (may be NULL),
// usage (may be NULL, accumulated), ze_out (may be NULL).  workspace[0] (int) = rows decided by the exact kernel.
static int tokenize_impl(const float* x, const float* packed, const void* packed16, const float* const* raw6,
                         const float* codebook, const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out,
                         void* workspace, int64_t N, int A, int J0, int J1, int D, int K, void* stream,
                         float* pre0 = nullptr, float* pre1 = nullptr, float* pre2 = nullptr) {
    if (N < 0) return fail(LIPVQ_EINVAL, "tokenize: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!x || !packed || !raw6 || !codebook || !prep || !idx || !workspace) return fail(LIPVQ_EINVAL, "tokenize: null pointer");
    for (int i = 0; i < 6; ++i)
        if (!raw6[i]) return fail(LIPVQ_EINVAL, "tokenize: raw encoder weight %d is null", i);
    if (!lipvq_tokenize_supported(A, J0, J1, D, K))
        return fail(LIPVQ_EUNSUPPORTED, "tokenize: unsupported shape A=%d J0=%d J1=%d D=%d K=%d", A, J0, J1, D, K);
    if (N > 2147483647LL) return fail(LIPVQ_EUNSUPPORTED, "tokenize: N too large");
    if ((((uintptr_t)codebook | (uintptr_t)zq | (uintptr_t)ze_out | (uintptr_t)workspace | (uintptr_t)packed16) & 15) != 0)
        return fail(LIPVQ_EINVAL, "tokenize: codebook, zq, ze_out, workspace and packed16 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    int* amb_count = (int*)ws + LQ_WS_LIVE;           // the header's live counters (lipvq_screen.h: zero between calls)
    int* amb_list = (int*)(ws + 64);
    // z_e goes to the caller's buffer when one is given (training), otherwise -- see below -- to a scratch in the workspace
    float* ze_buf = ze_out;
    // the one-product screen (parity instances only): its exact stage reads z_e rows, so one is always written
    const int coarse = !packed16 ? lq_screen_coarse(lq_screen_S(D), K) : 0;
    // ... and with the three-product screen too (round 3, late): the exact stage on stored rows (nearest_lists_kernel: a wave per
    // row, ~8 us for cfg2's 1 756 rows) against re-encoding the listed rows from x (nearest_rows_encode_kernel: 32 us behind a
    // 0.43 ms launch) -- the whole call 0.467 -> 0.456 ms at the metric's batch, same box, for 134 MB of stores the launch does
    // not feel (it is matrix-pipe bound at 0.4 TB/s of HBM traffic; `traffic` in bench.py's roofline shows them: 171 -> 305 MB).
    // Until then only batches of <= 131 072 rows stored z_e.  The fast mode keeps that limit: above it its uncertified rows are
    // re-encoded with the fp32 encoder and get the parity mode's answer.
    // (LIPVQ_TOK_ZE_ROWS: measurement knob, the batch size up to which a launch stores z_e when nothing else asks for it; per launch)
    int64_t ze_rows = packed16 ? 131072 : INT64_MAX;
    if (const char* ev = lq_knob("LIPVQ_TOK_ZE_ROWS")) ze_rows = atoll(ev);
    if ((coarse || N <= ze_rows) && !ze_buf) {
        size_t off = 64 + lq_lists_bytes(N) + sizeof(float) * w2q_floats(D);
        off = (off + 255) & ~(size_t)255;
        ze_buf = reinterpret_cast<float*>(ws + off);
    }
    // (no fill of the header here since round 4: the call's last kernel leaves the live counters at zero, lq_ws_finish;
    // lipvq_tokenize_workspace_init zeroes a fresh workspace once)
    float* w2q = nullptr;
    if (w2q_floats(D)) {
        // the streamed instance: re-lay out layer 2's packed weights into the workspace (one small launch; the weights may have
        // changed since the last call and the library keeps no state)
        w2q = reinterpret_cast<float*>(ws + 64 + lq_lists_bytes(N));
        const PackedLayout PL = packed_layout(A, J0, J1, D);
        const size_t n = w2q_floats(D);
        hipLaunchKernelGGL(w2q_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed + PL.oP2, w2q, PL.T2, PL.S2);
    }
    TokArgs a{x, packed, (const unsigned char*)packed16, (const unsigned char*)prep, codebook, idx, zq,
              (unsigned long long*)usage, ze_buf, amb_count, amb_list, w2q, pre0, pre1, pre2, N, A, D, K, LIPVQ_SCREEN_GAMMA, coarse,
              lq_inplace(ze_buf != nullptr, coarse, K, N), 0, 0, 0};
    a.ze_ring = lq_ze_ring(ze_out != nullptr, a.inplace);
    a.defer_ze = lq_schedule(a.ze_ring).defer_ze;
    a.nt_ze = lq_schedule(a.ze_ring).nt_ze;
    int rc;
    if (pre0) {
        if ((((uintptr_t)pre0 | (uintptr_t)pre1 | (uintptr_t)pre2) & 15) != 0)
            return fail(LIPVQ_EINVAL, "tokenize_train: the pre-activation buffers must be 16-byte aligned");
        switch (D) {
            case 32: rc = launch_tokenize<2, false, true>(a, st); break;
            case 64: rc = launch_tokenize<4, false, true>(a, st); break;
            case 128: rc = launch_tokenize<8, false, true>(a, st); break;
            default: rc = launch_tokenize<13, false, true>(a, st); break;
        }
    } else if (packed16) {
        switch (D) {
            case 32: rc = launch_tokenize<2, true>(a, st); break;
            case 64: rc = launch_tokenize<4, true>(a, st); break;
            case 128: rc = launch_tokenize<8, true>(a, st); break;
            default: return fail(LIPVQ_EUNSUPPORTED, "tokenize_fast: D=%d has no fast instance (32, 64, 128)", D);
        }
    } else {
        switch (D) {
            case 32: rc = launch_tokenize<2, false>(a, st); break;
            case 64: rc = launch_tokenize<4, false>(a, st); break;
            case 128: rc = launch_tokenize<8, false>(a, st); break;
            default: rc = launch_tokenize<13, false>(a, st); break;
        }
    }
    if (!rc && !a.inplace) {
        // uncertified rows (count on the device): exact decision, from the stored z_e rows or from x
        if (ze_buf) rc = lipvq_launch_rows(ze_buf, 0, codebook, idx, zq, usage, amb_list, amb_count, N, K, D, st);
        else rc = lipvq_launch_rows_encode(x, raw6, A, codebook, idx, zq, usage, amb_list, amb_count, N, K, D, st);
    }
    if (rc) (void)hipMemsetAsync(ws, 0, 64, st);     // a failed call must not leave counters behind for the next one
    return rc;
}

extern "C" int lipvq_tokenize_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                                  const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out,
                                  void* workspace, int64_t N, int A, int J0, int J1, int D, int K, void* stream) {
    return tokenize_impl(x, packed, nullptr, raw6, codebook, prep, idx, zq, usage, ze_out, workspace, N, A, J0, J1, D, K, stream);
}

// lipvq_tokenize_f32's schedule tuned on the caller's device, shape and data: the four (defer_ze, nt_ze) combinations, each warmed
// and then timed over `launches` back-to-back calls between HIP events, two alternating rounds, the minimum per combination; the
// fastest becomes this device's setting for every later lipvq_tokenize_* call of the process.  SYNCHRONOUS (it waits for the
// stream) and not capturable.  Every launch writes idx / zq / ze_out and accumulates into usage like lipvq_tokenize_f32 -- all four
// combinations write the same values.  choice (may be NULL): defer_ze | nt_ze << 1;  ms4 (may be NULL): the four times per launch.
extern "C" int lipvq_tokenize_tune_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                                       const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out, void* workspace,
                                       int64_t N, int A, int J0, int J1, int D, int K, void* stream, int launches, int* choice,
                                       float* ms4) {
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(LIPVQ_EINVAL, "tokenize_tune: the stream is capturing (the tuner synchronises)");
    if (launches < 1) return fail(LIPVQ_EINVAL, "tokenize_tune: launches < 1");
    if (lq_screen_S(D) > 4) {                                                // (wider latents: their instances have no such choice)
        if (choice) *choice = -1;
        if (ms4) for (int c = 0; c < 4; ++c) ms4[c] = 0.0f;
        return LIPVQ_OK;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(LIPVQ_EINVAL, "tokenize_tune: device index");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(LIPVQ_EHIP, "tokenize_tune: hipEventCreate");
    const int before = g_tuned[dev].load(std::memory_order_relaxed);
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int rc = LIPVQ_OK;
    auto run = [&](int n) {
        for (int i = 0; i < n && !rc; ++i)
            rc = tokenize_impl(x, packed, nullptr, raw6, codebook, prep, idx, zq, usage, ze_out, workspace, N, A, J0, J1, D, K, stream);
    };
    run(launches);                                                       // the chip's clock and power state of a running job
    for (int round = 0; round < 2 && !rc; ++round)
        for (int c = 0; c < 4 && !rc; ++c) {
            g_tuned[dev].store(1 + c, std::memory_order_relaxed);
            run((launches + 1) / 2);
            (void)hipEventRecord(e0, st);
            run(launches);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) rc = fail(LIPVQ_EHIP, "tokenize_tune: hipEventSynchronize");
            float ms = 0.0f;
            if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms / launches < best[c]) best[c] = ms / launches;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) { g_tuned[dev].store(before, std::memory_order_relaxed); return rc; }
    // the default of this launch's mode stays unless another combination is faster by more than half a percent (ties are common, and
    // the ring's plain stores are what keeps its rows out of HBM)
    const int ring = lq_ze_ring(ze_out != nullptr, lq_inplace(true, lq_screen_coarse(lq_screen_S(D), K), K, N));
    const int dflt = LQ_DEFAULT_DEFER_ZE | ((ring ? 0 : 1) << 1);
    int win = dflt;
    for (int c = 0; c < 4; ++c) if (best[c] < 0.995f * best[dflt] && best[c] < best[win]) win = c;
    g_tuned[dev].store(1 + win, std::memory_order_relaxed);
    if (choice) *choice = win;
    if (ms4) for (int c = 0; c < 4; ++c) ms4[c] = best[c];
    return LIPVQ_OK;
}

// The forward half of a training step (v5:71-74 with everything autograd saves): lipvq_tokenize_f32 that also writes the three
// pre-activations pre0 [N][J0], pre1 [N][J1], pre2 [N][D] and z_e -- the same values lipvq_mlp3_f32(.., pre0, pre1, pre2) stores.
extern "C" int lipvq_tokenize_train_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                                        const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out, float* pre0,
                                        float* pre1, float* pre2, void* workspace, int64_t N, int A, int J0, int J1, int D, int K,
                                        void* stream) {
    if (!ze_out || !pre0 || !pre1 || !pre2) return fail(LIPVQ_EINVAL, "tokenize_train: z_e and the three pre-activation buffers are required");
    return tokenize_impl(x, packed, nullptr, raw6, codebook, prep, idx, zq, usage, ze_out, workspace, N, A, J0, J1, D, K, stream,
                         pre0, pre1, pre2);
}

// The plain VQVAE's encode + quantize in one launch (reference backbone.py:40-66: encoder = Linear/ReLU x 3, then
// `(z_e.unsqueeze(1) - E).pow(2).sum(-1)`, argmin, embedding lookup).  packed: lipvq_mlp3_pack_f32 of the encoder (A -> 64 -> 128 -> D,
// plain weights); prep: lipvq_nearest_prepare_f32 of the embedding table; ze_out [N][D] is REQUIRED (the straight-through value
// z_e + (z_q - z_e) of vq:74 needs it, and so does the exact stage).  Same results as lipvq_mlp3_f32(relu, relu, relu) followed by
// lipvq_nearest_f32(LIPVQ_DIST_SQSUM).  workspace: lipvq_tokenize_workspace_bytes(N, D).
static int vq_tokenize_impl(const float* x, const float* packed, const float* codebook, const void* prep, int64_t* idx,
                            float* zq, int64_t* usage, float* ze_out, float* pre0, float* pre1, float* pre2, void* workspace, int64_t N,
                            int A, int J0, int J1, int D, int K, void* stream) {
    if (N < 0) return fail(LIPVQ_EINVAL, "vq_tokenize: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!x || !packed || !codebook || !prep || !idx || !ze_out || !workspace) return fail(LIPVQ_EINVAL, "vq_tokenize: null pointer");
    if (!lipvq_tokenize_supported(A, J0, J1, D, K))
        return fail(LIPVQ_EUNSUPPORTED, "vq_tokenize: unsupported shape A=%d J0=%d J1=%d D=%d K=%d", A, J0, J1, D, K);
    if (N > 2147483647LL) return fail(LIPVQ_EUNSUPPORTED, "vq_tokenize: N too large");
    if ((((uintptr_t)codebook | (uintptr_t)zq | (uintptr_t)ze_out | (uintptr_t)workspace) & 15) != 0)
        return fail(LIPVQ_EINVAL, "vq_tokenize: codebook, zq, ze_out and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    int* amb_count = (int*)ws + LQ_WS_LIVE;           // the header's live counters: zero between calls (lq_ws_finish)
    int* amb_list = (int*)(ws + 64);
    float* w2q = nullptr;
    if (w2q_floats(D)) {
        w2q = reinterpret_cast<float*>(ws + 64 + lq_lists_bytes(N));
        const PackedLayout PL = packed_layout(A, J0, J1, D);
        const size_t n = w2q_floats(D);
        hipLaunchKernelGGL(w2q_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed + PL.oP2, w2q, PL.T2, PL.S2);
    }
    const int coarse = lq_screen_coarse(lq_screen_S(D), K);
    TokArgs a{x, packed, nullptr, (const unsigned char*)prep, codebook, idx, zq, (unsigned long long*)usage, ze_out, amb_count,
              amb_list, w2q, pre0, pre1, pre2, N, A, D, K, LIPVQ_SCREEN_GAMMA, coarse, lq_inplace(true, coarse, K, N),
              lq_schedule().defer_ze, lq_schedule().nt_ze, 0};
    int rc;
    switch (D) {
        case 32: rc = launch_tokenize_vq<2>(a, st); break;
        case 64: rc = launch_tokenize_vq<4>(a, st); break;
        case 128: rc = launch_tokenize_vq<8>(a, st); break;
        default: rc = launch_tokenize_vq<13>(a, st); break;
    }
    if (!rc && !a.inplace) rc = lipvq_launch_rows(ze_out, 0, codebook, idx, zq, usage, amb_list, amb_count, N, K, D, st, LIPVQ_DIST_SQSUM);
    if (rc) (void)hipMemsetAsync(ws, 0, 64, st);     // a failed call must not leave counters behind for the next one
    return rc;
}

extern "C" int lipvq_vq_tokenize_f32(const float* x, const float* packed, const float* codebook, const void* prep, int64_t* idx,
                                     float* zq, int64_t* usage, float* ze_out, void* workspace, int64_t N, int A, int J0, int J1,
                                     int D, int K, void* stream) {
    return vq_tokenize_impl(x, packed, codebook, prep, idx, zq, usage, ze_out, nullptr, nullptr, nullptr, workspace, N, A, J0, J1, D, K,
                            stream);
}

// The training forward of the plain VQVAE in one launch: lipvq_vq_tokenize_f32 that also stores the three pre-activations
// [N][J0], [N][J1], [N][D] its backward needs (what lipvq_mlp3_f32 with pre0..2 would save, bit for bit).
extern "C" int lipvq_vq_tokenize_train_f32(const float* x, const float* packed, const float* codebook, const void* prep, int64_t* idx,
                                           float* zq, int64_t* usage, float* ze_out, float* pre0, float* pre1, float* pre2,
                                           void* workspace, int64_t N, int A, int J0, int J1, int D, int K, void* stream) {
    if (N > 0 && (!pre0 || !pre1 || !pre2)) return fail(LIPVQ_EINVAL, "vq_tokenize_train: the three pre-activation buffers are required");
    if ((((uintptr_t)pre0 | (uintptr_t)pre1 | (uintptr_t)pre2) & 15) != 0) return fail(LIPVQ_EINVAL, "vq_tokenize_train: pre0..2 must be 16-byte aligned");
    return vq_tokenize_impl(x, packed, codebook, prep, idx, zq, usage, ze_out, pre0, pre1, pre2, workspace, N, A, J0, J1, D, K, stream);
}

// Fast mode: the encoder's GEMMs on fp16 MFMAs (packed16 = lipvq_mlp3_pack_f16_f32 of the same weights; `packed` still
// supplies the fp32 biases).  Not bit-identical to lipvq_tokenize_f32: see tokenize_kernel.  Rows the screen cannot
// certify are decided by the exact kernel from the fp32 encoder (they get the parity-mode answer).
extern "C" int lipvq_tokenize_fast_f32(const float* x, const float* packed, const void* packed16, const float* const* raw6,
                                       const float* codebook, const void* prep, int64_t* idx, float* zq, int64_t* usage,
                                       void* workspace, int64_t N, int A, int J0, int J1, int D, int K, void* stream) {
    if (!packed16) return fail(LIPVQ_EINVAL, "tokenize_fast: packed16 is null");
    return tokenize_impl(x, packed, packed16, raw6, codebook, prep, idx, zq, usage, nullptr, workspace, N, A, J0, J1, D, K, stream);
}
