// lipvq_bwd.hip -- the parameter-gradient side of the tokenizer's backward pass (what autograd
// derives from reference backbone_lfqvae_v5.py:70-84 / backbone.py:38-76):
//   wgrad_kernel          gW = G^T . H and gb = sum_rows G  (contraction over the batch rows, fp32 MFMA,
//                         split over row chunks of 64...2048 rows into partial slabs)
//   wgrad_reduce_kernel   deterministic sum of the slabs (double)
//   scatter_add_kernel    codebook gradient: index_add_ of the gather's backward
//   lipschitz_bwd_kernel  backward of normalization() (v5:6-12)
//   scaled_diff_kernel    out = alpha * g * (a - b) + c   (the d mse / d input terms)
// ABI: include/lipvq.h.
#include <stdlib.h>

#include "lipvq_common.h"

// rows per chunk: enough chunks that the (chunk x tile) grid fills the chip even for training-step batches -- with one
// 2048-row chunk a 1024-row batch was ONE dependent chain of 512 load+MFMA steps per wave (248 us per call).
static inline int wgrad_chunk_rows(int64_t N) {
    static int forced = -1;                      // LIPVQ_WGRAD_CHUNK: measurement knob
    if (forced < 0) {
        const char* e = lq_knob("LIPVQ_WGRAD_CHUNK");
        forced = e ? atoi(e) : 0;
    }
    if (forced > 0) return forced;
    if (N <= 128) return 128;                    // ONE chunk: lipvq_wgrad_f32 then writes the result directly (no reduce launch)
    return N >= 262144 ? 1024 : (N >= 16384 ? 256 : 64);
}

// One wave = one 32x32 tile of gW over one chunk of rows.
//   A operand: lane (i = lane & 31, kh = lane >> 5) = G[row0 + 2s + kh][32 ti + i]
//   B operand: lane (j = lane & 31, kh)             = act(H[row0 + 2s + kh][32 tj + j])
//   D[i][j]  : col j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// Operands are fetched 8 steps (16 rows) ahead of the MFMAs that consume them.
__global__ __launch_bounds__(64) void wgrad_kernel(const float* __restrict__ G, const float* __restrict__ H,
                                                   const int64_t* __restrict__ hidx, int h_act,
                                                   float* __restrict__ partW, float* __restrict__ partB,
                                                   int64_t N, int J, int Kd, int TJ, int chunk_rows) {
    const int lane = threadIdx.x;
    const int li = lane & 31, kh = lane >> 5;
    const int ti = blockIdx.y / TJ, tj = blockIdx.y % TJ;
    const int fi = 32 * ti + li, fj = 32 * tj + li;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t r1 = r0 + chunk_rows;
    if (r1 > N) r1 = N;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;
    const bool ai = fi < J, bj = fj < Kd;
    auto fetch = [&](int64_t row, float& av, float& bv) {
        av = 0.0f;
        bv = 0.0f;
        if (row < r1) {
            if (ai) av = G[(size_t)row * J + fi];
            if (bj) {
                const int64_t hr = hidx ? hidx[row] : row;
                bv = H[(size_t)hr * Kd + fj];
            }
        }
    };
    float ca[8], cb[8], na[8], nb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fetch(r0 + 2 * j + kh, ca[j], cb[j]);
    for (int64_t base = r0; base < r1; base += 16) {
        if (base + 16 < r1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fetch(base + 16 + 2 * j + kh, na[j], nb[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (base + 2 * j < r1) {                     // wave-uniform: skip whole steps past the chunk
                const bool in = base + 2 * j + kh < r1;
                const float bv = (in && bj) ? lq_act_apply(cb[j], h_act) : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[j], bv, acc, 0, 0, 0);
                bsum += ca[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ca[j] = na[j]; cb[j] = nb[j]; }
    }
    float* pw = partW + (size_t)blockIdx.x * J * Kd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oi = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (oi < J && bj) pw[(size_t)oi * Kd + fj] = acc[r];
    }
    if (tj == 0) {
        const float tot = bsum + __shfl_xor(bsum, 32, 64);
        if (kh == 0 && ai) partB[(size_t)blockIdx.x * J + fi] = tot;
    }
}

// Workgroup-per-chunk variant (what lipvq_wgrad_f32 launches when the row tile fits LDS): the per-tile kernel above lets
// every wave stream 128-byte pieces of 512-byte rows (each G element is fetched by TJ waves, each H element by TI), which
// at N = 524 288 ran 620-740 us per call (12 TFLOP/s, ~1.7 TB/s of partial-row HBM reads).  Here the 8 waves of a
// workgroup own ALL 32x32 tiles of gW for one chunk of rows: 32 rows of G and act(H) at a time are staged into LDS with
// whole-row coalesced loads (activation and the hidx gather applied once per element), every wave then feeds its
// tiles' MFMAs from LDS.  Same chunk-ordered accumulation per tile as above.
#define WGW_WAVES 8
#define WGW_MAXT 4            // tiles per wave: TI * TJ <= 32

template <int CG, int TPW>        // column groups of 64 lanes covering max(Jp, Kp); 32x32 tiles per wave
__global__ __launch_bounds__(64 * WGW_WAVES) void wgrad_wg_kernel(const float* __restrict__ G, const float* __restrict__ H,
                                                                  const int64_t* __restrict__ hidx, int h_act,
                                                                  float* __restrict__ partW, float* __restrict__ partB,
                                                                  int64_t N, int J, int Kd, int TI, int TJ, int chunk_rows) {
    extern __shared__ float wg_lds[];
    const int Jp = 32 * TI, Kp = 32 * TJ;
    float* Gs = wg_lds;                   // [32][Jp]
    float* Hs = Gs + 32 * Jp;             // [32][Kp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int T = TI * TJ;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t r1 = r0 + chunk_rows;
    if (r1 > N) r1 = N;
    f32x16 acc[TPW];
    float bsum[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        bsum[q] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    }
    // staging map: wave w fetches rows 4w .. 4w+3 of the 32-row block, lanes along the columns (whole-row coalesced).
    // The loads of the next TWO blocks are in flight (registers) while the current block is multiplied: with one block of
    // look-ahead (the first version) the loads had one MFMA phase -- ~0.5 us -- to come back from HBM, and SQ counters showed
    // the waves parked in s_waitcnt / barriers 65 % of the time with the matrix pipe 27 % busy.
    constexpr int DEPTH = 3;                       // register sets: the block being staged + two in flight
    float gq[DEPTH][4][CG], hq[DEPTH][4][CG];
    // Branch-free on purpose: every load has a valid (clamped) address and is masked by a select afterwards, every block runs
    // its 16 k-steps (rows past the chunk were staged as zeros), and the block loop has no conditional between blocks -- at a
    // control-flow join hipcc's wait-count pass falls back to vmcnt(0), which would drain the look-ahead at every block.
    auto fetch = [&](int64_t rb, float (&gqs)[4][CG], float (&hqs)[4][CG]) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = rb + 4 * wave + rr;
            const bool in = row < r1;
            const int64_t rowc = in ? row : r1 - 1;
            const int64_t hr = hidx ? hidx[rowc] : rowc;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const int c = lane + 64 * cg;
#ifdef LQ_WG_NOLOAD                /* timing-only ablation */
                const float gv = (float)(rowc & 7) * 0.125f, hv = (float)(c & 3);
#else
                const float gv = G[(size_t)rowc * J + (c < J ? c : J - 1)];
                const float hv = H[(size_t)hr * Kd + (c < Kd ? c : Kd - 1)];
#endif
                gqs[rr][cg] = (in && c < J) ? gv : 0.0f;
                hqs[rr][cg] = (in && c < Kd) ? hv : 0.0f;
            }
        }
    };
    auto block = [&](int64_t rb, float (&gqs)[4][CG], float (&hqs)[4][CG]) {
        // stage the block held in (gqs, hqs), refill that register set with block rb + 32 DEPTH, multiply
        lq_wg_barrier();                                   // the previous block's operands have been consumed (raw barrier:
                                                           // the two blocks of loads in flight stay in flight)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const int c = lane + 64 * cg;
                if (c < Jp) Gs[(4 * wave + rr) * Jp + c] = gqs[rr][cg];
                if (c < Kp) Hs[(4 * wave + rr) * Kp + c] = (c < Kd) ? lq_act_apply(hqs[rr][cg], h_act) : 0.0f;
            }
        lq_wg_barrier();
        fetch(rb + 32 * DEPTH, gqs, hqs);                  // (past the chunk: clamped addresses, zeros)
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int t = wave + q * WGW_WAVES;
            if (t < T) {                                   // wave-uniform
                const int ti = t / TJ, tj = t - ti * TJ;
                const float* ga = Gs + kh * Jp + 32 * ti + li;
                const float* hb = Hs + kh * Kp + 32 * tj + li;
                // all 32 LDS reads first, then the 16 MFMAs back to back (the rolled loop read two values, waited for the LDS
                // round trip and issued one MFMA, sixteen times per block)
                float av[16], bv[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) { av[s2] = ga[2 * s2 * Jp]; bv[s2] = hb[2 * s2 * Kp]; }
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
#ifdef LQ_WG_NOMFMA                /* timing-only ablation */
                    acc[q][s2] += av[s2] * bv[s2];
#else
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc[q], 0, 0, 0);
#endif
                    bsum[q] += av[s2];
                }
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(r0 + 32 * d, gq[d], hq[d]);
    const int ntrip = (int)((r1 - r0 + 32 * DEPTH - 1) / (32 * DEPTH));      // whole trips of DEPTH blocks; the surplus blocks multiply zeros
    for (int trip = 0; trip < ntrip; ++trip) {
        const int64_t rb = r0 + (int64_t)trip * 32 * DEPTH;
        block(rb, gq[0], hq[0]);
        block(rb + 32, gq[1], hq[1]);
        block(rb + 64, gq[2], hq[2]);
    }
    float* pw = partW + (size_t)blockIdx.x * J * Kd;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * WGW_WAVES;
        if (t >= T) continue;
        const int ti = t / TJ, tj = t - ti * TJ;
        const int fj = 32 * tj + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oi = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (oi < J && fj < Kd) pw[(size_t)oi * Kd + fj] = acc[q][r];
        }
        if (tj == 0) {
            const float tot = bsum[q] + __shfl_xor(bsum[q], 32, 64);
            const int fi = 32 * ti + li;
            if (kh == 0 && fi < J) partB[(size_t)blockIdx.x * J + fi] = tot;
        }
    }
}

#define WG4_ROWS 64
// Double-buffered variant (TI, TJ in {1, 2, 4}: every layer of the tokenizer at D <= 128).  The ISA of wgrad_wg4_kernel showed
// why it never left ~250 us at N = 524 288: its look-ahead registers live in arrays captured by a lambda, hipcc keeps them in
// SCRATCH (scratch_store right behind every global load, scratch_load + vmcnt(0) in front of every LDS write), and the runtime
// activation switch inlines the whole erf/exp tail into the staging code.  Here everything is straight-line over compile-time
// shapes: a block of 64 rows is 512 TI + 512 TJ float4s, exactly TI + TJ per thread (no remainder tests); two LDS buffers, so one
// barrier per block: compute block b from buffer b & 1, then write block b+1 (already in registers) into the other buffer and
// issue the loads of block b+2 -- a whole block of MFMAs to land.  The GELU is the straight-line polynomial with the rare
// |x| >= sqrt(18) elements fixed up behind a branch.  Same chunk-ordered, row-ordered accumulation: the same bits.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 wg5_act(f32x4 v, int act) {
    if (act == LIPVQ_ACT_GELU) {
        f32x4 o;
        o.x = lq_gelu_poly(v.x); o.y = lq_gelu_poly(v.y); o.z = lq_gelu_poly(v.z); o.w = lq_gelu_poly(v.w);
        const float m = fmaxf(fmaxf(v.x * v.x, v.y * v.y), fmaxf(v.z * v.z, v.w * v.w));
        if (!(m < 18.0f)) {                                            // rare (also taken for NaN: lq_gelu propagates it)
            o.x = lq_gelu(v.x); o.y = lq_gelu(v.y); o.z = lq_gelu(v.z); o.w = lq_gelu(v.w);
        }
        return o;
    }
    if (act == LIPVQ_ACT_RELU) return (f32x4){v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f};
    if (act == LIPVQ_ACT_SIGMOID) return (f32x4){lq_sigmoid(v.x), lq_sigmoid(v.y), lq_sigmoid(v.z), lq_sigmoid(v.w)};
    return v;
}

__device__ __forceinline__ f32x4 wg5_sel(bool keep, f32x4 v) {        // (a ?: on HIP's float4 STRUCT goes through scratch memory)
    return (f32x4){keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f};
}

template <int TI, int TJ, int ROWS>
__global__ __launch_bounds__(64 * WGW_WAVES) void wgrad_wg5_kernel(const float* __restrict__ G, const float* __restrict__ H,
                                                                   const int64_t* __restrict__ hidx, int h_act,
                                                                   float* __restrict__ partW, float* __restrict__ partB,
                                                                   int64_t N, int J, int Kd, int chunk_rows, int ldG) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    constexpr int Jp = 32 * TI, Kp = 32 * TJ, Jp4 = Jp / 4, Kp4 = Kp / 4;
    // a G wider than 224 columns is processed in column blocks of Jp (grid y): ldG = G's row stride = the slabs' row count
    const int Jall = J, c0 = (int)blockIdx.y * Jp;
    G += c0;
    J = Jall - c0 < Jp ? Jall - c0 : Jp;
    constexpr int T = TI * TJ, TPW = (T + WGW_WAVES - 1) / WGW_WAVES;
    constexpr int BUF = ROWS * (Jp + Kp);                     // floats per LDS buffer: [ROWS][Jp] of G, then [ROWS][Kp] of act(H)
    constexpr int NG = (ROWS * Jp4 + 511) / 512, NH = (ROWS * Kp4 + 511) / 512;      // float4 slots per thread and block
    constexpr bool GFULL = (ROWS * Jp4) % 512 == 0, HFULL = (ROWS * Kp4) % 512 == 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int J4 = J >> 2, K4 = Kd >> 2;
    // whole-float4 rows (16-byte aligned bases: host check).  Only a one-tile operand may have odd rows (fan-in 7: element loads);
    // for the wide instantiations the test is a compile-time constant -- as a runtime flag it cost the hot shape 45 us
    const bool vecG = TI > 1 || (J & 3) == 0, vecH = TJ > 1 || (Kd & 3) == 0;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t r1 = r0 + chunk_rows;
    if (r1 > N) r1 = N;
    const int nblk = (int)((r1 - r0 + ROWS - 1) / ROWS);

    f32x16 acc[TPW];
    float bsum[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        bsum[q] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    }
    // this thread's float4 slots of a block (slot u: float4 tid + 512 u of the [ROWS][Jp4] / [ROWS][Kp4] image).  PF register
    // sets = blocks in flight; more than one bought nothing (PF = 4 at 64 rows: 178 -> 184 us), occupancy did (see the launch).
    constexpr int PF = 1;
    f32x4 gq[PF][NG], hq[PF][NH];
    auto fetchG = [&](int64_t rb, int u) -> f32x4 {                    // clamped address, masked value: no branch
        const int i = tid + 512 * u;
        const int row = (i / Jp4) < ROWS ? (i / Jp4) : ROWS - 1, col = i % Jp4;
        const int64_t rr = rb + row;
        const int64_t rc = rr < r1 ? rr : r1 - 1;
        if (vecG) {
            const f32x4 v = reinterpret_cast<const f32x4*>(G + (size_t)rc * ldG)[col < J4 ? col : J4 - 1];
            return wg5_sel(rr < r1 && col < J4, v);
        }
        f32x4 v;                                                       // rows that are not whole float4s (fan-in 7): element loads
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * col + e;
            const float t = G[(size_t)rc * ldG + (c < J ? c : J - 1)];
            v[e] = (rr < r1 && c < J) ? t : 0.0f;
        }
        return v;
    };
    auto fetchH = [&](int64_t rb, int u) -> f32x4 {
        const int i = tid + 512 * u;
        const int row = (i / Kp4) < ROWS ? (i / Kp4) : ROWS - 1, col = i % Kp4;
        const int64_t rr = rb + row;
        const int64_t rc = rr < r1 ? rr : r1 - 1;
        const int64_t hr = hidx ? hidx[rc] : rc;
        if (vecH) return reinterpret_cast<const f32x4*>(H + (size_t)hr * Kd)[col < K4 ? col : K4 - 1];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * col + e;
            v[e] = H[(size_t)hr * Kd + (c < Kd ? c : Kd - 1)];
        }
        return v;
    };
    auto fetch = [&](int slot, int64_t rb) {
#pragma unroll
        for (int u = 0; u < NG; ++u) gq[slot][u] = fetchG(rb, u);
#pragma unroll
        for (int u = 0; u < NH; ++u) hq[slot][u] = fetchH(rb, u);
    };
    auto stage = [&](float* buf, int slot, int64_t rb) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = tid + 512 * u;
            if (GFULL || i < ROWS * Jp4) reinterpret_cast<f32x4*>(buf)[i] = gq[slot][u];    // (i = row * Jp4 + col)
        }
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int i = tid + 512 * u;
            if (HFULL || i < ROWS * Kp4) {
                const int row = i / Kp4, col = i % Kp4;
                const f32x4 v = wg5_act(hq[slot][u], h_act);           // activation, then the masks (act(0) need not be 0)
                const bool in = rb + row < r1;
                reinterpret_cast<f32x4*>(buf + ROWS * Jp)[i] =
                    (f32x4){(in && 4 * col + 0 < Kd) ? v.x : 0.f, (in && 4 * col + 1 < Kd) ? v.y : 0.f,
                            (in && 4 * col + 2 < Kd) ? v.z : 0.f, (in && 4 * col + 3 < Kd) ? v.w : 0.f};
            }
        }
    };

#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, r0 + (int64_t)d * ROWS);     // (past the chunk: clamped addresses, staged as zeros)
    stage(wg_lds, 0, r0);
    fetch(0, r0 + (int64_t)PF * ROWS);
    lq_wg_barrier();
    for (int b = 0; b < nblk; b += PF) {
#pragma unroll
        for (int pu = 0; pu < PF; ++pu) {
            const int bb = b + pu;
            if (bb >= nblk) break;                                 // workgroup-uniform
            const float* cur = wg_lds + (bb & 1) * BUF;
            float* nxt = wg_lds + ((bb + 1) & 1) * BUF;
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int t = wave + q * WGW_WAVES;
                if (t < T) {                                       // wave-uniform
                    const int ti = t / TJ, tj = t % TJ;
                    const float* ga = cur + kh * Jp + 32 * ti + li;
                    const float* hb = cur + ROWS * Jp + kh * Kp + 32 * tj + li;
#pragma unroll
                    for (int part = 0; part < ROWS / 32; ++part) {
                        float av[16], bv[16];
#pragma unroll
                        for (int s2 = 0; s2 < 16; ++s2) { av[s2] = ga[2 * (16 * part + s2) * Jp]; bv[s2] = hb[2 * (16 * part + s2) * Kp]; }
#ifndef LQ_WG5_NO_PIN
                        // all 32 operand reads are REQUESTED before the first MFMA (round 3: left alone, hipcc sinks them to two in
                        // front of each MFMA pair behind `s_waitcnt lgkmcnt(0)` -- an LDS round trip exposed per 128 cycles of MFMA)
                        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                        for (int s2 = 0; s2 < 16; ++s2) {
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc[q], 0, 0, 0);
                            bsum[q] += av[s2];
                        }
                    }
                }
            }
            // block bb+1 (in registers for PF iterations) -> the other buffer, whose last readers passed the barrier that ended
            // iteration bb-1; then its register slot takes the loads of block bb+1+PF
            const int slot = (pu + 1) % PF;
            const int64_t rb1 = r0 + (int64_t)(bb + 1) * ROWS;
            stage(nxt, slot, rb1);
            fetch(slot, rb1 + (int64_t)PF * ROWS);
            lq_wg_barrier();
        }
    }
    float* pw = partW + (size_t)blockIdx.x * Jall * Kd + (size_t)c0 * Kd;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * WGW_WAVES;
        if (t >= T) continue;
        const int ti = t / TJ, tj = t % TJ;
        const int fj = 32 * tj + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oi = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (oi < J && fj < Kd) pw[(size_t)oi * Kd + fj] = acc[q][r];
        }
        if (tj == 0) {
            const float tot = bsum[q] + __shfl_xor(bsum[q], 32, 64);
            const int fi = 32 * ti + li;
            if (kh == 0 && fi < J) partB[(size_t)blockIdx.x * Jall + c0 + fi] = tot;
        }
    }
}

// gW / gb = sum over chunks of the partial slabs (one launch for both): 32 elements x 32 chunk groups per block (group g sums
// chunks g, g+32, ... in double, four loads in flight), the 32 group sums are combined in a fixed order: deterministic.  (With 8
// groups of 256 threads the 261 blocks of a 128x64 layer read their 16.7 MB at 0.9 TB/s: 18.6 us per layer, 112 us per step.)
#define WGR_GROUPS 32
__global__ __launch_bounds__(32 * WGR_GROUPS) void wgrad_reduce_kernel(const float* __restrict__ partW, const float* __restrict__ partB,
                                                                      float* __restrict__ gW, float* __restrict__ gb, int nchunks,
                                                                      size_t n_w, size_t n_b) {
    __shared__ double sh[WGR_GROUPS][33];
    const int le = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + le;             // [0, n_w): weight gradient, [n_w, n_w + n_b): bias gradient
    const bool isw = e < n_w, isb = !isw && e < n_w + n_b;
    const float* part = isw ? partW : partB;
    const size_t stride = isw ? n_w : n_b, off = isw ? e : e - n_w;
    double s = 0.0;
    if (isw || isb) {
        int c = g;
        for (; c + 3 * WGR_GROUPS < nchunks; c += 4 * WGR_GROUPS) {
            const float v0 = part[(size_t)c * stride + off], v1 = part[(size_t)(c + WGR_GROUPS) * stride + off];
            const float v2 = part[(size_t)(c + 2 * WGR_GROUPS) * stride + off], v3 = part[(size_t)(c + 3 * WGR_GROUPS) * stride + off];
            s += (double)v0;
            s += (double)v1;
            s += (double)v2;
            s += (double)v3;
        }
        for (; c < nchunks; c += WGR_GROUPS) s += (double)part[(size_t)c * stride + off];
    }
    sh[g][le] = s;
    __syncthreads();
    if (g == 0 && (isw || isb)) {
        double t = sh[0][le];
#pragma unroll
        for (int q = 1; q < WGR_GROUPS; ++q) t += sh[q][le];
        (isw ? gW : gb)[off] = (float)t;
    }
}

static inline int wgrad_chunks(int64_t N) { const int c = wgrad_chunk_rows(N); return (int)((N + c - 1) / c); }

extern "C" size_t lipvq_wgrad_workspace_bytes(int64_t N, int J, int Kd) {
    if (N <= 0 || J <= 0 || Kd <= 0) return 0;
    return (size_t)wgrad_chunks(N) * ((size_t)J * Kd + J) * sizeof(float);
}

// gW [J][Kd] = G^T . act(H),  gb [J] = column sums of G.   G [N][J];  H [N][Kd] or, with hidx,
// row n of H is H[hidx[n]].  h_act is applied to H elements on load (the forward activation of
// a saved pre-activation).  gb may be NULL.
extern "C" int lipvq_wgrad_f32(const float* G, const float* H, const int64_t* hidx, int h_act, float* gW,
                               float* gb, void* workspace, int64_t N, int J, int Kd, void* stream) {
    if (!G || !H || !gW || !workspace || N <= 0 || J <= 0 || Kd <= 0) return fail(LIPVQ_EINVAL, "wgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nch = wgrad_chunks(N);
    const int TI = (J + 31) / 32, TJ = (Kd + 31) / 32;
    float* partW = (float*)workspace;
    float* partB = partW + (size_t)nch * J * Kd;
    const size_t lds = (size_t)32 * 32 * (TI + TJ) * sizeof(float);
    static int use_wg = -1;                     // LIPVQ_WGRAD_PER_TILE=1 forces the one-wave-per-tile kernel (measurement knob)
    if (use_wg < 0) use_wg = lq_knob("LIPVQ_WGRAD_PER_TILE") ? 0 : 1;
    static int use_wg5 = -1;                    // LIPVQ_WGRAD_NO_WG5=1: the single-buffered kernels (measurement knob)
    if (use_wg5 < 0) use_wg5 = lq_knob("LIPVQ_WGRAD_NO_WG5") ? 0 : 1;
    // one chunk (training-step batches of <= 128 rows): the kernel's slab IS the result, no reduce launch
    const bool direct = nch == 1;
    if (direct) { partW = gW; if (gb) partB = gb; }
    auto tcode = [](int t) { return t == 1 ? 0 : t == 2 ? 1 : t == 4 ? 2 : t == 7 ? 3 : -1; };     // tile counts with an instance: 32, 64, 128, 208+ wide
    // G wider than 224 columns (the embedding Linear's 512): column blocks of 128 through the four-tile instance
    // (1.32 ms -> see DESIGN.md for N = 524 280, J = 512, Kd = 64; the per-tile kernel below read G at 0.9 TB/s)
    const bool wide = TI > 7 && J % 128 == 0;
    const int TIk = wide ? 4 : TI, ycount = wide ? J / 128 : 1;
    const int ci = tcode(TIk), cj = tcode(TJ);
    if (use_wg && use_wg5 && ci >= 0 && cj >= 0 && ((J & 3) == 0 || TI == 1) && ((Kd & 3) == 0 || TJ == 1) &&
        (((uintptr_t)G | (uintptr_t)H) & 15) == 0) {
        typedef void (*wg5_fn)(const float*, const float*, const int64_t*, int, float*, float*, int64_t, int, int, int, int);
        wg5_fn kfn = nullptr;
        // 64-row blocks for the narrow pairs; 32-row blocks from 192 columns on (TI + TJ >= 6): two 64-row buffers of those are
        // 98 KB and more -- one workgroup per CU, every barrier and staging phase exposed -- while at 32 rows two or three
        // workgroups share a CU and fill each other's gaps: 128x64 178 -> 146 us, 64x128 217 -> 174 us at N = 524 288 (the narrow
        // pairs lose 5-15 % at 32 rows; a deeper register prefetch instead of occupancy changed nothing).
        // LIPVQ_WGRAD_ROWS=32|64 forces one size (measurement knob).
        static int rows_knob = -1;
        if (rows_knob < 0) { const char* e = lq_knob("LIPVQ_WGRAD_ROWS"); rows_knob = e ? atoi(e) : 0; }
        const bool r64 = TIk + TJ <= 10 && (rows_knob == 64 || (rows_knob != 32 && TIk + TJ < 6));
#define LQ_W5(TI_, TJ_) if (TIk == TI_ && TJ == TJ_) kfn = r64 ? (wg5_fn)wgrad_wg5_kernel<TI_, TJ_, (TI_ + TJ_ <= 10 ? 64 : 32)> : (wg5_fn)wgrad_wg5_kernel<TI_, TJ_, 32>;
        LQ_W5(1, 1) LQ_W5(1, 2) LQ_W5(1, 4) LQ_W5(1, 7) LQ_W5(2, 1) LQ_W5(2, 2) LQ_W5(2, 4) LQ_W5(2, 7)
        LQ_W5(4, 1) LQ_W5(4, 2) LQ_W5(4, 4) LQ_W5(4, 7) LQ_W5(7, 1) LQ_W5(7, 2) LQ_W5(7, 4) LQ_W5(7, 7)
#undef LQ_W5
        const int rows5 = r64 ? 64 : 32;
        const size_t lds5 = (size_t)2 * rows5 * 32 * (TIk + TJ) * sizeof(float);
        static LqLdsReserve reserved5[32];          // per instantiation: per-device, thread-safe (lipvq_common.h)
        if (lds5 > 64 * 1024)
            if (int rc = lipvq_reserve_lds(reserved5[(r64 ? 16 : 0) + ci * 4 + cj], (const void*)kfn, lds5, "wgrad")) return rc;
        hipLaunchKernelGGL(kfn, dim3(nch, ycount), dim3(64 * WGW_WAVES), lds5, st, G, H, hidx, h_act, partW, partB, N, J, Kd,
                           wgrad_chunk_rows(N), J);
    } else if (use_wg && TI * TJ <= WGW_WAVES * WGW_MAXT && TI <= 8 && TJ <= 8 && lds <= 64 * 1024) {
        const int wide = 32 * (TI > TJ ? TI : TJ);
        const int tpw = (TI * TJ + WGW_WAVES - 1) / WGW_WAVES;
        typedef void (*wg_fn)(const float*, const float*, const int64_t*, int, float*, float*, int64_t, int, int, int, int, int);
        wg_fn kfn;
#define LQ_WG(cg_) (tpw <= 1 ? (wg_fn)wgrad_wg_kernel<cg_, 1> : tpw <= 2 ? (wg_fn)wgrad_wg_kernel<cg_, 2> : (wg_fn)wgrad_wg_kernel<cg_, 4>)
        kfn = wide <= 64 ? LQ_WG(1) : (wide <= 128 ? LQ_WG(2) : LQ_WG(4));
#undef LQ_WG
        hipLaunchKernelGGL(kfn, dim3(nch), dim3(64 * WGW_WAVES), lds, st, G, H, hidx, h_act, partW, partB, N, J, Kd, TI, TJ,
                           wgrad_chunk_rows(N));
    } else {
        hipLaunchKernelGGL(wgrad_kernel, dim3(nch, TI * TJ), dim3(64), 0, st, G, H, hidx, h_act, partW, partB, N, J, Kd, TJ,
                           wgrad_chunk_rows(N));
    }
    if (direct) return check_launch("wgrad");
    const size_t ne = (size_t)J * Kd, nb = gb ? (size_t)J : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((ne + nb + 31) / 32)), dim3(32 * WGR_GROUPS), 0, st, partW, partB, gW, gb, nch, ne,
                       nb);
    return check_launch("wgrad");
}

// gC[idx[n]][d] += g[n][d].  Float atomics (global_atomic_add_f32): the sum order, hence the last
// bits, can differ between runs; tests compare with 1e-5.  gC must be zeroed by the caller.
__global__ void scatter_add_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                   float* __restrict__ gC, int64_t N, int D) {
    const size_t n_elem = (size_t)N * D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (size_t)gridDim.x * blockDim.x) {
        const size_t n = e / D;
        const int d = (int)(e - n * D);
        atomicAdd(&gC[(size_t)idx[n] * D + d], g[e]);
    }
}

// Deterministic variant: one workgroup per code k, one wave per 64 columns; every wave walks idx in row order (ballot over
// 64 rows at a time) and adds the matching rows of g in ascending row order -- the result does not depend on scheduling
// and equals a sequential fp32 index_add_ over n = 0..N-1.  Cost O(K N / 64) ballots (idx stays in L2): ~0.5 ms at
// N = 524 288, K = 1024 against 0.24 ms for the atomic kernel; microseconds at training-step sizes.
__global__ __launch_bounds__(256) void scatter_add_det_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                                              float* __restrict__ gC, int64_t N, int D) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    for (int d0 = wave * 64; d0 < D; d0 += nw * 64) {
        const int d = d0 + lane;
        float acc = (d < D) ? gC[(size_t)k * D + d] : 0.0f;
        for (int64_t base = 0; base < N; base += 64) {
            const int64_t n = base + lane;
            unsigned long long m = __builtin_amdgcn_ballot_w64(n < N && idx[n] == (int64_t)k);
            while (m) {                                        // wave-uniform
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                if (d < D) acc = acc + g[(size_t)(base + b) * D + d];
            }
        }
        if (d < D) gC[(size_t)k * D + d] = acc;
    }
}

extern "C" int lipvq_scatter_add_det_f32(const float* g, const int64_t* idx, float* gC, int64_t N, int K, int D,
                                         void* stream) {
    if (!g || !idx || !gC || N < 0 || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "scatter_add_det: bad argument");
    if (N == 0) return LIPVQ_OK;
    const int waves = D <= 64 ? 1 : (D <= 128 ? 2 : 4);
    hipLaunchKernelGGL(scatter_add_det_kernel, dim3((unsigned)K), dim3(64 * waves), 0, (hipStream_t)stream, g, idx, gC, N, D);
    return check_launch("scatter_add_det");
}

// Large batches: a workgroup owns a slice of DC columns and a chunk of rows, accumulates its [K][DC] share in LDS (ds_add_f32:
// two rows of a wave collide only when they picked the same code) and flushes it with one global atomic per non-zero element.
// The plain kernel above issues N D global atomics onto K D addresses: 286 us at N = 524 288, K = 1024, D = 64 (L2 atomic
// throughput); this one 4-8x fewer, the rest at LDS speed.  What bounds it now is ds_add_f32 itself (~3 cycles per lane-add and
// CU: halving the workgroups doubles the time, and one 16-wave workgroup per CU with bank-padded rows and half the flush
// atomics ran the same 185-190 us).
template <int DC>
__global__ __launch_bounds__(256) void scatter_add_lds_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                                              float* __restrict__ gC, int64_t N, int K, int D, int64_t rows_per_wg) {
    extern __shared__ float sc_lds[];                         // [K][DC]
    constexpr int V = DC / 4;                                 // float4s per row of the slice
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * DC;
    for (int i = tid; i < K * DC; i += 256) sc_lds[i] = 0.0f;
    __syncthreads();
    const int64_t rb = (int64_t)blockIdx.x * rows_per_wg;
    int64_t re = rb + rows_per_wg;
    if (re > N) re = N;
    const int q = tid % V, rl = tid / V;                      // float4 q of the slice, local row
    // eight rows per thread at a time: all their loads first (the atomics are ordering points the compiler will not move a load
    // across: one row at a time paid one L2 round trip per row)
    constexpr int UNR = 8, STEP = 256 / V;
    for (int64_t r = rb + rl; r < re; r += (int64_t)UNR * STEP) {
        float4 v[UNR];
        int64_t k[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t rr = r + (int64_t)u * STEP;
            const int64_t rc = rr < re ? rr : re - 1;
            v[u] = *reinterpret_cast<const float4*>(g + (size_t)rc * D + c0 + 4 * q);
            k[u] = idx[rc];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (r + (int64_t)u * STEP < re) {
                float* dst = sc_lds + (size_t)k[u] * DC + 4 * q;
                atomicAdd(dst + 0, v[u].x); atomicAdd(dst + 1, v[u].y); atomicAdd(dst + 2, v[u].z); atomicAdd(dst + 3, v[u].w);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < K * DC; i += 256) {
        const float v = sc_lds[i];
        if (v != 0.0f) atomicAdd(&gC[(size_t)(i / DC) * D + c0 + (i % DC)], v);
    }
}

extern "C" int lipvq_scatter_add_f32(const float* g, const int64_t* idx, float* gC, int64_t N, int K, int D,
                                     void* stream) {
    if (!g || !idx || !gC || N < 0 || K <= 0 || D <= 0) return fail(LIPVQ_EINVAL, "scatter_add: bad argument");
    if (N == 0) return LIPVQ_OK;
    {
        int dc = 16;
        while (dc >= 4 && ((size_t)K * dc * sizeof(float) > 64 * 1024 || D % dc != 0)) dc >>= 1;
        if (N >= 32768 && dc >= 4 && (((uintptr_t)g) & 15) == 0) {
            const int slices = D / dc;
            int64_t chunks = 512 / slices;                     // ~512 workgroups in all (two per CU fit LDS): few, large flushes
            if (chunks < 1) chunks = 1;
            const int64_t rows_per_wg = (N + chunks - 1) / chunks;
            const dim3 grid((unsigned)((N + rows_per_wg - 1) / rows_per_wg), (unsigned)slices);
            const size_t lds = (size_t)K * dc * sizeof(float);
            if (dc == 16) hipLaunchKernelGGL(scatter_add_lds_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, g, idx, gC, N, K, D, rows_per_wg);
            else if (dc == 8) hipLaunchKernelGGL(scatter_add_lds_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, g, idx, gC, N, K, D, rows_per_wg);
            else hipLaunchKernelGGL(scatter_add_lds_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, g, idx, gC, N, K, D, rows_per_wg);
            return check_launch("scatter_add_lds");
        }
    }
    size_t ne = (size_t)N * D;
    size_t blocks = (ne + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, idx, gC, N, D);
    return check_launch("scatter_add");
}

// Backward of  Wn = W * sc,  sc = min(1, softplus(ci)/sum|W|)   (v5:6-12).  16 rows per workgroup, staged in LDS (coalesced
// loads and stores; 16 threads run the per-row sums -- the fp32 one in index order, because it must reproduce the forward
// kernel's `active` test bit for bit).  One thread per row from global memory took 33 us at D = 208.
#define LIPB_ROWS 16
#define LIPB_HMAX 256
__global__ __launch_bounds__(256) void lipschitz_bwd_kernel(const float* __restrict__ W, const float* __restrict__ ci,
                                                            const float* __restrict__ gWn, float* __restrict__ gW,
                                                            float* __restrict__ gci, int D, int H) {
    __shared__ float ws[LIPB_ROWS][LIPB_HMAX + 1], gs[LIPB_ROWS][LIPB_HMAX + 1];
    __shared__ double s_sc[LIPB_ROWS], s_k[LIPB_ROWS];
    const int tid = threadIdx.x, r0 = blockIdx.x * LIPB_ROWS;
    for (int i = tid; i < LIPB_ROWS * H; i += 256) {
        const int r = i / H, j = i - r * H;
        const bool in = r0 + r < D;
        ws[r][j] = in ? W[(size_t)(r0 + r) * H + j] : 0.0f;
        gs[r][j] = in ? gWn[(size_t)(r0 + r) * H + j] : 0.0f;
    }
    __syncthreads();
    if (tid < LIPB_ROWS && r0 + tid < D) {
        double s = 0.0, gsc = 0.0;
        float s32 = 0.0f;
        for (int j = 0; j < H; ++j) {
            const float w = ws[tid][j];
            s += fabs((double)w);
            s32 = s32 + lq_abs(w);
            gsc += (double)gs[tid][j] * (double)w;
        }
        const float cv = ci[r0 + tid];
        const float spf = lq_softplus(cv);
        const bool active = (spf / s32) < 1.0f;        // the same test the forward kernel makes
        const double sp = (double)spf;
        s_sc[tid] = active ? sp / s : 1.0;
        s_k[tid] = active ? gsc * (-sp / (s * s)) : 0.0;
        gci[r0 + tid] = active ? (float)(gsc * (double)lq_sigmoid(cv) / s) : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < LIPB_ROWS * H; i += 256) {
        const int r = i / H, j = i - r * H;
        if (r0 + r < D) {
            const double wj = (double)ws[r][j];
            gW[(size_t)(r0 + r) * H + j] = (float)((double)gs[r][j] * s_sc[r] + s_k[r] * ((wj > 0) - (wj < 0)));
        }
    }
}

extern "C" int lipvq_lipschitz_bwd_f32(const float* W, const float* ci, const float* gWn, float* gW, float* gci,
                                       int D, int H, void* stream) {
    if (!W || !ci || !gWn || !gW || !gci || D <= 0 || H <= 0) return fail(LIPVQ_EINVAL, "lipschitz_bwd: bad argument");
    if (H > LIPB_HMAX) return fail(LIPVQ_EUNSUPPORTED, "lipschitz_bwd: H=%d > %d", H, LIPB_HMAX);
    hipLaunchKernelGGL(lipschitz_bwd_kernel, dim3((D + LIPB_ROWS - 1) / LIPB_ROWS), dim3(256), 0, (hipStream_t)stream, W, ci, gWn, gW,
                       gci, D, H);
    return check_launch("lipschitz_bwd");
}

// out = alpha * (gscale ? *gscale : 1) * (a - b) + (c ? c : 0)
__global__ void scaled_diff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                   const float* __restrict__ c, float alpha, const float* __restrict__ gscale,
                                   float* __restrict__ out, int64_t n) {
    const float f = gscale ? alpha * gscale[0] : alpha;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = f * (a[i] - b[i]);                      // (built with -ffp-contract=off: product and sum round separately,
        out[i] = c ? v + c[i] : v;                              //  here and in mlp3_lds_kernel's folded terms)
    }
}

extern "C" int lipvq_scaled_diff_f32(const float* a, const float* b, const float* c, float alpha,
                                     const float* gscale, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return fail(LIPVQ_EINVAL, "scaled_diff: bad argument");
    if (n == 0) return LIPVQ_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scaled_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, alpha, gscale, out, n);
    return check_launch("scaled_diff");
}

// ---------------------------------------------------------------------------------------------------
// Opt-in EMA codebook update (NOT in the reference, which trains the codebook by gradient, v5:32-35,81; named by the
// north star and SURVEY 8e as an extension).  The standard VQ-VAE rule (van den Oord et al. 2017, appendix A.1 /
// sonnet's VectorQuantizerEMA):
//     cluster_size = decay * cluster_size + (1 - decay) * counts
//     embed_sum    = decay * embed_sum    + (1 - decay) * dw,       dw[k] = sum of z_e rows mapped to k
//     n = sum(cluster_size);  smoothed = (cluster_size + eps) / (n + K eps) * n;   codebook = embed_sum / smoothed
// ema_cluster_kernel: one workgroup; tree-sums n in double (deterministic) into workspace[0].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ema_cluster_kernel(float* __restrict__ cs, const int64_t* __restrict__ counts,
                                                           float decay, int K, double* __restrict__ n_out) {
    __shared__ double part[1024];
    const float omd = 1.0f - decay;
    double acc = 0.0;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float v = lq_fma(decay, cs[k], omd * (float)counts[k]);
        cs[k] = v;
        acc += (double)v;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = part[0];
}

__global__ __launch_bounds__(256) void ema_codebook_kernel(const float* __restrict__ cs, float* __restrict__ es,
                                                           const float* __restrict__ dw, float* __restrict__ codebook,
                                                           float decay, float eps, int K, int D,
                                                           const double* __restrict__ n_in) {
    const float omd = 1.0f - decay;
    const float n = (float)*n_in;
    const float denom = n + (float)K * eps;
    const int64_t total = (int64_t)K * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int k = (int)(e / D);
        const float sm = (cs[k] + eps) / denom * n;
        const float v = lq_fma(decay, es[e], omd * dw[e]);
        es[e] = v;
        codebook[e] = v / sm;
    }
}

extern "C" int lipvq_ema_update_f32(float* cluster_size, float* embed_sum, const int64_t* counts, const float* dw,
                                    float* codebook, float decay, float eps, int K, int D, void* workspace, void* stream) {
    if (!cluster_size || !embed_sum || !counts || !dw || !codebook || !workspace)
        return fail(LIPVQ_EINVAL, "lipvq_ema_update_f32: null pointer");
    if (K <= 0 || D <= 0 || !(decay >= 0.0f && decay <= 1.0f) || !(eps > 0.0f))
        return fail(LIPVQ_EINVAL, "lipvq_ema_update_f32: bad arguments (K=%d D=%d decay=%g eps=%g)", K, D, decay, eps);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ema_cluster_kernel, dim3(1), dim3(1024), 0, st, cluster_size, counts, decay, K, (double*)workspace);
    int rc = check_launch("ema_cluster_kernel");
    if (rc) return rc;
    int64_t g = ((int64_t)K * D + 1023) / 1024;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(ema_codebook_kernel, dim3((unsigned)g), dim3(256), 0, st, cluster_size, embed_sum, dw, codebook, decay,
                       eps, K, D, (const double*)workspace);
    return check_launch("ema_codebook_kernel");
}

// ---------------------------------------------------------------------------------------------------
// AdamW for the tokenizer's parameter list in TWO launches (reference robomimic/algo/icl.py:885-889 builds
// optim.AdamW(vq_vae_model.parameters(), lr=1e-3, weight_decay=1e-4); :970 vq_optimizer.step()).  torch's foreach AdamW
// is 8-10 launches over the 14 tensors (~80 us of GPU time at the ICRT step shape, as much as the encoder's forward and
// backward launches together, plus their Python issue cost); one launch per tensor LIST is what a launch-bound step wants.
// Same update as torch.optim.AdamW (amsgrad = False, maximize = False):
//     step += 1;  p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//     p -= (lr / (1 - b1^step)) m / (sqrt(v) / sqrt(1 - b2^step) + eps)
// `step` is a float32 device scalar PER TENSOR (torch's capturable layout: the optimizer state_dict stays interchangeable),
// so the whole step is capturable in a HIP graph.
// ---------------------------------------------------------------------------------------------------
#define LIPVQ_ADAMW_MAX 32
struct AdamwArgs {
    float* p[LIPVQ_ADAMW_MAX];
    const float* g[LIPVQ_ADAMW_MAX];
    float* m[LIPVQ_ADAMW_MAX];
    float* v[LIPVQ_ADAMW_MAX];
    float* step[LIPVQ_ADAMW_MAX];
    long long n[LIPVQ_ADAMW_MAX];
    int count;
};

__global__ void adamw_steps_kernel(AdamwArgs a, double beta1, double beta2, float* __restrict__ bc) {
    const int t = threadIdx.x;
    if (t >= a.count) return;
    const float s = a.step[t][0] + 1.0f;
    a.step[t][0] = s;
    // 1 - beta^step in double (torch computes the power in the step tensor's precision on the capturable path; the difference
    // is far below the 1e-5 the fixtures are held to)
    bc[2 * t] = (float)(1.0 - pow(beta1, (double)s));
    bc[2 * t + 1] = (float)sqrt(1.0 - pow(beta2, (double)s));
}

// the scalar constants are formed in double on the host, as torch forms them in Python, and rounded to fp32 once
// (1 - 0.999f is not 0.001f: the second moment drifted by 1.3e-5 relative)
__global__ __launch_bounds__(256) void adamw_kernel(AdamwArgs a, float lr, float decay, float omb1, float beta2, float omb2, float eps,
                                                    const float* __restrict__ bc) {
    const int t = blockIdx.y;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const long long n = a.n[t];
    const float step_size = lr / bc[2 * t], bc2s = bc[2 * t + 1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * decay;
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = v[i] * beta2 + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        pi = pi - step_size * (mi / (lq_sqrt(vi) / bc2s + eps));
        p[i] = pi;
    }
}

extern "C" size_t lipvq_adamw_workspace_bytes(void) { return 2 * LIPVQ_ADAMW_MAX * sizeof(float); }

extern "C" int lipvq_adamw_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                               float* const* steps, const int64_t* numels, int count, double lr, double beta1, double beta2, double eps,
                               double weight_decay, void* workspace, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !numels || !workspace)
        return fail(LIPVQ_EINVAL, "adamw: null pointer");
    if (count <= 0 || count > LIPVQ_ADAMW_MAX) return fail(LIPVQ_EINVAL, "adamw: %d tensors (1..%d per call)", count, LIPVQ_ADAMW_MAX);
    AdamwArgs a;
    long long nmax = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || !steps[i] || numels[i] <= 0)
            return fail(LIPVQ_EINVAL, "adamw: tensor %d has a null pointer or no elements", i);
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i]; a.step[i] = steps[i];
        a.n[i] = numels[i];
        if (numels[i] > nmax) nmax = numels[i];
    }
    a.count = count;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_steps_kernel, dim3(1), dim3(64), 0, st, a, beta1, beta2, (float*)workspace);
    long long gx = (nmax + 1023) / 1024;
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)gx, count), dim3(256), 0, st, a, (float)lr, (float)(1.0 - lr * weight_decay),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (const float*)workspace);
    return check_launch("adamw");
}
