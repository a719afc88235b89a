// lipvq_mlp.hip -- the three-layer perceptron stacks of the tokenizer, forward and backward-data, as fp32-MFMA kernels
// (v_mfma_f32_32x32x2_f32): mlp3_wg_kernel (8 waves per 32-row tile, activations cross layers through LDS; what the
// entry points launch) and mlp3_kernel (one wave per tile, activations never leave registers; fallback for very wide
// inputs).  ABI and reference citations: include/lipvq.h.  Arithmetic: lipvq_math.h.
//
// Layout.  The layers are evaluated TRANSPOSED, Y^T = W . X^T, with the weights as the MFMA
// A operand (32 output features x 2 k) and the activations as the B operand (2 k x 32 rows).
// The 32x32 result tile then has the batch row on the lane (col = lane & 31) and 16 output
// features in the lane's registers, which is exactly the B-operand shape of the next layer:
// register r of lane-half h feeds k-step r with k = 2r + h.  Choosing the feature <-> tile-row
// map  feature(i) = 2*((i & 3) + 4*(i >> 3)) + ((i >> 2) & 1)  makes that k order the natural
// 0,1,2,... order, so every output is ONE k-ordered fmaf chain starting from the bias -- the
// oracle's definition -- with no LDS round trip and no cross-lane traffic between layers.
//
// Backward-data reuses the same chain with the transposed weights (J2 -> J1 -> J0 -> K0, zero
// biases): the "activation" after a layer becomes a multiplication by act'(saved pre-activation)
// and the per-layer results g1, g0 are written out for the weight-gradient GEMMs (lipvq_bwd.hip).
#include <stdlib.h>

#include "lipvq_mlp.h"

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
// P[(t*S + s)*64 + lane] = Wv[32t + feat(lane & 31)][2s + (lane >> 5)]   (0 outside), where the
// virtual weight Wv[f][k] = W[f*sf + k*sk] (sf,sk select W or its transpose); J x K = its shape.
// One launch packs the three layers of a stack.
struct PackLayer {
    const float* W;
    const float* b;
    float* packed;             // the stack's packed buffer
    size_t oP, oB, n;          // offsets into the packed buffer; threads this layer needs
    int K, J, S, T, sf, sk;
};
#define PACK_MAX_LAYERS 6      // two stacks per launch
struct PackArgs {
    PackLayer l[PACK_MAX_LAYERS];
    int nl;
};

__global__ void mlp3_pack_kernel(const PackArgs a) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int li = 0;
    while (li < a.nl - 1 && gid >= a.l[li].n) { gid -= a.l[li].n; ++li; }
    const PackLayer& L = a.l[li];
    if (gid >= L.n) return;
    size_t nP = (size_t)L.T * L.S * 64;
    if (gid < nP) {
        int lane = (int)(gid & 63);
        size_t ts = gid >> 6;
        int s = (int)(ts % L.S), t = (int)(ts / L.S);
        int f = 32 * t + feat_of_tile_row(lane & 31);
        int k = 2 * s + (lane >> 5);
        L.packed[L.oP + gid] = (f < L.J && k < L.K) ? L.W[(size_t)f * L.sf + (size_t)k * L.sk] : 0.0f;
    }
    if (gid < (size_t)L.T * 32) L.packed[L.oB + gid] = (L.b && (int)gid < L.J) ? L.b[gid] : 0.0f;
}

extern "C" size_t lipvq_mlp3_packed_floats(int K0, int J0, int J1, int J2) {
    if (K0 <= 0 || J0 <= 0 || J1 <= 0 || J2 <= 0) return 0;
    return packed_layout(K0, J0, J1, J2).total;
}

static PackLayer pack_layer(const float* W, const float* b, float* packed, size_t oP, size_t oB, int K, int J, int S, int T, int sf, int sk) {
    size_t n = (size_t)T * S * 64;
    if (n < (size_t)T * 32) n = (size_t)T * 32;
    return PackLayer{W, b, packed, oP, oB, n, K, J, S, T, sf, sk};
}

static void pack_launch(const PackLayer* layers, int nl, hipStream_t st) {
    PackArgs a;
    size_t n = 0;
    for (int i = 0; i < PACK_MAX_LAYERS; ++i) {
        a.l[i] = layers[i < nl ? i : nl - 1];
        if (i < nl) n += layers[i].n;
    }
    a.nl = nl;
    hipLaunchKernelGGL(mlp3_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

static int check_hidden(const char* who, int a, int b) {
    if (a <= 0 || b <= 0 || (a & 31) || (b & 31) || a > 256 || b > 256)
        return fail(LIPVQ_EUNSUPPORTED, "%s: hidden widths must be multiples of 32 in [32,256] (got %d,%d)", who, a, b);
    return LIPVQ_OK;
}

extern "C" int lipvq_mlp3_pack_f32(const float* W0, const float* b0, const float* W1,
                                   const float* b1, const float* W2, const float* b2, float* packed,
                                   int K0, int J0, int J1, int J2, void* stream) {
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !packed) return fail(LIPVQ_EINVAL, "mlp3_pack: null pointer");
    if (K0 <= 0 || J2 <= 0) return fail(LIPVQ_EINVAL, "mlp3_pack: bad sizes");
    if (int e = check_hidden("mlp3_pack", J0, J1)) return e;
    PackedLayout L = packed_layout(K0, J0, J1, J2);
    hipStream_t st = (hipStream_t)stream;
    const PackLayer ls[3] = {pack_layer(W0, b0, packed, L.oP0, L.oB0, K0, J0, L.S0, L.T0, K0, 1),
                             pack_layer(W1, b1, packed, L.oP1, L.oB1, J0, J1, L.S1, L.T1, J0, 1),
                             pack_layer(W2, b2, packed, L.oP2, L.oB2, J1, J2, L.S2, L.T2, J1, 1)};
    pack_launch(ls, 3, st);
    return check_launch("mlp3_pack");
}

// Backward-data chain J2 -> J1 -> J0 -> K0 with W2^T, W1^T, W0^T and zero biases.
extern "C" size_t lipvq_mlp3_packed_bwd_floats(int K0, int J0, int J1, int J2) {
    if (K0 <= 0 || J0 <= 0 || J1 <= 0 || J2 <= 0) return 0;
    return packed_layout(J2, J1, J0, K0).total;
}

// the three layers of one backward-data chain: layer 0' has in J2, out J1 and the virtual weight [J1][J2] = W2^T, i.e.
// Wv[f][k] = W2[k][f] = W2[k*J1 + f]; and so on down the stack
static void pack_bwd_layers(PackLayer* out, const float* W0, const float* W1, const float* W2, float* packed, int K0, int J0, int J1, int J2) {
    const PackedLayout L = packed_layout(J2, J1, J0, K0);
    out[0] = pack_layer(W2, nullptr, packed, L.oP0, L.oB0, J2, J1, L.S0, L.T0, 1, J1);
    out[1] = pack_layer(W1, nullptr, packed, L.oP1, L.oB1, J1, J0, L.S1, L.T1, 1, J0);
    out[2] = pack_layer(W0, nullptr, packed, L.oP2, L.oB2, J0, K0, L.S2, L.T2, 1, K0);
}

extern "C" int lipvq_mlp3_pack_bwd_f32(const float* W0, const float* W1, const float* W2, float* packed,
                                       int K0, int J0, int J1, int J2, void* stream) {
    if (!W0 || !W1 || !W2 || !packed) return fail(LIPVQ_EINVAL, "mlp3_pack_bwd: null pointer");
    if (K0 <= 0 || J2 <= 0) return fail(LIPVQ_EINVAL, "mlp3_pack_bwd: bad sizes");
    if (int e = check_hidden("mlp3_pack_bwd", J0, J1)) return e;
    PackLayer ls[3];
    pack_bwd_layers(ls, W0, W1, W2, packed, K0, J0, J1, J2);
    pack_launch(ls, 3, (hipStream_t)stream);
    return check_launch("mlp3_pack_bwd");
}

// Two stacks (the tokenizer's decoder and encoder) in ONE launch: a training step of 80 rows is ~30 graph nodes of >= 4.5 us.
extern "C" int lipvq_mlp3_pack_bwd2_f32(const float* aW0, const float* aW1, const float* aW2, float* a_packed, int aK0, int aJ0,
                                        int aJ1, int aJ2, const float* bW0, const float* bW1, const float* bW2, float* b_packed,
                                        int bK0, int bJ0, int bJ1, int bJ2, void* stream) {
    if (!aW0 || !aW1 || !aW2 || !a_packed || !bW0 || !bW1 || !bW2 || !b_packed) return fail(LIPVQ_EINVAL, "mlp3_pack_bwd2: null pointer");
    if (aK0 <= 0 || aJ2 <= 0 || bK0 <= 0 || bJ2 <= 0) return fail(LIPVQ_EINVAL, "mlp3_pack_bwd2: bad sizes");
    if (int e = check_hidden("mlp3_pack_bwd2", aJ0, aJ1)) return e;
    if (int e = check_hidden("mlp3_pack_bwd2", bJ0, bJ1)) return e;
    PackLayer ls[6];
    pack_bwd_layers(ls, aW0, aW1, aW2, a_packed, aK0, aJ0, aJ1, aJ2);
    pack_bwd_layers(ls + 3, bW0, bW1, bW2, b_packed, bK0, bJ0, bJ1, bJ2);
    pack_launch(ls, 6, (hipStream_t)stream);
    return check_launch("mlp3_pack_bwd2");
}

// ------------------------------------------------------------------------------------------
// the fused three-layer kernel
// ------------------------------------------------------------------------------------------
struct Mlp3Args {
    const float* x;              // fwd: input rows (or gather table); bwd: gy
    const int64_t* gather_idx;
    const float* packed;
    float* y;                    // fwd: output; bwd: gx (NULL: skip the last layer)
    float* out0;                 // fwd: pre0 save; bwd: g1 = (gy' W2) * act1'(pre1)
    float* out1;                 // fwd: pre1 save; bwd: g0
    float* out2;                 // fwd: pre2 save; bwd: g2 = gy * act2'(pre2)
    const float* in_pre;         // bwd: pre2 (NULL when act2 is the identity)
    const float* mul0;           // bwd: pre1
    const float* mul1;           // bwd: pre0
    int64_t N;
    int K0, J0, J1, J2;          // widths of THIS chain (bwd: J2, J1, J0, K0 of the forward stack)
    int act0, act1, act2, act_in;
    // the VQ losses' gradient terms folded into the backward chain (mlp3_lds_kernel<.., FUSE> only; lipvq_mlp3_bwd_vq_f32):
    //   din_b  != NULL:  gy := (in_alpha  * *gscale) * (act_in(pre2) - B)   instead of reading gy (act_in(pre2) = the forward's output)
    //   dout_a != NULL:  gx := (out_alpha * *gscale) * (A - B) + gx         (product and sum rounded separately, as scaled_diff_kernel)
    // A / B = rows of dout_a / din_b / dout_b, picked by their index vector when one is given (codebook rows)
    const float* din_b; const int64_t* din_ib;
    const float* dout_a; const int64_t* dout_ia; const float* dout_b; const int64_t* dout_ib;
    const float* gscale;
    float in_alpha, out_alpha;
    // the forward with the tokenizer's two mean-squared errors folded in (mlp3_lds_kernel<.., 3>; lipvq_mlp3_loss_f32): per-wave
    // double sums of (y - loss_x)^2 and (layer-0 input rows - loss_z)^2 into loss_part[0 / MSE slots + 8 blockIdx.x + wave]
    const float* loss_x; const float* loss_z; double* loss_part;
    float* ste_out;              // (FUSE 3) non-NULL: the stack's input rows become loss_z + (rows - loss_z), stored here too: the plain
                                 // VQVAE's straight-through value (backbone.py:74), formed where its two operands already are
};

template <int T0, int T1, bool BWD>
__global__ __launch_bounds__(256) void mlp3_kernel(Mlp3Args a) {
    const PackedLayout L = packed_layout(a.K0, a.J0, a.J1, a.J2);
    const float* __restrict__ P0 = a.packed + L.oP0;
    const float* __restrict__ B0 = a.packed + L.oB0;
    const float* __restrict__ P1 = a.packed + L.oP1;
    const float* __restrict__ B1 = a.packed + L.oB1;
    const float* __restrict__ P2 = a.packed + L.oP2;
    const float* __restrict__ B2 = a.packed + L.oB2;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int64_t ntiles = (a.N + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t row = tile * 32 + (lane & 31);
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
            float bv = (k < a.K0) ? xr[k] : 0.0f;
            if (BWD) {
                if (a.in_pre && k < a.K0) bv = bv * lq_act_grad(a.in_pre[(size_t)rowc * a.K0 + k], a.act_in);
                if (a.out2 && valid && k < a.K0) a.out2[(size_t)row * a.K0 + k] = bv;
            }
#pragma unroll
            for (int t = 0; t < T0; ++t) {
                const float av = P0[((size_t)t * L.S0 + s) * 64 + lane];
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0[t], 0, 0, 0);
            }
        }
        if (!BWD) {
            if (a.out0 && valid) {
#pragma unroll
                for (int t = 0; t < T0; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a.out0[(size_t)row * a.J0 + 32 * t + 2 * r + h] = acc0[t][r];
            }
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[t][r] = lq_act_apply(acc0[t][r], a.act0);
        } else {
#pragma unroll
            for (int t = 0; t < T0; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const size_t o = (size_t)rowc * a.J0 + 32 * t + 2 * r + h;
                    acc0[t][r] = acc0[t][r] * lq_act_grad(a.mul0[o], a.act0);
                    if (valid) a.out0[o] = acc0[t][r];
                }
        }

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
        if (!BWD) {
            if (a.out1 && valid) {
#pragma unroll
                for (int t = 0; t < T1; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a.out1[(size_t)row * a.J1 + 32 * t + 2 * r + h] = acc1[t][r];
            }
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[t][r] = lq_act_apply(acc1[t][r], a.act1);
        } else {
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const size_t o = (size_t)rowc * a.J1 + 32 * t + 2 * r + h;
                    acc1[t][r] = acc1[t][r] * lq_act_grad(a.mul1[o], a.act1);
                    if (valid) a.out1[o] = acc1[t][r];
                }
            if (!a.y) continue;      // wave-uniform: the caller does not need d/d(input)
        }

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
                        if (!BWD && a.out2) a.out2[(size_t)row * a.J2 + f] = acc2[r];
                        a.y[(size_t)row * a.J2 + f] = BWD ? acc2[r] : lq_act_apply(acc2[r], a.act2);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The kernel the entry points launch.  With one wave per 32-row tile (mlp3_kernel above) the stack is a single dependent
// chain of ~600 MFMAs per wave whose A operands arrive from L2 (80-100 us even at N = 80, 0.78 ms at N = 524 288); here a
// workgroup of 8 waves owns the tile: every layer's 32-feature output tiles are dealt out to the waves, activations
// cross layers through LDS ([feature][row], stride 33: conflict-free as MFMA B operand) and the packed weights are
// prefetched 8 k-steps ahead.  Same packed layout, same k-ordered chains: bit-identical to mlp3_kernel, 5x faster at
// training-step sizes (19 us at N = 80...8192) and 2-2.7x at N = 524 288 (395 us forward, 782 us backward).
// mlp3_kernel remains the fallback when the input tile does not fit LDS (K0 > ~900).
// ------------------------------------------------------------------------------------------
#define MLPS_WAVES 8
#define MLPS_LD 33

#ifndef MLPS_PF
#define MLPS_PF 8           // k-steps of packed weights in flight ahead of the MFMAs (16 and 32 measure the same, round 3: at
                            // N = 80 ... 2 048 the launch is its dependent MFMA chains + barriers, 17.5 us forward whatever the depth)
#endif

__device__ __forceinline__ void mlps_chain(const float* __restrict__ Pt, int S, const float* __restrict__ bp, f32x16& acc) {
    float cur[MLPS_PF], nxt[MLPS_PF];
#pragma unroll
    for (int j = 0; j < MLPS_PF; ++j) cur[j] = (j < S) ? Pt[(size_t)j * 64] : 0.0f;
    for (int s0 = 0; s0 < S; s0 += MLPS_PF) {
        if (s0 + MLPS_PF < S) {
#pragma unroll
            for (int j = 0; j < MLPS_PF; ++j) nxt[j] = (s0 + MLPS_PF + j < S) ? Pt[(size_t)(s0 + MLPS_PF + j) * 64] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < MLPS_PF; ++j) {
            if (s0 + j < S)          // wave-uniform
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j], bp[(s0 + j) * 2 * MLPS_LD], acc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < MLPS_PF; ++j) cur[j] = nxt[j];
    }
}

// NSUB 32-row sub-tiles per workgroup: a "unit" of a layer is (output tile t, sub-tile), units are dealt to the 8 waves.  With one
// sub-tile the reference's widths keep 2 (layer 0), 4 (layer 1) and 2 (layer 2) of the 8 waves busy and pay three workgroup
// barriers per 32 rows; with two, 4 / 8 / 4 waves work and the barriers are shared by 64 rows.
template <bool BWD, int NSUB>
__global__ __launch_bounds__(64 * MLPS_WAVES) void mlp3_wg_kernel(Mlp3Args a) {
    extern __shared__ float mlps_lds[];
    const PackedLayout L = packed_layout(a.K0, a.J0, a.J1, a.J2);
    const int K0p = 2 * L.S0;
    const size_t plane = ((size_t)K0p + a.J0 + a.J1) * MLPS_LD;      // floats per sub-tile: xT [K0p][33], h0T [J0][33], h1T [J1][33]
    const float* __restrict__ P0 = a.packed + L.oP0;
    const float* __restrict__ B0 = a.packed + L.oB0;
    const float* __restrict__ P1 = a.packed + L.oP1;
    const float* __restrict__ B1 = a.packed + L.oB1;
    const float* __restrict__ P2 = a.packed + L.oP2;
    const float* __restrict__ B2 = a.packed + L.oB2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int64_t wg_row0 = (int64_t)blockIdx.x * (32 * NSUB);
    // 16-byte accesses to the saved pre-activations / gradients where base and row stride allow (lq_tile_store16)
    auto al16 = [](const void* p, int ld) { return p && ((((uintptr_t)p) & 15) == 0) && (ld & 3) == 0; };
    const bool vec0 = al16(a.out0, a.J0), vec1 = al16(a.out1, a.J1), vec2o = al16(a.out2, a.J2), vec2y = al16(a.y, a.J2);
    const bool vec0m = BWD && al16(a.mul0, a.J0), vec1m = BWD && al16(a.mul1, a.J1);

    // ---- stage the input rows, transposed; backward: fold act2'(pre2) in and save g2 ----
    for (int f = tid; f < 32 * NSUB * K0p; f += 64 * MLPS_WAVES) {
        const int r = f / K0p, k = f - r * K0p;                  // r: row of the workgroup's 32 NSUB
        const int64_t rw = wg_row0 + r;
        const bool ok = rw < a.N;
        const int64_t rc = ok ? rw : a.N - 1;
        float v = 0.0f;
        if (k < a.K0) {
            const float* src = a.gather_idx ? a.x + (size_t)a.gather_idx[rc] * a.K0 : a.x + (size_t)rc * a.K0;
            v = src[k];
            if (BWD) {
                if (a.in_pre) v = v * lq_act_grad(a.in_pre[(size_t)rc * a.K0 + k], a.act_in);
                if (a.out2 && ok) a.out2[(size_t)rw * a.K0 + k] = v;
            }
        }
        mlps_lds[(size_t)(r >> 5) * plane + (size_t)k * MLPS_LD + (r & 31)] = v;
    }
    __syncthreads();

    // ---- layer 0 ----
    for (int u = wave; u < L.T0 * NSUB; u += MLPS_WAVES) {
        const int t = u / NSUB, sub = u - t * NSUB;
        float* xT = mlps_lds + (size_t)sub * plane;
        float* h0T = xT + (size_t)K0p * MLPS_LD;
        const int64_t row = wg_row0 + 32 * sub + n;
        const bool valid = row < a.N;
        const int64_t rowc = valid ? row : a.N - 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = B0[32 * t + 2 * r + h];
        mlps_chain(P0 + (size_t)t * L.S0 * 64 + lane, L.S0, xT + h * MLPS_LD + n, acc);
        if (!BWD) {
            if (a.out0) lq_tile_store16(a.out0, a.J0, row, valid, t, h, acc, a.J0, vec0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = lq_act_apply(acc[r], a.act0);
        } else {
            const f32x16 m = lq_tile_load16(a.mul0, a.J0, rowc, t, h, a.J0, vec0m);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] * lq_act_grad(m[r], a.act0);
            lq_tile_store16(a.out0, a.J0, row, valid, t, h, acc, a.J0, vec0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h0T[(32 * t + 2 * r + h) * MLPS_LD + n] = acc[r];
    }
    __syncthreads();

    // ---- layer 1 ----
    for (int u = wave; u < L.T1 * NSUB; u += MLPS_WAVES) {
        const int t = u / NSUB, sub = u - t * NSUB;
        float* h0T = mlps_lds + (size_t)sub * plane + (size_t)K0p * MLPS_LD;
        float* h1T = h0T + (size_t)a.J0 * MLPS_LD;
        const int64_t row = wg_row0 + 32 * sub + n;
        const bool valid = row < a.N;
        const int64_t rowc = valid ? row : a.N - 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = B1[32 * t + 2 * r + h];
        mlps_chain(P1 + (size_t)t * L.S1 * 64 + lane, L.S1, h0T + h * MLPS_LD + n, acc);
        if (!BWD) {
            if (a.out1) lq_tile_store16(a.out1, a.J1, row, valid, t, h, acc, a.J1, vec1);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = lq_act_apply(acc[r], a.act1);
        } else {
            const f32x16 m = lq_tile_load16(a.mul1, a.J1, rowc, t, h, a.J1, vec1m);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] * lq_act_grad(m[r], a.act1);
            lq_tile_store16(a.out1, a.J1, row, valid, t, h, acc, a.J1, vec1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h1T[(32 * t + 2 * r + h) * MLPS_LD + n] = acc[r];
    }
    if (BWD && !a.y) return;         // the caller does not need d/d(input)
    __syncthreads();

    // ---- layer 2 ----
    for (int u = wave; u < L.T2 * NSUB; u += MLPS_WAVES) {
        const int t = u / NSUB, sub = u - t * NSUB;
        float* h1T = mlps_lds + (size_t)sub * plane + ((size_t)K0p + a.J0) * MLPS_LD;
        const int64_t row = wg_row0 + 32 * sub + n;
        const bool valid = row < a.N;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = B2[32 * t + 2 * r + h];
        mlps_chain(P2 + (size_t)t * L.S2 * 64 + lane, L.S2, h1T + h * MLPS_LD + n, acc);
        if (!BWD && a.out2) lq_tile_store16(a.out2, a.J2, row, valid, t, h, acc, a.J2, vec2o);
        if (!BWD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = lq_act_apply(acc[r], a.act2);
        }
        lq_tile_store16(a.y, a.J2, row, valid, t, h, acc, a.J2, vec2y);
    }
}

// ------------------------------------------------------------------------------------------
// Large batches: persistent workgroups with the packed weights RESIDENT IN LDS.  mlp3_wg_kernel pays, for every 64 rows, the
// L2 latency of its weight stream (8 k-steps of cover), three workgroup barriers with 2-4 of 8 waves idle in the narrow
// layers, and the HBM latency of the saved pre-activations after each chain; measured at N = 524 288 it ran 3-4x above both
// its matrix-pipe and its HBM time (backward of the encoder stack 562 us; 131 us of MFMA, ~150 us of HBM traffic).  Here one
// workgroup per CU copies the whole packed stack into LDS once (67 KB for the reference's widths), then every WAVE walks its
// own 32-row tiles through the three layers exactly like mlp3_kernel -- activations stay in registers (the MFMA result tile IS
// the next layer's B operand), no barrier after the prologue -- with the A operand one conflict-free ds_read_b32 per MFMA,
// the saved pre-activations of a layer requested BEFORE that layer's MFMA chain and consumed after it, and all row-major
// traffic in 64-byte pieces per lane (lq_tile_load16 / lq_tile_store16).  Same packed layout, same k-ordered chains from the
// bias: the same bits as the other two kernels.
// Measured at N = 524 288 (scripts/dev/measure_mlp3_bwd.py, measure_mlp3_fwd.py): encoder backward 562 -> 440 us, decoder
// backward 328 -> 300 us, decoder forward 385 -> 204 us (245 saving the pre-activations).  Ablations of the encoder backward
// (builds with one part removed): no stores 243, no multiplier loads 218 (identity activations), no MFMA 231 (memory alone:
// 940 MB at 4.1 TB/s, plain torch streams reach 5.4-5.9 here), MFMA alone 110 (= the pipe's peak), act' 160.  The three parts
// add up instead of overlapping: the fp32 MFMA and the wave's own VALU share issue (lipvq_fused.hip has the probe), and two
// waves per SIMD (245 VGPRs) is all the occupancy there is.  Tried and not kept, all correct: non-temporal loads/stores
// (+6-60 %), whole-line traffic through wave-private LDS patches (+20 %), next-tile input prefetch (+6 %), all multiplier loads
// at the tile top (+5 %), 12 waves per workgroup (+6 % forward, spills backward).
// ------------------------------------------------------------------------------------------
#define MLPL_WAVES 8
#define MLPL_LOSS_SLOTS 2048          // = MSE_BLOCKS of lipvq_misc.hip (mse_final_kernel sums that many partials per pair)

__device__ __forceinline__ void mlpl_act16(f32x16& v, int act) {
    if (act == LIPVQ_ACT_GELU) {
        f32x16 o;
        float m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[r] = lq_gelu_poly(v[r]); m = fmaxf(m, v[r] * v[r]); }
        if (!(m < 18.0f)) {                                    // rare: some |x| >= sqrt(18) in this lane (a NaN went through the polynomial as a NaN)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = lq_gelu(v[r]);
        }
        v = o;
    } else if (act != LIPVQ_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = lq_act_apply(v[r], act);
    }
}

// g[r] *= act'(m[r]).  Straight-line over the 16 elements (one rare, out-of-line fix-up per tile instead of a branch and an
// inlined erf/exp tail per element: with those the backward of the encoder stack took 629 us, 350 of them in here).
__device__ __noinline__ float mlpl_gelu_grad_slow(float x) { return lq_gelu_grad(x); }

__device__ __forceinline__ void mlpl_actgrad16(f32x16& g, f32x16 m, int act) {
    if (act == LIPVQ_ACT_GELU) {
        f32x16 d;
        float mx = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[r] = lq_gelu_grad_poly(m[r]); mx = fmaxf(mx, m[r] * m[r]); }
        if (!(mx < 18.0f)) {
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                float v = d[0];                                      // rotate through element 0: no dynamic register indexing
                if (!(m[0] * m[0] < 18.0f)) v = mlpl_gelu_grad_slow(m[0]);
                f32x16 dn, mn;
#pragma unroll
                for (int q = 0; q < 15; ++q) { dn[q] = d[q + 1]; mn[q] = m[q + 1]; }
                dn[15] = v; mn[15] = m[0];
                d = dn;
                m = mn;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = g[r] * d[r];
    } else if (act == LIPVQ_ACT_SIGMOID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float sg = lq_sigmoid(m[r]); g[r] = g[r] * (sg * (1.0f - sg)); }
    } else if (act == LIPVQ_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = g[r] * (m[r] > 0.0f ? 1.0f : 0.0f);
    }
}

// one 32 x 32 tile of a row-major operand in the MFMA B-operand layout (rows past N clamp to N - 1 on loads, are skipped on stores)
__device__ __forceinline__ f32x16 mlpl_load(const float* __restrict__ base, const int64_t* __restrict__ gather, int ld, int64_t row0,
                                            int64_t N, int t, int lane, int J, bool vec) {
    int64_t rr = row0 + (lane & 31);
    rr = rr < N ? rr : N - 1;
    if (gather) rr = gather[rr];
    return lq_tile_load16(base, ld, rr, t, lane >> 5, J, vec);
}
__device__ __forceinline__ void mlpl_store(float* __restrict__ base, int ld, int64_t row0, int64_t N, int t, int lane, const f32x16& v,
                                           int J, bool vec) {
    const int64_t rr = row0 + (lane & 31);
    lq_tile_store16(base, ld, rr, rr < N, t, lane >> 5, v, J, vec);
}

// FUSE: 0 = plain chain; backward: 1 = gy computed from a difference term (Mlp3Args::din_*), 2 = a difference term added to gx
// (dout_*); forward: 3 = squared-error sums of the output and of the layer-0 input against two more operands (loss_*).
// Separate instances: the plain encoder backward sits at 245 VGPRs, every live tile more is scratch traffic.
template <int T0, int T1, bool BWD, int FUSE = 0>
__global__ __launch_bounds__(64 * MLPL_WAVES) void mlp3_lds_kernel(Mlp3Args a) {
    extern __shared__ __attribute__((aligned(16))) float mlpl_w[];
    const PackedLayout L = packed_layout(a.K0, a.J0, a.J1, a.J2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    {   // the packed stack -> LDS (total is a multiple of 32 floats; the host checked the 16-byte alignment)
        // four loads in flight per thread and round (round 4: `dst[i] = src[i]` compiles to one load, one s_waitcnt vmcnt(0), one
        // ds_write per iteration -- eight dependent L2 round trips per thread for the cfg2 decoder's 64 KB, in front of every launch)
        const float4* __restrict__ src = reinterpret_cast<const float4*>(a.packed);
        float4* dst = reinterpret_cast<float4*>(mlpl_w);
        const int nv = (int)(L.total / 4);
        constexpr int NT = 64 * MLPL_WAVES;
        for (int i0 = tid; i0 < nv; i0 += 4 * NT) {
            float4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + q * NT; r[q] = src[i < nv ? i : nv - 1]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + q * NT; if (i < nv) dst[i] = r[q]; }
        }
    }
    __syncthreads();
    const float* P0 = mlpl_w + L.oP0 + lane;
    const float* B0 = mlpl_w + L.oB0 + h;
    const float* P1 = mlpl_w + L.oP1 + lane;
    const float* B1 = mlpl_w + L.oB1 + h;
    const float* P2 = mlpl_w + L.oP2 + lane;
    const float* B2 = mlpl_w + L.oB2 + h;
    auto al16 = [](const void* p, int ld) { return p && ((((uintptr_t)p) & 15) == 0) && (ld & 3) == 0; };
    const bool vecx = al16(a.x, a.K0), vecp = BWD && al16(a.in_pre, a.K0);
    constexpr bool fin = BWD && FUSE == 1, fout = BWD && FUSE == 2, floss = !BWD && FUSE == 3;
    const bool veclz = floss && al16(a.loss_z, a.K0), veclx = floss && al16(a.loss_x, a.J2), vecst = floss && al16(a.ste_out, a.K0);
    double lsum_x = 0.0, lsum_z = 0.0;                                  // (floss) this lane's share, over all of the wave's tiles
    const bool vecib = fin && al16(a.din_b, a.K0);
    const bool vecoa = fout && al16(a.dout_a, a.J2), vecob = fout && al16(a.dout_b, a.J2);
    const float gs = FUSE && a.gscale ? a.gscale[0] : 1.0f;
    const float f_in = a.in_alpha * gs, f_out = a.out_alpha * gs;      // scaled_diff_kernel's factor: alpha * *gscale
    const bool vec0 = al16(a.out0, a.J0), vec1 = al16(a.out1, a.J1), vec2o = al16(a.out2, BWD ? a.K0 : a.J2), vec2y = al16(a.y, a.J2);
    const bool vec0m = BWD && al16(a.mul0, a.J0), vec1m = BWD && al16(a.mul1, a.J1);
    const int KT0 = (a.K0 + 31) / 32;
    const int64_t ntiles = (a.N + 31) / 32;
    const int64_t nwaves = (int64_t)gridDim.x * MLPL_WAVES;

    for (int64_t tile = (int64_t)blockIdx.x * MLPL_WAVES + wave; tile < ntiles; tile += nwaves) {
        const int64_t row0 = tile * 32;
        // FUSE 1 (the one instance that spills): the lane index the tile's global addresses are formed with is opaque per tile.  As
        // loop invariants hipcc kept `base + lane offset` pairs from the kernel's first lines and spilled them; each reload in front
        // of a store or load is a vector-memory load whose `s_waitcnt vmcnt(0)` also waits for every multiplier tile requested ahead.
        int lane_tile = lane;
#ifndef LQ_MLPL_LANE_PLAIN
        if constexpr (FUSE == 1) asm volatile("" : "+v"(lane_tile));
#endif
        const int lane = lane_tile;      // (shadows the kernel's: everything below addresses with this one)
        // one 32-feature slice of the input rows as a B operand; backward: act2'(pre2) folded in and g2 saved
        // (forward: requesting the NEXT tile's first slice under this tile's second and third layer changed nothing -- decoder
        // forward with the loss 284 -> 280 us, plain 204 -> 204)
        f32x16 xraw, praw, qraw;
        auto in_issue = [&](int kt) {
            if (fin) {
                qraw = mlpl_load(a.din_b, a.din_ib, a.K0, row0, a.N, kt, lane, a.K0, vecib);
            } else {
                xraw = mlpl_load(a.x, a.gather_idx, a.K0, row0, a.N, kt, lane, a.K0, vecx);
            }
            if (BWD && a.in_pre) praw = mlpl_load(a.in_pre, nullptr, a.K0, row0, a.N, kt, lane, a.K0, vecp);
            if (floss) qraw = mlpl_load(a.loss_z, nullptr, a.K0, row0, a.N, kt, lane, a.K0, veclz);
        };
        auto in_finish = [&](int kt) -> f32x16 {
            f32x16 v;
            if (fin) {
                if (a.act_in == LIPVQ_ACT_SIGMOID) {                // LipVQ's encoder: z_e = sigmoid(pre2), one evaluation for z_e and act'
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sg = lq_sigmoid(praw[r]);
                        v[r] = (f_in * (sg - qraw[r])) * (sg * (1.0f - sg));
                    }
                    if (a.out2) mlpl_store(a.out2, a.K0, row0, a.N, kt, lane, v, a.K0, vec2o);
                    return v;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = f_in * (lq_act_apply(praw[r], a.act_in) - qraw[r]);
            } else {
                v = xraw;
            }
            if (floss && row0 + (lane & 31) < a.N) {            // (features past K0 load as 0 from both operands)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const double d = (double)xraw[r] - (double)qraw[r]; lsum_z += d * d; }
            }
            if (floss && a.ste_out) {                            // z_st = z_e + (z_q - z_e), ste_kernel's two roundings
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = qraw[r] + (xraw[r] - qraw[r]);
                mlpl_store(a.ste_out, a.K0, row0, a.N, kt, lane, v, a.K0, vecst);
            }
            if (BWD) {
                if (a.in_pre) mlpl_actgrad16(v, praw, a.act_in);
                if (a.out2) mlpl_store(a.out2, a.K0, row0, a.N, kt, lane, v, a.K0, vec2o);
            }
            return v;
        };

        // ---- layer 0: K0 -> 32 T0 ----
        in_issue(0);
        f32x16 m0[T0];                                           // requested before the chain, consumed after it
        if (BWD) {
#pragma unroll
            for (int t = 0; t < T0; ++t) m0[t] = mlpl_load(a.mul0, nullptr, a.J0, row0, a.N, t, lane, a.J0, vec0m);
        }
        f32x16 xb = in_finish(0);
        f32x16 acc0[T0];
#pragma unroll
        for (int t = 0; t < T0; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[t][r] = B0[32 * t + 2 * r];
        for (int kt = 0; kt < KT0; ++kt) {
            if (kt + 1 < KT0) in_issue(kt + 1);                  // (wave-uniform) the next slice is in flight under this one's MFMAs
            const int sl = L.S0 - 16 * kt;                       // k-steps left
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < sl) {
#pragma unroll
                    for (int t = 0; t < T0; ++t)
                        acc0[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(P0[((size_t)t * L.S0 + 16 * kt + r) * 64], xb[r], acc0[t], 0, 0, 0);
                }
            }
            if (kt + 1 < KT0) xb = in_finish(kt + 1);
        }
        f32x16 m1[T1];
        if (BWD) {
#pragma unroll
            for (int t = 0; t < T0; ++t) mlpl_actgrad16(acc0[t], m0[t], a.act0);
            // the next layer's multipliers: after m0 is dead (registers), before this layer's stores (vmcnt counts both in order)
#pragma unroll
            for (int t = 0; t < T1; ++t) m1[t] = mlpl_load(a.mul1, nullptr, a.J1, row0, a.N, t, lane, a.J1, vec1m);
#pragma unroll
            for (int t = 0; t < T0; ++t) mlpl_store(a.out0, a.J0, row0, a.N, t, lane, acc0[t], a.J0, vec0);
        } else {
#pragma unroll
            for (int t = 0; t < T0; ++t) {
                if (a.out0) mlpl_store(a.out0, a.J0, row0, a.N, t, lane, acc0[t], a.J0, vec0);
                mlpl_act16(acc0[t], a.act0);
            }
        }

        // ---- layer 1: 32 T0 -> 32 T1 ----
        f32x16 acc1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = B1[32 * t + 2 * r];
#pragma unroll
        for (int s = 0; s < 16 * T0; ++s) {
#pragma unroll
            for (int t = 0; t < T1; ++t)
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(P1[((size_t)t * (16 * T0) + s) * 64], acc0[s / 16][s % 16], acc1[t], 0, 0, 0);
        }
        if (BWD) {
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                mlpl_actgrad16(acc1[t], m1[t], a.act1);
                mlpl_store(a.out1, a.J1, row0, a.N, t, lane, acc1[t], a.J1, vec1);
            }
        } else {
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                if (a.out1) mlpl_store(a.out1, a.J1, row0, a.N, t, lane, acc1[t], a.J1, vec1);
                mlpl_act16(acc1[t], a.act1);
            }
        }

        // ---- layer 2: 32 T1 -> J2, one 32-feature output tile at a time (backward: only if the caller wants d/d input) ----
        if (!BWD || a.y) {
            for (int t2 = 0; t2 < L.T2; ++t2) {
                f32x16 acc2, ea, eb;
                if (fout) {
                    // requested before the chain, consumed after it.  (The 32 MFMAs of one output tile do not cover the round trip:
                    // the cfg2 decoder backward runs 359 us with the term, 303 without.  Requesting a layer ahead costs registers
                    // this instance does not have -- 8 to 23 spilled, 389 us.  LipVQ's codebook term therefore rides in the
                    // scatter instead, lipvq_scatter_add_sorted_vq_f32; the plain VQVAE's z_e term uses this.)
                    ea = mlpl_load(a.dout_a, a.dout_ia, a.J2, row0, a.N, t2, lane, a.J2, vecoa);
                    eb = mlpl_load(a.dout_b, a.dout_ib, a.J2, row0, a.N, t2, lane, a.J2, vecob);
                }
                f32x16 lx;
                if (floss) lx = mlpl_load(a.loss_x, nullptr, a.J2, row0, a.N, t2, lane, a.J2, veclx);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = B2[32 * t2 + 2 * r];
                const float* P2t = P2 + (size_t)t2 * (16 * T1) * 64;
#pragma unroll
                for (int s = 0; s < 16 * T1; ++s)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(P2t[(size_t)s * 64], acc1[s / 16][s % 16], acc2, 0, 0, 0);
                if (!BWD) {
                    if (a.out2) mlpl_store(a.out2, a.J2, row0, a.N, t2, lane, acc2, a.J2, vec2o);
                    mlpl_act16(acc2, a.act2);
                }
                if (fout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[r] = f_out * (ea[r] - eb[r]) + acc2[r];
                }
                if (floss && row0 + (lane & 31) < a.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const double d = (double)acc2[r] - (double)lx[r];
                        if (32 * t2 + 2 * r + h < a.J2) lsum_x += d * d;          // (padded features of the last tile hold no output)
                    }
                }
                mlpl_store(a.y, a.J2, row0, a.N, t2, lane, acc2, a.J2, vec2y);
            }
        }
    }
    if (floss) {                                                 // every wave of every workgroup owns one slot of each sum
        for (int o = 32; o > 0; o >>= 1) { lsum_x += __shfl_down(lsum_x, o, 64); lsum_z += __shfl_down(lsum_z, o, 64); }
        if (lane == 0) {
            const int slot = blockIdx.x * MLPL_WAVES + wave;
            a.loss_part[slot] = lsum_x;
            a.loss_part[MLPL_LOSS_SLOTS + slot] = lsum_z;
        }
    }
}
typedef void (*mlp3_fn)(Mlp3Args);
template <bool BWD>
static mlp3_fn mlp3_select(int T0, int T1) {
#define LQ_CASE(a_, b_) if (T0 == a_ && T1 == b_) return mlp3_kernel<a_, b_, BWD>;
    LQ_CASE(2, 4) LQ_CASE(4, 2)            // the reference's stacks: 64->128 and 128->64
    LQ_CASE(1, 1) LQ_CASE(2, 1) LQ_CASE(1, 2) LQ_CASE(2, 2) LQ_CASE(2, 3) LQ_CASE(3, 2)   // hidden_dim 32, 64, 96
#undef LQ_CASE
    return nullptr;
}

// tiles up to which the 8-waves-per-tile kernel is used: all of them, unless LIPVQ_MLP3_SMALL_TILES says otherwise
// (measurement knob: 0 forces the one-wave-per-tile kernel)
static int64_t mlp3_small_tiles() {
    static int64_t v = -1;
    if (v < 0) {
        const char* e = lq_knob("LIPVQ_MLP3_SMALL_TILES");
        v = e ? atoll(e) : INT64_MAX;
    }
    return v;
}

template <bool BWD>
static int launch_mlp3_wg(const Mlp3Args& a, hipStream_t st, const char* what, bool* done) {
    *done = false;
    const size_t plane = ((size_t)2 * ((a.K0 + 1) / 2) + a.J0 + a.J1) * MLPS_LD * sizeof(float);
    // two 32-row sub-tiles per workgroup from 4 096 rows on (more waves busy per layer, barriers shared by 64 rows); training-step
    // batches keep one, so that N = 80 still spreads over three workgroups.  LIPVQ_MLP3_SUB=1|2: measurement knob.
    static int forced = -1;
    if (forced < 0) { const char* e = lq_knob("LIPVQ_MLP3_SUB"); forced = e ? atoi(e) : 0; }
    // (two only while two such workgroups still share a CU's LDS: the 208-wide decoder input is faster with one; four is slower
    // everywhere: 3.75 -> 4.25 ms for the cfg2 training step)
    int nsub = (forced == 1 || forced == 2 || forced == 4) ? forced : ((a.N >= 4096 && 2 * plane <= 80 * 1024) ? 2 : 1);
    while (nsub > 1 && plane * nsub > 150 * 1024) nsub >>= 1;
    const int64_t ntiles = (a.N + 32 * nsub - 1) / (32 * nsub);
    const size_t lds = plane * nsub;
    if ((a.N + 31) / 32 > mlp3_small_tiles() || lds > 150 * 1024) return LIPVQ_OK;
    static LqLdsReserve reserved[3];            // per instantiation: per-device, thread-safe (lipvq_common.h)
    auto kfn = nsub == 4 ? mlp3_wg_kernel<BWD, 4> : nsub == 2 ? mlp3_wg_kernel<BWD, 2> : mlp3_wg_kernel<BWD, 1>;
    if (int rc = lipvq_reserve_lds(reserved[nsub == 4 ? 2 : nsub - 1], (const void*)kfn, 150 * 1024, what)) return rc;
    if (ntiles > 0x7fffffffLL) return LIPVQ_OK;
    hipLaunchKernelGGL(kfn, dim3((unsigned)ntiles), dim3(64 * MLPS_WAVES), lds, st, a);
    *done = true;
    return check_launch(what);
}


// rows from which the LDS-resident kernel is used (LIPVQ_MLP3_LDS_ROWS: measurement knob; 0 = never)
static int64_t mlp3_lds_rows() {
    static int64_t v = -1;
    if (v < 0) {
        const char* e = lq_knob("LIPVQ_MLP3_LDS_ROWS");
        v = e ? atoll(e) : 65536;              // crossover with mlp3_wg_kernel (backward: 35 vs 56 us at 32 768, 67 vs 65 at 65 536, 145 vs 117 at 131 072)
        if (v == 0) v = INT64_MAX;
    }
    return v;
}

template <bool BWD, int FUSE>
static mlp3_fn mlp3_lds_select(int T0, int T1) {      // the reference's hidden widths (64, 128) only; others keep mlp3_wg_kernel
#define LQ_CASE(a_, b_) if (T0 == a_ && T1 == b_) return mlp3_lds_kernel<a_, b_, BWD, FUSE>;
    LQ_CASE(2, 4) LQ_CASE(4, 2)
#undef LQ_CASE
    return nullptr;
}

// does the LDS-resident kernel take this chain?  (also what lipvq_mlp3_bwd_vq_supported answers: only that kernel folds the VQ terms)
static bool mlp3_lds_takes(int64_t N, int K0, int J0, int J1, int J2) {
    if (N < mlp3_lds_rows()) return false;
    const PackedLayout L = packed_layout(K0, J0, J1, J2);
    return L.total * sizeof(float) <= 156 * 1024 && ((J0 == 64 && J1 == 128) || (J0 == 128 && J1 == 64));
}

template <bool BWD, int FUSE = 0>
static int launch_mlp3_lds(const Mlp3Args& a, hipStream_t st, const char* what, bool* done) {
    *done = false;
    if (!mlp3_lds_takes(a.N, a.K0, a.J0, a.J1, a.J2) || (((uintptr_t)a.packed) & 15)) return LIPVQ_OK;
    const PackedLayout L = packed_layout(a.K0, a.J0, a.J1, a.J2);
    const size_t lds = L.total * sizeof(float);
    mlp3_fn fn = mlp3_lds_select<BWD, FUSE>(a.J0 / 32, a.J1 / 32);
    if (!fn) return LIPVQ_OK;
    static LqLdsReserve reserved[2];             // per instantiation: per-device, thread-safe (lipvq_common.h)
    if (int rc = lipvq_reserve_lds(reserved[a.J0 == 64 ? 0 : 1], (const void*)fn, 156 * 1024, what)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int64_t ntiles = (a.N + 31) / 32;
    int64_t blocks = (ntiles + MLPL_WAVES - 1) / MLPL_WAVES;
    if (blocks > cus) blocks = cus;                  // one persistent workgroup per CU
    if (FUSE == 3) {                                 // one slot per wave in the partial sums; unused slots must read 0
        if (blocks * MLPL_WAVES > MLPL_LOSS_SLOTS) blocks = MLPL_LOSS_SLOTS / MLPL_WAVES;
        // (every wave of the grid writes its slot: with 256 workgroups x 8 waves none is left over and nothing needs zeroing)
        if (blocks * MLPL_WAVES < MLPL_LOSS_SLOTS &&
            hipMemsetAsync(a.loss_part, 0, 2 * MLPL_LOSS_SLOTS * sizeof(double), st) != hipSuccess) return fail(LIPVQ_EHIP, "%s: memset", what);
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64 * MLPL_WAVES), lds, st, a);
    *done = true;
    return check_launch(what);
}

static int launch_mlp3(mlp3_fn fn, const Mlp3Args& a, hipStream_t st, const char* what) {
    int64_t ntiles = (a.N + 31) / 32;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;        // grid-stride beyond 8 blocks per CU
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return check_launch(what);
}

extern "C" int lipvq_mlp3_f32(const float* x, const int64_t* gather_idx, const float* packed, float* y,
                              float* pre0, float* pre1, float* pre2, int64_t N, int K0, int J0, int J1,
                              int J2, int act0, int act1, int act2, void* stream) {
    if (N < 0) return fail(LIPVQ_EINVAL, "mlp3: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!x || !packed || !y) return fail(LIPVQ_EINVAL, "mlp3: null pointer");
    if (K0 <= 0 || J2 <= 0) return fail(LIPVQ_EINVAL, "mlp3: bad sizes");
    if (int e = check_hidden("mlp3", J0, J1)) return e;
    Mlp3Args a{x, gather_idx, packed, y, pre0, pre1, pre2, nullptr, nullptr, nullptr,
               N, K0, J0, J1, J2, act0, act1, act2, LIPVQ_ACT_NONE};
    bool done;
    if (int e = launch_mlp3_lds<false>(a, (hipStream_t)stream, "mlp3_lds", &done)) return e;
    if (done) return LIPVQ_OK;
    if (int e = launch_mlp3_wg<false>(a, (hipStream_t)stream, "mlp3_wg", &done)) return e;
    if (done) return LIPVQ_OK;
    mlp3_fn fn = mlp3_select<false>(J0 / 32, J1 / 32);       // fallback (input tile wider than LDS): instantiated widths only
    if (!fn) return fail(LIPVQ_EUNSUPPORTED, "mlp3: no kernel instance for hidden widths %d,%d with K0=%d", J0, J1, K0);
    return launch_mlp3(fn, a, (hipStream_t)stream, "mlp3");
}

extern "C" int lipvq_mlp3_bwd_f32(const float* gy, const float* pre0, const float* pre1, const float* pre2,
                                  const float* packed_bwd, float* g2, float* g1, float* g0, float* gx,
                                  int64_t N, int K0, int J0, int J1, int J2, int act0, int act1, int act2,
                                  void* stream) {
    if (N < 0) return fail(LIPVQ_EINVAL, "mlp3_bwd: N < 0");
    if (N == 0) return LIPVQ_OK;
    if (!gy || !pre0 || !pre1 || !packed_bwd || !g1 || !g0) return fail(LIPVQ_EINVAL, "mlp3_bwd: null pointer");
    if (act2 != LIPVQ_ACT_NONE && !pre2) return fail(LIPVQ_EINVAL, "mlp3_bwd: pre2 required when act2 is not the identity");
    if (K0 <= 0 || J2 <= 0) return fail(LIPVQ_EINVAL, "mlp3_bwd: bad sizes");
    if (int e = check_hidden("mlp3_bwd", J0, J1)) return e;
    Mlp3Args a{gy, nullptr, packed_bwd, gx, g1, g0, g2, (act2 != LIPVQ_ACT_NONE) ? pre2 : nullptr, pre1, pre0,
               N, J2, J1, J0, K0, act1, act0, LIPVQ_ACT_NONE, act2};
    bool done;
    if (int e = launch_mlp3_lds<true>(a, (hipStream_t)stream, "mlp3_lds_bwd", &done)) return e;
    if (done) return LIPVQ_OK;
    if (int e = launch_mlp3_wg<true>(a, (hipStream_t)stream, "mlp3_wg_bwd", &done)) return e;
    if (done) return LIPVQ_OK;
    mlp3_fn fn = mlp3_select<true>(J1 / 32, J0 / 32);
    if (!fn) return fail(LIPVQ_EUNSUPPORTED, "mlp3_bwd: no kernel instance for hidden widths %d,%d with J2=%d", J0, J1, J2);
    return launch_mlp3(fn, a, (hipStream_t)stream, "mlp3_bwd");
}

// The forward stack with the tokenizer's loss folded in (round 3): the decoder of a training / evaluation step at large batches reads
// its input rows (z_q = codebook[idx]) and holds its output (x_rec) in registers anyway -- the separate first pass of
// lipvq_mse_pair_loss_f32 streamed 300 MB for the same sums (102 us at the metric's batch).
extern "C" int lipvq_mlp3_loss_supported(int64_t N, int K0, int J0, int J1, int J2) {
    return K0 > 0 && J2 > 0 && mlp3_lds_takes(N, K0, J0, J1, J2) ? 1 : 0;
}

extern "C" int lipvq_mlp3_loss_f32(const float* x, const int64_t* gather_idx, const float* packed, float* y, float* pre0, float* pre1,
                                   float* pre2, int64_t N, int K0, int J0, int J1, int J2, int act0, int act1, int act2,
                                   const float* target, const float* latent, float* ste_out, float* out3, float w, int form,
                                   void* workspace, void* stream) {
    if (N <= 0) return fail(LIPVQ_EINVAL, "mlp3_loss: N <= 0");
    if (!x || !packed || !y || !target || !latent || !out3 || !workspace) return fail(LIPVQ_EINVAL, "mlp3_loss: null pointer");
    if (form != LIPVQ_LOSS_LLFQ && form != LIPVQ_LOSS_VQ) return fail(LIPVQ_EINVAL, "mlp3_loss: unknown loss form %d", form);
    if (!lipvq_mlp3_loss_supported(N, K0, J0, J1, J2) || (((uintptr_t)packed) & 15))
        return fail(LIPVQ_EUNSUPPORTED, "mlp3_loss: N=%lld widths %d,%d,%d,%d (lipvq_mlp3_loss_supported)", (long long)N, K0, J0, J1, J2);
    Mlp3Args a{x, gather_idx, packed, y, pre0, pre1, pre2, nullptr, nullptr, nullptr,
               N, K0, J0, J1, J2, act0, act1, act2, LIPVQ_ACT_NONE};
    a.loss_x = target; a.loss_z = latent; a.loss_part = (double*)workspace; a.ste_out = ste_out;
    bool done;
    if (int e = launch_mlp3_lds<false, 3>(a, (hipStream_t)stream, "mlp3_lds_loss", &done)) return e;
    if (!done) return fail(LIPVQ_EUNSUPPORTED, "mlp3_loss: no kernel instance");
    return lipvq_mse_finish(a.loss_part, N * (int64_t)J2, N * (int64_t)K0, out3, out3 + 2, w, form, stream);
}

// The backward chain with the VQ losses' gradient terms folded in (round 3: the three scaled_diff launches of a training step were
// 171 us of streaming at the metric's batch).  Large batches on the reference's hidden widths only (the LDS-resident kernel).
extern "C" int lipvq_mlp3_bwd_vq_supported(int64_t N, int K0, int J0, int J1, int J2) {
    return K0 > 0 && J2 > 0 && mlp3_lds_takes(N, J2, J1, J0, K0) ? 1 : 0;
}

extern "C" int lipvq_mlp3_bwd_vq_f32(const float* gy, const float* pre0, const float* pre1, const float* pre2,
                                     const float* packed_bwd, float* g2, float* g1, float* g0, float* gx, int64_t N,
                                     int K0, int J0, int J1, int J2, int act0, int act1, int act2,
                                     const float* in_b, const int64_t* in_b_idx, float in_alpha, const float* out_a, const int64_t* out_a_idx, const float* out_b,
                                     const int64_t* out_b_idx, float out_alpha, const float* gscale, void* stream) {
    if (N <= 0) return N < 0 ? fail(LIPVQ_EINVAL, "mlp3_bwd_vq: N < 0") : LIPVQ_OK;
    if ((!gy && !in_b) || !pre0 || !pre1 || !packed_bwd || !g1 || !g0) return fail(LIPVQ_EINVAL, "mlp3_bwd_vq: null pointer");
    if (out_a && (!out_b || !gx)) return fail(LIPVQ_EINVAL, "mlp3_bwd_vq: the output term needs both operands and gx");
    if (in_b && out_a) return fail(LIPVQ_EUNSUPPORTED, "mlp3_bwd_vq: one folded term per launch");
    if (!in_b && !out_a) return fail(LIPVQ_EINVAL, "mlp3_bwd_vq: no term given (use lipvq_mlp3_bwd_f32)");
    if (in_b && act2 == LIPVQ_ACT_NONE) return fail(LIPVQ_EINVAL, "mlp3_bwd_vq: the input term is act2(pre2) - B: act2 must not be the identity");
    if (act2 != LIPVQ_ACT_NONE && !pre2) return fail(LIPVQ_EINVAL, "mlp3_bwd_vq: pre2 required when act2 is not the identity");
    if (!lipvq_mlp3_bwd_vq_supported(N, K0, J0, J1, J2) || (((uintptr_t)packed_bwd) & 15))
        return fail(LIPVQ_EUNSUPPORTED, "mlp3_bwd_vq: N=%lld widths %d,%d,%d,%d (lipvq_mlp3_bwd_vq_supported)", (long long)N, K0, J0, J1, J2);
    Mlp3Args a{gy, nullptr, packed_bwd, gx, g1, g0, g2, (act2 != LIPVQ_ACT_NONE) ? pre2 : nullptr, pre1, pre0,
               N, J2, J1, J0, K0, act1, act0, LIPVQ_ACT_NONE, act2,
               in_b, in_b_idx, out_a, out_a_idx, out_b, out_b_idx, gscale, in_alpha, out_alpha};
    bool done;
    if (int e = in_b ? launch_mlp3_lds<true, 1>(a, (hipStream_t)stream, "mlp3_lds_bwd_vq_in", &done)
                     : launch_mlp3_lds<true, 2>(a, (hipStream_t)stream, "mlp3_lds_bwd_vq_out", &done)) return e;
    return done ? LIPVQ_OK : fail(LIPVQ_EUNSUPPORTED, "mlp3_bwd_vq: no kernel instance");
}
