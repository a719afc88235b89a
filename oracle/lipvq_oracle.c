/* lipvq_oracle.c -- CPU oracle for the LipVQ-VAE action-tokenizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lipvq-vae_amd/ (the product) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and only as the checker.
 *
 * It restates, in plain C, what the reference computes on this path
 * (/root/reference/robomimic/models/vq_vae/backbone_lfqvae_v5.py = "v5",
 *  /root/reference/robomimic/models/vq_vae/backbone.py = "vq"):
 *   lq_ref_lipschitz_scale   v5:6-11   normalization(): min(1, softplus(ci)/sum|W|)
 *   lq_ref_mlp3              v5:54-59,22-24 (encoder+to_latent), v5:62-68 (decoder+to_output),
 *                            vq:17-32 (ReLU stacks)
 *   lq_ref_nearest           v5:37-48 (LFQQuantizer.forward), vq:55-66 (VQVAE.quantize)
 *   lq_ref_losses            v5:79-83, vq:50-51,69-71
 *   lq_ref_*_bwd             what autograd derives from v5:70-84 / vq:38-76
 *
 * Arithmetic contract ("canonical fp32"), shared with the gfx950 kernels via
 * lipvq-vae_amd/csrc/lipvq_math.h:
 *   - a Linear layer is, per output, ONE fused-multiply-add chain in natural k
 *     order that starts from the bias:  acc = b[j]; acc = fmaf(x[k], W[j][k], acc).
 *     (That is bit-for-bit what a gfx950 fp32 MFMA accumulation computes.)
 *     An odd fan-in is padded with one zero term, as the MFMA's K=2 step does.
 *   - activations use lq_gelu / lq_sigmoid / relu from lipvq_math.h.
 *   - distances use lq_sqdist8 (torch.norm order) or lq_sqdist32 (pow(2).sum order);
 *     for the LipVQ variant the comparison is made on sqrtf(distance) exactly as
 *     torch.norm + argmin do (two distinct squares may share one square root; the
 *     lower index then wins).
 *   - reductions whose order torch does not fix observably (mse means, weight
 *     gradients) are accumulated in double and rounded once: the GPU result is
 *     compared with a tolerance there (1e-5 relative, stated in the tests).
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against golden
 * vectors produced by importing the reference module itself in the build
 * container (oracle/gen_golden.py -> tests/golden/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../lipvq-vae_amd/csrc/lipvq_math.h"

#define LQ_EXPORT __attribute__((visibility("default")))

enum { LQ_ACT_NONE = 0, LQ_ACT_GELU = 1, LQ_ACT_SIGMOID = 2, LQ_ACT_RELU = 3 };
enum { LQ_DIST_NORM = 0 /* v5: torch.norm, compare sqrt */, LQ_DIST_SQSUM = 1 /* vq: pow(2).sum */ };

static inline float act_apply(float v, int act) {
    switch (act) {
        case LQ_ACT_GELU: return lq_gelu(v);
        case LQ_ACT_SIGMOID: return lq_sigmoid(v);
        case LQ_ACT_RELU: return v > 0.0f ? v : 0.0f;   /* torch relu: max(x,0); NaN not on this path */
        default: return v;
    }
}

/* scale[i] = min(1, softplus(ci[i]) / sum_j |W[i][j]|);  Wn = W * scale  (v5:6-12).
 * The row sum is a left-to-right fp32 sum (the GPU kernel does the same). */
LQ_EXPORT void lq_ref_lipschitz_scale(const float* W, const float* ci, float* scale, float* Wn,
                                      int D, int H) {
    for (int i = 0; i < D; ++i) {
        float s = 0.0f;
        for (int j = 0; j < H; ++j) s = s + lq_abs(W[(size_t)i * H + j]);
        float sc = lq_softplus(ci[i]) / s;
        if (!(sc < 1.0f)) sc = 1.0f;            /* torch.minimum(1, .): NaN (0/0) is not on this path */
        if (scale) scale[i] = sc;
        if (Wn)
            for (int j = 0; j < H; ++j) Wn[(size_t)i * H + j] = W[(size_t)i * H + j] * sc;
    }
}

static inline float chain(const float* x, const float* w, float b, int K) {
    float acc = b;
    for (int k = 0; k < K; ++k) acc = lq_fma(x[k], w[k], acc);
    if (K & 1) acc = lq_fma(0.0f, 0.0f, acc);
    return acc;
}

/* y = act2(L2(act1(L1(act0(L0(x))))));  W_l is [J_l][K_l] row-major (nn.Linear.weight).
 * pre0/pre1/pre2 (nullable) receive the pre-activations (saved for backward). */
LQ_EXPORT void lq_ref_mlp3(const float* x, const float* W0, const float* b0, const float* W1,
                           const float* b1, const float* W2, const float* b2, float* y,
                           float* pre0, float* pre1, float* pre2, int64_t N, int K0, int J0,
                           int J1, int J2, int act0, int act1, int act2) {
#pragma omp parallel
    {
        float* h0 = (float*)malloc(sizeof(float) * (size_t)(J0 + J1));
        float* h1 = h0 + J0;
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const float* xr = x + (size_t)n * K0;
            for (int j = 0; j < J0; ++j) {
                float a = chain(xr, W0 + (size_t)j * K0, b0[j], K0);
                if (pre0) pre0[(size_t)n * J0 + j] = a;
                h0[j] = act_apply(a, act0);
            }
            for (int j = 0; j < J1; ++j) {
                float a = chain(h0, W1 + (size_t)j * J0, b1[j], J0);
                if (pre1) pre1[(size_t)n * J1 + j] = a;
                h1[j] = act_apply(a, act1);
            }
            for (int j = 0; j < J2; ++j) {
                float a = chain(h1, W2 + (size_t)j * J1, b2[j], J1);
                if (pre2) pre2[(size_t)n * J2 + j] = a;
                y[(size_t)n * J2 + j] = act_apply(a, act2);
            }
        }
        free(h0);
    }
}

/* idx[n] = argmin_k dist(z[n], C[k]) with torch.argmin's first-minimum rule; zq[n] = C[idx[n]];
 * usage[k] += #rows mapped to k (nullable).  dist: LQ_DIST_NORM compares sqrtf(lq_sqdist8)
 * (v5:43-46; the sign mask of v5:39-40 multiplies every difference by +-1 and cannot change a
 * square, so it does not appear); LQ_DIST_SQSUM compares lq_sqdist32 (vq:58-63).
 * best_d (nullable) receives the winning compared value. */
LQ_EXPORT void lq_ref_nearest(const float* z, const float* C, int64_t* idx, float* zq,
                              int64_t* usage, float* best_d, int64_t N, int K, int D, int dist) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* zr = z + (size_t)n * D;
        float best = INFINITY;
        int bk = 0;
        for (int k = 0; k < K; ++k) {
            const float* cr = C + (size_t)k * D;
            float s = (dist == LQ_DIST_NORM) ? lq_sqdist8(zr, cr, D) : lq_sqdist32(zr, cr, D);
            float v = (dist == LQ_DIST_NORM) ? lq_sqrt(s) : s;
            if (v < best) { best = v; bk = k; }
        }
        idx[n] = bk;
        if (best_d) best_d[n] = best;
        if (zq) memcpy(zq + (size_t)n * D, C + (size_t)bk * D, sizeof(float) * (size_t)D);
    }
    if (usage)
        for (int64_t n = 0; n < N; ++n) usage[idx[n]] += 1;
}

/* Full distance row (small cases only): out[n][k] = compared value. */
LQ_EXPORT void lq_ref_distances(const float* z, const float* C, float* out, int64_t N, int K,
                                int D, int dist) {
    for (int64_t n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float s = (dist == LQ_DIST_NORM) ? lq_sqdist8(z + (size_t)n * D, C + (size_t)k * D, D)
                                             : lq_sqdist32(z + (size_t)n * D, C + (size_t)k * D, D);
            out[(size_t)n * K + k] = (dist == LQ_DIST_NORM) ? lq_sqrt(s) : s;
        }
}

/* Straight-through value of vq:74:  z_e + (z_q - z_e), rounded as fp32 evaluates it. */
LQ_EXPORT void lq_ref_ste(const float* ze, const float* zq, float* out, int64_t n_elem) {
    for (int64_t i = 0; i < n_elem; ++i) out[i] = ze[i] + (zq[i] - ze[i]);
}

/* out[0] = mean((xr-x)^2) over N*A; out[1] = mean((zq-ze)^2) over N*D   (F.mse_loss, v5:79-81).
 * Accumulated in double (see header). */
LQ_EXPORT void lq_ref_mse_pair(const float* xr, const float* x, int64_t nx, const float* zq,
                               const float* ze, int64_t nz, float* out) {
    double a = 0.0, b = 0.0;
    for (int64_t i = 0; i < nx; ++i) { double d = (double)xr[i] - (double)x[i]; a += d * d; }
    for (int64_t i = 0; i < nz; ++i) { double d = (double)zq[i] - (double)ze[i]; b += d * d; }
    out[0] = (float)(a / (double)nx);
    out[1] = (float)(b / (double)nz);
}

/* ---- backward (double accumulation; compared with a tolerance) ------------------------- */

static inline double act_grad(float pre, int act) {
    switch (act) {
        case LQ_ACT_GELU: return (double)lq_gelu_grad(pre);
        case LQ_ACT_SIGMOID: { double s = (double)lq_sigmoid(pre); return s * (1.0 - s); }
        case LQ_ACT_RELU: return pre > 0.0f ? 1.0 : 0.0;
        default: return 1.0;
    }
}

/* Backward of lq_ref_mlp3.  gy = dL/dy [N][J2].  Outputs (all nullable): gW0..2, gb0..2 (same shapes
 * as the weights and biases), gx [N][K0].  pre0/pre1/pre2 are the saved pre-activations. */
LQ_EXPORT void lq_ref_mlp3_bwd(const float* x, const float* W0, const float* W1, const float* W2,
                               const float* pre0, const float* pre1, const float* pre2,
                               const float* gy, float* gW0, float* gb0, float* gW1, float* gb1,
                               float* gW2, float* gb2, float* gx, int64_t N, int K0, int J0,
                               int J1, int J2, int act0, int act1, int act2) {
    double* aW0 = (double*)calloc((size_t)J0 * K0 + J0, sizeof(double));
    double* ab0 = aW0 + (size_t)J0 * K0;
    double* aW1 = (double*)calloc((size_t)J1 * J0 + J1, sizeof(double));
    double* ab1 = aW1 + (size_t)J1 * J0;
    double* aW2 = (double*)calloc((size_t)J2 * J1 + J2, sizeof(double));
    double* ab2 = aW2 + (size_t)J2 * J1;
    double* g2 = (double*)malloc(sizeof(double) * (size_t)(J2 + J1 + J0 + J1 + J0));
    double* g1 = g2 + J2;
    double* g0 = g1 + J1;
    double* h1 = g0 + J0;
    double* h0 = h1 + J1;
    for (int64_t n = 0; n < N; ++n) {
        for (int j = 0; j < J0; ++j) h0[j] = (double)act_apply(pre0[(size_t)n * J0 + j], act0);
        for (int j = 0; j < J1; ++j) h1[j] = (double)act_apply(pre1[(size_t)n * J1 + j], act1);
        for (int j = 0; j < J2; ++j)
            g2[j] = (double)gy[(size_t)n * J2 + j] * act_grad(pre2[(size_t)n * J2 + j], act2);
        for (int k = 0; k < J1; ++k) g1[k] = 0.0;
        for (int j = 0; j < J2; ++j) {
            ab2[j] += g2[j];
            for (int k = 0; k < J1; ++k) {
                aW2[(size_t)j * J1 + k] += g2[j] * h1[k];
                g1[k] += g2[j] * (double)W2[(size_t)j * J1 + k];
            }
        }
        for (int k = 0; k < J1; ++k) g1[k] *= act_grad(pre1[(size_t)n * J1 + k], act1);
        for (int k = 0; k < J0; ++k) g0[k] = 0.0;
        for (int j = 0; j < J1; ++j) {
            ab1[j] += g1[j];
            for (int k = 0; k < J0; ++k) {
                aW1[(size_t)j * J0 + k] += g1[j] * h0[k];
                g0[k] += g1[j] * (double)W1[(size_t)j * J0 + k];
            }
        }
        for (int k = 0; k < J0; ++k) g0[k] *= act_grad(pre0[(size_t)n * J0 + k], act0);
        for (int j = 0; j < J0; ++j) {
            ab0[j] += g0[j];
            for (int k = 0; k < K0; ++k) aW0[(size_t)j * K0 + k] += g0[j] * (double)x[(size_t)n * K0 + k];
        }
        if (gx)
            for (int k = 0; k < K0; ++k) {
                double s = 0.0;
                for (int j = 0; j < J0; ++j) s += g0[j] * (double)W0[(size_t)j * K0 + k];
                gx[(size_t)n * K0 + k] = (float)s;
            }
    }
    if (gW0) for (size_t i = 0; i < (size_t)J0 * K0; ++i) gW0[i] = (float)aW0[i];
    if (gb0) for (int i = 0; i < J0; ++i) gb0[i] = (float)ab0[i];
    if (gW1) for (size_t i = 0; i < (size_t)J1 * J0; ++i) gW1[i] = (float)aW1[i];
    if (gb1) for (int i = 0; i < J1; ++i) gb1[i] = (float)ab1[i];
    if (gW2) for (size_t i = 0; i < (size_t)J2 * J1; ++i) gW2[i] = (float)aW2[i];
    if (gb2) for (int i = 0; i < J2; ++i) gb2[i] = (float)ab2[i];
    free(aW0); free(aW1); free(aW2); free(g2);
}

/* Backward of lq_ref_lipschitz_scale: given gWn = dL/d(W*scale), produce gW and gci (v5:6-12).
 *   Wn = W*sc;  sc = min(1, sp/s), sp = softplus(ci), s = sum|W|.
 *   where sc < 1:  dsc/dci = sigmoid(ci)/s;  dsc/dW_ij = -sp/s^2 * sign(W_ij)
 *   (torch.minimum routes the gradient to the smaller argument; at a tie it splits it in half --
 *    a measure-zero case that is not reproduced). */
LQ_EXPORT void lq_ref_lipschitz_bwd(const float* W, const float* ci, const float* gWn, float* gW,
                                    float* gci, int D, int H) {
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
        for (int j = 0; j < H; ++j) s += fabs((double)W[(size_t)i * H + j]);
        float s32 = 0.0f;
        for (int j = 0; j < H; ++j) s32 = s32 + lq_abs(W[(size_t)i * H + j]);
        double sp = (double)lq_softplus(ci[i]);
        float scf = lq_softplus(ci[i]) / s32;
        int active = scf < 1.0f;
        double sc = active ? sp / s : 1.0;
        double gsc = 0.0;
        for (int j = 0; j < H; ++j) gsc += (double)gWn[(size_t)i * H + j] * (double)W[(size_t)i * H + j];
        for (int j = 0; j < H; ++j) {
            double w = (double)W[(size_t)i * H + j];
            double g = (double)gWn[(size_t)i * H + j] * sc;
            if (active) g += gsc * (-sp / (s * s)) * ((w > 0) - (w < 0));
            gW[(size_t)i * H + j] = (float)g;
        }
        gci[i] = active ? (float)(gsc * (double)lq_sigmoid(ci[i]) / s) : 0.0f;
    }
}

/* gC[k] += sum over rows n with idx[n]==k of g[n]   (index_add_ of the gather's backward). */
LQ_EXPORT void lq_ref_scatter_add(const float* g, const int64_t* idx, float* gC, int64_t N, int K,
                                  int D) {
    double* acc = (double*)calloc((size_t)K * D, sizeof(double));
    for (int64_t n = 0; n < N; ++n)
        for (int d = 0; d < D; ++d) acc[(size_t)idx[n] * D + d] += (double)g[(size_t)n * D + d];
    for (size_t i = 0; i < (size_t)K * D; ++i) gC[i] = (float)acc[i];
    free(acc);
}

/* ---- the step after the tokenizer: input_embedding + interleave --------------------
 * ob = /root/reference/robomimic/models/obs_nets.py */

/* ob:2536  y = x . W^T + b, the canonical Linear (nn.Linear `embed_encoder`). */
LQ_EXPORT void lq_ref_linear(const float* x, const float* W, const float* b, float* y, int64_t N,
                             int Kin, int E) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int e = 0; e < E; ++e)
            y[(size_t)n * E + e] = chain(x + (size_t)n * Kin, W + (size_t)e * Kin, b ? b[e] : 0.0f, Kin);
}

/* Sum of 16 lane values the way the kernel's 16-lane DPP butterfly does it: partners l^1, l^2, l^7, l^15. */
static float row_allsum(float* lane) {
    static const int partner[4] = {1, 2, 7, 15};
    for (int st = 0; st < 4; ++st) {
        float t[16];
        for (int l = 0; l < 16; ++l) t[l] = lane[l] + lane[l ^ partner[st]];
        memcpy(lane, t, sizeof(t));
    }
    return lane[0];
}

/* ob:2537-2540 + ob:2584-2596  out[slot(n)] = LayerNorm(src[idx ? idx[n] : n] + pos[n % T]) * w + b, with
 * slot(n = b*T + t) = b*bstride + t*tstride + offset (floats).  stats (nullable) [N][2] = (mean, rstd).
 * An index outside [0, src_rows) gives a NaN row. */
LQ_EXPORT void lq_ref_embed_rows(const float* src, const int64_t* idx, const float* pos, const float* ln_w,
                                 const float* ln_b, float eps, float* out, float* stats, int64_t N, int T,
                                 int E, int64_t src_rows, int64_t bstride, int64_t tstride, int64_t offset) {
    const int E4 = E / 4;
    const float invE = 1.0f / (float)E;
#pragma omp parallel
    {
        float* v = (float*)malloc(sizeof(float) * (size_t)E);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const int64_t k = idx ? idx[n] : n;
            const int64_t b = n / T;
            const int t = (int)(n - b * T);
            float* o = out + (size_t)(b * bstride + t * tstride + offset);
            if (k < 0 || k >= src_rows) {
                for (int e = 0; e < E; ++e) o[e] = NAN;
                if (stats) stats[2 * n] = stats[2 * n + 1] = NAN;
                continue;
            }
            float lane[16];                                  /* lane l owns the float4 groups q = l, l+16, ... */
            for (int l = 0; l < 16; ++l) {
                float s = 0.0f;
                for (int q = l; q < E4; q += 16)
                    for (int c = 0; c < 4; ++c) {
                        const int e = 4 * q + c;
                        float val = src[(size_t)k * E + e];
                        if (pos) val = val + pos[(size_t)t * E + e];
                        v[e] = val;
                        s = s + val;
                    }
                lane[l] = s;
            }
            const float mean = row_allsum(lane) * invE;
            for (int l = 0; l < 16; ++l) {
                float ss = 0.0f;
                for (int q = l; q < E4; q += 16)
                    for (int c = 0; c < 4; ++c) {
                        const int e = 4 * q + c;
                        const float d = v[e] - mean;
                        v[e] = d;
                        ss = lq_fma(d, d, ss);
                    }
                lane[l] = ss;
            }
            const float var = row_allsum(lane) * invE;
            const float rstd = 1.0f / lq_sqrt(var + eps);
            for (int e = 0; e < E; ++e) o[e] = lq_fma(v[e] * rstd, ln_w[e], ln_b[e]);
            if (stats) {
                stats[2 * n] = mean;
                stats[2 * n + 1] = rstd;
            }
        }
        free(v);
    }
}

/* ---- the sibling tokenizer behind `bin_enabled`: AdaptiveBinActionEmbedding ------------
 * bin = /root/reference/robomimic/models/bin_action/backbone.py */

/* ob:2536 / bin:29-30  y = act(x . W^T + b); pre (nullable) = the pre-activation. */
LQ_EXPORT void lq_ref_linear_act(const float* x, const float* W, const float* b, float* y, float* pre, int64_t N,
                                 int Kin, int E, int act) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int e = 0; e < E; ++e) {
            const float a = chain(x + (size_t)n * Kin, W + (size_t)e * Kin, b ? b[e] : 0.0f, Kin);
            if (pre) pre[(size_t)n * E + e] = a;
            y[(size_t)n * E + e] = act_apply(a, act);
        }
}

/* bin:37-40 update_running_stats(), in place. */
LQ_EXPORT void lq_ref_bin_minmax(const float* actions, float* rmin, float* rmax, int64_t N, int A) {
    for (int64_t n = 0; n < N; ++n)
        for (int i = 0; i < A; ++i) {
            const float v = actions[(size_t)n * A + i];
            if (v < rmin[i]) rmin[i] = v;
            if (v > rmax[i]) rmax[i] = v;
        }
}

/* bin:42-66 compute_bins() + discretize(); bins [A][N] int64; boundaries (nullable) [A][nb+1]. */
LQ_EXPORT void lq_ref_bin_discretize(const float* actions, const float* rmin, const float* rmax, int64_t* bins,
                                     float* boundaries, int64_t N, int A, int nb) {
    float* bd = (float*)malloc(sizeof(float) * (size_t)A * (nb + 1));
    for (int i = 0; i < A; ++i)
        for (int j = 0; j <= nb; ++j) bd[i * (nb + 1) + j] = lq_linspace(rmin[i], rmax[i], nb + 1, j);
    if (boundaries) memcpy(boundaries, bd, sizeof(float) * (size_t)A * (nb + 1));
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int i = 0; i < A; ++i)
            bins[(size_t)i * N + n] = lq_bin_index(actions[(size_t)n * A + i], bd + i * (nb + 1), nb);
    free(bd);
}

/* bin:77-86 through the P table: pre1[n][j] = b1[j] + sum_i P[i][bins[i][n]][j] (i ascending), h = gelu(pre1). */
LQ_EXPORT void lq_ref_bin_hidden(const int64_t* bins, const float* P, const float* b1, float* h, float* pre1,
                                 int64_t N, int A, int nb, int H) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int j = 0; j < H; ++j) {
            float acc = b1[j];
            for (int i = 0; i < A; ++i) acc = acc + P[((size_t)i * nb + bins[(size_t)i * N + n]) * H + j];
            if (pre1) pre1[(size_t)n * H + j] = acc;
            h[(size_t)n * H + j] = lq_gelu(acc);
        }
}

/* Opt-in EMA codebook update (extension, not in the reference): see include/lipvq.h lipvq_ema_update_f32. */
LQ_EXPORT void lq_ref_ema_update(float* cs, float* es, const int64_t* counts, const float* dw, float* codebook,
                                 float decay, float eps, int K, int D) {
    const float omd = 1.0f - decay;
    double nd = 0.0;
    for (int k = 0; k < K; ++k) {
        cs[k] = lq_fma(decay, cs[k], omd * (float)counts[k]);
        nd += (double)cs[k];
    }
    const float n = (float)nd, denom = n + (float)K * eps;
    for (int k = 0; k < K; ++k) {
        const float sm = (cs[k] + eps) / denom * n;
        for (int d = 0; d < D; ++d) {
            const size_t e = (size_t)k * D + d;
            es[e] = lq_fma(decay, es[e], omd * dw[e]);
            codebook[e] = es[e] / sm;
        }
    }
}

/* Probes for tests/test_oracle_math.py */
LQ_EXPORT void lq_ref_math_probe(const float* x, float* out, int64_t n, int fn) {
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i], r;
        switch (fn) {
            case 0: r = lq_expf(v); break;
            case 1: r = lq_erff(v); break;
            case 2: r = lq_gelu(v); break;
            case 3: r = lq_sigmoid(v); break;
            case 4: r = lq_softplus(v); break;
            case 5: r = lq_gelu_grad(v); break;
            default: r = v;
        }
        out[i] = r;
    }
}

LQ_EXPORT int lq_ref_abi_version(void) { return 1; }
