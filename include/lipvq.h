/* lipvq.h -- C ABI of the MI355X (gfx950) LipVQ-VAE action-tokenizer library.
 *
 * The reference is pure Python/PyTorch: its "FFI" for this path is the set of stock torch ops
 * that LLFQVAE_V4.forward / VQVAE.forward issue (reference files, relative to /root/reference:
 *   v5 = robomimic/models/vq_vae/backbone_lfqvae_v5.py,  vq = robomimic/models/vq_vae/backbone.py).
 * Each entry point below replaces one group of those ops with one hand-written HIP launch and
 * cites the lines it replaces.  The reference-side binding (a ctypes stub called from a
 * torch.autograd.Function) is shown in INTEGRATION.md and implemented in
 * lipvq-vae_amd/_capi.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates); the library
 *     never frees, retains or reallocates it.  Tensors are row-major contiguous fp32 unless a
 *     parameter says otherwise; indices and usage counts are int64 (torch.long).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     Work is only enqueued; no entry point synchronises the device.
 *   - return value: 0 on success, a negative LIPVQ_E* code otherwise; lipvq_last_error()
 *     returns a thread-local message.  Nothing throws across the ABI.
 *   - stateless and re-entrant; one process per GPU.
 *   - arithmetic: "canonical fp32" (lipvq-vae_amd/csrc/lipvq_math.h): every forward tensor is
 *     bit-identical to oracle/lipvq_oracle.c on the same inputs.
 */
#ifndef LIPVQ_H_
#define LIPVQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIPVQ_ABI_VERSION 1

enum { LIPVQ_OK = 0, LIPVQ_EINVAL = -1, LIPVQ_EUNSUPPORTED = -2, LIPVQ_EHIP = -3 };

/* activation codes of lipvq_mlp3_* */
enum { LIPVQ_ACT_NONE = 0, LIPVQ_ACT_GELU = 1, LIPVQ_ACT_SIGMOID = 2, LIPVQ_ACT_RELU = 3 };

/* distance rules of lipvq_nearest_f32 */
enum {
    LIPVQ_DIST_NORM = 0,  /* v5:43-46  torch.norm(.., dim=-1) then argmin (compares square roots) */
    LIPVQ_DIST_SQSUM = 1  /* vq:58-63  (..).pow(2).sum(-1) then argmin */
};

int lipvq_abi_version(void);
const char* lipvq_last_error(void);

/* Options: process-global switches for tests and measurements, set EXPLICITLY -- the library reads no environment variable.
 * Results are identical under every setting (the parity suites run both screens and every kernel shape); only speed changes.
 * value = NULL restores the default.  Unknown name: LIPVQ_EINVAL.  Read per launch unless noted.
 *   screen_mode       "coarse" | "fine": force the one-product / three-product screen (default: the shape's measured winner,
 *                     lipvq_screen_is_coarse)
 *   tok_shape         "w8rg1" | "w8rg2" | "w4rg2" | "w4rg1": (waves per workgroup, row groups per wave) of the fused launch
 *   tok_ze_rows       batch size up to which a fused launch stores z_e for its exact stage when nothing else asks for it
 *   tok_grid          workgroups of the fused launch's persistent grid (default 256 = one per CU)
 *   tok_inplace       "0" | "1": the fused launch never / whenever possible lets its waves decide their uncertified rows in place
 *                     instead of listing them for a second kernel (default: whenever possible = K <= 2048 under the three-product
 *                     screen with z_e rows stored)
 *   tok_defer_ze, tok_nt_ze   "0" | "1": the fused launch's two device-dependent schedule choices (the last z_e tile's stores issued
 *                     behind the screen's first stage copies; z_e rows stored nontemporal) -- overrides the defaults (0, 1) and
 *                     whatever lipvq_tokenize_tune_f32 found for the device
 *   tok_ze_ring       "0": the z_e scratch of an in-place launch is written in full (N x D floats) instead of as a 2 048-wave ring
 *   rows_grid, wgrad_chunk, wgrad_per_tile, wgrad_no_wg5, wgrad_rows, embed_bwd_grid, mlp3_small_tiles, mlp3_sub, mlp3_lds_rows
 *                     grid / route choices of the exact-rows, weight-gradient, embedding-backward and MLP kernels (read ONCE, at the
 *                     first launch of that kind: set them before it) */
int lipvq_set_option(const char* name, const char* value);
const char* lipvq_get_option(const char* name);        /* NULL = default */

/* v5:6-12 normalization():  scale[i] = min(1, softplus(ci[i]) / sum_j |W[i][j]|),  Wn = W * scale.
 * W [D][H], ci [D]; scale [D] and Wn [D][H] are outputs (either may be NULL). */
int lipvq_lipschitz_scale_f32(const float* W, const float* ci, float* scale, float* Wn, int D, int H,
                              void* stream);

/* A three-layer perceptron y = act2(L2(act1(L1(act0(L0(x)))))) with nn.Linear weights
 * W_l [J_l][K_l] -- the shape of every MLP stack on the path:
 *   v5:54-59 + v5:22-24   encoder + Lipschitz layer   (A -> 64 -> hidden -> D; gelu, gelu, sigmoid; W2 = Wn)
 *   v5:62-68              decoder + to_output         (D -> 64 -> hidden -> A; gelu, gelu, none)
 *   vq:17-24 / vq:25-32   ReLU encoder / decoder      (relu x3)
 * The weights are first re-laid out into MFMA A-operand order by lipvq_mlp3_pack_f32 (once
 * per parameter update); `packed` needs lipvq_mlp3_packed_floats() floats.
 * J0 and J1 must be multiples of 32 (<= 256); K0 and J2 are free. */
size_t lipvq_mlp3_packed_floats(int K0, int J0, int J1, int J2);
int lipvq_mlp3_pack_f32(const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, float* packed, int K0, int J0, int J1,
                        int J2, void* stream);
/* x [N][K0], or, when gather_idx != NULL, row n of the input is table x[gather_idx[n]] (the
 * codebook gather of v5:47 fused into the decoder's first layer).  y [N][J2].  pre0/pre1/pre2
 * (each may be NULL) receive the pre-activations [N][J_l] that the backward pass needs. */
int lipvq_mlp3_f32(const float* x, const int64_t* gather_idx, const float* packed, float* y,
                   float* pre0, float* pre1, float* pre2, int64_t N, int K0, int J0, int J1, int J2,
                   int act0, int act1, int act2, void* stream);

/* v5:37-48 LFQQuantizer.forward / vq:55-66 VQVAE.quantize (distance + argmin + gather):
 *   idx[n] = first k minimising dist(z[n], codebook[k]);  zq[n] = codebook[idx[n]];
 *   usage[k] += number of rows mapped to k (usage may be NULL; it is NOT zeroed here).
 * z [N][D], codebook [K][D], idx [N] int64, zq [N][D] (may be NULL), best [N] (may be NULL)
 * receives the winning compared value.  Never materialises [N][K] or [N][K][D]. */
int lipvq_nearest_f32(const float* z, const float* codebook, int64_t* idx, float* zq,
                      int64_t* usage, float* best, int64_t N, int K, int D, int dist, void* stream);

/* vq:74 straight-through value  out = z_e + (z_q - z_e)  (fp32, as torch rounds it). */
int lipvq_ste_f32(const float* ze, const float* zq, float* out, int64_t n_elem, void* stream);

/* v5:79-81 / vq:50,69-70  F.mse_loss pair:  out[0] = mean((xr-x)^2) over nx elements,
 * out[1] = mean((zq-ze)^2) over nz elements.  workspace: lipvq_mse_workspace_bytes() bytes. */
size_t lipvq_mse_workspace_bytes(void);
int lipvq_mse_pair_f32(const float* xr, const float* x, int64_t nx, const float* zq, const float* ze,
                       int64_t nz, float* out2, void* workspace, void* stream);

/* The same two means in out3[0], out3[1], plus out3[2] = the tokenizer's loss built from them on the device in the
 * reference's fp32 association:  LIPVQ_LOSS_LLFQ  (m0 + w m1) + w m1   (backbone_lfqvae_v5.py:83, w = 0.25)
 *                                LIPVQ_LOSS_VQ    m0 + (m1 + w m1)     (backbone.py:50-51, 69-71, w = commitment_cost). */
#define LIPVQ_LOSS_LLFQ 0
#define LIPVQ_LOSS_VQ 1
int lipvq_mse_pair_loss_f32(const float* xr, const float* x, int64_t nx, const float* zq, const float* ze, int64_t nz,
                            float* out3, float w, int form, void* workspace, void* stream);


/* ---- nearest code, fast path: MFMA screening + exact re-scoring (same results as
 *      lipvq_nearest_f32(.., LIPVQ_DIST_NORM); design: lipvq-vae_amd/csrc/lipvq_screen.hip) ------ */

/* Per-codebook preparation (centred, fp16 hi/lo split, MFMA fragment order, |e|^2, bounds).
 * Redo it whenever the codebook changes.  prep: lipvq_nearest_prep_bytes(K, D) bytes. */
size_t lipvq_nearest_prep_bytes(int K, int D);
int lipvq_nearest_prepare_f32(const float* codebook, void* prep, int K, int D, void* stream);
int lipvq_nearest_screened_supported(int K, int D);          /* 1 for any D in 1 ... 208 (widths other than 32 / 64 / 128 / 208 run
                                                               * the next larger instance on zero-padded columns) */
size_t lipvq_nearest_workspace_bytes(int64_t N);
/* The screen comes in two strengths with identical results: three fp16 products per algorithmic product (22-bit operands; a
 * fraction of a percent of the rows left to the exact kernel) or ONE (11-bit operands, a third of the matrix work, lower-bound
 * bookkeeping; 10-40 % of the rows left to the exact stage with their two or three candidates each).  The library picks per
 * shape (the one-product screen from K = 4096 on, and from K = 1024 on for D > 64); this query tells which one
 * a call with (K, D) would run now (environment LIPVQ_SCREEN_MODE=coarse|fine overrides: a measurement knob). */
int lipvq_screen_is_coarse(int K, int D);
/* idx / zq / usage exactly as lipvq_nearest_f32.  After the call the first int of `workspace`
 * holds how many rows were decided by the exact kernel (the rest were certified by the screen). */
int lipvq_nearest_screened_f32(const float* z, const float* codebook, const void* prep, int64_t* idx, float* zq,
                               int64_t* usage, void* workspace, int64_t N, int K, int D, void* stream);
/* Same contract again (idx / zq / usage as lipvq_nearest_f32 with LIPVQ_DIST_NORM) with every row decided by the exact
 * re-scoring kernel: needs no prepared codebook.  For batches of a few thousand rows (training steps, where the codebook
 * changes every step).  Any D (tuned instances for 32 / 64 / 128 / 208, a generic kernel otherwise). */
int lipvq_nearest_rows_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage, int64_t N,
                           int K, int D, void* stream);

/* The same decision for SMALL batches (the reference's training step and rollouts: B*T = 1 ... a few hundred rows) with the
 * whole chip busy: a workgroup scores 4 rows against 64 codes staged in LDS, the grid is (row groups) x (code groups), a row's
 * partial minima meet behind a counter.  dist: LIPVQ_DIST_NORM | LIPVQ_DIST_SQSUM.  Identical idx / zq / usage.
 * lipvq_nearest_small_supported: N <= 4096, D a multiple of 4 whose LDS image fits (D <= 240).
 * workspace: lipvq_nearest_small_workspace_bytes(N, K) bytes, 16-byte aligned, ZERO on entry; the call leaves it zero (one
 * zero-filled buffer serves every later call on a stream, whatever its shape). */
int lipvq_nearest_small_supported(int64_t N, int K, int D);
size_t lipvq_nearest_small_workspace_bytes(int64_t N, int K);
int lipvq_nearest_small_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage, void* workspace,
                            int64_t N, int K, int D, int dist, void* stream);
/* The plain VQVAE's quantizer (reference robomimic/models/vq_vae/backbone.py:55-63: `(z_e.unsqueeze(1) - E).pow(2).sum(-1)`,
 * argmin) through the same two routes: idx / zq / usage exactly as lipvq_nearest_f32(.., LIPVQ_DIST_SQSUM).  The screen is the
 * same certified MFMA screen (its margin covers the sum rule's rounding too); uncertified rows are decided by the exact kernel in
 * torch's cascade-sum order, first minimum.  prep / workspace as for lipvq_nearest_screened_f32.  D in 1 ... 208. */
int lipvq_vq_nearest_screened_f32(const float* z, const float* codebook, const void* prep, int64_t* idx, float* zq,
                                  int64_t* usage, void* workspace, int64_t N, int K, int D, void* stream);
int lipvq_vq_nearest_rows_f32(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t* usage, int64_t N,
                              int K, int D, void* stream);
/* Test hook: also dumps the approximate distances d~ [N][Kpad] (Kpad = K rounded up to 32) and takes
 * the error-bound factor gamma from the caller. */
int lipvq_screen_debug_f32(const float* z, const float* codebook, const void* prep, int64_t* idx, float* zq,
                           int64_t* usage, void* workspace, float* dtilde, float gamma, int64_t N, int K, int D,
                           void* stream);

/* ---- the metric's path in one launch: encode + quantize (v5:71-74), lipvq-vae_amd/csrc/lipvq_fused.hip ---- */

/* x [N][A] -> idx [N], zq [N][D] (may be NULL), usage [K] (may be NULL, accumulated), ze_out [N][D] (may be
 * NULL).  Same results as lipvq_mlp3_f32(gelu, gelu, sigmoid) followed by lipvq_nearest_f32(LIPVQ_DIST_NORM);
 * z_e stays in registers.  packed = lipvq_mlp3_pack_f32 of the encoder stack (A -> J0 -> J1 -> D, W2 already
 * Lipschitz-normalised); prep = lipvq_nearest_prepare_f32 of the codebook.  Supported: J0 = 64, J1 = 128
 * (the reference's widths), D in {32, 64, 128, 208} (208 = the width the reference itself runs, v5:89-92 / obs_nets.py:1225-1227:
 * its 112 KB of Lipschitz-layer weights are streamed through LDS), A <= 64.  raw6 = host array of the six device pointers {W0, b0, W1, b1,
 * W2 (normalised), b2}: the exact kernel re-encodes the few uncertified rows from x with them, so z_e is written to
 * HBM only when ze_out is given.  After the call the first int of `workspace` holds the number of rows the screen left to an
 * exact decision.  WORKSPACE CONTRACT (round 4): the first 64 bytes are a header whose counters are zero between calls -- zero a
 * fresh workspace ONCE with lipvq_tokenize_workspace_init (or a 64-byte memset); every lipvq_tokenize_* / lipvq_vq_tokenize_*
 * call leaves them at zero (its last kernel does that: no fill launch per call), and one that returns an error re-zeroes the
 * header.  One workspace serves one stream at a time.  Shapes that run the one-product screen (lipvq_screen_is_coarse) leave 10-20 % of the rows to the exact stage,
 * which then reads z_e rows: z_e is written to ze_out if given, else to a scratch inside the workspace -- which is why
 * lipvq_tokenize_workspace_bytes(N, D) always includes N x D floats (plus 72 bytes of row / candidate lists per row). */
int lipvq_tokenize_supported(int A, int J0, int J1, int D, int K);
int lipvq_tokenize_fast_supported(int A, int J0, int J1, int D, int K);      /* lipvq_tokenize_fast_f32: D in {32, 64, 128} */
size_t lipvq_tokenize_workspace_bytes(int64_t N, int D);
int lipvq_tokenize_workspace_init(void* workspace, void* stream);
int lipvq_tokenize_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                       const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out, void* workspace,
                       int64_t N, int A, int J0, int J1, int D, int K, void* stream);
/* MI355X devices hold different clocks under the same kernel, and two schedule choices of the fused launch (identical results) win
 * on some devices and lose on others (profiles/r04_i_clock_ab.txt).  This call measures the four combinations with the caller's
 * own arguments (those of lipvq_tokenize_f32) -- each warmed, then `launches` back-to-back calls between HIP events, two alternating
 * rounds, minimum per combination -- and keeps the fastest as the current device's setting for every later lipvq_tokenize_* call of
 * the process.  SYNCHRONOUS (waits for the stream), not capturable; every launch writes idx / zq / ze_out and accumulates into usage
 * as lipvq_tokenize_f32 does.  choice (may be NULL): defer_ze | nt_ze << 1; ms4 (may be NULL): ms per launch of the four.
 * Latent widths above 64 have no such choice (their instances fix both): the call returns at once with choice = -1. */
int lipvq_tokenize_tune_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                            const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out, void* workspace,
                            int64_t N, int A, int J0, int J1, int D, int K, void* stream, int launches, int* choice, float* ms4);
/* The forward half of a training step in the same launch: lipvq_tokenize_f32 that also stores what autograd saves for
 * the backward of v5:71-74 -- z_e [N][D] and the pre-activations pre0 [N][J0], pre1 [N][J1], pre2 [N][D] (all required,
 * 16-byte aligned), bit-identical to lipvq_mlp3_f32(x, .., pre0, pre1, pre2). */
int lipvq_tokenize_train_f32(const float* x, const float* packed, const float* const* raw6, const float* codebook,
                             const void* prep, int64_t* idx, float* zq, int64_t* usage, float* ze_out, float* pre0,
                             float* pre1, float* pre2, void* workspace, int64_t N, int A, int J0, int J1, int D, int K,
                             void* stream);

/* The plain VQVAE's encode + quantize in ONE launch (reference robomimic/models/vq_vae/backbone.py:40-66: encoder = three
 * Linear + ReLU, `(z_e.unsqueeze(1) - E).pow(2).sum(-1)`, argmin, embedding lookup): the same persistent kernel with ReLU
 * activations and a per-row fp16 scale (a ReLU latent is unbounded).  packed = lipvq_mlp3_pack_f32 of the encoder (plain weights),
 * prep = lipvq_nearest_prepare_f32 of the embedding table, ze_out [N][D] REQUIRED (the straight-through value of vq:74 and the
 * exact stage read it).  Same results as lipvq_mlp3_f32(relu, relu, relu) + lipvq_nearest_f32(LIPVQ_DIST_SQSUM).  Shapes as
 * lipvq_tokenize_supported; workspace lipvq_tokenize_workspace_bytes(N, D). */
int lipvq_vq_tokenize_f32(const float* x, const float* packed, const float* codebook, const void* prep, int64_t* idx,
                          float* zq, int64_t* usage, float* ze_out, void* workspace, int64_t N, int A, int J0, int J1, int D,
                          int K, void* stream);
/* The same launch as the forward half of a VQVAE training step: also stores the three pre-activations pre0 [N][J0], pre1 [N][J1],
 * pre2 [N][D] (16-byte aligned) that lipvq_mlp3_bwd_f32 consumes -- bit for bit what lipvq_mlp3_f32 would save. */
int lipvq_vq_tokenize_train_f32(const float* x, const float* packed, const float* codebook, const void* prep, int64_t* idx,
                                float* zq, int64_t* usage, float* ze_out, float* pre0, float* pre1, float* pre2, void* workspace,
                                int64_t N, int A, int J0, int J1, int D, int K, void* stream);

/* ---- fast mode (opt-in): the encoder's three GEMMs on fp16 MFMAs with fp32 accumulation -- the half-precision
 * encoder of BASELINE.json's config 2 / SURVEY section 7.  NOT bit-identical to lipvq_tokenize_f32: a fraction of a
 * percent of the indices differ, always between near-equidistant codes (the flip rate is reported by bench.py and
 * bounded in tests/test_gpu_fast.py); z_q rows are exact codebook rows; the quantizer (screen + exact re-scoring) is the
 * parity one, applied to the fp16-encoder z_e; rows the screen cannot certify are decided by the exact kernel from the
 * fp32 encoder.  packed16: lipvq_mlp3_pack_f16_f32 of {W0, W1, W2 (normalised)} (lipvq_mlp3_packed_f16_bytes() bytes,
 * 16-byte aligned); `packed` still supplies the fp32 biases.  Same shapes as lipvq_tokenize_supported(). */
size_t lipvq_mlp3_packed_f16_bytes(int A, int J0, int J1, int D);
int lipvq_mlp3_pack_f16_f32(const float* W0, const float* W1, const float* W2, void* packed16, int A, int J0, int J1,
                            int D, void* stream);
int lipvq_tokenize_fast_f32(const float* x, const float* packed, const void* packed16, const float* const* raw6,
                            const float* codebook, const void* prep, int64_t* idx, float* zq, int64_t* usage,
                            void* workspace, int64_t N, int A, int J0, int J1, int D, int K, void* stream);

/* ---- backward (what autograd derives from v5:70-84 / vq:38-76) --------------------------- */

/* Backward-data of lipvq_mlp3_f32.  gy [N][J2] = dL/dy.  pre0/pre1/pre2 are the saved
 * pre-activations (pre2 may be NULL when act2 is the identity).  packed_bwd holds the transposed
 * weights (lipvq_mlp3_pack_bwd_f32; lipvq_mlp3_packed_bwd_floats() floats).  Outputs:
 *   g2 [N][J2] = gy * act2'(pre2)      (may be NULL; equal to gy when act2 is the identity)
 *   g1 [N][J1], g0 [N][J0]             dL/d(pre-activation) of layers 1 and 0
 *   gx [N][K0] = dL/dx                 (may be NULL: the last GEMM is then skipped) */
size_t lipvq_mlp3_packed_bwd_floats(int K0, int J0, int J1, int J2);
int lipvq_mlp3_pack_bwd_f32(const float* W0, const float* W1, const float* W2, float* packed, int K0,
                            int J0, int J1, int J2, void* stream);
/* Two stacks' backward packs in one launch (small training steps are launch-count bound). */
int lipvq_mlp3_pack_bwd2_f32(const float* aW0, const float* aW1, const float* aW2, float* a_packed, int aK0, int aJ0, int aJ1,
                             int aJ2, const float* bW0, const float* bW1, const float* bW2, float* b_packed, int bK0, int bJ0,
                             int bJ1, int bJ2, void* stream);
int lipvq_mlp3_bwd_f32(const float* gy, const float* pre0, const float* pre1, const float* pre2,
                       const float* packed_bwd, float* g2, float* g1, float* g0, float* gx, int64_t N,
                       int K0, int J0, int J1, int J2, int act0, int act1, int act2, void* stream);

/* lipvq_mlp3_f32 with the tokenizer's loss folded in (the decoder of backbone_lfqvae_v5.py:76-83 at large batches): besides y
 * and the saved pre-activations,
 *   out3[0] = mean((y - target)^2)            target [N][J2]   (the reconstruction error)
 *   out3[1] = mean((input rows - latent)^2)   latent [N][K0]   (input rows = x, or x[gather_idx[n]]: z_q against z_e)
 *   out3[2] = the loss, as lipvq_mse_pair_loss_f32 forms it from the two means (w, form)
 * summed in double by the kernel itself -- no second pass over the operands.  workspace: lipvq_mse_workspace_bytes().
 * ste_out [N][K0], optional: the stack then runs on latent + (input rows - latent) -- the plain VQVAE's straight-through value
 * z_e + (z_q - z_e).detach() of backbone.py:74, lipvq_ste_f32's two roundings -- and stores it there (out3[1] stays the mean over
 * the rows as given: z_q against z_e).
 * Large batches on the reference's hidden widths only: lipvq_mlp3_loss_supported() (LIPVQ_EUNSUPPORTED otherwise). */
int lipvq_mlp3_loss_supported(int64_t N, int K0, int J0, int J1, int J2);
int lipvq_mlp3_loss_f32(const float* x, const int64_t* gather_idx, const float* packed, float* y, float* pre0, float* pre1,
                        float* pre2, int64_t N, int K0, int J0, int J1, int J2, int act0, int act1, int act2,
                        const float* target, const float* latent, float* ste_out, float* out3, float w, int form,
                        void* workspace, void* stream);

/* The same chain with the VQ losses' gradient terms folded in (what autograd derives from backbone_lfqvae_v5.py:79-83 /
 * backbone.py:69-74 next to the stack's own backward; three lipvq_scaled_diff_f32 launches per step otherwise):
 *   in_b  != NULL:  gy := (in_alpha  * *gscale) * (act2(pre2) - B)   -- gy is not read; act2(pre2) is the forward's own output
 *                                                                       (z_e), re-evaluated from the saved pre-activation
 *   out_a != NULL:  gx := (out_alpha * *gscale) * (A - B) + gx
 * (one of the two per launch) with A / B the rows of out_a / in_b / out_b, [N][K0] ([N][J2] for in_b) floats, or -- when the
 * matching index vector is given -- rows of a table picked by it (z_q = codebook[idx] without materialising it).  gscale: device scalar or NULL (= 1).
 * Same numbers as lipvq_scaled_diff_f32 followed by lipvq_mlp3_bwd_f32.  Large batches on the reference's hidden widths
 * only: ask lipvq_mlp3_bwd_vq_supported() first (LIPVQ_EUNSUPPORTED otherwise). */
int lipvq_mlp3_bwd_vq_supported(int64_t N, int K0, int J0, int J1, int J2);
int lipvq_mlp3_bwd_vq_f32(const float* gy, const float* pre0, const float* pre1, const float* pre2,
                          const float* packed_bwd, float* g2, float* g1, float* g0, float* gx, int64_t N,
                          int K0, int J0, int J1, int J2, int act0, int act1, int act2,
                          const float* in_b, const int64_t* in_b_idx, float in_alpha, const float* out_a, const int64_t* out_a_idx, const float* out_b,
                          const int64_t* out_b_idx, float out_alpha, const float* gscale, void* stream);

/* Weight/bias gradient of one Linear layer:  gW [J][Kd] = G^T . act(H),  gb [J] = column sums of G.
 * G [N][J] = dL/d(pre-activation); H [N][Kd] = the layer's input, given as a saved pre-activation
 * plus the activation code h_act to re-apply (LIPVQ_ACT_NONE for raw inputs), or, with hidx,
 * rows H[hidx[n]] of a table (the codebook gather).  workspace: lipvq_wgrad_workspace_bytes(). */
size_t lipvq_wgrad_workspace_bytes(int64_t N, int J, int Kd);
int lipvq_wgrad_f32(const float* G, const float* H, const int64_t* hidx, int h_act, float* gW, float* gb,
                    void* workspace, int64_t N, int J, int Kd, void* stream);

/* Codebook gradient: gC[idx[n]] += g[n]  (the index_add_ behind v5:47 / vq:66).  gC [K][D] must be
 * zeroed by the caller.  Uses float atomics: the last bits may differ between runs. */
int lipvq_scatter_add_f32(const float* g, const int64_t* idx, float* gC, int64_t N, int K, int D,
                          void* stream);

/* The same sum, deterministically: rows are added in ascending row order per code (no atomics), so repeated runs give
 * bit-identical gradients (what torch.use_deterministic_algorithms(True) asks of index_add_).  gC is accumulated into. */
int lipvq_scatter_add_det_f32(const float* g, const int64_t* idx, float* gC, int64_t N, int K, int D, void* stream);

/* The same sum for a large batch without floating-point atomics (csrc/lipvq_scatter.hip): rows are counting-sorted by code
 * (stable), then
 *   sequential = 0: every code's rows are summed in segments of 256 rows in row order and the segment sums added in segment
 *                   order.  The order depends on (idx, N) only: repeated runs give bit-identical results;
 *   sequential = 1: one chain per (code, column) over ALL its rows in ascending row order -- the order of
 *                   lipvq_scatter_add_det_f32 and of torch's deterministic index_add_, bit for bit (a hot code is one long chain).
 * gC is accumulated into.  lipvq_scatter_add_sorted_supported: N >= 32768, K <= 16384.
 * workspace: lipvq_scatter_add_sorted_workspace_bytes() (0 when unsupported). */
int lipvq_scatter_add_sorted_supported(int64_t N, int K, int D);
size_t lipvq_scatter_add_sorted_workspace_bytes(int64_t N, int K, int D);
int lipvq_scatter_add_sorted_f32(const float* g, const int64_t* idx, float* gC, void* workspace, int64_t N, int K, int D,
                                 int sequential, void* stream);
/* The same scatter with the rows formed inside the summing kernel instead of read (the codebook gradient of a training step,
 * backbone_lfqvae_v5.py:81-83 / backbone.py:69-70, without its [N][D] intermediate):
 *   row n contributes  (alpha * *gscale) * (table[idx[n]] - ze[n])  (+ g[n] when g is not NULL)
 * with the roundings of lipvq_scaled_diff_f32; gscale: device scalar or NULL (= 1).  Same support, workspace and orders. */
int lipvq_scatter_add_sorted_vq_f32(const float* g, const float* ze, const float* table, float alpha, const float* gscale,
                                    const int64_t* idx, float* gC, void* workspace, int64_t N, int K, int D, int sequential,
                                    void* stream);

/* Backward of lipvq_lipschitz_scale_f32: gWn [D][H] -> gW [D][H], gci [D]. */
int lipvq_lipschitz_bwd_f32(const float* W, const float* ci, const float* gWn, float* gW, float* gci, int D,
                            int H, void* stream);

/* out = alpha * (gscale ? *gscale : 1) * (a - b) + (c ? c : 0): the d mse / d input terms of
 * v5:79-81 with the upstream gradient of the loss read from device memory (no host sync). */
int lipvq_scaled_diff_f32(const float* a, const float* b, const float* c, float alpha, const float* gscale,
                          float* out, int64_t n, void* stream);

/* ---- the step after the tokenizer: input embedding + interleave, lipvq-vae_amd/csrc/lipvq_embed.hip ----
 * ob = robomimic/models/obs_nets.py.  ob:2525-2543 input_embedding():
 *     e = embed_drop(embed_ln(embed_encoder(inputs) + time_embeddings))
 * ob:2580-2596: stack/view/cat of the three streams into transformer_embeddings [B][3T][E]
 *     (context_obs at 2t, context_actions at 2t+1, obs at 2T+t).
 * Dropout (ob:2541) is left to the caller: identity in eval, torch's own RNG in training. */

/* ob:2536  y [N][E] = x [N][Kin] . W [E][Kin]^T + b [E] (b may be NULL) -- the canonical Linear (one k-ordered fmaf
 * chain per output, fp32 MFMA).  Used (i) once per parameter update on the codebook
 * (x = codebook, N = K) to build the [K][E] table that replaces the Linear over the tokenizer's output rows, since
 * z_latent[n] = codebook[idx[n]] (v5:47,84); (ii) on rows that are not codebook rows (observation streams). */
int lipvq_linear_f32(const float* x, const float* W, const float* b, float* y, int64_t N, int Kin, int E,
                     void* stream);

/* ob:2537-2540 + ob:2584-2596  For n = b*T + t (n < N):
 *     out[b*out_batch_stride + t*out_t_stride + out_offset + e] =
 *         LayerNorm_e(src[idx ? idx[n] : n][e] + (pos ? pos[t][e] : 0)) * ln_w[e] + ln_b[e]
 * src [src_rows][E] is the table (idx = the tokenizer's int64 indices) or dense pre-LayerNorm rows (idx NULL).
 * pos [T][E] is the time embedding the reference adds (ob:2485-2523: nn.Parameter [1][T][E], nn.Embedding rows 0..T-1
 * or the sinusoidal table).  The strides/offset (in floats, multiples of 4) place each row directly in its interleaved
 * slot: context_actions use (3T*E, 2E, E), context_obs (3T*E, 2E, 0), obs (3T*E, E, 2T*E).  stats [N][2] (may be
 * NULL) receives (mean, rstd) for the backward.  E: multiple of 4, <= 1024.  An index outside [0, src_rows) yields a
 * NaN row (no fault). */
int lipvq_embed_rows_f32(const float* src, const int64_t* idx, const float* pos, const float* ln_w, const float* ln_b,
                         float eps, float* out, float* stats, int64_t N, int T, int E, int64_t src_rows,
                         int64_t out_batch_stride, int64_t out_t_stride, int64_t out_offset, void* stream);

/* Backward of lipvq_embed_rows_f32 (what autograd derives from ob:2536-2540).  gout is addressed like out.
 * ACCUMULATES (caller zero-fills): g_src [src_rows][E] (gradient of the table rows; with idx NULL it is written, not
 * accumulated), g_pos [T][E], g_lnw [E], g_lnb [E]; any of them may be NULL.  The Linear's own gradients follow from
 * g_src with lipvq_wgrad_f32 (g_src^T . codebook) and lipvq_linear_f32 (g_src . W). */
int lipvq_embed_rows_bwd_f32(const float* gout, const float* src, const int64_t* idx, const float* pos,
                             const float* stats, const float* ln_w, float* g_src, float* g_pos, float* g_lnw,
                             float* g_lnb, int64_t N, int T, int E, int64_t src_rows, int64_t out_batch_stride,
                             int64_t out_t_stride, int64_t out_offset, void* stream);

/* The same backward for a large batch of INDEXED rows without atomics on the table or the time embedding: a wave owns one time
 * step (its time-embedding gradient stays in registers), the rows' gradients are written once and summed per table row by the
 * counting-sort scatter (lipvq_scatter_add_sorted_f32): reproducible, and independent of how the indices are distributed
 * (the atomic kernel: 8.7 ms at N = 524 280, E = 512, T = 10; 55 ms when every row picks the same table row).  idx == NULL
 * (dense rows, N >= 32768): the row gradients are g_src itself, workspace may be NULL.  workspace: lipvq_embed_rows_bwd_workspace_bytes() (0 = unsupported: N < 32768 or more than 16384 table rows). */
int lipvq_embed_rows_bwd_ws_supported(int64_t N, int T, int E, int64_t src_rows);
size_t lipvq_embed_rows_bwd_workspace_bytes(int64_t N, int T, int E, int64_t src_rows);
int lipvq_embed_rows_bwd_ws_f32(const float* gout, const float* src, const int64_t* idx, const float* pos, const float* stats,
                                const float* ln_w, float* g_src, float* g_pos, float* g_lnw, float* g_lnb, void* workspace,
                                int64_t N, int T, int E, int64_t src_rows, int64_t out_batch_stride, int64_t out_t_stride,
                                int64_t out_offset, void* stream);

/* lipvq_linear_f32 with an activation epilogue: y = act(x . W^T + b); pre (may be NULL) receives the pre-activation
 * for the backward.  bin:29-30 (second Linear + GELU of AdaptiveBinActionEmbedding.output_layer). */
int lipvq_linear_act_f32(const float* x, const float* W, const float* b, float* y, float* pre, int64_t N, int Kin,
                         int E, int act, void* stream);

/* ---- the sibling tokenizer behind `bin_enabled` (obs_nets.py:1214-1217): AdaptiveBinActionEmbedding,
 *      bin = robomimic/models/bin_action/backbone.py; lipvq-vae_amd/csrc/lipvq_bin.hip ---- */

/* bin:37-40 update_running_stats(): running_min[i] = min(running_min[i], min_n actions[n][i]), running_max likewise,
 * updated IN PLACE (actions [N][A], buffers [A]). */
int lipvq_bin_minmax_f32(const float* actions, float* running_min, float* running_max, int64_t N, int A, void* stream);

/* bin:42-66 compute_bins() + discretize(): bins[i][n] = clamp(bucketize(actions[n][i], linspace(running_min[i],
 * running_max[i], num_bins + 1)) - 1, 0, num_bins - 1).  bins is int64 [A][N] (dimension-major: each dimension's
 * indices are contiguous; the reference's [N][A] stack is its transpose).  Bit-exact restatement of torch's linspace
 * rounding and lower-bound search.  A * (num_bins + 1) <= 4096. */
int lipvq_bin_discretize_f32(const float* actions, const float* running_min, const float* running_max, int64_t* bins,
                             int64_t N, int A, int num_bins, void* stream);

/* bin:42-53 compute_bins() alone: boundaries [A][num_bins + 1] = linspace(running_min[i], running_max[i], num_bins + 1). */
int lipvq_bin_boundaries_f32(const float* running_min, const float* running_max, float* boundaries, int A, int num_bins,
                             void* stream);

/* bin:77-86 embeddings + cat + output_layer[0..1]:  pre1[n][j] = b1[j] + sum_i P[i][bins[i][n]][j], h = gelu(pre1).
 * P [A][num_bins][H] = per-dimension product of the embedding table with its 64-column block of the first Linear's
 * weight (lipvq_linear_f32(emb_i, W1[:, 64 i : 64 i + 64], NULL)), rebuilt when a parameter changes.  h [N][H];
 * pre1 [N][H] may be NULL. */
int lipvq_bin_hidden_f32(const int64_t* bins, const float* P, const float* b1, float* h, float* pre1, int64_t N, int A,
                         int num_bins, int H, void* stream);

/* out = g * act'(pre), elementwise over n floats (backward of the GELUs of bin:28,30). */
int lipvq_act_bwd_f32(const float* g, const float* pre, float* out, int64_t n, int act, void* stream);

/* ---- the DEFAULT action branch (obs_nets.py:1244-1260, the `else` of the tokenizer switch; ob = robomimic/models/obs_nets.py):
 *      nn.Sequential(spectral_norm(Linear(A,64)), GELU, spectral_norm(Linear(64,128)), GELU, spectral_norm(Linear(128,D)),
 *                    nn.TransformerEncoder(TransformerEncoderLayer(d_model=D, nhead=8, dim_feedforward=256, activation="gelu"), 4),
 *                    Linear(D,D))
 *      called on the 2-D [B*T][A] action tensor (ob:1344), i.e. the encoder sees ONE unbatched sequence of S = B*T actions.
 *      Linears: lipvq_linear_act_f32 / lipvq_wgrad_f32 above.  lipvq-vae_amd/csrc/lipvq_xf.hip, lipvq-vae_amd/default_branch.py ---- */

/* ob:1253-1257 torch.nn.utils.spectral_norm (torch/nn/utils/spectral_norm.py compute_weight), W [J][K] = weight_orig,
 * u [J] = weight_u, v [K] = weight_v.  do_power_iteration (module.training): v = normalize(W^T u), u = normalize(W v)
 * (x / max(|x|, eps), one iteration) written back in place.  Always: sigma[0] = u . (W v), Wsn = W / sigma.  J, K <= 256. */
int lipvq_spectral_norm_f32(const float* W, float* u, float* v, float* Wsn, float* sigma, int J, int K,
                            int do_power_iteration, float eps, void* stream);
/* Backward of W -> Wsn with u, v held constant (as torch does):  gW = (gWsn - <gWsn, Wsn> u v^T) / sigma. */
int lipvq_spectral_norm_bwd_f32(const float* gWsn, const float* Wsn, const float* u, const float* v, const float* sigma,
                                float* gW, int J, int K, void* stream);

/* ob:1245-1258 nn.MultiheadAttention inside TransformerEncoderLayer on an unbatched sequence: qkv [S][3D] = in_proj(x)
 * (q | k | v column blocks; head h owns columns [h D/H, (h+1) D/H) of each), out [S][D] = concat_h softmax(Q_h K_h^T /
 * sqrt(D/H)) V_h, lse [H][S] = log-sum-exp of the scaled scores (for the backward).  keep [H][S][S] bytes (may be NULL):
 * the attention-probability dropout mask of training mode, kept entries scaled by 1 / keep_prob.  D/H <= 32. */
int lipvq_attention_f32(const float* qkv, float* out, float* lse, const unsigned char* keep, float keep_prob, int64_t S,
                        int D, int H, void* stream);
/* Its backward: gqkv [S][3D] from gout [S][D]; delta [H][S] is scratch. */
int lipvq_attention_bwd_f32(const float* qkv, const float* out, const float* gout, const float* lse, float* gqkv,
                            float* delta, const unsigned char* keep, float keep_prob, int64_t S, int D, int H, void* stream);

/* ob:1245 post-norm residual of TransformerEncoderLayer (norm_first = False): y = LayerNorm(a + b) * w + bias, rows of
 * E <= 256 floats; b may be NULL.  xhat [N][E] and rstd [N] (either may be NULL) are saved for the backward. */
int lipvq_add_layernorm_f32(const float* a, const float* b, const float* w, const float* bias, float eps, float* y,
                            float* xhat, float* rstd, int64_t N, int E, void* stream);
/* gx [N][E] (the gradient of BOTH a and b), and gw [E], gb [E] ACCUMULATED (caller zero-fills). */
int lipvq_layernorm_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* w, float* gx, float* gw,
                            float* gb, int64_t N, int E, void* stream);

/* icl.py:885-889, :970  optim.AdamW(vq_vae_model.parameters(), lr=1e-3, weight_decay=1e-4).step() for a LIST of tensors in
 * two launches (torch's foreach form is 8-10): params / grads / exp_avg / exp_avg_sq / steps are HOST arrays of `count`
 * DEVICE pointers (count <= 32), numels their element counts; steps[i] is a float32 device scalar (torch's capturable layout),
 * incremented here.  Standard AdamW (amsgrad off); the hyper-parameters are doubles (derived constants such as 1 - beta2 are
 * formed in double and rounded to fp32 once, as torch forms them).  workspace: lipvq_adamw_workspace_bytes() device bytes. */
size_t lipvq_adamw_workspace_bytes(void);
int lipvq_adamw_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    float* const* steps, const int64_t* numels, int count, double lr, double beta1, double beta2, double eps,
                    double weight_decay, void* workspace, void* stream);

/* ---- opt-in extension: EMA codebook update (not in the reference; named by BASELINE.json's north star) ----
 * cluster_size [K] and embed_sum [K][D] are the running statistics (updated in place), counts [K] int64 = this batch's
 * code usage (lipvq_nearest_f32 / lipvq_tokenize_f32 `usage`, summed over ranks), dw [K][D] = sum of the z_e rows mapped
 * to each code (lipvq_scatter_add_f32, summed over ranks).  Writes the new codebook:
 *     cluster_size = decay cluster_size + (1-decay) counts;  embed_sum = decay embed_sum + (1-decay) dw;
 *     n = sum cluster_size;  codebook[k] = embed_sum[k] / ((cluster_size[k] + eps) / (n + K eps) * n).
 * workspace: 8 bytes. */
int lipvq_ema_update_f32(float* cluster_size, float* embed_sum, const int64_t* counts, const float* dw, float* codebook,
                         float decay, float eps, int K, int D, void* workspace, void* stream);

/* ---- multi-GPU: the path's only cross-GPU exchange (the reference is single-GPU,
 *      robomimic/utils/torch_utils.py:48-50, so there is no reference line to cite) ----
 * One process per GPU, rows sharded, parameters replicated.  Per batch the code-usage histogram (and, with the EMA
 * extension, the per-code sums) is summed over the ranks by RCCL over xGMI.  The library binds RCCL at run time
 * (dlopen of librccl.so.1 -- the copy PyTorch has already mapped when there is one, so both share one RCCL), hence the
 * tokenizer library itself loads on hosts without RCCL; these four entries then return LIPVQ_EUNSUPPORTED.
 *
 * lipvq_allreduce_counts takes ANY ncclComm_t (as void*): the application's own communicator, or one made by
 * lipvq_comm_init from a 128-byte unique id that rank 0 obtains with lipvq_comm_unique_id and hands to the other ranks
 * over any out-of-band channel (torch.distributed's store, a file, MPI).  In place, sum, enqueued on `stream`. */
#define LIPVQ_COMM_ID_BYTES 128
int lipvq_comm_unique_id(void* id128);
int lipvq_comm_init(void** comm, const void* id128, int rank, int world);
int lipvq_comm_destroy(void* comm);
int lipvq_allreduce_counts(int64_t* counts, int K, void* comm /* ncclComm_t */, void* stream);
/* the same for fp32 payloads (EMA per-code sums [K][D], flat data-parallel gradients) */
int lipvq_allreduce_f32(float* buf, int64_t n, void* comm /* ncclComm_t */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIPVQ_H_ */
