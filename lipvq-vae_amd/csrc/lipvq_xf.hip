// lipvq_xf.hip -- the pieces of the DEFAULT action branch (reference robomimic/models/obs_nets.py:1244-1260, the `else` of the
// tokenizer switch: spectral-norm MLP + a 4-layer post-norm nn.TransformerEncoder attending over the flattened batch + Linear)
// that the tokenizer library did not have yet:
//   spectral_kernel        torch.nn.utils.spectral_norm's power iteration, sigma and W / sigma for one small weight matrix
//   spectral_bwd_kernel    backward of W -> W / sigma(W)  (u, v are constants, as in torch)
//   attention_kernel       softmax(Q K^T / sqrt(dh)) V per head on an UNBATCHED sequence [S][3D] (the reference hands the
//                          encoder a 2-D [B*T][D] tensor: every action of the batch attends to every other), fp32, online softmax
//   attention_bwd_q/kv     its backward (probabilities recomputed from the saved log-sum-exp)
//   add_layernorm_kernel   y = LayerNorm(a + b) * w + bias  (post-norm residual), and its backward
// The Linears run on the existing lipvq_linear_act_f32 / lipvq_wgrad_f32.  All sizes here are small (S = B*T rows of a training
// step, D <= 256): the kernels are written for launch latency and occupancy of a few workgroups, not for a roofline.
// ABI: include/lipvq.h ("default action branch").  Tolerance against torch: 1e-5 of each tensor's scale (tests/test_gpu_default.py).
#include "lipvq_common.h"

// ---------------------------------------------------------------------------------------------------
// spectral norm (torch/nn/utils/spectral_norm.py compute_weight): W [J][K] = weight_orig, u [J], v [K]
//   training:  v = normalize(W^T u), u = normalize(W v)   (one iteration, eps = 1e-12: x / max(|x|, eps)), written back
//   always:    sigma = u . (W v);  Wsn = W / sigma
// One workgroup; every reduction is a sequential fp32 chain in index order (deterministic; J, K <= 256).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spectral_kernel(const float* __restrict__ W, float* __restrict__ u, float* __restrict__ v,
                                                       float* __restrict__ Wsn, float* __restrict__ sigma_out, int J, int K,
                                                       int do_iter, float eps) {
    __shared__ float su[256], sv[256], sw[256];
    __shared__ float s_norm;
    const int tid = threadIdx.x;
    if (tid < J) su[tid] = u[tid];
    if (tid < K) sv[tid] = v[tid];
    __syncthreads();
    if (do_iter) {
        if (tid < K) {
            float a = 0.0f;
            for (int j = 0; j < J; ++j) a = lq_fma(W[(size_t)j * K + tid], su[j], a);
            sw[tid] = a;
        }
        __syncthreads();
        if (tid == 0) {
            float n2 = 0.0f;
            for (int k = 0; k < K; ++k) n2 = lq_fma(sw[k], sw[k], n2);
            const float n = lq_sqrt(n2);
            s_norm = n > eps ? n : eps;
        }
        __syncthreads();
        if (tid < K) { sv[tid] = sw[tid] / s_norm; v[tid] = sv[tid]; }
        __syncthreads();
    }
    if (tid < J) {                                   // (W v)[j]
        float a = 0.0f;
        for (int k = 0; k < K; ++k) a = lq_fma(W[(size_t)tid * K + k], sv[k], a);
        sw[tid] = a;
    }
    __syncthreads();
    if (do_iter) {
        if (tid == 0) {
            float n2 = 0.0f;
            for (int j = 0; j < J; ++j) n2 = lq_fma(sw[j], sw[j], n2);
            const float n = lq_sqrt(n2);
            s_norm = n > eps ? n : eps;
        }
        __syncthreads();
        if (tid < J) { su[tid] = sw[tid] / s_norm; u[tid] = su[tid]; }
        __syncthreads();
    }
    if (tid == 0) {
        float s = 0.0f;
        for (int j = 0; j < J; ++j) s = lq_fma(su[j], sw[j], s);
        s_norm = s;
        *sigma_out = s;
    }
    __syncthreads();
    const float sg = s_norm;
    for (int i = tid; i < J * K; i += 256) Wsn[i] = W[i] / sg;
}

extern "C" int lipvq_spectral_norm_f32(const float* W, float* u, float* v, float* Wsn, float* sigma, int J, int K,
                                       int do_power_iteration, float eps, void* stream) {
    if (!W || !u || !v || !Wsn || !sigma) return fail(LIPVQ_EINVAL, "spectral_norm: null pointer");
    if (J <= 0 || K <= 0 || J > 256 || K > 256) return fail(LIPVQ_EUNSUPPORTED, "spectral_norm: J=%d K=%d (1..256)", J, K);
    hipLaunchKernelGGL(spectral_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, W, u, v, Wsn, sigma, J, K,
                       do_power_iteration, eps);
    return check_launch("spectral_norm");
}

// Wsn = W / sigma with sigma = u^T W v, u and v constants:  gW = (gWsn - <gWsn, Wsn> u v^T) / sigma
__global__ __launch_bounds__(256) void spectral_bwd_kernel(const float* __restrict__ gWsn, const float* __restrict__ Wsn,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ sigma, float* __restrict__ gW, int J, int K) {
    __shared__ double part[256];
    const int tid = threadIdx.x;
    double a = 0.0;
    for (int i = tid; i < J * K; i += 256) a += (double)gWsn[i] * (double)Wsn[i];
    part[tid] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) part[tid] += part[tid + s];
        __syncthreads();
    }
    const float dot = (float)part[0];
    const float sg = *sigma;
    for (int i = tid; i < J * K; i += 256) {
        const int j = i / K, k = i - j * K;
        gW[i] = (gWsn[i] - dot * u[j] * v[k]) / sg;
    }
}

extern "C" int lipvq_spectral_norm_bwd_f32(const float* gWsn, const float* Wsn, const float* u, const float* v, const float* sigma,
                                           float* gW, int J, int K, void* stream) {
    if (!gWsn || !Wsn || !u || !v || !sigma || !gW) return fail(LIPVQ_EINVAL, "spectral_norm_bwd: null pointer");
    if (J <= 0 || K <= 0) return fail(LIPVQ_EINVAL, "spectral_norm_bwd: bad sizes");
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gWsn, Wsn, u, v, sigma, gW, J, K);
    return check_launch("spectral_norm_bwd");
}

// ---------------------------------------------------------------------------------------------------
// attention over an unbatched sequence.  qkv [S][3D]: q = columns [0, D), k = [D, 2D), v = [2D, 3D); head h owns columns
// [h dh, (h+1) dh) of each.  Workgroup = 16 queries x 16 key lanes of one head: lane p of a query takes keys p, p+16, ... of a
// 64-key LDS tile, keeps an online softmax (m, l, acc[dh]); the 16 partials are merged by xor-shuffles at the end.
// keep [H][S][S] bytes (1 = keep) + inv_keep = 1 / (1 - p): the attention-probability dropout of nn.MultiheadAttention in
// training mode (NULL in eval); the normaliser uses every key, the value sum only the kept ones.
// ---------------------------------------------------------------------------------------------------
#define XF_DH 32
#define XF_QB 16
#define XF_KT 64
__device__ __forceinline__ float xf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                        const unsigned char* __restrict__ keep, float inv_keep, int S, int D, int H) {
    __shared__ float sk[XF_KT][DH + 1], svv[XF_KT][DH + 1];
    const int dh = D / H, h = blockIdx.y;
    const int tid = threadIdx.x, qi = tid >> 4, p = tid & 15;
    const int q = blockIdx.x * XF_QB + qi;
    const int qc = q < S ? q : S - 1;
    const float scale = 1.0f / lq_sqrt((float)dh);
    float qr[DH], acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        qr[d] = d < dh ? qkv[(size_t)qc * 3 * D + h * dh + d] * scale : 0.0f;
        acc[d] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    for (int k0 = 0; k0 < S; k0 += XF_KT) {
        __syncthreads();
        for (int i = tid; i < XF_KT * DH; i += 256) {
            const int j = i / DH, d = i - j * DH;
            const int kj = k0 + j < S ? k0 + j : S - 1;
            sk[j][d] = d < dh ? qkv[(size_t)kj * 3 * D + D + h * dh + d] : 0.0f;            // (columns past the head width: zeros)
            svv[j][d] = d < dh ? qkv[(size_t)kj * 3 * D + 2 * D + h * dh + d] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < XF_KT / 16; ++jj) {
            const int j = p + 16 * jj;
            if (k0 + j < S) {
                float s = 0.0f;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    s = lq_fma(qr[d], sk[j][d], s);
                const float mn = fmaxf(m, s);
                const float c = xf_exp(m - mn), e = xf_exp(s - mn);
                l = lq_fma(l, c, e);
                const float w = keep ? (keep[((size_t)h * S + qc) * S + k0 + j] ? e * inv_keep : 0.0f) : e;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    acc[d] = lq_fma(acc[d], c, w * svv[j][d]);
                m = mn;
            }
        }
    }
    // merge the 16 key lanes of the query (lanes qi*16 .. qi*16+15 of the workgroup: 16 consecutive lanes of one wave)
    float mall = m;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) mall = fmaxf(mall, __shfl_xor(mall, off, 64));
    const float c = (m == -INFINITY) ? 0.0f : xf_exp(m - mall);
    l *= c;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        float a = acc[d] * c;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) a += __shfl_xor(a, off, 64);
        acc[d] = a;
    }
    if (q < S) {
        const float inv = 1.0f / l;
        for (int d = p; d < dh; d += 16) out[(size_t)q * D + h * dh + d] = acc[d] * inv;       // (acc[d] is the same in all 16 lanes)
        if (p == 0) lse[(size_t)h * S + q] = mall + __logf(l);
    }
}

extern "C" int lipvq_attention_f32(const float* qkv, float* out, float* lse, const unsigned char* keep, float keep_prob,
                                   int64_t S, int D, int H, void* stream) {
    if (!qkv || !out || !lse) return fail(LIPVQ_EINVAL, "attention: null pointer");
    if (S <= 0 || D <= 0 || H <= 0 || D % H != 0 || D / H > XF_DH)
        return fail(LIPVQ_EUNSUPPORTED, "attention: S=%lld D=%d H=%d (head width 1..%d)", (long long)S, D, H, XF_DH);
    if (S > 65535LL * XF_QB || (keep && !(keep_prob > 0.0f))) return fail(LIPVQ_EINVAL, "attention: bad S / keep_prob");
    // head width padded to 8, 16 or 32 columns (compile-time loops: the reference's 8 heads give 8 at D = 64, 26 at D = 208)
    const int dh = D / H;
    auto kfn = dh <= 8 ? attention_kernel<8> : (dh <= 16 ? attention_kernel<16> : attention_kernel<32>);
    hipLaunchKernelGGL(kfn, dim3((unsigned)((S + XF_QB - 1) / XF_QB), H), dim3(256), 0, (hipStream_t)stream, qkv, out, lse,
                       keep, keep ? 1.0f / keep_prob : 1.0f, (int)S, D, H);
    return check_launch("attention");
}

// Backward.  delta[h][i] = sum_d dO[i][hd] O[i][hd];  P_ij = exp(s_ij - lse_i);  dP_ij = (keep_ij / keep_prob) dO_i . V_j;
// dS_ij = P_ij (dP_ij - delta_i);  dQ_i = scale sum_j dS_ij K_j;  dK_j = scale sum_i dS_ij Q_i;  dV_j = sum_i (keep_ij/keep_prob) P_ij dO_i.
// attention_bwd_q: workgroup = 16 queries x 16 key lanes (as the forward) -> dQ;  attention_bwd_kv: 16 keys x 16 query lanes -> dK, dV.
template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                              const float* __restrict__ go, const float* __restrict__ lse,
                                                              float* __restrict__ gqkv, float* __restrict__ delta,
                                                              const unsigned char* __restrict__ keep, float inv_keep, int S, int D, int H) {
    __shared__ float sk[XF_KT][DH + 1], svv[XF_KT][DH + 1];
    const int dh = D / H, h = blockIdx.y;
    const int tid = threadIdx.x, qi = tid >> 4, p = tid & 15;
    const int q = blockIdx.x * XF_QB + qi;
    const int qc = q < S ? q : S - 1;
    const float scale = 1.0f / lq_sqrt((float)dh);
    float qr[DH], gor[DH], acc[DH];
    float dl = 0.0f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        qr[d] = d < dh ? qkv[(size_t)qc * 3 * D + h * dh + d] * scale : 0.0f;
        gor[d] = d < dh ? go[(size_t)qc * D + h * dh + d] : 0.0f;
        if (d < dh) dl = lq_fma(gor[d], o[(size_t)qc * D + h * dh + d], dl);
        acc[d] = 0.0f;
    }
    const float ls = lse[(size_t)h * S + qc];
    if (p == 0 && q < S) delta[(size_t)h * S + q] = dl;
    for (int k0 = 0; k0 < S; k0 += XF_KT) {
        __syncthreads();
        for (int i = tid; i < XF_KT * DH; i += 256) {
            const int j = i / DH, d = i - j * DH;
            const int kj = k0 + j < S ? k0 + j : S - 1;
            sk[j][d] = d < dh ? qkv[(size_t)kj * 3 * D + D + h * dh + d] : 0.0f;
            svv[j][d] = d < dh ? qkv[(size_t)kj * 3 * D + 2 * D + h * dh + d] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < XF_KT / 16; ++jj) {
            const int j = p + 16 * jj;
            if (k0 + j < S) {
                float s = 0.0f, dp = 0.0f;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    { s = lq_fma(qr[d], sk[j][d], s); dp = lq_fma(gor[d], svv[j][d], dp); }
                const float pr = xf_exp(s - ls);
                if (keep) dp = keep[((size_t)h * S + qc) * S + k0 + j] ? dp * inv_keep : 0.0f;
                const float ds = pr * (dp - dl) * scale;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    acc[d] = lq_fma(ds, sk[j][d], acc[d]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        float a = acc[d];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) a += __shfl_xor(a, off, 64);
        acc[d] = a;
    }
    if (q < S)
        for (int d = p; d < dh; d += 16) gqkv[(size_t)q * 3 * D + h * dh + d] = acc[d];
}

template <int DH>
__global__ __launch_bounds__(256) void attention_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ go,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ gqkv, const unsigned char* __restrict__ keep,
                                                               float inv_keep, int S, int D, int H) {
    __shared__ float sq[XF_KT][DH + 1], sg[XF_KT][DH + 1];
    __shared__ float sl[XF_KT], sd[XF_KT];
    const int dh = D / H, h = blockIdx.y;
    const int tid = threadIdx.x, ki = tid >> 4, p = tid & 15;
    const int k = blockIdx.x * XF_QB + ki;
    const int kc = k < S ? k : S - 1;
    const float scale = 1.0f / lq_sqrt((float)dh);
    float kr[DH], vr[DH], ak[DH], av[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        kr[d] = d < dh ? qkv[(size_t)kc * 3 * D + D + h * dh + d] : 0.0f;
        vr[d] = d < dh ? qkv[(size_t)kc * 3 * D + 2 * D + h * dh + d] : 0.0f;
        ak[d] = 0.0f; av[d] = 0.0f;
    }
    for (int q0 = 0; q0 < S; q0 += XF_KT) {
        __syncthreads();
        for (int i = tid; i < XF_KT * DH; i += 256) {
            const int j = i / DH, d = i - j * DH;
            const int qj = q0 + j < S ? q0 + j : S - 1;
            sq[j][d] = d < dh ? qkv[(size_t)qj * 3 * D + h * dh + d] * scale : 0.0f;
            sg[j][d] = d < dh ? go[(size_t)qj * D + h * dh + d] : 0.0f;
        }
        if (tid < XF_KT) {
            const int qj = q0 + tid < S ? q0 + tid : S - 1;
            sl[tid] = lse[(size_t)h * S + qj];
            sd[tid] = delta[(size_t)h * S + qj];
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < XF_KT / 16; ++jj) {
            const int j = p + 16 * jj;
            if (q0 + j < S) {
                float s = 0.0f, dp = 0.0f;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    { s = lq_fma(sq[j][d], kr[d], s); dp = lq_fma(sg[j][d], vr[d], dp); }
                const float pr = xf_exp(s - sl[j]);
                float kp = 1.0f;
                if (keep) kp = keep[((size_t)h * S + q0 + j) * S + kc] ? inv_keep : 0.0f;
                const float ds = pr * (dp * kp - sd[j]);          // (the forward's q was pre-scaled: dK_j = sum_i dS_ij (scale Q_i))
                const float pv = pr * kp;
#pragma unroll
                for (int d = 0; d < DH; ++d)
                    { ak[d] = lq_fma(ds, sq[j][d], ak[d]); av[d] = lq_fma(pv, sg[j][d], av[d]); }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        float a = ak[d], b = av[d];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        ak[d] = a; av[d] = b;
    }
    if (k < S)
        for (int d = p; d < dh; d += 16) {
            gqkv[(size_t)k * 3 * D + D + h * dh + d] = ak[d];
            gqkv[(size_t)k * 3 * D + 2 * D + h * dh + d] = av[d];
        }
}

extern "C" int lipvq_attention_bwd_f32(const float* qkv, const float* out, const float* gout, const float* lse, float* gqkv,
                                       float* delta, const unsigned char* keep, float keep_prob, int64_t S, int D, int H, void* stream) {
    if (!qkv || !out || !gout || !lse || !gqkv || !delta) return fail(LIPVQ_EINVAL, "attention_bwd: null pointer");
    if (S <= 0 || D <= 0 || H <= 0 || D % H != 0 || D / H > XF_DH)
        return fail(LIPVQ_EUNSUPPORTED, "attention_bwd: S=%lld D=%d H=%d (head width 1..%d)", (long long)S, D, H, XF_DH);
    if (S > 65535LL * XF_QB || (keep && !(keep_prob > 0.0f))) return fail(LIPVQ_EINVAL, "attention_bwd: bad S / keep_prob");
    const dim3 grid((unsigned)((S + XF_QB - 1) / XF_QB), H);
    const float ik = keep ? 1.0f / keep_prob : 1.0f;
    const int dh = D / H;
    auto kq = dh <= 8 ? attention_bwd_q_kernel<8> : (dh <= 16 ? attention_bwd_q_kernel<16> : attention_bwd_q_kernel<32>);
    auto kkv = dh <= 8 ? attention_bwd_kv_kernel<8> : (dh <= 16 ? attention_bwd_kv_kernel<16> : attention_bwd_kv_kernel<32>);
    hipLaunchKernelGGL(kq, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, gout, lse, gqkv, delta, keep, ik, (int)S, D, H);
    hipLaunchKernelGGL(kkv, grid, dim3(256), 0, (hipStream_t)stream, qkv, gout, lse, delta, gqkv, keep, ik, (int)S, D, H);
    return check_launch("attention_bwd");
}

// ---------------------------------------------------------------------------------------------------
// y = LayerNorm(a + b) * w + bias over rows of E <= 256 floats (one wave per row; two-pass moments in registers).
// Saves xhat [N][E] (normalised rows) and rstd [N] for the backward.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ w, const float* __restrict__ bias, float eps,
                                                            float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd,
                                                            int64_t N, int E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float x[4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        x[i] = e < E ? a[(size_t)row * E + e] + (b ? b[(size_t)row * E + e] : 0.0f) : 0.0f;
        s += x[i];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)E;
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        const float d = e < E ? x[i] - mean : 0.0f;
        x[i] = d;
        v = lq_fma(d, d, v);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    const float rs = 1.0f / lq_sqrt(v / (float)E + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        if (e < E) {
            const float xh = x[i] * rs;
            if (xhat) xhat[(size_t)row * E + e] = xh;
            y[(size_t)row * E + e] = lq_fma(xh, w[e], bias[e]);
        }
    }
    if (rstd && lane == 0) rstd[row] = rs;
}

extern "C" int lipvq_add_layernorm_f32(const float* a, const float* b, const float* w, const float* bias, float eps, float* y,
                                       float* xhat, float* rstd, int64_t N, int E, void* stream) {
    if (!a || !w || !bias || !y) return fail(LIPVQ_EINVAL, "add_layernorm: null pointer");
    if (N <= 0 || E <= 0 || E > 256) return fail(LIPVQ_EUNSUPPORTED, "add_layernorm: N=%lld E=%d (E <= 256)", (long long)N, E);
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, w, bias, eps, y,
                       xhat, rstd, N, E);
    return check_launch("add_layernorm");
}

// gx = rstd (g w - mean(g w) - xhat mean(g w xhat));  gw += sum_rows g xhat;  gb += sum_rows g   (gw, gb: caller zero-fills;
// per-workgroup partial sums in LDS, then one atomic per column and workgroup)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, const float* __restrict__ w,
                                                            float* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb,
                                                            int64_t N, int E, int rows_per_block) {
    __shared__ float s_gw[4][256], s_gb[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float pgw[4] = {0.f, 0.f, 0.f, 0.f}, pgb[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    int64_t rend = rbeg + rows_per_block;
    if (rend > N) rend = N;
    for (int64_t row = rbeg + wv; row < rend; row += 4) {
        float g[4], xh[4];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            g[i] = e < E ? gy[(size_t)row * E + e] : 0.0f;
            xh[i] = e < E ? xhat[(size_t)row * E + e] : 0.0f;
            pgw[i] = lq_fma(g[i], xh[i], pgw[i]);
            pgb[i] += g[i];
            g[i] = e < E ? g[i] * w[e] : 0.0f;
            s1 += g[i];
            s2 = lq_fma(g[i], xh[i], s2);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        const float m1 = s1 / (float)E, m2 = s2 / (float)E, rs = rstd[row];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            if (e < E) gx[(size_t)row * E + e] = rs * (g[i] - m1 - xh[i] * m2);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_gw[wv][lane + 64 * i] = pgw[i]; s_gb[wv][lane + 64 * i] = pgb[i]; }
    __syncthreads();
    const int e = threadIdx.x;
    if (e < E) {
        atomicAdd(&gw[e], ((s_gw[0][e] + s_gw[1][e]) + s_gw[2][e]) + s_gw[3][e]);
        atomicAdd(&gb[e], ((s_gb[0][e] + s_gb[1][e]) + s_gb[2][e]) + s_gb[3][e]);
    }
}

extern "C" int lipvq_layernorm_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* w, float* gx, float* gw,
                                       float* gb, int64_t N, int E, void* stream) {
    if (!gy || !xhat || !rstd || !w || !gx || !gw || !gb) return fail(LIPVQ_EINVAL, "layernorm_bwd: null pointer");
    if (N <= 0 || E <= 0 || E > 256) return fail(LIPVQ_EUNSUPPORTED, "layernorm_bwd: N=%lld E=%d (E <= 256)", (long long)N, E);
    const int rpb = N >= 65536 ? 256 : (N >= 1024 ? 32 : 4);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((N + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, gy, xhat, rstd, w,
                       gx, gw, gb, N, E, rpb);
    return check_launch("layernorm_bwd");
}
