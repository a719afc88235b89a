// lipvq_bin.hip -- the sibling tokenizer behind the reference's `bin_enabled` switch:
// AdaptiveBinActionEmbedding (reference robomimic/models/bin_action/backbone.py = "bin"; selected at
// robomimic/models/obs_nets.py:1214-1217, called like the LipVQ tokenizer at obs_nets.py:1330-1333):
//     running min/max per action dimension (bin:37-40) -> linspace boundaries (bin:42-53) -> bucketize + clamp
//     (bin:55-66) -> one nn.Embedding per dimension, concatenated (bin:77-83) -> Linear . GELU . Linear . GELU (bin:26-31)
//
// MI355X design.  The first Linear acts on a concatenation of A embedding rows, so
//     Linear1(cat_i emb_i[bin_i]) = b1 + sum_i P[i][bin_i],     P[i][bin] = W1[:, 64 i : 64 i + 64] . emb_i[bin]
// and the [N][64 A] x [64 A][32 A] GEMM (590 kflop per action at A = 12) collapses into A gathered adds per hidden
// unit from a [A][num_bins][H] table (360 KiB at A = 12) rebuilt only when a parameter changes (lipvq_linear_f32 per
// dimension).  bin_hidden_kernel keeps a column slice of that table in LDS and streams the rows through it;
// the second Linear is the canonical MFMA Linear with the GELU in its epilogue (lipvq_linear_act_f32).
// Bin indices are integer work: bit-exact against the oracle / the reference (torch's linspace rounding and
// lower-bound search are restated in lipvq_math.h).  Canonical layer 1: acc = b1[j]; acc = acc + P[i][bin_i][j], i ascending.
// ABI: include/lipvq.h.
#include "lipvq_common.h"

// ---------------------------------------------------------------------------------------------------
// bin:37-40  running_min = minimum(running_min, actions.min(0)), running_max likewise (in place, atomics on the
// monotone integer image of the floats)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    int* ia = reinterpret_cast<int*>(addr);
    int old = __atomic_load_n(ia, __ATOMIC_RELAXED);
    while (v < __int_as_float(old)) {
        if (__atomic_compare_exchange_n(ia, &old, __float_as_int(v), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
    }
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    int* ia = reinterpret_cast<int*>(addr);
    int old = __atomic_load_n(ia, __ATOMIC_RELAXED);
    while (v > __int_as_float(old)) {
        if (__atomic_compare_exchange_n(ia, &old, __float_as_int(v), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
    }
}

__global__ __launch_bounds__(256) void bin_minmax_kernel(const float* __restrict__ x, float* rmin, float* rmax,
                                                         int64_t N, int A) {
    __shared__ float smin[256], smax[256];
    const int64_t total = N * A;
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    const int64_t stride = (nthreads / A) * A;                // a multiple of A: every thread stays in one column
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    if (gtid < stride)
        for (int64_t i = gtid; i < total; i += stride) {
            const float v = x[i];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    // thread c < A folds the block's threads of its column (same gtid % A)
    if (threadIdx.x < A) {
        const int c0 = (int)(((int64_t)blockIdx.x * 256) % A);     // column of thread 0
        const int first = (threadIdx.x - c0 + A) % A;               // first thread of this block in column threadIdx.x
        float l = INFINITY, h = -INFINITY;
        for (int t = first; t < 256; t += A) {
            l = smin[t] < l ? smin[t] : l;
            h = smax[t] > h ? smax[t] : h;
        }
        if (l < INFINITY) atomic_min_f32(rmin + threadIdx.x, l);
        if (h > -INFINITY) atomic_max_f32(rmax + threadIdx.x, h);
    }
}

// ---------------------------------------------------------------------------------------------------
// bin:42-66  bins[i][n] = clamp(bucketize(x[n][i], linspace(min_i, max_i, nb + 1)) - 1, 0, nb - 1)   (int64, [A][N])
// ---------------------------------------------------------------------------------------------------
#define BIN_MAX_BOUNDS 4096          // floats of LDS for the boundaries: A * (nb + 1) <= 4096

__global__ __launch_bounds__(256) void bin_discretize_kernel(const float* __restrict__ x, const float* __restrict__ rmin,
                                                             const float* __restrict__ rmax, int64_t* __restrict__ bins,
                                                             int64_t N, int A, int nb) {
    __shared__ float bounds[BIN_MAX_BOUNDS];
    for (int f = threadIdx.x; f < A * (nb + 1); f += 256) {
        const int i = f / (nb + 1), j = f - i * (nb + 1);
        bounds[f] = lq_linspace(rmin[i], rmax[i], nb + 1, j);
    }
    __syncthreads();
    const int64_t total = N * A;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / A;
        const int i = (int)(e - n * A);
        bins[(size_t)i * N + n] = lq_bin_index(x[e], bounds + i * (nb + 1), nb);
    }
}

__global__ void bin_boundaries_kernel(const float* __restrict__ rmin, const float* __restrict__ rmax,
                                      float* __restrict__ out, int A, int nb) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A * (nb + 1)) return;
    const int i = f / (nb + 1), j = f - i * (nb + 1);
    out[f] = lq_linspace(rmin[i], rmax[i], nb + 1, j);
}

// ---------------------------------------------------------------------------------------------------
// bin:77-86 (embeddings + cat + first Linear + GELU) through the P table:
//     pre1[n][j] = b1[j] + sum_i P[i][bins[i][n]][j],   h[n][j] = gelu(pre1[n][j])
// grid.y = column slices of width SW (the slice of P lives in LDS: A * nb * SW floats); one wave per row,
// rows grid-strided over grid.x workgroups of 16 waves.
// ---------------------------------------------------------------------------------------------------
struct BinHiddenArgs {
    const int64_t* bins;
    const float* P;
    const float* b1;
    float* h;
    float* pre1;
    int64_t N;
    int A, nb, H, SW;
};

#define BINH_ROWS 4          // rows per wave step (independent gather chains in flight)

__global__ __launch_bounds__(1024) void bin_hidden_kernel(const BinHiddenArgs a) {
    extern __shared__ float pl[];                                // [A * nb][SW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.y * a.SW;
    const int sw = (a.H - s0 < a.SW) ? (a.H - s0) : a.SW;       // width of this slice
    const int rowsP = a.A * a.nb;
    for (int f = tid; f < rowsP * a.SW; f += 1024) {
        const int r = f / a.SW, c = f - r * a.SW;
        pl[f] = (c < sw) ? a.P[(size_t)r * a.H + s0 + c] : 0.0f;
    }
    __syncthreads();
    const int64_t nw = (int64_t)gridDim.x * 16;
    for (int64_t n0 = ((int64_t)blockIdx.x * 16 + wave) * BINH_ROWS; n0 < a.N; n0 += nw * BINH_ROWS) {
        // lane i < A holds the LDS row offset (i * nb + bin) * SW of dimension i for each of the step's rows
        int off[BINH_ROWS];
#pragma unroll
        for (int r = 0; r < BINH_ROWS; ++r) {
            off[r] = 0;
            if (lane < a.A && n0 + r < a.N) off[r] = (lane * a.nb + (int)a.bins[(size_t)lane * a.N + n0 + r]) * a.SW;
        }
        for (int c = lane; c < sw; c += 64) {
            const float b1 = a.b1[s0 + c];
            float acc[BINH_ROWS];
#pragma unroll
            for (int r = 0; r < BINH_ROWS; ++r) acc[r] = b1;
            for (int i = 0; i < a.A; ++i) {
#pragma unroll
                for (int r = 0; r < BINH_ROWS; ++r) acc[r] = acc[r] + pl[__builtin_amdgcn_readlane(off[r], i) + c];
            }
#pragma unroll
            for (int r = 0; r < BINH_ROWS; ++r) {
                if (n0 + r < a.N) {
                    if (a.pre1) a.pre1[(size_t)(n0 + r) * a.H + s0 + c] = acc[r];
                    a.h[(size_t)(n0 + r) * a.H + s0 + c] = lq_gelu(acc[r]);
                }
            }
        }
    }
}

// out = g * act'(pre)   (elementwise; the backward of the two GELUs of bin:28,30)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pre,
                                                      float* __restrict__ out, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = g[i] * lq_act_grad(pre[i], act);
}

static int stream_grid(int64_t work_items, int per_block, int cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int lipvq_bin_minmax_f32(const float* actions, float* running_min, float* running_max, int64_t N, int A, void* stream) {
    if (!actions || !running_min || !running_max) return fail(LIPVQ_EINVAL, "lipvq_bin_minmax_f32: null pointer");
    if (N < 0 || A <= 0 || A > 256) return fail(LIPVQ_EINVAL, "lipvq_bin_minmax_f32: bad sizes N=%lld A=%d", (long long)N, A);
    if (N == 0) return LIPVQ_OK;
    const dim3 grid(stream_grid(N * A, 256 * 8, 1024));
    hipLaunchKernelGGL(bin_minmax_kernel, grid, dim3(256), 0, (hipStream_t)stream, actions, running_min, running_max, N, A);
    return check_launch("bin_minmax_kernel");
}

int lipvq_bin_discretize_f32(const float* actions, const float* running_min, const float* running_max, int64_t* bins,
                             int64_t N, int A, int num_bins, void* stream) {
    if (!actions || !running_min || !running_max || !bins) return fail(LIPVQ_EINVAL, "lipvq_bin_discretize_f32: null pointer");
    if (N < 0 || A <= 0 || num_bins < 1) return fail(LIPVQ_EINVAL, "lipvq_bin_discretize_f32: bad sizes");
    if ((int64_t)A * (num_bins + 1) > BIN_MAX_BOUNDS)
        return fail(LIPVQ_EUNSUPPORTED, "lipvq_bin_discretize_f32: A * (num_bins + 1) = %lld exceeds %d",
                    (long long)A * (num_bins + 1), BIN_MAX_BOUNDS);
    if (N == 0) return LIPVQ_OK;
    const dim3 grid(stream_grid(N * A, 256 * 4, 2048));
    hipLaunchKernelGGL(bin_discretize_kernel, grid, dim3(256), 0, (hipStream_t)stream, actions, running_min, running_max,
                       bins, N, A, num_bins);
    return check_launch("bin_discretize_kernel");
}

int lipvq_bin_boundaries_f32(const float* running_min, const float* running_max, float* boundaries, int A, int num_bins,
                             void* stream) {
    if (!running_min || !running_max || !boundaries) return fail(LIPVQ_EINVAL, "lipvq_bin_boundaries_f32: null pointer");
    if (A <= 0 || num_bins < 1) return fail(LIPVQ_EINVAL, "lipvq_bin_boundaries_f32: bad sizes");
    const int total = A * (num_bins + 1);
    hipLaunchKernelGGL(bin_boundaries_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_min,
                       running_max, boundaries, A, num_bins);
    return check_launch("bin_boundaries_kernel");
}

int lipvq_bin_hidden_f32(const int64_t* bins, const float* P, const float* b1, float* h, float* pre1, int64_t N, int A,
                         int num_bins, int H, void* stream) {
    if (!bins || !P || !b1 || !h) return fail(LIPVQ_EINVAL, "lipvq_bin_hidden_f32: null pointer");
    if (N < 0 || A <= 0 || num_bins < 1 || H <= 0) return fail(LIPVQ_EINVAL, "lipvq_bin_hidden_f32: bad sizes");
    if (N == 0) return LIPVQ_OK;
    // widest column slice (a multiple of 64) whose part of P lets two workgroups share a CU's LDS
    const int64_t rowsP = (int64_t)A * num_bins;
    int64_t SW = (80 * 1024 / 4) / rowsP / 64 * 64;       // <= 80 KiB: two 16-wave workgroups per CU
    if (SW < 64) SW = (150 * 1024 / 4) / rowsP / 64 * 64;   // large A * num_bins: one workgroup per CU
    if (SW < 64) return fail(LIPVQ_EUNSUPPORTED, "lipvq_bin_hidden_f32: A * num_bins = %lld is too large for the LDS table slice",
                             (long long)rowsP);
    const int Hpad = (H + 63) / 64 * 64;
    if (SW > Hpad) SW = Hpad;
    const int slices = (int)((H + SW - 1) / SW);
    const size_t lds = (size_t)rowsP * SW * sizeof(float);
    static LqLdsReserve reserved;               // per-device, thread-safe (lipvq_common.h)
    if (int rc = lipvq_reserve_lds(reserved, (const void*)bin_hidden_kernel, 160 * 1024 - 64, "lipvq_bin_hidden_f32")) return rc;
    int gx = 512 / slices;                               // two 16-wave workgroups per CU
    if (gx < 1) gx = 1;
    const int64_t need = (N + 16 * BINH_ROWS - 1) / (16 * BINH_ROWS);
    if (gx > need) gx = (int)need;
    BinHiddenArgs a{bins, P, b1, h, pre1, N, A, num_bins, H, (int)SW};
    hipLaunchKernelGGL(bin_hidden_kernel, dim3(gx, slices), dim3(1024), lds, (hipStream_t)stream, a);
    return check_launch("bin_hidden_kernel");
}

int lipvq_act_bwd_f32(const float* g, const float* pre, float* out, int64_t n, int act, void* stream) {
    if (!g || !pre || !out) return fail(LIPVQ_EINVAL, "lipvq_act_bwd_f32: null pointer");
    if (n < 0 || act < LIPVQ_ACT_NONE || act > LIPVQ_ACT_RELU) return fail(LIPVQ_EINVAL, "lipvq_act_bwd_f32: bad arguments");
    if (n == 0) return LIPVQ_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, g, pre, out, n, act);
    return check_launch("act_bwd_kernel");
}

}  // extern "C"
