// lipvq_scatter.hip -- the codebook gradient of a LARGE batch (gC[idx[n]] += g[n], the index_add_ behind
// backbone_lfqvae_v5.py:47 / backbone.py:66) without floating-point atomics: rows are counting-sorted by code (stable),
// every code's rows are cut into segments of at most SCS_SEG rows, one wave sums a segment in row order, and a last
// kernel adds a code's segment sums in segment order.  The sum order is a function of (idx, N) alone: bit-identical
// results run after run (lipvq_scatter_add_det_f32 remains the strictly sequential order torch's deterministic
// index_add_ has).  Why: the LDS-privatised atomic kernel (lipvq_bwd.hip) is bound by ds_add_f32 itself, ~3 cycles per
// lane-add and CU -- 185 us at N = 524 288, D = 64 and 563 us at D = 208, whatever the launch shape.
//
//   sc_count_kernel    per block of 512 x waves rows: histogram of its codes (LDS integer atomics)   -> cnt[b][k]
//   sc_scan_blocks_kernel  cnt[b][k] -> rows of code k in earlier blocks, tot[k]
//   sc_scan_codes_kernel   offsets[k] (rows before code k), itemoff[k] (segments before code k)
//   sc_place_kernel    stable placement: perm[offsets[k] + rank of the row among the rows of code k] = row
//   sc_sum_kernel      one wave per (segment, 64-column slice): sequential fp32 sum of its rows, lane = column
//   sc_combine_kernel  gC[k] += the code's segment sums, in order (codes of one segment were written directly)
// ABI: include/lipvq.h.
#include "lipvq_common.h"

#define SCS_GROUPS 8               // groups of 64 consecutive rows per wave of a count / place block
#define SCS_SEG 256                // rows per segment
#define SCS_MAX_K 16384

// waves per count / place block: their K-int histograms share the LDS (128 KB at most); rows per block = 512 x waves
static inline int scs_waves(int K) { return K <= 2048 ? 16 : (K <= 8192 ? 4 : 2); }
static inline int scs_block_rows(int K) { return scs_waves(K) * 64 * SCS_GROUPS; }
static inline int scs_blocks(int64_t N, int K) { return (int)((N + scs_block_rows(K) - 1) / scs_block_rows(K)); }
static inline int64_t scs_max_items(int64_t N, int K) { return N / SCS_SEG + K; }

struct ScsLayout {
    size_t cnt, tot, offsets, itemoff, perm, partial, total;      // byte offsets
};
static inline ScsLayout scs_layout(int64_t N, int K, int D) {
    ScsLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.cnt = take((size_t)scs_blocks(N, K) * K * sizeof(int));
    L.tot = take((size_t)K * sizeof(int));
    L.offsets = take((size_t)(K + 1) * sizeof(int));
    L.itemoff = take((size_t)(K + 1) * sizeof(int));
    L.perm = take((size_t)N * sizeof(int));
    L.partial = take((size_t)scs_max_items(N, K) * D * sizeof(float));
    L.total = o;
    return L;
}

__global__ __launch_bounds__(1024) void sc_count_kernel(const int64_t* __restrict__ idx, int* __restrict__ cnt, int64_t N, int K) {
    extern __shared__ int sc_hist[];                          // [K]
    const int nt = blockDim.x;
    for (int i = threadIdx.x; i < K; i += nt) sc_hist[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * nt * SCS_GROUPS;
#pragma unroll
    for (int j = 0; j < SCS_GROUPS; ++j) {
        const int64_t r = r0 + (int64_t)j * nt + threadIdx.x;
        if (r < N) atomicAdd(&sc_hist[(int)idx[r]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += nt) cnt[(size_t)blockIdx.x * K + i] = sc_hist[i];
}

// 64 codes per workgroup, 16 threads per code: cnt[b][k] becomes the number of rows of code k in blocks before b, tot[k] all of them.
__global__ __launch_bounds__(1024) void sc_scan_blocks_kernel(int* __restrict__ cnt, int* __restrict__ tot, int nb, int K) {
    __shared__ int psum[16][64];
    const int c = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    const bool kv = k < K;
    const int per = (nb + 15) / 16;
    const int b0 = l * per;
    int b1 = b0 + per;
    if (b1 > nb) b1 = nb;
    int sum = 0;
    if (kv) {
        for (int b = b0; b < b1; b += 8) {
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = b + u < b1 ? cnt[(size_t)(b + u) * K + k] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
    }
    psum[l][c] = sum;
    __syncthreads();
    int run = 0;
    for (int q = 0; q < l; ++q) run += psum[q][c];
    if (kv) {
        for (int b = b0; b < b1; b += 8) {
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = b + u < b1 ? cnt[(size_t)(b + u) * K + k] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (b + u < b1) cnt[(size_t)(b + u) * K + k] = run;
                run += t[u];
            }
        }
        if (l == 15) tot[k] = run;                             // (threads of empty ranges carry the prefix through)
    }
}

// One workgroup: offsets / itemoff = exclusive prefix sums over the codes of (rows of k) and (segments of k); C consecutive codes per thread.
__global__ __launch_bounds__(1024) void sc_scan_codes_kernel(const int* __restrict__ tot, int* __restrict__ offsets, int* __restrict__ itemoff,
                                                             int K) {
    __shared__ int wsum[2][16];
    const int C = (K + 1023) / 1024;                           // <= 16
    const int k0 = C * threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int a[16], sa = 0, si = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        a[u] = (u < C && k0 + u < K) ? tot[k0 + u] : 0;
        sa += a[u];
        si += (a[u] + SCS_SEG - 1) / SCS_SEG;
    }
    const int mine_a = sa, mine_i = si;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(sa, o, 64), ti = __shfl_up(si, o, 64);
        if (lane >= o) { sa += ta; si += ti; }
    }
    if (lane == 63) { wsum[0][w] = sa; wsum[1][w] = si; }
    __syncthreads();
    int ea = sa - mine_a, ei = si - mine_i;                    // exclusive prefix at k0
    for (int q = 0; q < w; ++q) { ea += wsum[0][q]; ei += wsum[1][q]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u < C && k0 + u < K) {
            offsets[k0 + u] = ea;
            itemoff[k0 + u] = ei;
            ea += a[u];
            ei += (a[u] + SCS_SEG - 1) / SCS_SEG;
            if (k0 + u == K - 1) { offsets[K] = ea; itemoff[K] = ei; }
        }
    }
}

// lanes of the wave holding the same code as this lane (among `active`): log2(K) ballots
__device__ __forceinline__ unsigned long long scs_same_code(int k, bool active, int bits) {
    unsigned long long m = __ballot(active);
    for (int b = 0; b < bits; ++b) {
        const bool one = (k >> b) & 1;
        const unsigned long long v = __ballot(active && one);
        m &= one ? v : ~v;
    }
    return m;
}

__global__ __launch_bounds__(1024) void sc_place_kernel(const int64_t* __restrict__ idx, const int* __restrict__ cnt,
                                                        const int* __restrict__ offsets, int* __restrict__ perm,
                                                        int64_t N, int K, int bits) {
    extern __shared__ int sc_wh[];                            // [waves][K]: wave histograms, then wave cursors
    const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x, waves = nt >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int G = SCS_GROUPS;                             // groups of 64 consecutive rows per wave
    for (int i = tid; i < waves * K; i += nt) sc_wh[i] = 0;
    __syncthreads();
    const int64_t rw = (int64_t)blockIdx.x * nt * G + (int64_t)wave * (64 * G);
    int kk[G];
    int* mine = sc_wh + (size_t)wave * K;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int64_t r = rw + 64 * j + lane;
        kk[j] = r < N ? (int)idx[r] : -1;
        if (kk[j] >= 0) atomicAdd(&mine[kk[j]], 1);
    }
    __syncthreads();
    // rows of code k before this wave = offsets[k] + earlier blocks + earlier waves of this block
    for (int k = tid; k < K; k += nt) {
        int run = offsets[k] + cnt[(size_t)blockIdx.x * K + k];
        for (int w = 0; w < waves; ++w) {
            const int t = sc_wh[(size_t)w * K + k];
            sc_wh[(size_t)w * K + k] = run;
            run += t;
        }
    }
    __syncthreads();
    // stable: the groups of a wave in row order (LDS operations of one wave execute in order), lanes in lane order
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const bool active = kk[j] >= 0;
        const unsigned long long same = scs_same_code(kk[j], active, bits);
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        const int rank = __popcll(below), n = __popcll(same);
        if (active) {
            const int base = mine[kk[j]];
            perm[base + rank] = (int)(rw + 64 * j + lane);
            if (rank == n - 1) mine[kk[j]] = base + n;        // one writer per code: the last lane holding it
        }
        asm volatile("" ::: "memory");
    }
}

// which code does segment `item` belong to: the last k with itemoff[k] <= item (codes without rows have no segment)
__device__ __forceinline__ int scs_code_of_item(const int* itemoff, int K, int item) {
    int lo = 0, hi = K;                                       // itemoff[lo] <= item < itemoff[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (itemoff[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}

// One wave per (segment, slice of 64 columns): lane = column; rows in sorted (= ascending row) order, one fp32 chain per column.
// SEQUENTIAL: a segment is ALL rows of a code (item = code): the strictly sequential order of lipvq_scatter_add_det_f32.
// FOLD (round 3, lipvq_scatter_add_sorted_vq_f32): the row's value is not read but formed here,
//   (alpha * *gscale) * (tab[k][col] - sub[row][col]) (+ g[row][col] when g is given)
// -- the codebook-loss gradient of the row plus what the decoder sent back, with the roundings of lipvq_scaled_diff_f32 -- so
// that a training step needs neither that stream (97 us at the metric's batch) nor its [N][D] result.
template <bool SEQUENTIAL, bool FOLD = false>
__global__ __launch_bounds__(256) void sc_sum_kernel(const float* __restrict__ g, const int* __restrict__ perm,
                                                     const int* __restrict__ offsets, const int* __restrict__ itemoff,
                                                     float* __restrict__ partial, float* __restrict__ gC, int K, int D, int slices,
                                                     const float* __restrict__ sub = nullptr, const float* __restrict__ tab = nullptr,
                                                     float alpha = 0.0f, const float* __restrict__ gscale = nullptr) {
    extern __shared__ int sc_io[];                            // itemoff [K + 1]: the binary search below runs on LDS
    if (!SEQUENTIAL) {
        for (int i = threadIdx.x; i <= K; i += 256) sc_io[i] = itemoff[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int item = (int)(wid / slices), slice = (int)(wid % slices);
    if (item >= (SEQUENTIAL ? K : sc_io[K])) return;          // (wave-uniform; no barrier below)
    const int k = SEQUENTIAL ? item : scs_code_of_item(sc_io, K, item);
    const int p0 = SEQUENTIAL ? offsets[k] : offsets[k] + (item - sc_io[k]) * SCS_SEG;
    int p1 = SEQUENTIAL ? offsets[k + 1] : p0 + SCS_SEG;
    if (p1 > offsets[k + 1]) p1 = offsets[k + 1];
    if (p1 <= p0) return;                                     // (a code without rows)
    const int col = 64 * slice + lane;
    const bool cv = col < D;
    const float* __restrict__ gc = g + (cv ? col : 0);
    float acc = SEQUENTIAL && cv ? gC[(size_t)k * D + col] : 0.0f;     // sequential: the chain starts from what gC holds, like index_add_
    const float* __restrict__ sc = FOLD ? sub + (cv ? col : 0) : nullptr;
    const float tk = FOLD ? tab[(size_t)k * D + (cv ? col : 0)] : 0.0f;
    const float f = FOLD ? (gscale ? alpha * gscale[0] : alpha) : 0.0f;
    for (int p = p0; p < p1; p += 64) {
        const int mine = p + lane < p1 ? perm[p + lane] : 0;  // (lanes past the end: row 0, loaded and not added)
        const int n = p1 - p < 64 ? p1 - p : 64;
        if (FOLD) {                                           // two gathered operands per row: 32 rows in flight
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (32 * half >= n) break;                    // (wave-uniform)
                float v[32], u[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const size_t row = (size_t)__builtin_amdgcn_readlane(mine, 32 * half + r) * D;
                    u[r] = sc[row];
                    v[r] = g ? gc[row] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    if (32 * half + r < n) {
                        const float d = f * (tk - u[r]);
                        acc += g ? d + v[r] : d;
                    }
            }
        } else if (n > 16) {
            float v[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) v[r] = gc[(size_t)__builtin_amdgcn_readlane(mine, r) * D];
#pragma unroll
            for (int r = 0; r < 64; ++r)
                if (r < n) acc += v[r];                       // (wave-uniform)
        } else {                                              // short segments (large codebooks: a code has few rows)
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = gc[(size_t)__builtin_amdgcn_readlane(mine, r) * D];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < n) acc += v[r];
        }
    }
    if (!cv) return;
    if (SEQUENTIAL) { gC[(size_t)k * D + col] = acc; return; }
    const bool single = sc_io[k + 1] - sc_io[k] == 1;         // the code's only segment: this wave owns gC[k][col]
    if (single) gC[(size_t)k * D + col] += acc;
    else partial[(size_t)item * D + col] = acc;
}

// codes with more than one segment: gC[k] += their segment sums in segment order
__global__ __launch_bounds__(256) void sc_combine_kernel(const float* __restrict__ partial, const int* __restrict__ itemoff,
                                                         float* __restrict__ gC, int K, int D) {
    const int k = blockIdx.x;
    const int i0 = itemoff[k], i1 = itemoff[k + 1];
    if (i1 - i0 < 2) return;
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = gC[(size_t)k * D + c];
        int i = i0;
        for (; i + 16 <= i1; i += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(i + u) * D + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; i < i1; ++i) acc += partial[(size_t)i * D + c];
        gC[(size_t)k * D + c] = acc;
    }
}

extern "C" int lipvq_scatter_add_sorted_supported(int64_t N, int K, int D) {
    return N >= 32768 && N < (1ll << 31) - 65536 && K >= 2 && K <= SCS_MAX_K && D >= 1 &&
           (size_t)scs_blocks(N, K) * K < (1ull << 31);
}

extern "C" size_t lipvq_scatter_add_sorted_workspace_bytes(int64_t N, int K, int D) {
    if (!lipvq_scatter_add_sorted_supported(N, K, D)) return 0;
    return scs_layout(N, K, D).total;
}

static int scatter_sorted_impl(const float* g, const float* sub, const float* tab, float alpha, const float* gscale, const int64_t* idx,
                               float* gC, void* workspace, int64_t N, int K, int D, int sequential, void* stream) {
    const bool fold = sub != nullptr;
    if ((!g && !fold) || (fold && !tab) || !idx || !gC || !workspace) return fail(LIPVQ_EINVAL, "scatter_add_sorted: null pointer");
    if (!lipvq_scatter_add_sorted_supported(N, K, D))
        return fail(LIPVQ_EUNSUPPORTED, "scatter_add_sorted: N=%lld K=%d D=%d outside the supported range (N >= 32768, K <= %d)",
                    (long long)N, K, D, SCS_MAX_K);
    hipStream_t st = (hipStream_t)stream;
    const ScsLayout L = scs_layout(N, K, D);
    char* ws = (char*)workspace;
    int* cnt = (int*)(ws + L.cnt);
    int* tot = (int*)(ws + L.tot);
    int* offsets = (int*)(ws + L.offsets);
    int* itemoff = (int*)(ws + L.itemoff);
    int* perm = (int*)(ws + L.perm);
    float* partial = (float*)(ws + L.partial);
    const int waves = scs_waves(K), nb = scs_blocks(N, K);
    int bits = 1;
    while ((1 << bits) < K) ++bits;
    static LqLdsReserve reserved_place, reserved_count, reserved_sum, reserved_sum_fold;      // per device, thread-safe (lipvq_common.h)
    const size_t lds_place = (size_t)waves * K * sizeof(int), lds_count = (size_t)K * sizeof(int), lds_sum = (size_t)(K + 1) * sizeof(int);
    if (lds_place > 64 * 1024)
        if (int rc = lipvq_reserve_lds(reserved_place, (const void*)sc_place_kernel, lds_place, "scatter_add_sorted")) return rc;
    if (lds_count > 64 * 1024)
        if (int rc = lipvq_reserve_lds(reserved_count, (const void*)sc_count_kernel, lds_count, "scatter_add_sorted")) return rc;
    if (lds_sum > 64 * 1024 && !sequential)
        if (int rc = fold ? lipvq_reserve_lds(reserved_sum_fold, (const void*)sc_sum_kernel<false, true>, lds_sum, "scatter_add_sorted")
                          : lipvq_reserve_lds(reserved_sum, (const void*)sc_sum_kernel<false, false>, lds_sum, "scatter_add_sorted")) return rc;
    hipLaunchKernelGGL(sc_count_kernel, dim3(nb), dim3(64 * waves), lds_count, st, idx, cnt, N, K);
    hipLaunchKernelGGL(sc_scan_blocks_kernel, dim3((K + 63) / 64), dim3(1024), 0, st, cnt, tot, nb, K);
    hipLaunchKernelGGL(sc_scan_codes_kernel, dim3(1), dim3(1024), 0, st, tot, offsets, itemoff, K);
    hipLaunchKernelGGL(sc_place_kernel, dim3(nb), dim3(64 * waves), lds_place, st, idx, cnt, offsets, perm, N, K, bits);
    const int slices = (D + 63) / 64;
    if (sequential) {                                          // one chain per (code, column) over ALL its rows in ascending row order
        const int64_t nwaves = (int64_t)K * slices;
        if (fold)
            hipLaunchKernelGGL((sc_sum_kernel<true, true>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, g, perm, offsets, itemoff,
                               partial, gC, K, D, slices, sub, tab, alpha, gscale);
        else
            hipLaunchKernelGGL((sc_sum_kernel<true, false>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, g, perm, offsets, itemoff,
                               partial, gC, K, D, slices, nullptr, nullptr, 0.0f, nullptr);
        return check_launch("scatter_add_sorted");
    }
    const int64_t nwaves = scs_max_items(N, K) * slices;
    if (fold)
        hipLaunchKernelGGL((sc_sum_kernel<false, true>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), lds_sum, st, g, perm, offsets, itemoff,
                           partial, gC, K, D, slices, sub, tab, alpha, gscale);
    else
        hipLaunchKernelGGL((sc_sum_kernel<false, false>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), lds_sum, st, g, perm, offsets, itemoff,
                           partial, gC, K, D, slices, nullptr, nullptr, 0.0f, nullptr);
    hipLaunchKernelGGL(sc_combine_kernel, dim3(K), dim3(256), 0, st, partial, itemoff, gC, K, D);
    return check_launch("scatter_add_sorted");
}

extern "C" int lipvq_scatter_add_sorted_f32(const float* g, const int64_t* idx, float* gC, void* workspace, int64_t N, int K, int D,
                                            int sequential, void* stream) {
    if (!g) return fail(LIPVQ_EINVAL, "scatter_add_sorted: null pointer");
    return scatter_sorted_impl(g, nullptr, nullptr, 0.0f, nullptr, idx, gC, workspace, N, K, D, sequential, stream);
}

extern "C" int lipvq_scatter_add_sorted_vq_f32(const float* g, const float* ze, const float* table, float alpha, const float* gscale,
                                               const int64_t* idx, float* gC, void* workspace, int64_t N, int K, int D,
                                               int sequential, void* stream) {
    if (!ze || !table) return fail(LIPVQ_EINVAL, "scatter_add_sorted_vq: null pointer");
    return scatter_sorted_impl(g, ze, table, alpha, gscale, idx, gC, workspace, N, K, D, sequential, stream);
}
