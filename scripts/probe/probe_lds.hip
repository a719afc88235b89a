// probe_lds.hip -- LDS read throughput of the screen loop's access pattern: every lane reads 16 bytes at lane*16 of a 1 KiB
// fragment (ds_read_b128, conflict-free), fragments streamed from a 64 KiB window; 4 or 8 waves per CU (1 or 2 per SIMD),
// alone and beside fp16 MFMAs.  Reports bytes per cycle and CU.
//   build: hipcc --offload-arch=gfx950 -O3 -o probe_lds probe_lds.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 2000;

// MODE 0: 8 ds_read_b128 per iteration; 1: 8 reads + 12 fp16 MFMAs (the screen loop's ratio at S = 4: 8 reads per 12 MFMAs);
// 2: 12 MFMAs only; 3: 4 reads + 12 MFMAs (two row tiles per B fragment);
// 5: mode 1 + the screen loop's bookkeeping (per MFMA: 16/12 elements x {fma, and_or, med3, min} on the PREVIOUS tile's accumulator);
// 6: mode 5 + a workgroup barrier every second iteration (= per stage of two tiles); 7: mode 6 + three 1 KiB LDS-DMA copies per stage
template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, const float* __restrict__ gsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = in[i & 1023];
    f16x8 ha;
    for (int j = 0; j < 8; ++j) ha[j] = (_Float16)in[(tid + j) & 1023];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 sum = ha;
    f32x16 prev;
    float m1[16], m2[16];
    for (int r = 0; r < 16; ++r) { prev[r] = in[(tid + r) & 1023]; m1[r] = 1e30f; m2[r] = 1e30f; }
    const float e2 = in[tid & 63], fr = in[(tid + 7) & 63];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        const unsigned char* base = lds + ((it * 8 + wave) & 7) * 8192 + lane * 16;
        f16x8 f[8];
        if constexpr (MODE == 0 || MODE == 1 || MODE >= 5) {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = *reinterpret_cast<const f16x8*>(base + q * 1024);
        }
        if constexpr (MODE >= 6) {
            if ((it & 1) == 0) {
                if constexpr (MODE == 7) {
                    typedef __attribute__((address_space(3))) void* lds_ptr_t;
                    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gsrc + ((it * 3 + j) & 63) * 4096 + wave * 256 + lane * 4),
                                                         (lds_ptr_t)(lds + 65536 - 8192 + wave * 1024), 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { f[q] = *reinterpret_cast<const f16x8*>(base + q * 1024); f[q + 4] = f[q]; }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = sum;
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += f[q];
        } else if constexpr (MODE >= 5) {
            // two named accumulators, as in the kernel: even iterations run their chain into accA and book accB, odd ones the reverse
            auto tile = [&](f32x16& cur, const f32x16& prv) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[r] = 0.0f;
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    __builtin_amdgcn_sched_barrier(0);
                    cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, f[q & 7], cur, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r >= (16 * q) / 12 && r < (16 * (q + 1)) / 12) {
                            if constexpr (MODE == 8) {                     // plain independent fmas instead of the bookkeeping
                                m1[r] = __builtin_fmaf(m1[r], e2, fr); m2[r] = __builtin_fmaf(m2[r], e2, fr);
                                m1[r] = __builtin_fmaf(m1[r], e2, fr); m2[r] = __builtin_fmaf(m2[r], e2, fr);
                            } else {
                                const float v = __builtin_fmaf(e2, fr, prv[r]);
                                float key;
                                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(v), "s"(0xffffff00u), "v"(it));
                                m2[r] = __builtin_amdgcn_fmed3f(key, m1[r], m2[r]);
                                asm("v_min_f32 %0, %1, %2" : "=v"(m1[r]) : "v"(key), "v"(m1[r]));
                            }
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (it & 1) tile(prev, acc); else tile(acc, prev);
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, f[q & 7], acc, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += (float)sum[j];
    for (int j = 0; j < 16; ++j) r += acc[j] + m1[j] + m2[j] + prev[j];
    out[blockIdx.x * 512 + tid] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, const float* in, float* out, long long* cyc, const float* gsrc = nullptr) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 65536, 0, in, out, cyc, gsrc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2048);
    CHECK(hipMemcpy(h.data(), cyc, 16384, hipMemcpyDeviceToHost));
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / ITER);
    std::sort(c.begin(), c.end());
    const double per_it = c[c.size() / 2];
    const int reads = (MODE == 0 || MODE == 1 || MODE >= 5) ? 8 : (MODE == 3 ? 4 : 0);
    printf("%-58s %7.1f cycles per iteration and wave; LDS %6.1f B/clk/CU   wall %.3f ms\n", name, per_it,
           reads * 1024.0 * (threads / 64) / per_it, ms);
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 16384));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)(rand() % 2001) / 2000.0f - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("4 waves/CU: 8 ds_read_b128", 256, in, out, cyc);
    run<0>("8 waves/CU: 8 ds_read_b128", 512, in, out, cyc);
    run<2>("4 waves/CU: 12 fp16 MFMA", 256, in, out, cyc);
    run<2>("8 waves/CU: 12 fp16 MFMA", 512, in, out, cyc);
    run<1>("4 waves/CU: 8 ds_read_b128 + 12 fp16 MFMA", 256, in, out, cyc);
    run<1>("8 waves/CU: 8 ds_read_b128 + 12 fp16 MFMA", 512, in, out, cyc);
    run<3>("4 waves/CU: 4 ds_read_b128 + 12 fp16 MFMA", 256, in, out, cyc);
    run<3>("8 waves/CU: 4 ds_read_b128 + 12 fp16 MFMA", 512, in, out, cyc);
    float* gsrc;
    CHECK(hipMalloc(&gsrc, 64 * 4096 * 4 + 65536));
    CHECK(hipMemset(gsrc, 0, 64 * 4096 * 4 + 65536));
    run<8>("4 waves/CU: 8 reads + 12 MFMA + 64 independent fma", 256, in, out, cyc);
    run<8>("8 waves/CU: 8 reads + 12 MFMA + 64 independent fma", 512, in, out, cyc);
    run<5>("4 waves/CU: 8 reads + 12 MFMA + 64 bookkeeping VALU", 256, in, out, cyc);
    run<5>("8 waves/CU: 8 reads + 12 MFMA + 64 bookkeeping VALU", 512, in, out, cyc);
    run<6>("8 waves/CU: ... + barrier every 2 tiles", 512, in, out, cyc);
    run<7>("8 waves/CU: ... + barrier + 3 LDS-DMA per stage", 512, in, out, cyc, gsrc);
    return 0;
}
