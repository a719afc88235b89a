// probe_pipes.hip -- how do the fp32 MFMA (v_mfma_f32_32x32x2_f32), the fp16 MFMA (v_mfma_f32_32x32x16_f16) and plain
// fp32 VALU work share one SIMD on gfx950?  Decides how tokenize_kernel's encoder phase can be scheduled.
//   build: hipcc --offload-arch=gfx950 -O3 -o probe_pipes probe_pipes.hip ; run: ./probe_pipes
// Every test: 256 workgroups (one per CU), W waves per SIMD, each wave runs ITER iterations of a body and reports
// s_memtime cycles per iteration (median over waves) and the wall time.  Operands are random (DVFS).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

// role: 0 = fp32 MFMA chain, 1 = fp16 MFMA chain, 2 = VALU only, 3 = idle (exit at once)
// NV = independent v_fma_f32 per MFMA (roles 0/1), or per iteration (role 2)
template <int ROLE_A, int ROLE_B, int NV>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc,
                                             int waves_per_simd) {
    const int tid = threadIdx.x, wave = tid >> 6;
    // waves 0-3 land on the four SIMDs, waves 4-7 are their partners: role A for the first four, role B for the rest
    const int role = (wave < 4) ? ROLE_A : ROLE_B;
    float a = in[tid], b = in[tid + 512];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = in[(tid + r * 7) & 1023];
    f16x8 ha, hb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)in[(tid + j) & 1023]; hb[j] = (_Float16)in[(tid * 3 + j) & 1023]; }
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = in[(tid + 31 * j) & 1023];
    const float c0 = in[5], c1 = in[6];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], c0, c1);
            }
        }
    } else if (role == 1) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], c0, c1);
            }
        }
    } else if (role == 2) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c0, c1);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = (role == 3 || (wave >= 4 && waves_per_simd == 1)) ? -1 : (t1 - t0);
}

template <int RA, int RB, int NV>
void run(const char* name, int threads, const float* in, float* out, long long* cyc) {
    const int W = threads / 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<RA, RB, NV>), dim3(256), dim3(threads), 0, 0, in, out, cyc, W);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < threads / 64; ++w) {
            long long c = h[b * 8 + w];
            if (c < 0) continue;
            (w < 4 ? ca : cb).push_back((double)c / (ITER * 4));
        }
    auto med = [](std::vector<double>& x) { if (x.empty()) return 0.0; std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    // s_memtime ticks at 100 MHz on gfx9 (constant), wall from events: report both; cycles = wall * clock unknown, so give ns per group
    printf("%-58s W=%d  wavesA: %8.2f ticks/grp  wavesB: %8.2f ticks/grp   wall %.3f ms  (%.1f ns per group per wave)\n", name, W, med(ca),
           med(cb), ms, ms * 1e6 / (ITER * 4));
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    printf("group = 1 MFMA + NV v_fma (MFMA roles) or 16 v_fma (VALU role); ns per group per wave is the number to read\n");
    // 1. bare chains
    run<0, 3, 0>("fp32 MFMA chain alone, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 0, 0>("fp32 MFMA chain, 2 waves/SIMD", 512, in, out, cyc);
    run<1, 3, 0>("fp16 MFMA chain alone, 1 wave/SIMD", 256, in, out, cyc);
    run<1, 1, 0>("fp16 MFMA chain, 2 waves/SIMD", 512, in, out, cyc);
    run<2, 3, 0>("VALU only (16 fma/grp), 1 wave/SIMD", 256, in, out, cyc);
    run<2, 2, 0>("VALU only (16 fma/grp), 2 waves/SIMD", 512, in, out, cyc);
    // 2. fp32 MFMA + VALU in the same wave
    run<0, 3, 4>("fp32 MFMA + 4 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 3, 8>("fp32 MFMA + 8 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 3, 12>("fp32 MFMA + 12 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 3, 16>("fp32 MFMA + 16 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 3, 24>("fp32 MFMA + 24 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<0, 0, 4>("fp32 MFMA + 4 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<0, 0, 8>("fp32 MFMA + 8 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<0, 0, 12>("fp32 MFMA + 12 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<0, 0, 16>("fp32 MFMA + 16 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<0, 0, 24>("fp32 MFMA + 24 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<0, 0, 32>("fp32 MFMA + 32 fma, 2 waves/SIMD", 512, in, out, cyc);
    // 3. fp16 MFMA + VALU in the same wave
    run<1, 3, 4>("fp16 MFMA + 4 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<1, 3, 8>("fp16 MFMA + 8 fma, 1 wave/SIMD", 256, in, out, cyc);
    run<1, 1, 4>("fp16 MFMA + 4 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<1, 1, 8>("fp16 MFMA + 8 fma, 2 waves/SIMD", 512, in, out, cyc);
    run<1, 1, 16>("fp16 MFMA + 16 fma, 2 waves/SIMD", 512, in, out, cyc);
    // 4. different roles on the two waves of a SIMD
    run<0, 2, 0>("A: fp32 MFMA chain | B: VALU only (16 fma/grp)", 512, in, out, cyc);
    run<1, 2, 0>("A: fp16 MFMA chain | B: VALU only (16 fma/grp)", 512, in, out, cyc);
    run<0, 1, 0>("A: fp32 MFMA chain | B: fp16 MFMA chain", 512, in, out, cyc);
    run<0, 1, 8>("A: fp32 MFMA + 8 fma | B: fp16 MFMA + 8 fma", 512, in, out, cyc);
    return 0;
}
