// probe_pipes2.hip -- asm-pinned version of probe_pipes.hip: exact instruction order, no compiler rescheduling.
// Question: how many independent v_fma_f32 fit "under" one MFMA of the same wave (dependent accumulator chain), for the
// fp32 MFMA (v_mfma_f32_32x32x2_f32, 16 passes) and the fp16 MFMA (v_mfma_f32_32x32x16_f16, 8 passes), at 1 and 2 waves
// per SIMD, and what a VALU-only partner wave gets beside an MFMA-only wave.
//   build: hipcc --offload-arch=gfx950 -O3 -o probe_pipes2 probe_pipes2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 4000;

#define FMA1 "v_fma_f32 %[v0], %[v0], %[c0], %[c1]\n\t"
#define FMA2 FMA1 "v_fma_f32 %[v1], %[v1], %[c0], %[c1]\n\t"
#define FMA4 FMA2 "v_fma_f32 %[v2], %[v2], %[c0], %[c1]\n\tv_fma_f32 %[v3], %[v3], %[c0], %[c1]\n\t"
#define FMA8 FMA4 "v_fma_f32 %[v4], %[v4], %[c0], %[c1]\n\tv_fma_f32 %[v5], %[v5], %[c0], %[c1]\n\tv_fma_f32 %[v6], %[v6], %[c0], %[c1]\n\tv_fma_f32 %[v7], %[v7], %[c0], %[c1]\n\t"
#define MF32 "v_mfma_f32_32x32x2_f32 %[acc], %[a], %[b], %[acc]\n\t"
#define MF16 "v_mfma_f32_32x32x16_f16 %[acc], %[ha], %[hb], %[acc]\n\t"
#define MF32B "v_mfma_f32_32x32x2_f32 %[acc2], %[a], %[b], %[acc2]\n\t"
#define MS0 "v_mfma_f32_16x16x4_f32 %[q0], %[a], %[b], %[q0]\n\t"
#define MS1 "v_mfma_f32_16x16x4_f32 %[q1], %[a], %[b], %[q1]\n\t"
#define MS2 "v_mfma_f32_16x16x4_f32 %[q2], %[a], %[b], %[q2]\n\t"
#define MS3 "v_mfma_f32_16x16x4_f32 %[q3], %[a], %[b], %[q3]\n\t"

#define OPS : [acc] "+v"(acc), [acc2] "+v"(acc2), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [v4] "+v"(v4), [v5] "+v"(v5), [v6] "+v"(v6), [v7] "+v"(v7) \
            : [a] "v"(a), [b] "v"(b), [ha] "v"(ha), [hb] "v"(hb), [c0] "v"(c0), [c1] "v"(c1)

// BODY codes: 0 = MF32 only; 1 = MF32+4; 2 = MF32+8; 3 = MF32+12; 4 = MF32+16; 10..14 the same with MF16;
// 20 = 16 fma only; 30 = two independent fp32 chains alternating + 8 fma each; 99 = idle
template <int BODY>
__device__ __forceinline__ void body(f32x16& acc, f32x16& acc2, float __attribute__((ext_vector_type(4)))& q0, float __attribute__((ext_vector_type(4)))& q1,
                                     float __attribute__((ext_vector_type(4)))& q2, float __attribute__((ext_vector_type(4)))& q3, float& v0, float& v1, float& v2, float& v3, float& v4, float& v5,
                                     float& v6, float& v7, float a, float b, f16x8 ha, f16x8 hb, float c0, float c1) {
    for (int it = 0; it < ITER; ++it) {
        if constexpr (BODY == 0) asm volatile(MF32 MF32 MF32 MF32 OPS);
        if constexpr (BODY == 1) asm volatile(MF32 FMA4 MF32 FMA4 MF32 FMA4 MF32 FMA4 OPS);
        if constexpr (BODY == 2) asm volatile(MF32 FMA8 MF32 FMA8 MF32 FMA8 MF32 FMA8 OPS);
        if constexpr (BODY == 3) asm volatile(MF32 FMA8 FMA4 MF32 FMA8 FMA4 MF32 FMA8 FMA4 MF32 FMA8 FMA4 OPS);
        if constexpr (BODY == 4) asm volatile(MF32 FMA8 FMA8 MF32 FMA8 FMA8 MF32 FMA8 FMA8 MF32 FMA8 FMA8 OPS);
        if constexpr (BODY == 10) asm volatile(MF16 MF16 MF16 MF16 OPS);
        if constexpr (BODY == 11) asm volatile(MF16 FMA2 MF16 FMA2 MF16 FMA2 MF16 FMA2 OPS);
        if constexpr (BODY == 12) asm volatile(MF16 FMA4 MF16 FMA4 MF16 FMA4 MF16 FMA4 OPS);
        if constexpr (BODY == 13) asm volatile(MF16 FMA4 FMA2 MF16 FMA4 FMA2 MF16 FMA4 FMA2 MF16 FMA4 FMA2 OPS);
        if constexpr (BODY == 14) asm volatile(MF16 FMA8 MF16 FMA8 MF16 FMA8 MF16 FMA8 OPS);
        if constexpr (BODY == 20) asm volatile(FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 OPS);      // 64 fma = "4 groups of 16"
        if constexpr (BODY == 30) asm volatile(MF32 FMA8 MF32B FMA8 MF32 FMA8 MF32B FMA8 OPS);
        if constexpr (BODY == 31) asm volatile(MF32 MF32B MF32 MF32B OPS);
        // fp32 16x16x4 (8 passes): group = ONE such MFMA (half the flops of a 32x32x2) + n fma
        if constexpr (BODY == 40) asm volatile(MS0 MS0 MS0 MS0 OPS);                                   // one dependent chain
        if constexpr (BODY == 41) asm volatile(MS0 MS1 MS0 MS1 OPS);                                   // two chains
        if constexpr (BODY == 42) asm volatile(MS0 MS1 MS2 MS3 OPS);                                   // four chains
        if constexpr (BODY == 43) asm volatile(MS0 FMA2 MS1 FMA2 MS2 FMA2 MS3 FMA2 OPS);
        if constexpr (BODY == 44) asm volatile(MS0 FMA4 MS1 FMA4 MS2 FMA4 MS3 FMA4 OPS);
        if constexpr (BODY == 45) asm volatile(MS0 FMA4 FMA2 MS1 FMA4 FMA2 MS2 FMA4 FMA2 MS3 FMA4 FMA2 OPS);
        if constexpr (BODY == 46) asm volatile(MS0 FMA8 MS1 FMA8 MS2 FMA8 MS3 FMA8 OPS);
    }
}

template <int BA, int BB>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc) {
    const int tid = threadIdx.x, wave = tid >> 6;
    float a = in[tid], b = in[tid + 512];
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = in[(tid + r * 7) & 1023]; acc2[r] = in[(tid + r * 11) & 1023]; }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 q0, q1, q2, q3;
#pragma unroll
    for (int r = 0; r < 4; ++r) { q0[r] = in[(tid + r) & 1023]; q1[r] = in[(tid + r + 9) & 1023]; q2[r] = in[(tid + r + 19) & 1023]; q3[r] = in[(tid + r + 29) & 1023]; }
    f16x8 ha, hb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)in[(tid + j) & 1023]; hb[j] = (_Float16)in[(tid * 3 + j) & 1023]; }
    float v0 = in[tid & 1023], v1 = in[(tid + 1) & 1023], v2 = in[(tid + 2) & 1023], v3 = in[(tid + 3) & 1023];
    float v4 = in[(tid + 4) & 1023], v5 = in[(tid + 5) & 1023], v6 = in[(tid + 6) & 1023], v7 = in[(tid + 7) & 1023];
    const float c0 = in[5] * 0.5f, c1 = in[6];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) body<BA>(acc, acc2, q0, q1, q2, q3, v0, v1, v2, v3, v4, v5, v6, v7, a, b, ha, hb, c0, c1);
    else body<BB>(acc, acc2, q0, q1, q2, q3, v0, v1, v2, v3, v4, v5, v6, v7, a, b, ha, hb, c0, c1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    for (int r = 0; r < 4; ++r) s += q0[r] + q1[r] + q2[r] + q3[r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int BA, int BB>
void run(const char* name, int threads, const float* in, float* out, long long* cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<BA, BB>), dim3(256), dim3(threads), 0, 0, in, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < threads / 64; ++w) (w < 4 ? ca : cb).push_back((double)h[b * 8 + w] / (ITER * 4));
    auto med = [](std::vector<double>& x) { if (x.empty()) return 0.0; std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("%-62s A %7.1f cyc/grp   B %7.1f cyc/grp   wall %.3f ms\n", name, med(ca), med(cb), ms);
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)(rand() % 20001) / 20000.0f - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    printf("grp = one MFMA + its n v_fma (MFMA bodies) or 16 v_fma (VALU body); cycles of wave lifetime per group\n");
    run<0, 99>("1w: fp32 MFMA", 256, in, out, cyc);
    run<1, 99>("1w: fp32 MFMA + 4 fma", 256, in, out, cyc);
    run<2, 99>("1w: fp32 MFMA + 8 fma", 256, in, out, cyc);
    run<3, 99>("1w: fp32 MFMA + 12 fma", 256, in, out, cyc);
    run<4, 99>("1w: fp32 MFMA + 16 fma", 256, in, out, cyc);
    run<31, 99>("1w: two independent fp32 chains alternating", 256, in, out, cyc);
    run<30, 99>("1w: two independent fp32 chains alternating + 8 fma each", 256, in, out, cyc);
    run<40, 99>("1w: fp32 16x16x4, one chain", 256, in, out, cyc);
    run<41, 99>("1w: fp32 16x16x4, two chains", 256, in, out, cyc);
    run<42, 99>("1w: fp32 16x16x4, four chains", 256, in, out, cyc);
    run<43, 99>("1w: fp32 16x16x4 x4 chains + 2 fma", 256, in, out, cyc);
    run<44, 99>("1w: fp32 16x16x4 x4 chains + 4 fma", 256, in, out, cyc);
    run<45, 99>("1w: fp32 16x16x4 x4 chains + 6 fma", 256, in, out, cyc);
    run<46, 99>("1w: fp32 16x16x4 x4 chains + 8 fma", 256, in, out, cyc);
    run<42, 42>("2w: fp32 16x16x4 x4 chains | same", 512, in, out, cyc);
    run<43, 43>("2w: fp32 16x16x4 x4 chains + 2 fma | same", 512, in, out, cyc);
    run<44, 44>("2w: fp32 16x16x4 x4 chains + 4 fma | same", 512, in, out, cyc);
    run<46, 46>("2w: fp32 16x16x4 x4 chains + 8 fma | same", 512, in, out, cyc);
    run<42, 20>("2w: A fp32 16x16x4 x4 chains | B 16 fma", 512, in, out, cyc);
    run<10, 99>("1w: fp16 MFMA", 256, in, out, cyc);
    run<11, 99>("1w: fp16 MFMA + 2 fma", 256, in, out, cyc);
    run<12, 99>("1w: fp16 MFMA + 4 fma", 256, in, out, cyc);
    run<13, 99>("1w: fp16 MFMA + 6 fma", 256, in, out, cyc);
    run<14, 99>("1w: fp16 MFMA + 8 fma", 256, in, out, cyc);
    run<20, 99>("1w: 16 fma", 256, in, out, cyc);
    run<0, 0>("2w: fp32 MFMA | same", 512, in, out, cyc);
    run<1, 1>("2w: fp32 MFMA + 4 fma | same", 512, in, out, cyc);
    run<2, 2>("2w: fp32 MFMA + 8 fma | same", 512, in, out, cyc);
    run<4, 4>("2w: fp32 MFMA + 16 fma | same", 512, in, out, cyc);
    run<10, 10>("2w: fp16 MFMA | same", 512, in, out, cyc);
    run<12, 12>("2w: fp16 MFMA + 4 fma | same", 512, in, out, cyc);
    run<14, 14>("2w: fp16 MFMA + 8 fma | same", 512, in, out, cyc);
    run<20, 20>("2w: 16 fma | same", 512, in, out, cyc);
    run<0, 20>("2w: A fp32 MFMA | B 16 fma", 512, in, out, cyc);
    run<10, 20>("2w: A fp16 MFMA | B 16 fma", 512, in, out, cyc);
    run<0, 10>("2w: A fp32 MFMA | B fp16 MFMA", 512, in, out, cyc);
    run<2, 14>("2w: A fp32 MFMA + 8 fma | B fp16 MFMA + 8 fma", 512, in, out, cyc);
    return 0;
}
