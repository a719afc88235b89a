// probe_track.hip -- does the screen loop's bookkeeping (per element: v_fma, v_and_or, v_med3, v_min -- a dependent chain of four)
// hide under the wave's own fp16 MFMA chain, and what do two such waves on one SIMD get?  Instruction order pinned in asm.
//   build: hipcc --offload-arch=gfx950 -O3 -o probe_track probe_track.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 4000;

#define MF "v_mfma_f32_32x32x16_f16 %[acc], %[ha], %[hb], %[acc]\n\t"
// one element's bookkeeping (E = 0..3): t = fma(e2, fr, pE); t = (t & mask) | tile; m2E = med3(t, m1E, m2E); m1E = min(t, m1E)
#define TRK(E) "v_fma_f32 %[t" #E "], %[e2], %[fr], %[p" #E "]\n\tv_and_or_b32 %[t" #E "], %[t" #E "], %[mask], %[tile]\n\t" \
               "v_med3_f32 %[b" #E "], %[t" #E "], %[a" #E "], %[b" #E "]\n\tv_min_f32 %[a" #E "], %[t" #E "], %[a" #E "]\n\t"
// the same four instructions, INDEPENDENT of each other (no chain through t)
#define IND(E) "v_fma_f32 %[t" #E "], %[e2], %[fr], %[p" #E "]\n\tv_and_or_b32 %[u" #E "], %[p" #E "], %[mask], %[tile]\n\t" \
               "v_med3_f32 %[b" #E "], %[p" #E "], %[fr], %[b" #E "]\n\tv_min_f32 %[a" #E "], %[e2], %[a" #E "]\n\t"
#define OPS : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), \
              [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3) \
            : [ha] "v"(ha), [hb] "v"(hb), [e2] "v"(e2), [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [mask] "s"(0xffffff00u), [tile] "v"(tile)

template <int BODY>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc) {
    const int tid = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = in[(tid + 7 * r) & 1023];
    f16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)in[(tid + j) & 1023]; hb[j] = (_Float16)in[(3 * tid + j) & 1023]; }
    float t0, t1, t2, t3, u0, u1, u2, u3;
    float a0 = in[tid & 63], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 + 4, b1 = a0 + 5, b2 = a0 + 6, b3 = a0 + 7;
    const float e2 = in[5], fr = in[6], p0 = in[(tid + 1) & 63], p1 = in[(tid + 2) & 63], p2 = in[(tid + 3) & 63], p3 = in[(tid + 4) & 63];
    const int tile = tid & 31;
    __syncthreads();
    const long long s0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (BODY == 0) asm volatile(MF MF MF MF OPS);
        if constexpr (BODY == 1) asm volatile(MF TRK(0) MF TRK(1) MF TRK(2) MF TRK(3) OPS);                       // 4 per MFMA, chained
        if constexpr (BODY == 2) asm volatile(MF TRK(0) TRK(1) MF TRK(2) MF TRK(3) TRK(0) MF TRK(1) OPS);         // 6 per MFMA on average (the loop's 5.3 rounded up)
        if constexpr (BODY == 3) asm volatile(MF IND(0) MF IND(1) MF IND(2) MF IND(3) OPS);                       // 4 per MFMA, independent
        if constexpr (BODY == 4) asm volatile(TRK(0) TRK(1) TRK(2) TRK(3) OPS);                                   // the bookkeeping alone
    }
    asm volatile("s_nop 15" ::: "memory");
    const long long s1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
    for (int j = 0; j < 16; ++j) r += acc[j];
    out[blockIdx.x * 512 + tid] = r;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = s1 - s0;
}

template <int BODY>
void run(const char* name, int threads, const float* in, float* out, long long* cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<BODY>, dim3(256), dim3(threads), 0, 0, in, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2048);
    CHECK(hipMemcpy(h.data(), cyc, 16384, hipMemcpyDeviceToHost));
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[b * 8 + w] / ITER);
    std::sort(c.begin(), c.end());
    printf("%-64s %7.1f cycles per iteration and wave (4 MFMAs = 128)   wall %.3f ms\n", name, c[c.size() / 2], ms);
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 16384));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)(rand() % 2001) / 2000.0f - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("1 wave/SIMD: 4 MFMA", 256, in, out, cyc);
    run<4>("1 wave/SIMD: 16 bookkeeping instructions alone", 256, in, out, cyc);
    run<1>("1 wave/SIMD: 4 x (MFMA + 4 chained bookkeeping)", 256, in, out, cyc);
    run<3>("1 wave/SIMD: 4 x (MFMA + 4 independent of the same kinds)", 256, in, out, cyc);
    run<2>("1 wave/SIMD: 4 MFMA + 24 chained bookkeeping (6 per MFMA)", 256, in, out, cyc);
    run<0>("2 waves/SIMD: 4 MFMA", 512, in, out, cyc);
    run<1>("2 waves/SIMD: 4 x (MFMA + 4 chained bookkeeping)", 512, in, out, cyc);
    run<3>("2 waves/SIMD: 4 x (MFMA + 4 independent of the same kinds)", 512, in, out, cyc);
    run<2>("2 waves/SIMD: 4 MFMA + 24 chained bookkeeping (6 per MFMA)", 512, in, out, cyc);
    return 0;
}
