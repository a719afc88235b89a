// probe_pk.hip -- two questions behind the packed GELU / sigmoid of the fused tokenizer:
//  (1) issue rate of v_pk_fma_f32 against v_fma_f32 (one wave per SIMD, independent chains), and beside an fp32 / fp16 MFMA;
//  (2) is  r = rcp(d); e = fma(-d,r,1); r = fma(e,r,r); err = fma(-d,r,1); q = fma(err,r,r); err = fma(-d,q,1); fma(err,r,q)
//      (hipcc's IEEE division sequence for 1/d with the range handling removed) bit-identical to 1.0f / d for every d in [1, 2^127)?
//      Exhaustive over the 2^23 mantissas at several exponents.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o probe_pk probe_pk.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 4000;

#define PK1(v) "v_pk_fma_f32 %[" #v "], %[" #v "], %[c0], %[c1]\n\t"
#define PK8 PK1(p0) PK1(p1) PK1(p2) PK1(p3) PK1(p4) PK1(p5) PK1(p6) PK1(p7)
#define SC1(v) "v_fma_f32 %[" #v "], %[" #v "], %[d0], %[d1]\n\t"
#define SC8 SC1(s0) SC1(s1) SC1(s2) SC1(s3) SC1(s4) SC1(s5) SC1(s6) SC1(s7)
#define MF32 "v_mfma_f32_32x32x2_f32 %[acc], %[a], %[b], %[acc]\n\t"
#define MF16 "v_mfma_f32_32x32x16_f16 %[acc], %[ha], %[hb], %[acc]\n\t"
#define OPS : [acc] "+v"(acc), [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]), \
              [s0] "+v"(s[0]), [s1] "+v"(s[1]), [s2] "+v"(s[2]), [s3] "+v"(s[3]), [s4] "+v"(s[4]), [s5] "+v"(s[5]), [s6] "+v"(s[6]), [s7] "+v"(s[7]) \
            : [a] "v"(a), [b] "v"(b), [ha] "v"(ha), [hb] "v"(hb), [c0] "v"(c0), [c1] "v"(c1), [d0] "v"(d0), [d1] "v"(d1)

template <int BODY>
__device__ __forceinline__ void body(f32x16& acc, v2f (&p)[8], float (&s)[8], float a, float b, f16x8 ha, f16x8 hb, v2f c0, v2f c1, float d0, float d1);

template <int BA, int BB>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc) {
    const int tid = threadIdx.x;
    float a = in[tid], b = in[tid + 512];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = in[(tid + 7 * r) & 1023];
    f16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)in[(tid + j) & 1023]; hb[j] = (_Float16)in[(3 * tid + j) & 1023]; }
    v2f p[8]; float s[8];
    for (int j = 0; j < 8; ++j) { p[j] = (v2f){in[(tid + j) & 1023], in[(tid + j + 8) & 1023]}; s[j] = in[(tid + j + 16) & 1023]; }
    const v2f c0 = {in[5] * 0.5f, in[5] * 0.5f}, c1 = {in[6], in[6]};
    const float d0 = in[5] * 0.5f, d1 = in[6];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (tid < 256) body<BA>(acc, p, s, a, b, ha, hb, c0, c1, d0, d1); else body<BB>(acc, p, s, a, b, ha, hb, c0, c1, d0, d1);
    asm volatile("s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += p[j].x + p[j].y + s[j];
    for (int j = 0; j < 16; ++j) r += acc[j];
    out[blockIdx.x * 512 + tid] = r;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int BODY>
__device__ __forceinline__ void body(f32x16& acc, v2f (&p)[8], float (&s)[8], float a, float b, f16x8 ha, f16x8 hb, v2f c0, v2f c1, float d0, float d1) {
    for (int it = 0; it < ITER; ++it) {
        if constexpr (BODY == 20) asm volatile(MF32 MF32 OPS);
        if constexpr (BODY == 21) asm volatile(MF16 MF16 MF16 MF16 OPS);
        if constexpr (BODY == 0) asm volatile(SC8 SC8 OPS);              // 16 v_fma_f32
        if constexpr (BODY == 1) asm volatile(PK8 PK8 OPS);              // 16 v_pk_fma_f32 (32 fmas)
        if constexpr (BODY == 2) asm volatile(MF32 PK8 OPS);             // fp32 MFMA + 8 pk
        if constexpr (BODY == 3) asm volatile(MF32 SC8 OPS);             // fp32 MFMA + 8 scalar
        if constexpr (BODY == 4) asm volatile(MF16 PK8 OPS);             // fp16 MFMA + 8 pk
        if constexpr (BODY == 5) asm volatile(MF16 SC8 OPS);
        if constexpr (BODY == 6) asm volatile(MF16 MF16 PK8 OPS);
        if constexpr (BODY == 7) asm volatile(PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) PK1(p0) OPS);
        if constexpr (BODY == 8) asm volatile(SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) SC1(s0) OPS);
        if constexpr (BODY == 9) asm volatile(PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) PK1(p0) PK1(p1) OPS);
        if constexpr (BODY == 10) asm volatile(SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) SC1(s0) SC1(s1) OPS);
        if constexpr (BODY == 11) asm volatile(PK1(p0) PK1(p1) PK1(p2) PK1(p3) PK1(p0) PK1(p1) PK1(p2) PK1(p3) PK1(p0) PK1(p1) PK1(p2) PK1(p3) PK1(p0) PK1(p1) PK1(p2) PK1(p3) OPS);
    }
}

template <int BA, int BB = 99>
void run(const char* name, const float* in, float* out, long long* cyc) {
    const int threads = BB == 99 ? 256 : 512;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<BA, BB>), dim3(256), dim3(threads), 0, 0, in, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2048);
    CHECK(hipMemcpy(h.data(), cyc, 16384, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < threads / 64; ++w) (w < 4 ? ca : cb).push_back((double)h[b * 8 + w] / ITER);
    std::sort(ca.begin(), ca.end()); std::sort(cb.begin(), cb.end());
    printf("%-52s A %8.1f  B %8.1f cycles per iteration   wall %.3f ms\n", name, ca[ca.size() / 2], cb.empty() ? 0.0 : cb[cb.size() / 2], ms);
}

__device__ __forceinline__ float rcp_seq(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float err = __builtin_fmaf(-d, r, 1.0f);
    float q = __builtin_fmaf(err, r, r);
    err = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(err, r, q);
}
__global__ void div_check(unsigned expo, unsigned long long* bad, unsigned* first_bad) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;       // 2^23 mantissas
    const float d = __uint_as_float((expo << 23) | m);
    const float ref = 1.0f / d;
    const float got = rcp_seq(d);
    if (__float_as_uint(ref) != __float_as_uint(got)) { if (atomicAdd(bad, 1ull) == 0) *first_bad = __float_as_uint(d); }
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 16384));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)(rand() % 20001) / 20000.0f - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("16 v_fma_f32", in, out, cyc);
    run<1>("16 v_pk_fma_f32", in, out, cyc);
    run<2>("fp32 MFMA + 8 v_pk_fma_f32", in, out, cyc);
    run<3>("fp32 MFMA + 8 v_fma_f32", in, out, cyc);
    run<4>("fp16 MFMA + 8 v_pk_fma_f32", in, out, cyc);
    run<5>("fp16 MFMA + 8 v_fma_f32", in, out, cyc);
    run<6>("2 fp16 MFMA + 8 v_pk_fma_f32", in, out, cyc);
    run<7>("16 DEPENDENT v_pk_fma_f32", in, out, cyc);
    run<8>("16 DEPENDENT v_fma_f32", in, out, cyc);
    run<9>("16 v_pk_fma_f32, two chains", in, out, cyc);
    run<10>("16 v_fma_f32, two chains", in, out, cyc);
    run<11>("16 v_pk_fma_f32, four chains", in, out, cyc);
    run<20, 0>("2w: A 2 fp32 MFMA | B 16 v_fma_f32", in, out, cyc);
    run<20, 1>("2w: A 2 fp32 MFMA | B 16 v_pk_fma_f32", in, out, cyc);
    run<21, 0>("2w: A 4 fp16 MFMA | B 16 v_fma_f32", in, out, cyc);
    run<21, 1>("2w: A 4 fp16 MFMA | B 16 v_pk_fma_f32", in, out, cyc);
    run<0, 0>("2w: 16 v_fma_f32 | same", in, out, cyc);
    run<1, 1>("2w: 16 v_pk_fma_f32 | same", in, out, cyc);
    run<3, 3>("2w: fp32 MFMA + 8 v_fma_f32 | same", in, out, cyc);
    run<2, 2>("2w: fp32 MFMA + 8 v_pk_fma_f32 | same", in, out, cyc);
    run<5, 5>("2w: fp16 MFMA + 8 v_fma_f32 | same", in, out, cyc);
    run<4, 4>("2w: fp16 MFMA + 8 v_pk_fma_f32 | same", in, out, cyc);
    unsigned long long* bad; unsigned* fb;
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&fb, 4));
    const unsigned expos[] = {127, 128, 129, 140, 200, 250, 252, 253};     // d in [1, 2), [2, 4), ... up to 2^126
    for (unsigned e : expos) {
        CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(fb, 0, 4));
        hipLaunchKernelGGL(div_check, dim3(1 << 15), dim3(256), 0, 0, e, bad, fb);
        unsigned long long nb; unsigned f;
        CHECK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost));
        printf("1/d, exponent %3u: %llu of 8388608 mantissas differ from IEEE division (first 0x%08x)\n", e, nb, f);
    }
    return 0;
}
