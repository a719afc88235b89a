// probe_roles.hip -- round 4.  Two waves per SIMD, each owes T "tiles" of work; a tile = 32 dependent fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, one accumulator) + 152 packed fmas (eight independent Horner chains of 19: the encoder's GELU of a
// 16-element tile).  How long does a PAIR of tiles (one per wave) take under different arrangements of the same instructions?
//   I    interleaved: MFMA, ~5 pk_fma, MFMA, ...                       (the kernel until round 3)
//   L    lumped:      32 MFMAs, then 152 pk_fma                         (round 4)
//   Loff lumped, waves 4-7 start half a period late (once)
//   S    specialised: waves 0-3 issue the 64 MFMAs of both tiles, waves 4-7 the 304 pk_fma of both (no data exchange modelled)
//   SL   super-lumped: 4 tiles of MFMAs (128), then 4 tiles of VALU (608)
// MFMA-pipe bound: 64 x 64 = 4096 cycles per pair.   build: hipcc --offload-arch=gfx950 -O3 -o probe_roles probe_roles.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int TILES = 512;

#define MF "v_mfma_f32_32x32x2_f32 %[acc], %[a], %[b], %[acc]\n\t"
#define MF4 MF MF MF MF
#define MF32 MF4 MF4 MF4 MF4 MF4 MF4 MF4 MF4
// one "round" of the eight Horner chains: 8 packed fmas
#define PK(i) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[u], %[c]\n\t"
#define PK8 PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
#define PK5 PK(0) PK(1) PK(2) PK(3) PK(4)
#define PK4b PK(5) PK(6) PK(7) PK(0)
#define PK152 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8
#define OPS : [acc] "+v"(acc), [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]) \
            : [a] "v"(a), [b] "v"(b), [u] "v"(u), [c] "v"(c)
// interleaved: 32 x (MFMA + 4.75 pk) = 152 pk: pattern of 5,5,5,4 per 4 MFMAs = 19 per 4 -> x 8 = 152
#define I4 MF PK5 MF PK(5) PK(6) PK(7) PK(0) PK(1) MF PK(2) PK(3) PK(4) PK(5) PK(6) MF PK(7) PK(0) PK(1) PK(2)
#define I32 I4 I4 I4 I4 I4 I4 I4 I4

// ---- encoder wave beside screen wave (round 4): the stream of tokenize_kernel's screen loop, one k-step =
//   s_waitcnt lgkmcnt(0); MFMA16; 2 x ds_read_b128; 5-6 bookkeeping VALU; MFMA16; 5 VALU; MFMA16; 5 VALU   (16 VALU: 4 x fmac, and_or, med3, min)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define BK(m1, m2, x) "v_fmac_f32 %[" #x "], %[sc], %[e2]\n\tv_and_or_b32 %[" #x "], %[" #x "], %[km], %[id]\n\tv_med3_f32 %[" #m2 "], %[" #x "], %[" #m1 "], %[" #m2 "]\n\tv_min_f32 %[" #m1 "], %[" #x "], %[" #m1 "]\n\t"
#define MH "v_mfma_f32_32x32x16_f16 %[hacc], %[ha], %[hb], %[hacc]\n\t"
#define KSTEP "s_waitcnt lgkmcnt(0)\n\t" MH "ds_read_b128 %[hb], %[la]\n\tds_read_b128 %[hb2], %[la] offset:1024\n\t" BK(ma, mb, x0) BK(mc, md, x1) MH BK(ma, mb, x2) MH BK(mc, md, x3)
#define KOPS : [hacc] "+v"(hacc), [hb] "+v"(hb), [hb2] "+v"(hb2), [ma] "+v"(ma), [mb] "+v"(mb), [mc] "+v"(mc), [md] "+v"(md), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3) \
             : [ha] "v"(ha), [la] "v"(la), [sc] "s"(sc), [e2] "v"(e2), [km] "s"(km), [id] "v"(id)

template <int MODE>
__global__ __launch_bounds__(512) void probe2(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int ksteps_per_tile) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8 * 1024; i += 512) reinterpret_cast<float*>(lds)[i] = in[i & 1023];
    float a = in[tid], b = in[tid + 512];
    f32x16 acc, hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = in[(tid + r * 7) & 1023]; hacc[r] = in[(tid + r * 5) & 1023]; }
    v2f p[8], u = {in[3] * 0.25f, in[4] * 0.25f}, c = {in[5], in[6]};
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (v2f){in[(tid + i) & 1023], in[(tid + 2 * i + 1) & 1023]};
    f16x8 ha, hb, hb2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)in[(tid + j) & 1023]; hb[j] = (_Float16)in[(tid * 3 + j) & 1023]; hb2[j] = hb[j]; }
    float ma = 1e30f, mb = 1e30f, mc = 1e30f, md = 1e30f, x0 = in[7], x1 = in[8], x2 = in[9], x3 = in[10], e2 = in[11];
    const float sc = in[12];
    const unsigned km = 0xffffffe0u;
    int id = lane & 31;
    const unsigned la = (unsigned)(uintptr_t)(lds + (wave & 3) * 4096 + lane * 16) & 0xffffu;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const bool enc = (MODE == 0) ? (wave < 4) : (MODE == 1) ? (wave >= 4) : (MODE == 2) ? true : (MODE == 3) ? false : (MODE == 4);
    const bool idle = (MODE == 2 && wave >= 4) || (MODE == 3 && wave >= 4);
    if (!idle) {
        if (enc) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(PK152 OPS); }
        else for (int it = 0; it < TILES * ksteps_per_tile; ++it) asm volatile(KSTEP KOPS);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = ma + mb + mc + md + x0 + x1 + x2 + x3;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r] + hacc[r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)hb[i] + (float)hb2[i];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run2(const char* name, const float* in, float* out, long long* cyc, int kpt) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe2<MODE>), dim3(256), dim3(512), 0, 0, in, out, cyc, kpt);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ca : cb).push_back((double)h[b * 8 + w] / TILES);
    auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("%-74s waves 0-3 %8.0f   waves 4-7 %8.0f cycles per unit   wall %.3f ms\n", name, med(ca), med(cb), ms);
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int delay) {
    const int tid = threadIdx.x, wave = tid >> 6;
    float a = in[tid], b = in[tid + 512];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = in[(tid + r * 7) & 1023];
    v2f p[8], u = {in[3] * 0.25f, in[4] * 0.25f}, c = {in[5], in[6]};
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (v2f){in[(tid + i) & 1023], in[(tid + 2 * i + 1) & 1023]};
    __syncthreads();
    if (MODE == 2 && wave >= 4) { const long long t = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t < delay) __builtin_amdgcn_s_sleep(2); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) for (int it = 0; it < TILES; ++it) asm volatile(I32 OPS);
    if (MODE == 1 || MODE == 2) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(PK152 OPS); }
    if (MODE == 3) {
        if (wave < 4) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(MF32 OPS); }
        else for (int it = 0; it < TILES; ++it) { asm volatile(PK152 OPS); asm volatile(PK152 OPS); }
    }
    if (MODE == 4) for (int it = 0; it < TILES / 4; ++it) { asm volatile(MF32 MF32 OPS); asm volatile(MF32 MF32 OPS); asm volatile(PK152 PK152 OPS); asm volatile(PK152 PK152 OPS); }
    if (MODE == 5) {      // lumped, but waves 4-7 run VALU first, then MFMA (opposite order inside a tile)
        if (wave < 4) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(PK152 OPS); }
        else for (int it = 0; it < TILES; ++it) { asm volatile(PK152 OPS); asm volatile(MF32 OPS); }
    }
    if (MODE == 6) {      // one wave per SIMD, lumped (reference: what a lone wave needs per tile)
        if (wave < 4) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(PK152 OPS); }
    }
    if (MODE == 7) {      // specialised, VALU side on the OLDER waves
        if (wave >= 4) for (int it = 0; it < TILES; ++it) { asm volatile(MF32 OPS); asm volatile(MF32 OPS); }
        else for (int it = 0; it < TILES; ++it) { asm volatile(PK152 OPS); asm volatile(PK152 OPS); }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* in, float* out, long long* cyc, int delay = 0) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 0, 0, in, out, cyc, delay);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ca, cb;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ca : cb).push_back((double)h[b * 8 + w] / TILES);
    auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("%-66s waves 0-3 %8.0f   waves 4-7 %8.0f cycles per tile of its own   wall %.3f ms (%.0f ns per pair of tiles)\n", name, med(ca), med(cb), ms, ms * 1e6 / TILES);
}

int main() {
    float *in, *out; long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    std::vector<float> h(4096);
    srand(1);
    for (auto& x : h) x = (float)(rand() % 20001) / 20000.0f - 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    printf("tile = 32 dependent fp32 MFMAs + 152 v_pk_fma_f32; two waves per SIMD unless noted; MFMA-pipe bound per pair of tiles: 4096 cycles\n");
    for (int rep = 0; rep < 2; ++rep) {
    run<6>("one wave per SIMD, lumped (waves 4-7 idle)", in, out, cyc);
    run<0>("I    interleaved (MFMA, ~5 pk, MFMA, ...)", in, out, cyc);
    run<1>("L    lumped (32 MFMA, 152 pk)", in, out, cyc);
    run<2>("Loff lumped, waves 4-7 start 1400 cycles late", in, out, cyc, 1400);
    run<2>("Loff lumped, waves 4-7 start 2100 cycles late", in, out, cyc, 2100);
    run<5>("L/R  lumped, waves 4-7 in the opposite order (pk first)", in, out, cyc);
    run<4>("SL   super-lumped (128 MFMA, 608 pk)", in, out, cyc);
    run<3>("S    specialised: waves 0-3 all MFMAs, waves 4-7 all pk", in, out, cyc);
    run<7>("S'   specialised: waves 4-7 all MFMAs, waves 0-3 all pk", in, out, cyc);
    }
    printf("\nunit = one encoder tile (32 fp32 MFMA + 152 pk_fma) or KPT screen k-steps (3 fp16 MFMA + 16 bookkeeping VALU + 2 ds_read_b128 each)\n");
    for (int kpt : {12, 16, 24}) {
        printf("-- %d k-steps per unit\n", kpt);
        run2<2>("encoder stream alone (one wave per SIMD)", in, out, cyc, kpt);
        run2<3>("screen stream alone (one wave per SIMD, waves 0-3)", in, out, cyc, kpt);
        run2<4>("encoder | encoder (two waves per SIMD)", in, out, cyc, kpt);
        run2<5>("screen | screen (two waves per SIMD)", in, out, cyc, kpt);
        run2<0>("waves 0-3 encoder | waves 4-7 screen", in, out, cyc, kpt);
        run2<1>("waves 0-3 screen | waves 4-7 encoder", in, out, cyc, kpt);
    }
    return 0;
}
