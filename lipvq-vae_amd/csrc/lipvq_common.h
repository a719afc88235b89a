// lipvq_common.h -- shared by every translation unit of the gfx950 tokenizer library.
#ifndef LIPVQ_COMMON_H_
#define LIPVQ_COMMON_H_
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lipvq.h"
#include "lipvq_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Dynamic-LDS reservation of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize): the attribute belongs to the (function,
// device) pair, so the "already raised to" bookkeeping is kept PER DEVICE and in atomics -- a process that drives several
// GPUs, or several host threads, must not launch with more LDS than the current device has been told about.  (The first
// version kept one `static size_t` per kernel: per process, unsynchronised.)  The attribute call costs ~10 us of host
// time, hence the bookkeeping; a lost race only repeats the call.
#ifdef __cplusplus
#include <atomic>
struct LqLdsReserve {
    static constexpr int kMaxDev = 64;
    std::atomic<size_t> got[kMaxDev];
};
int lipvq_reserve_lds(LqLdsReserve& r, const void* kernel, size_t bytes, const char* what);
#endif

#if defined(__HIPCC__)
// Workgroup barrier that does NOT drain vector-memory operations: this wave's LDS accesses are done (lgkmcnt(0)), then
// s_barrier.  __syncthreads() makes hipcc wait vmcnt(0) first, i.e. for every global load / LDS-DMA still in flight --
// which defeats any prefetch that is meant to stay in flight across the barrier.  Global data is waited for where it is
// consumed (hipcc's own counted vmcnt for ordinary loads; lq_wait_vmcnt for LDS-DMA).
__device__ __forceinline__ void lq_wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

#if defined(__HIPCC__)
// usage[k] += 1 for every lane with `active`, with duplicates inside the wave combined first: when many rows map
// to few codes (the reference's default initialisation maps EVERY row to one code) per-row atomics on one address
// serialise chip-wide (measured: 1.7 ms instead of 0.57 ms for a 524 288-row batch with 32 codes).  Up to four
// leader rounds (each: the first active lane's code, a ballot of the lanes that share it, ONE atomic of the count)
// then plain atomics for whatever is left (the typical well-spread case pays four cheap rounds).
__device__ __forceinline__ void lq_usage_add(unsigned long long* __restrict__ usage, int k, bool active) {
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const unsigned long long act = __ballot(active);
        if (act == 0ull) return;                                   // wave-uniform
        const int leader = __ffsll((long long)act) - 1;
        const int kl = __shfl(k, leader, 64);
        const unsigned long long same = __ballot(active && k == kl);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&usage[kl], (unsigned long long)__popcll(same));
        active = active && k != kl;
    }
    if (active) atomicAdd(&usage[k], 1ull);
}
#endif

// measurement / test knobs: ONE place reads them (lipvq_misc.hip)
const char* lq_knob(const char* name);

// error plumbing (defined in lipvq_misc.hip)
int lipvq_fail(int code, const char* fmt, ...);
int lipvq_check_launch(const char* what);
// lipvq_misc.hip: the second pass of the mean-squared-error pair over 2 x 2048 double partial sums (what lipvq_mse_pair_loss_f32
// runs after its own first pass; lipvq_mlp3_loss_f32's first pass is the decoder kernel itself)
int lipvq_mse_finish(const double* partial, int64_t nx, int64_t nz, float* out, float* loss, float w, int form, void* stream);
#define fail lipvq_fail
#define check_launch lipvq_check_launch

__host__ __device__ static inline float lq_act_apply(float v, int act) {
    switch (act) {
        case LIPVQ_ACT_GELU: return lq_gelu(v);
        case LIPVQ_ACT_SIGMOID: return lq_sigmoid(v);
        case LIPVQ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        default: return v;
    }
}

// d act(v) / d v evaluated at the pre-activation v
__host__ __device__ static inline float lq_act_grad(float v, int act) {
    switch (act) {
#if defined(__HIP_DEVICE_COMPILE__)
        case LIPVQ_ACT_GELU: return lq_gelu_grad_dev(v);          // straight-line form (lipvq_math.h), 2e-7 from lq_gelu_grad
#else
        case LIPVQ_ACT_GELU: return lq_gelu_grad(v);
#endif
        case LIPVQ_ACT_SIGMOID: { const float s = lq_sigmoid(v); return s * (1.0f - s); }
        case LIPVQ_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}
#endif
