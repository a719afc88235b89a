// lipvq_screen.h -- layout of the prepared codebook and constants shared by the screening kernels
// (lipvq_screen.hip, lipvq_fused.hip).  Design notes: lipvq_screen.hip.
#ifndef LIPVQ_SCREEN_H_
#define LIPVQ_SCREEN_H_
#include <hip/hip_fp16.h>

#include "lipvq_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LIPVQ_SCREEN_GAMMA 7.62939453125e-06f   /* 2^-17 */
#define SCREEN_WAVES 8

struct PrepLayout {
    int S, Dpad, Kpad, ntiles;
    size_t o_hdr, o_mu, o_tiles, tile_bytes, total;
};

__host__ __device__ static inline PrepLayout prep_layout(int K, int D) {
    PrepLayout L;
    L.S = (D + 15) / 16;
    L.Dpad = L.S * 16;
    L.Kpad = ((K + 31) / 32) * 32;
    L.ntiles = L.Kpad / 32;
    L.o_hdr = 0;                       // 16 floats: [0] E2max bits, [1] Emax^2 bits, [2] max|2e'| bits
    L.o_mu = 64;
    L.o_tiles = L.o_mu + sizeof(float) * (size_t)L.Dpad;
    L.o_tiles = (L.o_tiles + 255) & ~(size_t)255;
    L.tile_bytes = (size_t)L.S * 2048 + 128;    // S steps x {hi,lo} x 32 codes x 2 halves x 16 B, then 32 x e2
    L.total = L.o_tiles + (size_t)L.ntiles * L.tile_bytes;
    return L;
}


// exact decision for listed rows (lipvq_screen.hip); z_by_slot: z is a compact [count][D] buffer
int lipvq_launch_rows(const float* z, int z_by_slot, const float* cb, int64_t* idx, float* zq, int64_t* usage,
                      const int* amb_list, const int* amb_count, int64_t N, int K, int D, hipStream_t st);
#endif
