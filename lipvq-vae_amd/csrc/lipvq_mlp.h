// lipvq_mlp.h -- MFMA A-operand packing of a three-layer stack (shared by lipvq_mlp.hip and
// lipvq_fused.hip).  Layout notes: lipvq_mlp.hip.
#ifndef LIPVQ_MLP_H_
#define LIPVQ_MLP_H_
#include "lipvq_common.h"

__host__ __device__ static inline int feat_of_tile_row(int i) {
    return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1);
}

struct PackedLayout {
    int S0, S1, S2;      // k-steps per layer (k pairs)
    int T0, T1, T2;      // 32-feature output tiles per layer
    size_t oP0, oB0, oP1, oB1, oP2, oB2, total;   // offsets in floats
};

__host__ __device__ static inline PackedLayout packed_layout(int K0, int J0, int J1, int J2) {
    PackedLayout L;
    L.T0 = (J0 + 31) / 32; L.T1 = (J1 + 31) / 32; L.T2 = (J2 + 31) / 32;
    L.S0 = (K0 + 1) / 2; L.S1 = 16 * L.T0; L.S2 = 16 * L.T1;
    size_t o = 0;
    L.oP0 = o; o += (size_t)L.T0 * L.S0 * 64;
    L.oB0 = o; o += (size_t)L.T0 * 32;
    L.oP1 = o; o += (size_t)L.T1 * L.S1 * 64;
    L.oB1 = o; o += (size_t)L.T1 * 32;
    L.oP2 = o; o += (size_t)L.T2 * L.S2 * 64;
    L.oB2 = o; o += (size_t)L.T2 * 32;
    L.total = o;
    return L;
}

#endif
