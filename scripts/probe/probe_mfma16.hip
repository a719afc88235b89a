// Dev probe (GPU): what does v_mfma_f32_32x32x16_f16 compute, bit for bit?  Compares the hardware result of 1024 independent
// 16-term dot products (+ C) per trial with candidate accumulation models evaluated in exact fixed-point arithmetic.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probe/probe_mfma16 scripts/probe/probe_mfma16.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A[m][k] (32 x 16), B[k][n] (16 x 32), C/D[m][n]: lane l supplies A[m = l & 31][8 (l >> 5) + j], B[8 (l >> 5) + j][n = l & 31]
__global__ void probe(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int l = threadIdx.x, i = l & 31, q = l >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 16 + 8 * q + j]; b[j] = B[(8 * q + j) * 32 + i]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * q) * 32 + i];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * q) * 32 + i] = c[r];
}

typedef __int128 i128;
static const int FRAC = 60;                       // fixed point: value * 2^60
static i128 to_fix(double v) { return (i128)ldexp(v, FRAC); }     // exact for our magnitudes (|v| < 2^40, lsb >= 2^-60)
static float round_fix(i128 s, int mode) {        // mode 0: RNE, 1: RTZ
    if (s == 0) return 0.0f;
    const bool neg = s < 0;
    unsigned __int128 m = neg ? (unsigned __int128)(-s) : (unsigned __int128)s;
    int hb = 127;
    while (!((m >> hb) & 1)) --hb;
    int sh = hb - 23;                             // keep 24 bits
    uint64_t mant;
    if (sh <= 0) mant = (uint64_t)(m << (-sh));
    else {
        mant = (uint64_t)(m >> sh);
        if (mode == 0) {
            const unsigned __int128 rem = m & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
            if (rem > half || (rem == half && (mant & 1))) ++mant;
        }
    }
    const float f = ldexpf((float)mant, sh - FRAC);
    return neg ? -f : f;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 200;
    const int spread = argc > 2 ? atoi(argv[2]) : 6;      // exponent spread of the operands (+- spread)
    _Float16 *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, 512 * 2); (void)hipMalloc(&dB, 512 * 2); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    _Float16 hA[512], hB[512]; float hC[1024], hD[1024];
    long n = 0, okA = 0, okT = 0, okSeq = 0, okG4 = 0, okHalf = 0, okG4x = 0, okPair = 0;
    srand(1);
    int shown = 0;
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < 512; ++i) {
            double m = 1.0 + (rand() % 1024) / 1024.0;
            int e = rand() % (2 * spread + 1) - spread;
            hA[i] = (_Float16)((rand() & 1 ? -1 : 1) * ldexp(m, e));
            m = 1.0 + (rand() % 1024) / 1024.0; e = rand() % (2 * spread + 1) - spread;
            hB[i] = (_Float16)((rand() & 1 ? -1 : 1) * ldexp(m, e));
        }
        for (int i = 0; i < 1024; ++i) {
            double m = 1.0 + (rand() % (1 << 23)) / (double)(1 << 23);
            int e = rand() % (2 * spread + 1) - spread;
            hC[i] = (float)((rand() & 1 ? -1 : 1) * ldexp(m, e));
            if (t % 4 == 0) hC[i] = 0.0f;
        }
        (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        for (int m = 0; m < 32; ++m)
            for (int c = 0; c < 32; ++c) {
                double p[16];
                for (int k = 0; k < 16; ++k) p[k] = (double)hA[m * 16 + k] * (double)hB[k * 32 + c];     // exact in double
                const float C0 = hC[m * 32 + c], got = hD[m * 32 + c];
                i128 tot = to_fix(C0);
                for (int k = 0; k < 16; ++k) tot += to_fix(p[k]);
                const float mA = round_fix(tot, 0), mT = round_fix(tot, 1);
                float seq = C0;
                for (int k = 0; k < 16; ++k) seq = fmaf((float)hA[m * 16 + k], (float)hB[k * 32 + c], seq);
                float g4 = C0;                                     // groups of 4 consecutive k, each added exactly then rounded
                for (int g = 0; g < 4; ++g) { i128 s = to_fix(g4); for (int k = 4 * g; k < 4 * g + 4; ++k) s += to_fix(p[k]); g4 = round_fix(s, 0); }
                float hf = C0;                                     // the two k-halves (lane halves), exact inside
                for (int g = 0; g < 2; ++g) { i128 s = to_fix(hf); for (int k = 8 * g; k < 8 * g + 8; ++k) s += to_fix(p[k]); hf = round_fix(s, 0); }
                float g4x = C0;                                    // groups {j, j+8} interleaved: k = 4g'.. from both halves
                for (int g = 0; g < 4; ++g) { i128 s = to_fix(g4x); for (int j = 0; j < 2; ++j) { s += to_fix(p[2 * g + j]); s += to_fix(p[8 + 2 * g + j]); } g4x = round_fix(s, 0); }
                float pr = C0;                                     // pairs
                for (int g = 0; g < 8; ++g) { i128 s = to_fix(pr); s += to_fix(p[2 * g]); s += to_fix(p[2 * g + 1]); pr = round_fix(s, 0); }
                ++n;
                okA += memcmp(&got, &mA, 4) == 0; okT += memcmp(&got, &mT, 4) == 0; okSeq += memcmp(&got, &seq, 4) == 0;
                okG4 += memcmp(&got, &g4, 4) == 0; okHalf += memcmp(&got, &hf, 4) == 0; okG4x += memcmp(&got, &g4x, 4) == 0;
                okPair += memcmp(&got, &pr, 4) == 0;
                if (memcmp(&got, &mA, 4) != 0 && shown < 6) {
                    ++shown;
                    printf("mismatch: got %a exactRNE %a RTZ %a seq %a g4 %a half %a C %a\n", got, mA, mT, seq, g4, hf, C0);
                }
            }
    }
    printf("cases %ld: exact-RNE %.4f exact-RTZ %.4f seq-fma %.4f groups4 %.4f halves %.4f g4-interleaved %.4f pairs %.4f\n", n,
           okA / (double)n, okT / (double)n, okSeq / (double)n, okG4 / (double)n, okHalf / (double)n, okG4x / (double)n, okPair / (double)n);
    return 0;
}
