/* lipvq_math.h -- canonical fp32 arithmetic of the LipVQ-VAE action tokenizer.
 *
 * ONE definition of every transcendental and of the distance reduction, shared
 * verbatim by the gfx950 device code (hipcc) and by the CPU oracle (gcc).  It
 * uses only +, -, *, /, fmaf, comparisons and integer bit casts, so that both
 * compilers (built with -ffp-contract=off; IEEE division; denormals kept)
 * produce bit-identical results.  That is what lets the GPU path be compared
 * with the oracle with `==`, not with a tolerance, at any batch size.
 *
 * What each function restates (reference: /root/reference/robomimic/models/
 * vq_vae/backbone_lfqvae_v5.py, "v5" below):
 *   lq_gelu       nn.GELU() default = exact erf form        (v5:56,58,64,66)
 *   lq_sigmoid    torch.sigmoid                             (v5:24)
 *   lq_softplus   F.softplus, beta 1, threshold 20          (v5:10)
 *   lq_sqdist8    torch.norm(..., dim=-1) BEFORE the sqrt   (v5:43-45)
 * torch's CPU kernels use vendor libm/Sleef for erf/exp; these polynomials
 * agree with them to a few ulp (tests/test_oracle_math.py), which is inside
 * the 1e-5 budget of the north star.  The distance reduction, in contrast, is
 * reproduced EXACTLY: torch's vectorised L2-norm reduction on x86 (AVX2 and
 * AVX512 builds alike) keeps 8 partial sums acc[j] += d[8i+j]^2 with fused
 * multiply-add, then adds the 8 lanes left to right, folds the elements past
 * the last whole 8-vector (lq_sqdist8 says how) and takes the square root
 * (established empirically: oracle/probe_torch_norm.py, 0 mismatches for every
 * D in 1..263).  Coefficients: oracle/fit_coeffs.py.
 */
#ifndef LIPVQ_MATH_H_
#define LIPVQ_MATH_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LQ_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define LQ_HD static inline __attribute__((always_inline))
#endif

LQ_HD uint32_t lq_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
LQ_HD float lq_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
LQ_HD float lq_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
/* IEEE correctly rounded square root: sqrtss on the host; on gfx950 the refined sequence hipcc
 * emits under -fhip-fp32-correctly-rounded-divide-sqrt (NOT __fsqrt_rn, which is the 1-ulp native op). */
LQ_HD float lq_sqrt(float x) { return __builtin_sqrtf(x); }
LQ_HD float lq_abs(float x) { return lq_u2f(lq_f2u(x) & 0x7fffffffu); }

/* 2^n for n in [-126, 127] */
LQ_HD float lq_pow2i(int n) { return lq_u2f((uint32_t)(n + 127) << 23); }

/* exp(x): n = rint(x/ln2) by the 1.5*2^23 trick, r = x - n ln2 (two-term
 * Cody-Waite), exp(r) = 1 + r + r^2 q(r), scaled by 2^n in two exact steps so
 * that results in the denormal range are rounded once. */
LQ_HD float lq_expf(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283f) x = 88.72283f;       /* exp -> 3.4028e38, no inf */
    if (x < -104.0f) return 0.0f;            /* below half the smallest denormal */
    const float magic = 12582912.0f;         /* 1.5 * 2^23 */
    float n = lq_fma(x, 1.44269504088896341f, magic) - magic;
    float r = lq_fma(n, -0.693145751953125f, x);          /* ln2 high: 0x3f317200 */
    r = lq_fma(n, -1.42860682030941723e-6f, r);           /* ln2 low */
    float q = 0.00019907570094801486f;
    q = lq_fma(q, r, 0.0013933652080595493f);
    q = lq_fma(q, r, 0.00833328627049923f);
    q = lq_fma(q, r, 0.04166646674275398f);
    q = lq_fma(q, r, 0.1666666716337204f);
    q = lq_fma(q, r, 0.5f);
    float p = lq_fma(r * r, q, r) + 1.0f;
    int ni = (int)n;
    int h = ni / 2;
    return (p * lq_pow2i(h)) * lq_pow2i(ni - h);
}

/* erf(x).  Main range |x| < 3: x * s(u), s a degree-12 polynomial in u = x^2/4.5 - 1 (shifted so that
 * the fp32 Horner evaluation is well conditioned).  3 <= |x| < 4: 1 - exp(-p(|x|)).  Else +-1.
 * Max abs error 1.7e-7 (tests/test_oracle_math.py).  The main range is wide on purpose: in GELU it covers
 * |pre-activation| < 4.24, so on the GPU practically every wavefront takes ONLY this branch (no exp). */
LQ_HD float lq_erff(float x) {
    if (!(x == x)) return x;
    float a = lq_abs(x);
    float r;
    if (a < 3.0f) {
        float u = lq_fma(a * a, 0.22222222222222222222f, -1.0f);
        float s = 0.00012666420661844313f;
        s = lq_fma(s, u, -0.00043783686123788357f);
        s = lq_fma(s, u, 0.0008924771682359278f);
        s = lq_fma(s, u, -0.002175821689888835f);
        s = lq_fma(s, u, 0.005515238270163536f);
        s = lq_fma(s, u, -0.01217574905604124f);
        s = lq_fma(s, u, 0.02415713667869568f);
        s = lq_fma(s, u, -0.043842192739248276f);
        s = lq_fma(s, u, 0.07253222167491913f);
        s = lq_fma(s, u, -0.11009667813777924f);
        s = lq_fma(s, u, 0.15749694406986237f);
        s = lq_fma(s, u, -0.2287982553243637f);
        s = lq_fma(s, u, 0.4701318144798279f);
        r = a * s;
    } else if (a < 4.0f) {
        float p = -1.4354437780639273e-6f;
        p = lq_fma(p, a, 4.2518615373410285e-5f);
        p = lq_fma(p, a, -0.0005705535295419395f);
        p = lq_fma(p, a, 0.004652169067412615f);
        p = lq_fma(p, a, -0.02615582011640072f);
        p = lq_fma(p, a, 0.10967476665973663f);
        p = lq_fma(p, a, 0.6322006583213806f);
        p = lq_fma(p, a, 1.1300415992736816f);
        p = lq_fma(p, a, -0.0002783838426694274f);
        r = 1.0f - lq_expf(-p);
    } else {
        r = 1.0f;
    }
    return lq_u2f(lq_f2u(r) | (lq_f2u(x) & 0x80000000u));
}

/* nn.GELU() (approximate='none'): 0.5 x (1 + erf(x / sqrt 2)).
 * With erf(a) = a s(a^2/4.5 - 1) on |a| < 3 and a = x/sqrt 2:  x erf(x/sqrt 2) = (x^2/sqrt 2) s(x^2/9 - 1), so
 *   gelu(x) = 0.5 x + (x^2 / sqrt 8) s(x^2/9 - 1)            for |x| < 3 sqrt 2
 * -- the same polynomial as lq_erff, but no |x| / sign handling: 17 instructions on the path that practically
 * every GPU wavefront takes.  Outside, the textbook form with lq_erff (which then takes its exp branch). */
/* the polynomial branch of lq_gelu, valid for x*x < 18 (straight-line: the fused kernel interleaves it with MFMAs) */
LQ_HD float lq_gelu_poly(float x) {
    float t = x * x;
    float u = lq_fma(t, 0.11111111111111111111f, -1.0f);
    float s = 0.00012666420661844313f;
    s = lq_fma(s, u, -0.00043783686123788357f);
    s = lq_fma(s, u, 0.0008924771682359278f);
    s = lq_fma(s, u, -0.002175821689888835f);
    s = lq_fma(s, u, 0.005515238270163536f);
    s = lq_fma(s, u, -0.01217574905604124f);
    s = lq_fma(s, u, 0.02415713667869568f);
    s = lq_fma(s, u, -0.043842192739248276f);
    s = lq_fma(s, u, 0.07253222167491913f);
    s = lq_fma(s, u, -0.11009667813777924f);
    s = lq_fma(s, u, 0.15749694406986237f);
    s = lq_fma(s, u, -0.2287982553243637f);
    s = lq_fma(s, u, 0.4701318144798279f);
    return lq_fma(t * 0.35355339059327376220f, s, 0.5f * x);
}
/* the other branch: the textbook form with lq_erff (which then takes its exp branch); also propagates NaN */
LQ_HD float lq_gelu_tail(float x) { return (0.5f * x) * (1.0f + lq_erff(x * 0.70710678118654752440f)); }

LQ_HD float lq_gelu(float x) { return (x * x < 18.0f) ? lq_gelu_poly(x) : lq_gelu_tail(x); }

/* d/dx gelu(x) = Phi(x) + x phi(x) */
LQ_HD float lq_gelu_grad(float x) {
    float cdf = 0.5f * (1.0f + lq_erff(x * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * lq_expf(-0.5f * (x * x));
    return lq_fma(x, pdf, cdf);
}

/* torch.sigmoid: 1 / (1 + exp(-x)), straight-line (the fused kernel evaluates it 16 times per accumulator tile: lq_expf's two
 * early returns cost a divergent-branch sequence per value).  The argument of exp is clamped to [-87, 87], where exp stays
 * a NORMAL fp32 number: one exact scaling by 2^n instead of lq_expf's two, same polynomial, hence the same bits as
 * 1 / (1 + lq_expf(-x)) for |x| <= 87; beyond, the result is 1 or below 2e-38 either way.  NaN propagates (a NaN fails both
 * clamp comparisons and poisons p). */
LQ_HD float lq_sigmoid(float x) {
    float t = -x;
    t = (t > 87.0f) ? 87.0f : t;
    t = (t < -87.0f) ? -87.0f : t;
    const float magic = 12582912.0f;         /* 1.5 * 2^23 */
    const float nf = lq_fma(t, 1.44269504088896341f, magic);
    const uint32_t nb = lq_f2u(nf) - 0x4B400000u + 127u;          /* rint(t / ln 2) + 127 from the low mantissa bits: in [1, 253] */
    const float n = nf - magic;
    float r = lq_fma(n, -0.693145751953125f, t);
    r = lq_fma(n, -1.42860682030941723e-6f, r);
    float q = 0.00019907570094801486f;
    q = lq_fma(q, r, 0.0013933652080595493f);
    q = lq_fma(q, r, 0.00833328627049923f);
    q = lq_fma(q, r, 0.04166646674275398f);
    q = lq_fma(q, r, 0.1666666716337204f);
    q = lq_fma(q, r, 0.5f);
    const float p = lq_fma(r * r, q, r) + 1.0f;
    const float e = p * lq_u2f(nb << 23);
    return 1.0f / (1.0f + e);
}

#if defined(__HIPCC__) || defined(__HIP__)
/* ---- two elements per instruction (device code only) ------------------------------------------------------------
 * gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- the same IEEE operation on both halves of a 64-bit
 * register pair -- at the rate of their one-element forms (scripts/probe/probe_pk.hip: 16 of either take 78 cycles),
 * so the polynomials below cost half the issue slots of lq_gelu_poly / lq_sigmoid and return the same bits per
 * element.  They matter where nothing can hide vector instructions: beside fp32 MFMAs (which run on the same ALUs). */
typedef float lq_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lq_v2f lq_fma2(lq_v2f a, lq_v2f b, lq_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ lq_v2f lq_bc2(float c) { return (lq_v2f){c, c}; }

/* lq_gelu_poly on a pair (valid for x*x < 18 per element; the caller fixes the others up with lq_gelu_tail) */
__device__ __forceinline__ lq_v2f lq_gelu_poly2(lq_v2f x) {
    const lq_v2f t = x * x;
    const lq_v2f u = lq_fma2(t, lq_bc2(0.11111111111111111111f), lq_bc2(-1.0f));
    lq_v2f s = lq_bc2(0.00012666420661844313f);
    s = lq_fma2(s, u, lq_bc2(-0.00043783686123788357f));
    s = lq_fma2(s, u, lq_bc2(0.0008924771682359278f));
    s = lq_fma2(s, u, lq_bc2(-0.002175821689888835f));
    s = lq_fma2(s, u, lq_bc2(0.005515238270163536f));
    s = lq_fma2(s, u, lq_bc2(-0.01217574905604124f));
    s = lq_fma2(s, u, lq_bc2(0.02415713667869568f));
    s = lq_fma2(s, u, lq_bc2(-0.043842192739248276f));
    s = lq_fma2(s, u, lq_bc2(0.07253222167491913f));
    s = lq_fma2(s, u, lq_bc2(-0.11009667813777924f));
    s = lq_fma2(s, u, lq_bc2(0.15749694406986237f));
    s = lq_fma2(s, u, lq_bc2(-0.2287982553243637f));
    s = lq_fma2(s, u, lq_bc2(0.4701318144798279f));
    return lq_fma2(t * lq_bc2(0.35355339059327376220f), s, lq_bc2(0.5f) * x);
}

/* 1 / d for d in [1, 2^126): hipcc's IEEE division sequence (v_div_scale, v_rcp, four refinement fmas, v_div_fmas,
 * v_div_fixup) with the range handling removed -- for such d and numerator 1 the scaling is the identity and the fix-up
 * passes the value through, so what remains is v_rcp + six fmas, the same bits as 1.0f / d (probe_pk.hip: all 2^23
 * mantissas at exponents 0, 1, 2, 13, 73, 123, 125).  NaN propagates. */
__device__ __forceinline__ lq_v2f lq_rcp2_ge1(lq_v2f d) {
    lq_v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const lq_v2f nd = -d, one = lq_bc2(1.0f);
    const lq_v2f e = lq_fma2(nd, r, one);
    r = lq_fma2(e, r, r);
    lq_v2f err = lq_fma2(nd, r, one);
    const lq_v2f q = lq_fma2(err, r, r);
    err = lq_fma2(nd, q, one);
    return lq_fma2(err, r, q);
}

/* lq_sigmoid on a pair of FINITE elements (the clamp is a v_med3, which would turn a NaN into a number: callers route
 * tiles that hold a NaN or an infinity to lq_sigmoid -- see lq_nonfinite_acc2) */
__device__ __forceinline__ lq_v2f lq_sigmoid2_finite(lq_v2f x) {
    const lq_v2f t = {__builtin_amdgcn_fmed3f(-x.x, -87.0f, 87.0f), __builtin_amdgcn_fmed3f(-x.y, -87.0f, 87.0f)};
    const lq_v2f magic = lq_bc2(12582912.0f);
    const lq_v2f nf = lq_fma2(t, lq_bc2(1.44269504088896341f), magic);
    /* (bits(nf) - 0x4B400000 + 127) << 23 = (bits(nf) << 23) + 0x3F800000: the low nine bits of 0x4B400000 are zero */
    const lq_v2f sc = {lq_u2f((lq_f2u(nf.x) << 23) + 0x3F800000u), lq_u2f((lq_f2u(nf.y) << 23) + 0x3F800000u)};
    const lq_v2f n = nf - magic;
    lq_v2f r = lq_fma2(n, lq_bc2(-0.693145751953125f), t);
    r = lq_fma2(n, lq_bc2(-1.42860682030941723e-6f), r);
    lq_v2f q = lq_bc2(0.00019907570094801486f);
    q = lq_fma2(q, r, lq_bc2(0.0013933652080595493f));
    q = lq_fma2(q, r, lq_bc2(0.00833328627049923f));
    q = lq_fma2(q, r, lq_bc2(0.04166646674275398f));
    q = lq_fma2(q, r, lq_bc2(0.1666666716337204f));
    q = lq_fma2(q, r, lq_bc2(0.5f));
    const lq_v2f p = lq_fma2(r * r, q, r) + lq_bc2(1.0f);
    const lq_v2f e = p * sc;
    return lq_rcp2_ge1(lq_bc2(1.0f) + e);
}
/* d/dx gelu(x) = Phi(x) + x phi(x), straight-line on x*x < 18 (device code: the backward chains evaluate it for every saved
 * pre-activation, 192 per row and stack): Phi through the SAME erf polynomial as lq_gelu_poly -- erf(x/sqrt 2) = (x/sqrt 2) s(x^2/9 - 1)
 * -- and exp(-x^2/2) through lq_expf's polynomial with ONE exact scaling (the argument stays in (-9, 0]: no clamps, no
 * denormal care).  Within 2e-7 of lq_gelu_grad (which takes erf's and exp's general branches and is what the oracle runs);
 * gradients are compared with 1e-5 of their scale.  Elsewhere: lq_gelu_grad itself. */
__device__ __forceinline__ float lq_gelu_grad_poly(float x) {      /* the straight-line part: valid for x * x < 18 */
    const float t = x * x;
    const float u = lq_fma(t, 0.11111111111111111111f, -1.0f);
    float s = 0.00012666420661844313f;
    s = lq_fma(s, u, -0.00043783686123788357f);
    s = lq_fma(s, u, 0.0008924771682359278f);
    s = lq_fma(s, u, -0.002175821689888835f);
    s = lq_fma(s, u, 0.005515238270163536f);
    s = lq_fma(s, u, -0.01217574905604124f);
    s = lq_fma(s, u, 0.02415713667869568f);
    s = lq_fma(s, u, -0.043842192739248276f);
    s = lq_fma(s, u, 0.07253222167491913f);
    s = lq_fma(s, u, -0.11009667813777924f);
    s = lq_fma(s, u, 0.15749694406986237f);
    s = lq_fma(s, u, -0.2287982553243637f);
    s = lq_fma(s, u, 0.4701318144798279f);
    const float cdf = lq_fma(x * 0.35355339059327376220f, s, 0.5f);
    const float a = -0.5f * t;                                   /* (-9, 0] */
    const float magic = 12582912.0f;
    const float nf = lq_fma(a, 1.44269504088896341f, magic);
    const float n = nf - magic;
    float r = lq_fma(n, -0.693145751953125f, a);
    r = lq_fma(n, -1.42860682030941723e-6f, r);
    float q = 0.00019907570094801486f;
    q = lq_fma(q, r, 0.0013933652080595493f);
    q = lq_fma(q, r, 0.00833328627049923f);
    q = lq_fma(q, r, 0.04166646674275398f);
    q = lq_fma(q, r, 0.1666666716337204f);
    q = lq_fma(q, r, 0.5f);
    const float p = lq_fma(r * r, q, r) + 1.0f;
    const float e = p * lq_u2f((lq_f2u(nf) << 23) + 0x3F800000u);
    return lq_fma(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float lq_gelu_grad_dev(float x) { return (x * x < 18.0f) ? lq_gelu_grad_poly(x) : lq_gelu_grad(x); }

/* running check for lq_sigmoid2_finite: chk stays 0 (or -0) while every element seen is finite, and turns NaN for good
 * once one is NaN or +-inf (0 * inf = NaN) */
__device__ __forceinline__ lq_v2f lq_nonfinite_acc2(lq_v2f x, lq_v2f chk) { return lq_fma2(x, lq_bc2(0.0f), chk); }
#endif

/* log(u) for finite u >= 1 (only use: softplus). */
LQ_HD float lq_logf_ge1(float u) {
    uint32_t b = lq_f2u(u);
    int e = (int)(b >> 23) - 127;
    float m = lq_u2f((b & 0x007fffffu) | 0x3f800000u);   /* [1,2) */
    if (m > 1.41421356237309504880f) { m = 0.5f * m; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float r = 0.19365468621253967f;
    r = lq_fma(r, z, 0.2219124436378479f);
    r = lq_fma(r, z, 0.28571757674217224f);
    r = lq_fma(r, z, 0.3999999761581421f);
    r = lq_fma(r, z, 0.6666666865348816f);
    float lm = lq_fma(s * z, r, 2.0f * s);
    float fe = (float)e;
    return lq_fma(fe, 0.693145751953125f, lq_fma(fe, 1.42860682030941723e-6f, lm));
}

/* F.softplus(x) with beta = 1, threshold = 20: x if x > 20 else log1p(exp(x)) */
LQ_HD float lq_softplus(float x) {
    if (!(x <= 20.0f)) return x;
    float t = lq_expf(x);
    float u = 1.0f + t;
    if (u == 1.0f) return t;
    return lq_logf_ge1(u) + (t - (u - 1.0f)) / u;    /* log1p correction term */
}

/* Squared L2 distance in torch's CPU reduction order (see header comment), any D >= 1.
 * z, c: D contiguous floats.  The 8-lane part is torch's vectorised loop; what follows the lane fold is the
 * kernel's scalar remainder loop AS THE SHIPPED BINARY EXECUTES IT (torch 2.10.0 CPU, the AVX2 kernel -- also
 * what an AVX512 host dispatches for this op): while at least four elements remain, four are squared (rounded
 * products) and added in index order -- the compiler turned the remainder loop into a 4-wide multiply followed
 * by an in-order add --, and the last one to three elements are folded with fused multiply-add.  Established by
 * oracle/probe_torch_norm.py for every D in 1..263 (0 mismatches) and pinned by the reference-decided near-tie
 * fixtures tests/golden/llfq_nearties_d{7,20,37,100,203}.npz. */
LQ_HD float lq_sqdist8(const float* z, const float* c, int D) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int i = 0;
    for (; i + 8 <= D; i += 8) {
        float d0 = z[i + 0] - c[i + 0], d1 = z[i + 1] - c[i + 1];
        float d2 = z[i + 2] - c[i + 2], d3 = z[i + 3] - c[i + 3];
        float d4 = z[i + 4] - c[i + 4], d5 = z[i + 5] - c[i + 5];
        float d6 = z[i + 6] - c[i + 6], d7 = z[i + 7] - c[i + 7];
        a0 = lq_fma(d0, d0, a0); a1 = lq_fma(d1, d1, a1);
        a2 = lq_fma(d2, d2, a2); a3 = lq_fma(d3, d3, a3);
        a4 = lq_fma(d4, d4, a4); a5 = lq_fma(d5, d5, a5);
        a6 = lq_fma(d6, d6, a6); a7 = lq_fma(d7, d7, a7);
    }
    float s = ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7;
    if (i + 4 <= D) {
        const float d0 = z[i + 0] - c[i + 0], d1 = z[i + 1] - c[i + 1];
        const float d2 = z[i + 2] - c[i + 2], d3 = z[i + 3] - c[i + 3];
        const float p0 = d0 * d0, p1 = d1 * d1, p2 = d2 * d2, p3 = d3 * d3;
        s = (((s + p0) + p1) + p2) + p3;
        i += 4;
    }
    for (; i < D; ++i) { float d = z[i] - c[i]; s = lq_fma(d, d, s); }
    return s;
}

/* Squared L2 distance of the plain VQVAE variant,
 * `(z_e.unsqueeze(1) - E).pow(2).sum(-1)` (reference: robomimic/models/vq_vae/
 * backbone.py:57-60): every d*d is rounded first (pow materialises a tensor), then torch's CPU sum kernel
 * (cascade sum over the contiguous inner dimension) adds them.  Restated for any D >= 1 (probe: every D in
 * 1..139 and a spread of widths up to 4100, 0 mismatches):
 *   D >= 8 : the row is D/8 vectors of 8 lanes; vector v goes to accumulator v mod 4 while whole groups of
 *            four remain, left-over vectors to accumulator 0; each accumulator is a four-level cascade (16
 *            additions per level -- it only shows from D = 512 on); accumulators 1..3 are added to 0 in that
 *            order; the result starts from the scalar tail (elements past the last whole vector, summed in
 *            order from 0) and then takes the 8 lanes left to right.
 *   D <  8 : the same scheme with one-element "vectors" (torch's scalar inner-sum path) and no tail. */
LQ_HD float lq_sqdist32(const float* z, const float* c, int D) {
    const int W = (D >= 8) ? 8 : 1;
    const int nvec = D / W, size_ilp = nvec / 4;
    int lg = 1;                                        /* torch's CeilLog2(size_ilp) */
    if (size_ilp > 2) { lg = 0; while ((1 << lg) < size_ilp) ++lg; }
    const int level_power = (lg / 4 > 4) ? lg / 4 : 4;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    float acc[4][4][8];
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 8; ++l) acc[j][k][l] = 0.f;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < W; ++l) {
                    const int e = (4 * i + k) * W + l;
                    const float d = z[e] - c[e];
                    acc[0][k][l] = acc[0][k][l] + d * d;
                }
        for (int j = 1; j < 4; ++j) {
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < W; ++l) { acc[j][k][l] = acc[j][k][l] + acc[j - 1][k][l]; acc[j - 1][k][l] = 0.f; }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < W; ++l) {
                const int e = (4 * i + k) * W + l;
                const float d = z[e] - c[e];
                acc[0][k][l] = acc[0][k][l] + d * d;
            }
    for (int j = 1; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < W; ++l) acc[0][k][l] = acc[0][k][l] + acc[j][k][l];
    for (int v = 4 * size_ilp; v < nvec; ++v)
        for (int l = 0; l < W; ++l) {
            const float d = z[v * W + l] - c[v * W + l];
            acc[0][0][l] = acc[0][0][l] + d * d;
        }
    for (int k = 1; k < 4; ++k)
        for (int l = 0; l < W; ++l) acc[0][0][l] = acc[0][0][l] + acc[0][k][l];
    if (W == 1) return acc[0][0][0];
    float s = 0.f;
    for (int t = nvec * 8; t < D; ++t) { const float d = z[t] - c[t]; s = s + d * d; }
    for (int l = 0; l < 8; ++l) s = s + acc[0][0][l];
    return s;
}

/* ---- AdaptiveBinActionEmbedding (reference robomimic/models/bin_action/backbone.py = "bin") -------------
 * bin:47-49 torch.linspace(min, max, steps) in fp32 as torch's CPU kernel rounds it (probed for 2..257 steps):
 * step = (end - start) / (steps - 1);  element i = fma(step, i, start) below the midpoint, fma(-step, steps-1-i, end)
 * from it on. */
LQ_HD float lq_linspace(float start, float end, int steps, int i) {
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? lq_fma(step, (float)i, start) : lq_fma(-step, (float)(steps - i - 1), end);
}

/* bin:60-63 torch.bucketize(v, boundaries) (right = False) followed by clamp(idx - 1, 0, nb - 1): the same binary
 * search as torch's lower bound (the test is !(mid >= v), so a NaN lands in the last bin), over the nb + 1 boundaries. */
LQ_HD int lq_bin_index(float v, const float* boundaries, int nb) {
    int lo = 0, hi = nb + 1;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (!(boundaries[mid] >= v)) lo = mid + 1; else hi = mid;
    }
    int b = lo - 1;
    if (b < 0) b = 0;
    if (b > nb - 1) b = nb - 1;
    return b;
}

#endif /* LIPVQ_MATH_H_ */
