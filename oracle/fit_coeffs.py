#!/usr/bin/env python3
"""Development tool (not shipped in any product path): derives the polynomial
coefficients used by lipvq-vae_amd/csrc/lipvq_math.h.

The canonical math header implements expf/erff/log1pf with nothing but
+,-,*,/,fmaf and integer bit manipulation so that gcc (oracle) and hipcc
(gfx950 device code) produce bit-identical results.  The polynomials are
near-minimax fits computed here in float64 (Chebyshev-node least squares
followed by a few Remez-style exchanges) against scipy's reference functions.

Run:  python oracle/fit_coeffs.py      (prints C initialisers)
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy import special


def remez_like(f, lo, hi, deg, weight=None, iters=30, n=4000):
    """Near-minimax polynomial (power basis on the raw variable) for f on [lo,hi]."""
    xs = np.cos(np.pi * (np.arange(n) + 0.5) / n) * (hi - lo) / 2 + (hi + lo) / 2
    w = np.ones_like(xs) if weight is None else weight(xs)
    # iteratively re-weighted least squares (Lawson) -> minimax
    lw = np.ones_like(xs)
    best = None
    for _ in range(iters):
        t = (2 * xs - (hi + lo)) / (hi - lo)
        V = C.chebvander(t, deg)
        sw = np.sqrt(lw) * w
        coef, *_ = np.linalg.lstsq(V * sw[:, None], f(xs) * sw, rcond=None)
        err = np.abs(V @ coef - f(xs)) * w
        m = err.max()
        if best is None or m < best[0]:
            best = (m, coef.copy())
        lw = lw * (err / err.mean() + 1e-30)
        lw /= lw.sum()
    m, coef = best
    # convert Chebyshev-on-[-1,1] to power basis in x
    cheb_poly = C.Chebyshev(coef, domain=[lo, hi])
    pw = cheb_poly.convert(kind=P.Polynomial, domain=[-1, 1], window=[-1, 1])
    return pw.coef, m


def cfmt(name, coefs):
    body = ", ".join(f"{float(np.float32(c))!r}f".replace("e-0", "e-").replace("e+0", "e+") for c in coefs)
    print(f"static const float {name}[{len(coefs)}] = {{ {body} }};")


DEG_S, DEG_P = 12, 8

if __name__ == "__main__":
    # exp(r) on [-ln2/2, ln2/2]:  exp(r) = 1 + r + r^2 * q(r)
    L = np.log(2) / 2 * 1.0001
    q = lambda r: np.where(np.abs(r) < 1e-8, 0.5, (np.expm1(r) - r) / np.where(r == 0, 1, r * r))
    ce, me = remez_like(q, -L, L, 5)
    print("// exp: q(r) deg5, max abs err of q:", me)
    cfmt("LQ_EXP_Q", ce)

    # erf main range: erf(a) = a * s(u), u = a^2/4.5 - 1 in [-1, 1] (|a| < 3); the shifted variable keeps
    # the fp32 Horner evaluation well conditioned (max abs error 1.6e-7 including rounding)
    half = 4.5
    s = lambda u: np.where((u * half + half) < 1e-16, 2 / np.sqrt(np.pi),
                           special.erf(np.sqrt(np.maximum(u * half + half, 0))) / np.sqrt(np.maximum(u * half + half, 1e-300)))
    cs, ms = remez_like(s, -1.0, 1.0, DEG_S, weight=lambda u: np.sqrt(np.maximum(u * half + half, 1e-12)))
    print("// erf main: s(u), weighted max abs err:", ms)
    cfmt("LQ_ERF_S", cs)

    # erf large: erfc(a) = exp(-p(a)), a in [1, 4.0]
    p = lambda a: -np.log(special.erfc(a))
    # weight: abs error of erfc = erfc * err(p)  -> weight = erfc / erfc(1)
    cl, ml = remez_like(p, 1.0, 4.0, DEG_P, weight=lambda a: special.erfc(a) / special.erfc(1.0) + 1e-3)
    print("// erf large: p(a), weighted max err:", ml)
    cfmt("LQ_ERF_P", cl)

    # log(m) on m in [sqrt(.5), sqrt(2)): s=(m-1)/(m+1), log m = 2s + s^3 * r(s^2)
    smax = (np.sqrt(2) - 1) / (np.sqrt(2) + 1)
    r = lambda z: sum(2.0 * z ** k / (2 * k + 3) for k in range(24))  # series of (log((1+s)/(1-s)) - 2s)/s^3, z = s^2
    cr, mr = remez_like(r, 0.0, smax * smax * 1.001, 4)
    print("// log: r(z) deg4, max abs err:", mr)
    cfmt("LQ_LOG_R", cr)
