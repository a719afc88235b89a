"""CPU: accuracy of the canonical transcendental functions (lipvq_math.h) against float64 references.
They must sit within a few ulp of libm, well inside the path's 1e-5 budget."""
import numpy as np
from scipy import special


def _ulp_err(got, ref):
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)


def test_exp(oracle):
    x = np.concatenate([np.linspace(-103, 88.7, 400001), np.linspace(-1, 1, 100001)]).astype(np.float32)
    got = oracle.math_probe(x, 0)
    ref = np.exp(x.astype(np.float64))
    normal = ref > 1.2e-38
    assert _ulp_err(got[normal], ref[normal]).max() < 2.0
    assert np.abs(got[~normal].astype(np.float64) - ref[~normal]).max() < 3e-45
    assert oracle.math_probe(np.array([-200.0, 0.0], np.float32), 0).tolist() == [0.0, 1.0]


def test_erf_gelu(oracle):
    x = np.linspace(-6, 6, 600001).astype(np.float32)
    got = oracle.math_probe(x, 1)
    ref = special.erf(x.astype(np.float64))
    assert np.abs(got - ref).max() < 2.5e-7            # absolute: erf is O(1); measured 1.6e-7
    assert (np.abs(got) <= 1.0).all() and np.array_equal(got, -oracle.math_probe(-x, 1))
    g = oracle.math_probe(x, 2)
    gref = 0.5 * x.astype(np.float64) * (1 + special.erf(x.astype(np.float64) / np.sqrt(2)))
    assert np.abs(g - gref).max() < 8e-7           # measured 6e-7
    gg = oracle.math_probe(x, 5)
    xd = x.astype(np.float64)
    ggref = 0.5 * (1 + special.erf(xd / np.sqrt(2))) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    assert np.abs(gg - ggref).max() < 6e-7


def test_sigmoid_softplus(oracle):
    x = np.linspace(-40, 40, 400001).astype(np.float32)
    s = oracle.math_probe(x, 3)
    sref = special.expit(x.astype(np.float64))
    assert _ulp_err(s, sref)[sref > 1e-30].max() < 3.0
    sp = oracle.math_probe(x, 4)
    spref = np.where(x > 20, x.astype(np.float64), np.log1p(np.exp(x.astype(np.float64))))
    assert _ulp_err(sp, spref)[spref > 1e-30].max() < 3.0
    assert oracle.math_probe(np.array([25.0], np.float32), 4)[0] == 25.0      # threshold = 20 branch
