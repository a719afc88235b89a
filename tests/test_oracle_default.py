"""CPU: the default action branch (reference obs_nets.py:1244-1260).  The oracle's explicit-op restatement against the
fixtures the stock torch modules produced (oracle/gen_golden.py::run_default_branch), and the host-side drop-in facts of
DefaultActionNetwork (same state_dict keys, shapes and constructor RNG consumption as the reference's nn.Sequential)."""
import hashlib
import warnings

import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

CASES = ["default_icrt", "default_a7_d64"]


def _load(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    seed, A, D, N = (int(g[k]) for k in ("seed", "A", "D", "N"))
    p = O.make_default_branch_params(seed, A, D)
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    assert h.hexdigest() == str(g["params_sha256"]), "seeded parameters drifted from the ones the fixture was made with"
    return g, p, A, D, N


@pytest.mark.parametrize("name", CASES)
def test_restatement_matches_stock_modules(golden_dir, name):
    torch.set_num_threads(1)
    g, p, A, D, N = _load(golden_dir, name)
    y, _, _ = O.torch_default_branch(p, g["x"])
    scale = np.abs(g["y"]).max()
    assert np.abs(y.detach().numpy() - g["y"]).max() <= 1e-5 * scale
    yt, uv, P = O.torch_default_branch(p, g["x"], training=True)
    assert np.abs(yt.detach().numpy() - g["y_train"]).max() <= 1e-5 * np.abs(g["y_train"]).max()
    (yt * torch.from_numpy(g["r"])).sum().backward()
    for k in [k for k in g.files if k.startswith("gdig/")]:
        want, got = g[k], O.grad_digest(P[k[5:]].grad.numpy())
        assert np.abs(got - want).max() <= 1e-4 * max(1e-6, want[1]), k          # relative to the gradient's L2 norm
    for i in (0, 2, 4):
        assert np.allclose(uv[i][0], g[f"train_u/{i}"], rtol=0, atol=1e-6) and np.allclose(uv[i][1], g[f"train_v/{i}"], rtol=0, atol=1e-6)


def test_constructor_matches_the_reference_sequential():
    """Same keys, shapes and values under one seed: the module consumes the RNG exactly as the reference's constructor text."""
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.default_branch import DefaultActionNetwork
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(77)
        ours = DefaultActionNetwork(12, 208)
        torch.manual_seed(77)
        ref = O.build_default_branch_modules(12, 208)
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys()) and len(a) == 62
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    ours.load_state_dict(b)                                   # a reference checkpoint of this branch loads unchanged
    assert [n for n, _ in ours.named_parameters()] == [n for n, _ in ref.named_parameters()]


def test_cpu_input_is_refused():
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.default_branch import DefaultActionNetwork
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DefaultActionNetwork(7, 64)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 7))
    with pytest.raises(ValueError):
        DefaultActionNetwork(7, 60)                           # 60 is not a multiple of 8 heads
