"""GPU: the sibling `bin_enabled` tokenizer (lipvq_bin_* kernels, lipvq_vae_amd.binning.AdaptiveBinActionEmbedding)
against the canonical oracle (bit-exact: statistics, bin indices AND floats) and against fixtures produced by the
reference class itself (tests/golden/bin_*.npz; indices exact, floats 1e-5, gradients 1e-4 of their scale)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O
from test_oracle_bin import NAMES, _close, load_bin

pytestmark = pytest.mark.gpu


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _module(meta, bp):
    from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding
    m = AdaptiveBinActionEmbedding(meta["A"], meta["D"], num_bins=meta["nb"]).cuda()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in bp.items()}
    sd["running_min"], sd["running_max"] = m.running_min.cpu(), m.running_max.cpu()
    m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("name", NAMES)
def test_module_matches_reference_and_oracle(oracle, golden_dir, name):
    g, meta, bp = load_bin(golden_dir, name)
    m = _module(meta, bp)
    A = meta["A"]
    rmin, rmax = np.full(A, np.inf, np.float32), np.full(A, -np.inf, np.float32)
    for step in range(3):
        x = g[f"x{step}"]
        if step == 2:
            m._update_enabled = False
        with torch.no_grad():
            out = m(_cuda(x))
            idx = m.discretize(_cuda(x))
        ref = oracle.bin_forward(bp, x, rmin, rmax, update=step < 2)
        rmin, rmax = ref["running_min"], ref["running_max"]
        assert np.array_equal(m.running_min.cpu().numpy(), g[f"rmin{step}"])
        assert np.array_equal(m.running_max.cpu().numpy(), g[f"rmax{step}"])
        assert np.array_equal(idx.cpu().numpy(), g[f"bins{step}"])               # the reference's own indices
        assert np.array_equal(m.last_bins.cpu().numpy(), ref["bins"])
        assert _close(out.cpu().numpy(), g[f"out{step}"])                        # the reference's own outputs
        assert np.array_equal(out.cpu().numpy(), ref["out"])                     # canonical oracle: bit for bit
    bd = torch.stack(m.compute_bins()).cpu().numpy()
    _, bd_ref = oracle.bin_discretize(g["x2"], rmin, rmax, meta["nb"], want_boundaries=True)
    assert np.array_equal(bd, bd_ref)


@pytest.mark.parametrize("name", NAMES)
def test_gradients_match_reference_autograd(golden_dir, name):
    g, meta, bp = load_bin(golden_dir, name)
    m = _module(meta, bp)
    with torch.no_grad():
        m(_cuda(g["x0"]))
        m(_cuda(g["x1"]))
    m._update_enabled = False
    out = m(_cuda(g["x2"]))
    assert out.requires_grad and _close(out.detach().cpu().numpy(), g["out2"])
    (out * _cuda(g["R"])).sum().backward()
    for k, p in m.named_parameters():
        want = g["grad/" + k]
        scale = np.abs(want).max() + 1e-12
        err = np.abs(p.grad.cpu().numpy() - want).max() / scale
        assert err < 1e-4, (k, err)


def test_constructor_matches_reference_rng_and_keys(golden_dir):
    """Same submodule construction order as the reference: identical RNG consumption and state_dict keys."""
    from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding
    torch.manual_seed(77)
    m = AdaptiveBinActionEmbedding(3, 16, num_bins=5, embedding_dim=8)
    torch.manual_seed(77)
    embs = [torch.nn.Embedding(5, 8) for _ in range(3)]
    l0, l2 = torch.nn.Linear(24, 12), torch.nn.Linear(12, 16)
    assert all(torch.equal(a.weight, b.weight) for a, b in zip(m.embedding_layers, embs))
    assert torch.equal(m.output_layer[0].weight, l0.weight) and torch.equal(m.output_layer[2].bias, l2.bias)
    assert list(m.state_dict()) == ["running_min", "running_max", "embedding_layers.0.weight", "embedding_layers.1.weight",
                                    "embedding_layers.2.weight", "output_layer.0.weight", "output_layer.0.bias",
                                    "output_layer.2.weight", "output_layer.2.bias"]


def test_update_stops_after_num_step_stop():
    from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding
    m = AdaptiveBinActionEmbedding(2, 8, num_step_stop=2).cuda()
    with torch.no_grad():
        m(torch.tensor([[0.0, 0.0], [1.0, 1.0]], device="cuda"))
        m(torch.tensor([[-1.0, 0.5], [0.5, 2.0]], device="cuda"))
        assert not m._update_enabled
        m(torch.tensor([[-9.0, 9.0]], device="cuda"))
    assert m.running_min.tolist() == [-1.0, 0.0] and m.running_max.tolist() == [1.0, 2.0]


@pytest.mark.parametrize("N,A,D,nb", [(1, 7, 64, 20), (4097, 12, 208, 20), (1000, 1, 32, 3), (70000, 7, 64, 20)])
def test_kernels_bit_exact_vs_oracle(oracle, N, A, D, nb):
    from lipvq_vae_amd import ops
    bp = O.make_bin_params(N + A, A, D, nb)
    x = O.make_inputs(N, N, A)
    x[N // 2:] *= 0.5
    rmin, rmax = np.full(A, np.inf, np.float32), np.full(A, -np.inf, np.float32)
    tmin, tmax = _cuda(rmin), _cuda(rmax)
    half = x[: max(1, N // 2)]                                    # statistics from half the rows: the rest clamps
    ops.bin_minmax(_cuda(half), tmin, tmax)
    rmin, rmax = oracle.bin_minmax(half, rmin, rmax)
    assert np.array_equal(tmin.cpu().numpy(), rmin) and np.array_equal(tmax.cpu().numpy(), rmax)
    bins = ops.bin_discretize(_cuda(x), tmin, tmax, nb)
    bins_ref = oracle.bin_discretize(x, rmin, rmax, nb)
    assert np.array_equal(bins.cpu().numpy(), bins_ref)
    P_ref = oracle.bin_table(bp)
    ed = 64
    W1 = _cuda(bp["output_layer.0.weight"])
    P = torch.stack([ops.linear(_cuda(bp[f"embedding_layers.{i}.weight"]), W1[:, ed * i:ed * (i + 1)].contiguous())
                     for i in range(A)])
    assert np.array_equal(P.cpu().numpy(), P_ref)
    h, pre = ops.bin_hidden(bins, P, _cuda(bp["output_layer.0.bias"]), save_pre=True)
    h_ref, pre_ref = oracle.bin_hidden(bins_ref, P_ref, bp["output_layer.0.bias"], save_pre=True)
    assert np.array_equal(pre.cpu().numpy(), pre_ref) and np.array_equal(h.cpu().numpy(), h_ref)
    y, pre2 = ops.linear(h, _cuda(bp["output_layer.2.weight"]), _cuda(bp["output_layer.2.bias"]), act=ops.ACT_GELU, save_pre=True)
    y_ref, pre2_ref = oracle.linear_act(h_ref, bp["output_layer.2.weight"], bp["output_layer.2.bias"], O.ACT_GELU, save_pre=True)
    assert np.array_equal(pre2.cpu().numpy(), pre2_ref) and np.array_equal(y.cpu().numpy(), y_ref)


def test_full_size_histogram_property(oracle):
    """BASELINE config 2's batch (524 288 actions, A = 7): bins must equal the oracle's on every element and each
    dimension's histogram must sum to N (size-independent checks at full size)."""
    from lipvq_vae_amd import ops
    N, A, nb = 524288, 7, 20
    x = O.make_inputs(99, N, A)
    tmin, tmax = torch.full((A,), float("inf"), device="cuda"), torch.full((A,), float("-inf"), device="cuda")
    xt = _cuda(x)
    ops.bin_minmax(xt, tmin, tmax)
    assert np.array_equal(tmin.cpu().numpy(), x.min(0)) and np.array_equal(tmax.cpu().numpy(), x.max(0))
    bins = ops.bin_discretize(xt, tmin, tmax, nb)
    assert np.array_equal(bins.cpu().numpy(), oracle.bin_discretize(x, x.min(0), x.max(0), nb))
    hist = torch.stack([torch.bincount(bins[i], minlength=nb) for i in range(A)])
    assert (hist.sum(1) == N).all() and int(bins.min()) == 0 and int(bins.max()) == nb - 1
