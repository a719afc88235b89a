"""CPU: the canonical oracle of the sibling `bin_enabled` tokenizer (AdaptiveBinActionEmbedding, reference
robomimic/models/bin_action/backbone.py) against fixtures produced by the REFERENCE CLASS ITSELF
(tests/golden/bin_*.npz, oracle/gen_golden.py --only-bin).  Bin indices and running statistics: bit-exact.
Floats: |diff| <= 1e-5 * (1 + |ref|)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

NAMES = ["bin_icrt", "bin_a7", "bin_nb5"]


def _close(got, ref):
    return np.all(np.abs(got.astype(np.float64) - ref) <= 1e-5 * (1.0 + np.abs(ref)))


def load_bin(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    meta = dict(eval(str(g["meta"])))
    bp = O.make_bin_params(meta["seed"], meta["A"], meta["D"], meta["nb"])
    assert O.params_digest(bp) == str(g["digest"]), "seeded parameter generator drifted"
    return g, meta, bp


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(oracle, golden_dir, name):
    g, meta, bp = load_bin(golden_dir, name)
    A = meta["A"]
    rmin, rmax = np.full(A, np.inf, np.float32), np.full(A, -np.inf, np.float32)
    for step in range(3):
        r = oracle.bin_forward(bp, g[f"x{step}"], rmin, rmax, update=step < 2)
        rmin, rmax = r["running_min"], r["running_max"]
        assert np.array_equal(rmin, g[f"rmin{step}"]) and np.array_equal(rmax, g[f"rmax{step}"])
        assert np.array_equal(r["bins"].T, g[f"bins{step}"])                     # integer work: bit-exact
        assert _close(r["out"], g[f"out{step}"])
    # step 2 ran with frozen statistics on wider-range actions: both edge bins must be populated by clamping
    assert (g["bins2"] == 0).any() and (g["bins2"] == meta["nb"] - 1).any()


@pytest.mark.parametrize("name", NAMES)
def test_torch_restatement_bitwise(golden_dir, name):
    g, meta, bp = load_bin(golden_dir, name)
    tp = O.to_torch(bp)
    rmin, rmax = torch.full((meta["A"],), float("inf")), torch.full((meta["A"],), float("-inf"))
    for step in range(3):
        out, idx, rmin, rmax = O.torch_bin_forward(tp, torch.from_numpy(g[f"x{step}"]), rmin, rmax, update=step < 2)
        assert np.array_equal(out.numpy(), g[f"out{step}"]) and np.array_equal(idx.numpy(), g[f"bins{step}"])


def test_boundaries_are_torch_linspace(oracle):
    rng = np.random.default_rng(0)
    for nb in (1, 2, 5, 20, 64, 255):
        lo = rng.standard_normal(9).astype(np.float32)
        hi = (lo + np.abs(rng.standard_normal(9)).astype(np.float32) * 3).astype(np.float32)
        hi[0] = lo[0]                                                         # degenerate: constant column
        _, bd = oracle.bin_discretize(np.zeros((1, 9), np.float32), lo, hi, nb, want_boundaries=True)
        want = np.stack([torch.linspace(torch.tensor(a), torch.tensor(b), nb + 1).numpy() for a, b in zip(lo, hi)])
        assert np.array_equal(bd, want)


def test_bucketize_edges(oracle):
    """Values exactly on boundaries, outside the range, and a constant column, against torch.bucketize + clamp."""
    nb = 20
    lo, hi = np.array([-1.0, 2.0], np.float32), np.array([1.0, 2.0], np.float32)
    _, bd = oracle.bin_discretize(np.zeros((1, 2), np.float32), lo, hi, nb, want_boundaries=True)
    col0 = np.concatenate([bd[0], np.nextafter(bd[0], np.float32(9)), np.nextafter(bd[0], np.float32(-9)), [-5.0, 5.0]]).astype(np.float32)
    col1 = np.resize(np.array([1.0, 2.0, 3.0], np.float32), col0.size)
    x = np.stack([col0, col1], 1)
    bins = oracle.bin_discretize(x, lo, hi, nb)
    for i in range(2):
        want = torch.clamp(torch.bucketize(torch.from_numpy(x[:, i].copy()), torch.from_numpy(bd[i])) - 1, 0, nb - 1).numpy()
        assert np.array_equal(bins[i], want)


def test_table_sum_equals_concat_linear_within_tolerance(oracle):
    """b1 + sum_i P[i][bin_i] vs the k-ordered chain over the concatenated embeddings: same value to fp32 rounding."""
    bp = O.make_bin_params(5, 4, 16)
    rng = np.random.default_rng(1)
    bins = rng.integers(0, 20, (4, 50))
    h, pre = oracle.bin_hidden(bins, oracle.bin_table(bp), bp["output_layer.0.bias"], save_pre=True)
    cat = np.concatenate([bp[f"embedding_layers.{i}.weight"][bins[i]] for i in range(4)], 1)
    ref = oracle.linear(cat, bp["output_layer.0.weight"], bp["output_layer.0.bias"])
    assert _close(pre, ref)
