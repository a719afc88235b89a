"""CPU: the canonical oracle of the embedding stage (oracle/lipvq_oracle.c lq_ref_linear / lq_ref_embed_rows) against
the committed fixtures of the reference's op sequence (obs_nets.py:2525-2543, 2580-2596; tests/golden/embed_*.npz,
made by oracle/gen_golden.py --only-embed) and against its own torch restatement.  Floating point: |diff| <=
1e-5 * (1 + |ref|) (the tolerance north_star states for floats); layout/interleave checks are exact."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

CASES = {"embed_parameter": "parameter", "embed_embedding": "embedding", "embed_sinusoidal": "sinusoidal"}


def _close(got, ref):
    return np.all(np.abs(got.astype(np.float64) - ref) <= 1e-5 * (1.0 + np.abs(ref)))


def _load(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    meta = dict(eval(str(g["meta"])))
    ep = O.make_embed_params(meta["seed"], meta["Din"], meta["E"], meta["T"], meta["mode"])
    assert O.params_digest(ep) == str(g["digest"]), "seeded parameter generator drifted"
    return g, meta, ep


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(oracle, golden_dir, name):
    g, meta, ep = _load(golden_dir, name)
    got = oracle.transformer_embeddings(ep, g["obs"], g["context_obs"], g["codebook"], g["indices"].astype(np.int64))
    assert got.shape == g["embeddings"].shape == (meta["B"], 3 * meta["T"], meta["E"])
    assert _close(got, g["embeddings"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_torch_restatement_matches_golden_bitwise(golden_dir, name):
    g, meta, ep = _load(golden_dir, name)
    tp = O.to_torch(ep)
    acts = torch.from_numpy(g["codebook"][g["indices"].astype(np.int64)])
    out = O.torch_transformer_embeddings(tp, torch.from_numpy(g["obs"]), torch.from_numpy(g["context_obs"]), acts)
    assert np.array_equal(out.numpy(), g["embeddings"])


def test_table_path_equals_dense_path(oracle):
    """Linear(codebook)[idx] == Linear(codebook[idx]) bit for bit: the identity the table design rests on."""
    rng = np.random.default_rng(5)
    cb = rng.uniform(0, 1, (97, 33)).astype(np.float32)           # odd fan-in: the zero pad term
    ep = O.make_embed_params(9, 33, 64, 4, "embedding")
    idx = rng.integers(0, 97, 50)
    W, b = ep["embed_encoder.weight"], ep["embed_encoder.bias"]
    assert np.array_equal(oracle.linear(cb, W, b)[idx], oracle.linear(cb[idx], W, b))


def test_linear_is_the_mlp_chain(oracle):
    """lq_ref_linear is the same k-ordered chain as a layer of lq_ref_mlp3 (one definition of Linear on the path)."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((40, 12)).astype(np.float32)
    W0, b0 = rng.standard_normal((32, 12)).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    W1, b1 = np.eye(32, dtype=np.float32), np.zeros(32, np.float32)
    _, pre = oracle.mlp3(x, W0, b0, W1, b1, W1, b1, (O.ACT_NONE,) * 3, save_pre=True)
    assert np.array_equal(pre[0], oracle.linear(x, W0, b0))


def test_interleave_slots_and_bad_index(oracle):
    E, T, B, K = 8, 3, 2, 5
    rng = np.random.default_rng(7)
    table = rng.standard_normal((K, E)).astype(np.float32)
    w, b = np.ones(E, np.float32), np.zeros(E, np.float32)
    out = np.full((B, 3 * T, E), 7.0, np.float32)
    idx = np.array([0, 1, 2, 3, 4, 99], np.int64)                  # 99 is out of range
    stats = oracle.embed_rows(table, idx, None, w, b, 1e-5, out, T, 3 * T * E, 2 * E, E, want_stats=True)
    assert np.all(out[:, 0::2][:, :T] == 7.0) and np.all(out[:, 2 * T:] == 7.0)     # other streams' slots untouched
    rows = out[:, 1:2 * T:2].reshape(B * T, E)
    assert np.all(np.isnan(rows[5])) and np.all(np.isnan(stats[5]))
    ref = (table[:5] - table[:5].mean(1, keepdims=True)) / np.sqrt(table[:5].var(1, keepdims=True) + 1e-5)
    assert _close(rows[:5], ref)
    assert _close(stats[:5, 0], table[:5].mean(1))


def test_layernorm_degenerate_rows(oracle):
    """A constant row has zero variance: the output is exactly ln_b (0 * rstd * w + b), no NaN."""
    E = 256
    src = np.full((2, E), 3.25, np.float32)
    w = np.linspace(0.5, 1.5, E).astype(np.float32)
    b = np.linspace(-1, 1, E).astype(np.float32)
    out = np.empty((2, 1, E), np.float32)
    oracle.embed_rows(src, None, None, w, b, 1e-5, out, 1, E, E, 0)
    assert np.array_equal(out[0, 0], b) and np.array_equal(out[1, 0], b)
