"""CPU: pin the oracle against the golden vectors the REFERENCE produced (oracle/gen_golden.py).

 - oracle.torch_* (the restatement bench.py times) must reproduce the reference's outputs: exactly when
   torch build and CPU capability equal the fixture's, else within 1e-6;
 - the canonical C oracle: indices identical (a mismatch is tolerated only on a row whose reference
   top-2 distance gap is below fp32 noise), floats within 1e-5; its quantizer fed the reference's own
   z_e must be EXACT (the distance arithmetic is torch's, bit for bit)."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
TOL = 1e-5
LLFQ = sorted(Path(p).stem for p in glob.glob(str(GOLD / "llfq_*.npz"))
              if "nearest_edge" not in p and "nearties" not in p and "neartri" not in p and "_train_" not in p and not p.endswith("_big.npz"))
NEARTIES = sorted(Path(p).stem for p in glob.glob(str(GOLD / "llfq_nearties_*.npz")) + glob.glob(str(GOLD / "llfq_neartri_*.npz")))
VQ_NEARTIES = sorted(Path(p).stem for p in glob.glob(str(GOLD / "vq_nearties_*.npz")))
BIG = sorted(Path(p).stem for p in glob.glob(str(GOLD / "llfq_*_big.npz")))
NEAR_TIE = 1e-6      # relative top-2 distance gap (in the reference's own fp32 distances) below which an index may differ
VQ = sorted(Path(p).stem for p in glob.glob(str(GOLD / "vq_*.npz")) if "nearties" not in p)


def _meta(g):
    return dict(eval(str(g["meta"])))


def _same_platform(g):
    m = _meta(g)
    return m["torch_version"] == torch.__version__ and m["cpu_capability"] == torch.backends.cpu.get_cpu_capability()


def _check_indices(got, g):
    ref = g["indices"].astype(np.int64)
    bad = np.nonzero(got != ref)[0]
    if bad.size:
        gap = (g["d_second"][bad] - g["d_best"][bad]) / np.maximum(g["d_second"][bad], 1e-30)
        assert (gap < 1e-5).all(), f"{bad.size} index mismatches that are not near-ties (gaps {gap})"


def test_fixture_inventory():
    assert len(LLFQ) >= 9 and len(VQ) >= 3 and (GOLD / "llfq_nearest_edge.npz").exists()
    assert {"llfq_cfg2_big", "llfq_cfg3_big", "llfq_icrt_big"} <= set(BIG)


def big_fixture(name, oracle):
    """(params, x, reference indices int64, relative top-2 gap per row) of a full-size fixture the REFERENCE produced
    (oracle/gen_golden.py::run_llfq_big).  Shared with tests/test_gpu_big_parity.py."""
    g = np.load(GOLD / f"{name}.npz")
    A, D, K, N = int(g["A"]), int(g["D"]), int(g["K"]), int(g["N"])
    p = O.make_params(int(g["seed"]), A, D, K, regime="trained", oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"]), "parameter generator drifted from the fixture"
    x = O.make_inputs(int(g["seed"]), N, A)
    gap = (g["d_second"] - g["d_best"]) / np.maximum(g["d_second"], 1e-30)
    return p, x, g["indices"].astype(np.int64), gap


def assert_indices_match_reference(got, ref, gap, what):
    """Exact, except rows whose REFERENCE top-2 relative gap is below NEAR_TIE; returns (mismatches, near-tie rows)."""
    bad = np.nonzero(got != ref)[0]
    near = int((gap < NEAR_TIE).sum())
    assert (gap[bad] < NEAR_TIE).all(), (f"{what}: {bad.size} index mismatches vs the reference, "
                                         f"{int((gap[bad] >= NEAR_TIE).sum())} of them not near-ties (gaps {gap[bad][:8]})")
    return int(bad.size), near


@pytest.mark.parametrize("name", BIG)
def test_canonical_oracle_vs_reference_full_size(name, oracle):
    """The canonical C oracle -- what the GPU is held to bit for bit at ANY size -- against the reference itself on
    65 536 / 4 096 / 16 384 rows: indices exact except reference near-ties (none exist in these fixtures)."""
    p, x, ref, gap = big_fixture(name, oracle)
    r_idx, _, _ = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    mism, near = assert_indices_match_reference(r_idx, ref, gap, name)
    print(f"{name}: {mism} mismatches, {near} reference near-tie rows of {ref.size}")


@pytest.mark.parametrize("name", LLFQ)
def test_llfq_oracles_vs_reference(name, oracle):
    g = np.load(GOLD / f"{name}.npz")
    m = _meta(g)
    A, D, K, N = int(g["A"]), int(g["D"]), int(g["K"]), int(g["N"])
    p = O.make_params(int(g["seed"]), A, D, K, regime=m["regime"], oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"])
    x = O.make_inputs(int(g["seed"]), N, A, clamp=m["clamp"])
    # canonical C oracle
    r = oracle.llfq_forward(p, x)
    assert np.abs(r["z_e"] - g["z_e"]).max() <= TOL
    _check_indices(r["indices"], g)
    idx2, _, _ = oracle.nearest(g["z_e"], p["quantizer.codebook"])
    assert np.array_equal(idx2, g["indices"].astype(np.int64)), "quantizer on the reference's z_e must be exact"
    assert np.array_equal(r["z_latent"], p["quantizer.codebook"][r["indices"]])
    if "loss" in g.files:
        assert np.abs(r["x_recon"] - g["x_recon"]).max() <= TOL
        for k in ("recon_loss", "commitment_loss", "loss"):
            assert abs(r[k] - float(g[k])) <= TOL * abs(float(g[k])), k
    # torch-CPU restatement
    torch.set_num_threads(1)
    chunk = 256 if K * D <= 65536 else 32
    idx_t, _ = O.torch_llfq_tokenize(O.to_torch(p), torch.from_numpy(x), chunk=chunk)
    if _same_platform(g):
        assert np.array_equal(idx_t.numpy(), g["indices"].astype(np.int64))
    else:
        _check_indices(idx_t.numpy(), g)
    if m["full"]:
        og = oracle.llfq_grads(p, x, fwd=r)
        for k in O.LLFQ_KEYS:
            ref = g["grad/" + k]
            assert np.abs(og[k] - ref).max() <= TOL * max(np.abs(ref).max(), 1e-12), k
        # restatement backward + AdamW (icl.py:887-889, 968-970) reproduces the reference's step
        tp = {k: v.requires_grad_(True) for k, v in O.to_torch(p).items()}
        opt = torch.optim.AdamW(list(tp.values()), lr=1e-3, weight_decay=1e-4)
        _, loss, _ = O.torch_llfq_forward(tp, torch.from_numpy(x))
        loss.backward()
        opt.step()
        for k in O.LLFQ_KEYS:
            tol = 0 if _same_platform(g) else 1e-6
            assert np.abs(tp[k].detach().numpy() - g["post/" + k]).max() <= tol, k


@pytest.mark.parametrize("name", VQ)
def test_vq_oracles_vs_reference(name, oracle):
    g = np.load(GOLD / f"{name}.npz")
    m = _meta(g)
    A, D, K, N = int(g["A"]), int(g["D"]), int(g["K"]), int(g["N"])
    p = O.make_params(int(g["seed"]), A, D, K, regime=m["regime"], variant="vq", oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"])
    x = O.make_inputs(int(g["seed"]), N, A)
    r = oracle.vq_forward(p, x)
    _check_indices(r["indices"], g)
    idx2, _, _ = oracle.nearest(g["z_e"], p["embedding.weight"], O.DIST_SQSUM)
    assert np.array_equal(idx2, g["indices"].astype(np.int64))
    assert np.abs(r["z_latent"] - g["z_latent"]).max() <= TOL
    assert np.abs(r["x_recon"] - g["x_recon"]).max() <= TOL
    assert abs(r["loss"] - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    og = oracle.vq_grads(p, x, fwd=r)
    for k in O.VQ_KEYS:
        ref = g["grad/" + k]
        assert np.abs(og[k] - ref).max() <= TOL * max(np.abs(ref).max(), 1e-12), k
    z_t, loss_t, _ = O.torch_vq_forward(O.to_torch(p), torch.from_numpy(x))
    tol = 0 if _same_platform(g) else 1e-6
    assert np.abs(z_t.numpy() - g["z_latent"]).max() <= tol


def test_quantizer_edge_cases_exact(oracle):
    """Duplicate codes, zero distance, sign mask, and two squares sharing one fp32 square root."""
    g = np.load(GOLD / "llfq_nearest_edge.npz")
    idx, zq, _ = oracle.nearest(g["z_e"], g["codebook"])
    assert np.array_equal(idx, g["indices"].astype(np.int64))
    assert np.array_equal(zq, g["z_q"])
    assert np.array_equal(oracle.distances(g["z_e"], g["codebook"]), g["distances"]), "distances must be torch's, bitwise"
    assert idx[0] == 7          # rows equal to duplicated codes 7/40/41 -> lowest index
    assert idx[70] == 50        # sqrt merge: code 51 has the smaller square, but both share one fp32 root -> lower index
    d = g["distances"]
    assert d[70, 50] == d[70, 51]


@pytest.mark.parametrize("name", NEARTIES)
def test_adversarial_near_ties_exact(name, oracle):
    """Rows on the bisector of two codes (+- k * 1e-8), D = 64/128/208, K up to 8192: every row is a near-tie (hundreds are
    exact fp32 ties), so only the reference's exact arithmetic -- 8-accumulator sum, sqrt, first minimum -- decides them.
    The canonical quantizer must reproduce the REFERENCE's index on every row."""
    g = np.load(GOLD / f"{name}.npz")
    # (llfq_neartri_*: round 4 -- a third code within the one-product screen's margin, oracle.make_neartie3_case)
    make = O.make_neartie3_case if "neartri" in name else O.make_neartie_case
    z, cb = make(int(g["seed"]), int(g["N"]), int(g["K"]), int(g["D"]))
    idx, _, _ = oracle.nearest(z, cb)
    assert np.array_equal(idx, g["indices"].astype(np.int64))
    # ... and its two smallest distances are the reference's, bit for bit (a wrong summation order shows here first)
    d = np.sort(oracle.distances(z, cb), axis=1)[:, :2]
    assert np.array_equal(d[:, 0], g["d_best"]) and np.array_equal(d[:, 1], g["d_second"])
    assert len(NEARTIES) >= 8 and {"llfq_nearties_d7_k512", "llfq_nearties_d20_k512", "llfq_nearties_d37_k512",
                                   "llfq_nearties_d100_k512", "llfq_nearties_d203_k512"} <= set(NEARTIES)


@pytest.mark.parametrize("name", VQ_NEARTIES)
def test_vq_adversarial_near_ties_exact(name, oracle):
    """The plain VQVAE's rule -- `(z_e.unsqueeze(1) - E).pow(2).sum(-1)` then argmin, vq:57-63 -- on bisector rows at
    D = 7 / 20 / 64 / 100 / 203, decided by the REFERENCE's quantize(): torch's cascade sum adds the scalar tail BEFORE the
    eight lanes, and below eight columns takes its scalar path; lq_sqdist32 restates both."""
    g = np.load(GOLD / f"{name}.npz")
    z, cb = O.make_neartie_case(int(g["seed"]), int(g["N"]), int(g["K"]), int(g["D"]))
    idx, _, _ = oracle.nearest(z, cb, dist=O.DIST_SQSUM)
    assert np.array_equal(idx, g["indices"].astype(np.int64))
    d = np.sort(oracle.distances(z, cb, dist=O.DIST_SQSUM), axis=1)[:, :2]
    assert np.array_equal(d[:, 0], g["d_best"]) and np.array_equal(d[:, 1], g["d_second"])
    assert len(VQ_NEARTIES) >= 5


def test_icrt_training_steps_k1024_restatement_vs_reference(oracle):
    """BASELINE config 5's tokenizer step at the real shape (A = 12, D = 208, K = 1024, N = 80): three zero_grad / forward /
    backward / AdamW iterations by the torch restatement reproduce what the REFERENCE module produced (losses, indices,
    first-step gradients, final parameters)."""
    g = np.load(GOLD / "llfq_icrt_train_k1024.npz")
    A, D, K, N, steps, seed = (int(g[k]) for k in ("A", "D", "K", "N", "steps", "seed"))
    p = O.make_params(seed, A, D, K, regime="trained", oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"])
    torch.set_num_threads(1)
    tp = {k: v.requires_grad_(True) for k, v in O.to_torch(p).items()}
    opt = torch.optim.AdamW(list(tp.values()), lr=1e-3, weight_decay=1e-4)
    tol = 0 if _same_platform(g) else 1e-6
    for st in range(steps):
        x = torch.from_numpy(O.make_inputs(seed + st, N, A))
        opt.zero_grad()
        _, loss, ex = O.torch_llfq_forward(tp, x)
        loss.backward()
        assert abs(loss.item() - float(g[f"loss{st}"])) <= max(tol, 1e-7) * abs(float(g[f"loss{st}"]))
        assert np.array_equal(ex["indices"].numpy(), g[f"indices{st}"].astype(np.int64))
        if st == 0:
            for k in O.LLFQ_KEYS:
                ref = g["grad0/" + k]
                got = tp[k].grad.numpy()
                got = got[g["grad0_rows"]] if k == "quantizer.codebook" else got
                assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()) + (0 if tol == 0 else 1e-9), k
        opt.step()
    for k in O.LLFQ_KEYS:
        got = tp[k].detach().numpy()
        got = got[g["post_rows"]] if k == "quantizer.codebook" else got
        assert np.abs(got - g["post/" + k]).max() <= tol, k
