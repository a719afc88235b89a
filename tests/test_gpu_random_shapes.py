"""GPU: seeded random shape sweep of the newer kernels against the canonical oracle (bit-exact) -- ragged rows, odd fan-ins,
non-multiple-of-32 widths, tiny and mid-size batches; complements the hand-picked cases of the per-kernel test files."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(12))
def test_linear_and_mlp3_random_shapes(oracle, seed):
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 2, 31, 33, 100, 777, 4099, 20011]))
    K0 = int(rng.integers(1, 70))
    J0, J1 = int(rng.choice([32, 64, 96, 128])), int(rng.choice([32, 64, 96, 128]))
    J2 = int(rng.integers(1, 230))
    x = rng.standard_normal((N, K0)).astype(np.float32)
    Ws = [(rng.standard_normal((J0, K0)) / np.sqrt(K0)).astype(np.float32), rng.standard_normal(J0).astype(np.float32),
          (rng.standard_normal((J1, J0)) / np.sqrt(J0)).astype(np.float32), rng.standard_normal(J1).astype(np.float32),
          (rng.standard_normal((J2, J1)) / np.sqrt(J1)).astype(np.float32), rng.standard_normal(J2).astype(np.float32)]
    acts = tuple(int(a) for a in rng.choice([O.ACT_NONE, O.ACT_GELU, O.ACT_SIGMOID, O.ACT_RELU], 3))
    y_ref, pre_ref = oracle.mlp3(x, *Ws, acts, save_pre=True)
    y, pre = ops.mlp3(_cuda(x), ops.mlp3_pack(*[_cuda(w) for w in Ws]), acts, save_pre=True)
    assert np.array_equal(y.cpu().numpy(), y_ref)
    for a, b in zip(pre, pre_ref):
        assert np.array_equal(a.cpu().numpy(), b)
    # the single Linear with an activation epilogue, same data
    act = int(rng.choice([O.ACT_NONE, O.ACT_GELU]))
    l_ref = oracle.linear_act(x, Ws[0], Ws[1], act)
    assert np.array_equal(ops.linear(_cuda(x), _cuda(Ws[0]), _cuda(Ws[1]), act=act).cpu().numpy(), l_ref)


@pytest.mark.parametrize("seed", range(10))
def test_embed_rows_random_shapes(oracle, seed):
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(2000 + seed)
    B, T = int(rng.integers(1, 40)), int(rng.integers(1, 23))
    E = 4 * int(rng.integers(1, 257))
    K = int(rng.integers(1, 300))
    S = int(rng.integers(1, 4))                      # streams sharing the output rows
    slot = int(rng.integers(0, S))
    table = rng.standard_normal((K, E)).astype(np.float32)
    pos = rng.standard_normal((T, E)).astype(np.float32) if rng.random() < 0.8 else None
    w, b = rng.standard_normal(E).astype(np.float32), rng.standard_normal(E).astype(np.float32)
    idx = rng.integers(0, K, B * T).astype(np.int64)
    N = B * T - (int(rng.integers(0, T)) if B > 1 and rng.random() < 0.3 else 0)      # sometimes a ragged last batch entry
    ref = np.full((B, S * T, E), 3.0, np.float32)
    st_ref = oracle.embed_rows(table, idx[:N], pos, w, b, 1e-5, ref, T, S * T * E, S * E, slot * E, N=N, want_stats=True)
    out = torch.full((B, S * T, E), 3.0, device="cuda")
    st = ops.embed_rows(_cuda(table), _cuda(idx[:N]), None if pos is None else _cuda(pos), _cuda(w), _cuda(b), 1e-5, out, N, T,
                        S * T * E, S * E, slot * E, want_stats=True)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(st.cpu().numpy(), st_ref)


@pytest.mark.parametrize("seed", range(8))
def test_bin_kernels_random_shapes(oracle, seed):
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(3000 + seed)
    N, A = int(rng.choice([1, 5, 64, 1000, 9001])), int(rng.integers(1, 16))
    nb, D = int(rng.integers(1, 40)), int(rng.integers(1, 220))
    bp = O.make_bin_params(seed, A, D, nb)
    x = (rng.standard_normal((N, A)) * rng.uniform(0.1, 5)).astype(np.float32)
    if N > 4:
        x[:, 0] = x[0, 0]                            # a constant column: degenerate boundaries
    rmin, rmax = oracle.bin_minmax(x[: max(1, N // 2)], np.full(A, np.inf, np.float32), np.full(A, -np.inf, np.float32))
    tmin, tmax = _cuda(np.full(A, np.inf, np.float32)), _cuda(np.full(A, -np.inf, np.float32))
    ops.bin_minmax(_cuda(x[: max(1, N // 2)]), tmin, tmax)
    assert np.array_equal(tmin.cpu().numpy(), rmin) and np.array_equal(tmax.cpu().numpy(), rmax)
    bins = ops.bin_discretize(_cuda(x), tmin, tmax, nb)
    bins_ref = oracle.bin_discretize(x, rmin, rmax, nb)
    assert np.array_equal(bins.cpu().numpy(), bins_ref)
    P_ref = oracle.bin_table(bp)
    h_ref = oracle.bin_hidden(bins_ref, P_ref, bp["output_layer.0.bias"])
    h = ops.bin_hidden(bins, _cuda(P_ref), _cuda(bp["output_layer.0.bias"]))
    assert np.array_equal(h.cpu().numpy(), h_ref)


@pytest.mark.parametrize("seed", range(10))
def test_tokenize_random_shapes(oracle, seed):
    """Fused tokenize (parity mode) and the screened quantizer against the oracle on random N / A / D / K / input scales."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    rng = np.random.default_rng(4000 + seed)
    N = int(rng.choice([1, 31, 257, 2049, 5000, 30011]))
    A, D = int(rng.integers(1, 17)), int(rng.choice([32, 64, 128]))
    K = int(rng.integers(2, 3000))
    p = O.make_params(seed, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    x = (O.make_inputs(seed, N, A) * rng.choice([0.01, 1.0, 30.0])).astype(np.float32)
    ze_ref = oracle.llfq_encode(p, x)
    idx_ref, zq_ref, usage_ref = oracle.nearest(ze_ref, p["quantizer.codebook"])
    model.code_usage.zero_()
    idx, zq = model.tokenize(_cuda(x))
    assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(zq.cpu().numpy(), zq_ref)
    assert np.array_equal(model.code_usage.cpu().numpy(), usage_ref)
    idx2, _ = model._quantize(model.encode(_cuda(x)), None)            # unfused: mlp3_wg + exact rows / screen
    assert np.array_equal(idx2.cpu().numpy(), idx_ref)


@pytest.mark.parametrize("hidden,N", [(256, 300), (32, 77), (192, 1000)])
def test_module_other_hidden_widths(oracle, hidden, N):
    """hidden_dim beyond the reference default: forward values and parameter gradients against the oracle."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K = 9, 48, 200
    p = O.make_params(hidden, A, D, K, hidden=hidden, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K, hidden_dim=hidden).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    x = O.make_inputs(N, N, A)
    f = oracle.llfq_forward(p, x)
    z, loss = model(_cuda(x))
    assert np.array_equal(z.detach().cpu().numpy(), f["z_q"]) and np.array_equal(model.last_indices.cpu().numpy(), f["indices"])
    assert abs(loss.item() - f["loss"]) <= 1e-5 * abs(f["loss"])
    loss.backward()
    g = oracle.llfq_grads(p, x, fwd=f)
    for k, prm in model.named_parameters():
        scale = np.abs(g[k]).max() + 1e-12
        assert np.abs(prm.grad.cpu().numpy() - g[k]).max() <= 2e-5 * scale, k


@pytest.mark.parametrize("A,D,K,N", [(64, 64, 512, 3000), (33, 32, 100, 700), (17, 128, 300, 1500)])
def test_tokenize_wide_actions(oracle, A, D, K, N):
    """Action widths up to the fused kernel's limit (A <= 64): parity mode against the oracle, fast mode well-formed."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(A + D, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    x = O.make_inputs(A, N, A)
    idx_ref, zq_ref, _ = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    idx, zq = model.tokenize(_cuda(x), count_usage=False)
    assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(zq.cpu().numpy(), zq_ref)
    idx_f, zq_f = model.tokenize(_cuda(x), count_usage=False, mode="fast")
    assert (idx_f.cpu().numpy() != idx_ref).mean() <= 0.01
    assert torch.equal(zq_f, model.quantizer.codebook.detach()[idx_f])
