"""GPU: the embedding stage after the tokenizer (lipvq_linear_f32, lipvq_embed_rows_f32, lipvq_embed_rows_bwd_f32 and
lipvq_vae_amd.embedding.ICLInputEmbedding) against the canonical oracle (bit-exact), the committed fixtures of the
reference's op sequence (obs_nets.py:2525-2543, 2580-2596; 1e-5 tolerance) and torch autograd on the CPU."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=1e-5):
    return np.all(np.abs(got.astype(np.float64) - ref) <= tol * (1.0 + np.abs(ref)))


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N,Kin,E", [(1024, 64, 512), (77, 33, 96), (1, 7, 32), (300, 208, 384), (4096, 512, 512),
                                     (65, 64, 40), (33, 5, 8),
                                     # the 128 x 128-tile kernel (>= 512 workgroups, Kin % 4 == 0): ragged N / E / K chunk
                                     (70000, 64, 512), (33000, 224, 208), (66000, 36, 130), (65536, 33, 512), (140000, 224, 64),
                                     (131100, 8, 33)])
def test_linear_bit_exact(oracle, N, Kin, E):
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(N + Kin)
    x = rng.standard_normal((N, Kin)).astype(np.float32)
    W = (rng.standard_normal((E, Kin)) / np.sqrt(Kin)).astype(np.float32)
    b = rng.standard_normal(E).astype(np.float32)
    y = ops.linear(_cuda(x), _cuda(W), _cuda(b)).cpu().numpy()
    assert np.array_equal(y, oracle.linear(x, W, b))
    y0 = ops.linear(_cuda(x), _cuda(W)).cpu().numpy()
    assert np.array_equal(y0, oracle.linear(x, W, None))


@pytest.mark.parametrize("B,T,E,K", [(3, 10, 512, 1024), (2, 7, 256, 256), (2, 5, 384, 128), (1, 1, 4, 3), (5, 3, 1024, 64),
                                     (700, 10, 512, 1024), (2, 500, 128, 50)])
def test_embed_rows_bit_exact(oracle, B, T, E, K):
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(B * 1000 + T + E)
    table = rng.standard_normal((K, E)).astype(np.float32)
    pos = (0.1 * rng.standard_normal((T, E))).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(E)).astype(np.float32)
    b = (0.1 * rng.standard_normal(E)).astype(np.float32)
    idx = rng.integers(0, K, B * T).astype(np.int64)
    dense = rng.standard_normal((B * T, E)).astype(np.float32)
    ref = np.zeros((B, 3 * T, E), np.float32)
    st_ref = oracle.embed_rows(table, idx, pos, w, b, 1e-5, ref, T, 3 * T * E, 2 * E, E, want_stats=True)
    oracle.embed_rows(dense, None, pos, w, b, 1e-5, ref, T, 3 * T * E, 2 * E, 0)
    oracle.embed_rows(dense, None, None, w, b, 1e-5, ref, T, 3 * T * E, E, 2 * T * E)
    out = torch.zeros((B, 3 * T, E), device="cuda")
    tw, tb, tpos, td = _cuda(w), _cuda(b), _cuda(pos), _cuda(dense)
    st = ops.embed_rows(_cuda(table), _cuda(idx), tpos, tw, tb, 1e-5, out, B * T, T, 3 * T * E, 2 * E, E, want_stats=True)
    ops.embed_rows(td, None, tpos, tw, tb, 1e-5, out, B * T, T, 3 * T * E, 2 * E, 0)
    ops.embed_rows(td, None, None, tw, tb, 1e-5, out, B * T, T, 3 * T * E, E, 2 * T * E)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(st.cpu().numpy(), st_ref)


def test_embed_rows_bad_index_and_errors():
    from lipvq_vae_amd import ops
    from lipvq_vae_amd._capi import LipvqLibraryError
    E, T = 64, 2
    table = torch.randn(5, E, device="cuda")
    w, b = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
    idx = torch.tensor([0, 7, -1, 4], device="cuda")
    out = torch.zeros((2, T, E), device="cuda")
    ops.embed_rows(table, idx, None, w, b, 1e-5, out, 4, T, T * E, E, 0)
    o = out.reshape(4, E)
    assert torch.isnan(o[1]).all() and torch.isnan(o[2]).all() and torch.isfinite(o[0]).all() and torch.isfinite(o[3]).all()
    with pytest.raises(ValueError):
        ops.embed_rows(table, idx, None, w, b, 1e-5, out, 4, T, 3 * T * E, E, 0)       # does not fit
    with pytest.raises(LipvqLibraryError):
        ops.embed_rows(table, idx, None, w, b, 1e-5, torch.zeros(1024, device="cuda"), 4, T, T * E, E + 2, 0)   # stride % 4
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(3, 4), torch.randn(5, 4))                                # CPU tensors: no fallback


def _module(meta, ep):
    from lipvq_vae_amd.embedding import ICLInputEmbedding
    mode = meta["mode"]
    m = ICLInputEmbedding(meta["Din"], meta["E"], meta["T"], emb_dropout=0.1, sinusoidal_embedding=mode == "sinusoidal",
                          nn_parameter_for_timesteps=mode == "parameter").cuda().eval()
    sd = {}
    for k, v in ep.items():
        sd[("params." if k == "embed_timestep" else "nets.") + k] = torch.from_numpy(np.ascontiguousarray(v))
    m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("name", ["embed_parameter", "embed_embedding", "embed_sinusoidal"])
def test_module_matches_golden_and_oracle(oracle, golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    meta = dict(eval(str(g["meta"])))
    ep = O.make_embed_params(meta["seed"], meta["Din"], meta["E"], meta["T"], meta["mode"])
    m = _module(meta, ep)
    idx = g["indices"].astype(np.int64)
    obs, cobs, cb = _cuda(g["obs"]), _cuda(g["context_obs"]), _cuda(g["codebook"])
    with torch.no_grad():
        out_tok = m(obs, cobs, action_indices=_cuda(idx), codebook=cb)
        out_dense = m(obs, cobs, _cuda(g["codebook"][idx]))
        single = m.input_embedding(obs)
        single_tok = m.input_embedding_tokens(_cuda(idx), cb)
    assert torch.equal(out_tok, out_dense)                       # table path == dense path, bit for bit
    assert _close(out_tok.cpu().numpy(), g["embeddings"])        # the reference's op sequence
    want = oracle.transformer_embeddings(ep, g["obs"], g["context_obs"], g["codebook"], idx)
    if meta["mode"] != "sinusoidal":                             # sin/cos table: libm vs torch, tolerance only
        assert np.array_equal(out_tok.cpu().numpy(), want)
    T = meta["T"]
    assert torch.equal(single, out_tok[:, 2 * T:]) and torch.equal(single_tok, out_tok[:, 1:2 * T:2])


def test_state_dict_keys_are_the_references():
    from lipvq_vae_amd.embedding import ICLInputEmbedding
    keys = set(ICLInputEmbedding(64, 512, 10).state_dict())
    assert keys == {"nets.embed_encoder.weight", "nets.embed_encoder.bias", "params.embed_timestep",
                    "nets.embed_ln.weight", "nets.embed_ln.bias"}
    keys = set(ICLInputEmbedding(64, 512, 10, nn_parameter_for_timesteps=False).state_dict())
    assert "nets.embed_timestep.weight" in keys and "params.embed_timestep" not in keys


@pytest.mark.parametrize("mode,tokens", [("parameter", True), ("embedding", True), ("parameter", False), ("sinusoidal", True)])
def test_backward_matches_torch_autograd(mode, tokens):
    """Gradients of sum(out * R) w.r.t. embed_encoder, time embedding, LayerNorm and the dense inputs, against torch
    autograd of the restated op sequence on the CPU (rtol 1e-4 of the gradient's scale: fp32 atomics, other order)."""
    B, T, Din, E, K = 4, 6, 64, 256, 32
    meta = dict(mode=mode, Din=Din, E=E, T=T)
    ep = O.make_embed_params(11, Din, E, T, mode)
    m = _module(meta, ep).train()
    m.nets["embed_drop"].p = 0.0
    rng = np.random.default_rng(3)
    cb = rng.uniform(0, 1, (K, Din)).astype(np.float32)
    idx = rng.integers(0, K, (B, T)).astype(np.int64)
    obs, cobs = rng.standard_normal((B, T, Din)).astype(np.float32), rng.standard_normal((B, T, Din)).astype(np.float32)
    R = rng.standard_normal((B, 3 * T, E)).astype(np.float32)
    # torch CPU reference
    tp = {k: v.clone().requires_grad_(True) for k, v in O.to_torch(ep).items()}
    t_obs, t_cobs = torch.from_numpy(obs).requires_grad_(True), torch.from_numpy(cobs).requires_grad_(True)
    ref = O.torch_transformer_embeddings(tp, t_obs, t_cobs, torch.from_numpy(cb[idx]))
    (ref * torch.from_numpy(R)).sum().backward()
    # HIP path
    g_obs, g_cobs = _cuda(obs).requires_grad_(True), _cuda(cobs).requires_grad_(True)
    if tokens:
        out = m(g_obs, g_cobs, action_indices=_cuda(idx), codebook=_cuda(cb))
    else:
        out = m(g_obs, g_cobs, _cuda(cb[idx]))
    assert _close(out.detach().cpu().numpy(), ref.detach().numpy())
    (out * _cuda(R)).sum().backward()

    def chk(got, want, what):
        want = want.numpy()
        scale = np.abs(want).max() + 1e-12
        err = np.abs(got.cpu().numpy() - want).max() / scale
        assert err < 1e-4, (what, err)

    chk(m.nets["embed_encoder"].weight.grad, tp["embed_encoder.weight"].grad, "W")
    chk(m.nets["embed_encoder"].bias.grad, tp["embed_encoder.bias"].grad, "b")
    chk(m.nets["embed_ln"].weight.grad, tp["embed_ln.weight"].grad, "ln_w")
    chk(m.nets["embed_ln"].bias.grad, tp["embed_ln.bias"].grad, "ln_b")
    if mode == "parameter":
        chk(m.params["embed_timestep"].grad, tp["embed_timestep"].grad, "pos")
    elif mode == "embedding":
        chk(m.nets["embed_timestep"].weight.grad, tp["embed_timestep.weight"].grad, "pos")
    chk(g_obs.grad, t_obs.grad, "obs")
    chk(g_cobs.grad, t_cobs.grad, "context_obs")


def test_tokenizer_to_embedding_end_to_end(oracle):
    """tokenize() -> indices -> embedding, against oracle tokenizer + oracle embedding: the whole hand-over."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, B, T, E = 7, 64, 1024, 16, 10, 512
    p = O.make_params(21, A, D, K, oracle=oracle)
    tok = LLFQVAE_V4(A, D, num_codes=K).cuda()
    tok.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    ep = O.make_embed_params(22, D, E, T, "parameter")
    emb = _module(dict(mode="parameter", Din=D, E=E, T=T), ep)
    x = O.make_inputs(23, B * T, A)
    rng = np.random.default_rng(24)
    obs, cobs = rng.standard_normal((B, T, D)).astype(np.float32), rng.standard_normal((B, T, D)).astype(np.float32)
    with torch.no_grad():
        idx, _ = tok.tokenize(_cuda(x), count_usage=False)
        out = emb(_cuda(obs), _cuda(cobs), action_indices=idx.view(B, T), codebook=tok.quantizer.codebook)
    idx_ref, _, _ = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    want = oracle.transformer_embeddings(ep, obs, cobs, p["quantizer.codebook"], idx_ref.reshape(B, T))
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(out.cpu().numpy(), want)


def test_full_size_gather_property():
    """BASELINE config 2's batch (524 288 actions), E = 512: every output row must equal, bit for bit, the row the
    same kernel produces for its (code, timestep) pair in a K*T-row launch (size-independent property)."""
    from lipvq_vae_amd import ops
    N, T, E, K = 524288, 8, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    table = torch.randn((K, E), device="cuda", generator=g)
    pos = 0.1 * torch.randn((T, E), device="cuda", generator=g)
    w = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    b = 0.1 * torch.randn(E, device="cuda", generator=g)
    idx = torch.randint(0, K, (N,), device="cuda", generator=g)
    out = torch.empty((N // T, T, E), device="cuda")
    ops.embed_rows(table, idx, pos, w, b, 1e-5, out, N, T, T * E, E, 0)
    combos = torch.empty((K, T, E), device="cuda")
    all_idx = torch.arange(K, device="cuda").repeat_interleave(T)
    ops.embed_rows(table, all_idx, pos, w, b, 1e-5, combos, K * T, T, T * E, E, 0)
    t = torch.arange(N, device="cuda") % T
    assert torch.equal(out.view(N, E), combos.view(K * T, E)[idx * T + t])


@pytest.mark.parametrize("B,T,E,K,kind", [(3300, 10, 512, 1024, "uniform"), (4100, 8, 256, 37, "collapsed"), (40000, 1, 64, 300, "bad"),
                                          (3641, 9, 320, 2000, "uniform")])
def test_embed_rows_bwd_large_batch_route(B, T, E, K, kind):
    """lipvq_embed_rows_bwd_ws_f32 (N >= 32 768 indexed rows: no atomics on the table / time embedding) against the atomic kernel
    and a float64 torch autograd reference; reproducible run after run; rows with a bad index contribute nothing."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd._capi import lib
    N = B * T
    g = torch.Generator(device="cuda").manual_seed(B + E)
    table = torch.randn(K, E, device="cuda", generator=g)
    pos = 0.1 * torch.randn(T, E, device="cuda", generator=g)
    w = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    b = 0.1 * torch.randn(E, device="cuda", generator=g)
    idx = torch.randint(0, K, (N,), device="cuda", generator=g)
    if kind == "collapsed":
        idx[torch.rand(N, device="cuda", generator=g) < 0.9] = 5
    if kind == "bad":
        idx[::1000] = K + 3
        idx[7] = -1
    out = torch.zeros(B, 3 * T, E, device="cuda")
    gout = torch.randn(B, 3 * T, E, device="cuda", generator=g)
    args = (N, T, 3 * T * E, 2 * E, E)
    st = ops.embed_rows(table, idx, pos, w, b, 1e-5, out, *args, want_stats=True)
    assert lib.lipvq_embed_rows_bwd_ws_supported(N, T, E, K)

    def grads(route_ws):
        gs = [torch.zeros_like(table), torch.zeros_like(pos), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")]
        if route_ws:
            ops.embed_rows_bwd(gout, table, idx, pos, st, w, *gs, *args)
        else:
            from lipvq_vae_amd.ops import _ptr, _stream, check
            check(lib.lipvq_embed_rows_bwd_f32(_ptr(gout), _ptr(table), _ptr(idx), _ptr(pos), _ptr(st), _ptr(w), *(_ptr(t) for t in gs),
                                               N, T, E, K, 3 * T * E, 2 * E, E, _stream()), "lipvq_embed_rows_bwd_f32")
        return gs
    a1, a2, old = grads(True), grads(True), grads(False)
    assert torch.equal(a1[0], a2[0])                                   # the table gradient has no atomics left
    # float64 reference through torch autograd on the valid rows
    ok = (idx >= 0) & (idx < K)
    tb, ps, ww, bb = (t.double().requires_grad_(True) for t in (table, pos, w, b))
    tt = torch.arange(N, device="cuda") % T
    x = tb[idx.clamp(0, K - 1)] + ps[tt]
    y = torch.nn.functional.layer_norm(x, (E,), ww, bb, 1e-5)
    sel = gout.view(B, 3 * T, E)[:, 1:2 * T:2, :].reshape(N, E).double()
    (y * sel * ok[:, None]).sum().backward()
    for got, ref, o, name in zip(a1, (tb.grad, ps.grad, ww.grad, bb.grad), old, ("table", "pos", "ln_w", "ln_b")):
        scale = float(ref.abs().max()) + 1e-30
        assert float((got.double() - ref).abs().max()) <= 2e-4 * scale, name
        assert float((got - o).abs().max()) <= 2e-4 * scale, name


def test_embed_rows_bwd_large_dense_batch_route():
    """Dense rows (idx = None) of a large batch: the row gradients are written straight into g_src, the time-embedding gradient
    has no per-row atomics; equals the atomic kernel to fp32 summation accuracy and float64 autograd."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd._capi import lib
    from lipvq_vae_amd.ops import _ptr, _stream, check
    B, T, E = 5000, 8, 192
    N = B * T
    g = torch.Generator(device="cuda").manual_seed(11)
    src = torch.randn(N, E, device="cuda", generator=g)
    pos = 0.1 * torch.randn(T, E, device="cuda", generator=g)
    w = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    b = 0.1 * torch.randn(E, device="cuda", generator=g)
    out = torch.zeros(B, 3 * T, E, device="cuda")
    gout = torch.randn(B, 3 * T, E, device="cuda", generator=g)
    args = (N, T, 3 * T * E, 2 * E, 0)
    st = ops.embed_rows(src, None, pos, w, b, 1e-5, out, *args, want_stats=True)
    new = [torch.zeros_like(src), torch.zeros_like(pos), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")]
    ops.embed_rows_bwd(gout, src, None, pos, st, w, *new, *args)
    old = [torch.zeros_like(src), torch.zeros_like(pos), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")]
    check(lib.lipvq_embed_rows_bwd_f32(_ptr(gout), _ptr(src), None, _ptr(pos), _ptr(st), _ptr(w), *(_ptr(t) for t in old),
                                       N, T, E, N, 3 * T * E, 2 * E, 0, _stream()), "lipvq_embed_rows_bwd_f32")
    assert torch.equal(new[0], old[0])                                  # per-row arithmetic is the same code
    sd, ps, ww, bb = (t.double().requires_grad_(True) for t in (src, pos, w, b))
    tt = torch.arange(N, device="cuda") % T
    y = torch.nn.functional.layer_norm(sd + ps[tt], (E,), ww, bb, 1e-5)
    sel = gout.view(B, 3 * T, E)[:, 0:2 * T:2, :].reshape(N, E).double()
    (y * sel).sum().backward()
    for got, ref, o, name in zip(new, (sd.grad, ps.grad, ww.grad, bb.grad), old, ("src", "pos", "ln_w", "ln_b")):
        scale = float(ref.abs().max()) + 1e-30
        assert float((got.double() - ref).abs().max()) <= 2e-4 * scale, name
        assert float((got - o).abs().max()) <= 2e-4 * scale, name
