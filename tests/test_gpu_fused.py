"""GPU: the fused encode+quantize launch (lipvq_tokenize_f32) equals the oracle bit for bit and equals the
unfused path (lipvq_mlp3_f32 + lipvq_nearest_f32), including z_e when requested."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def _setup(seed, A, D, K, oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(seed, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    return p, model


@pytest.mark.parametrize("N,A,D,K", [(5000, 7, 64, 1024), (1024, 7, 32, 256), (777, 12, 128, 1000), (33, 7, 64, 37),
                                     (1, 7, 32, 256), (256 * 300 + 5, 7, 64, 1024), (600, 3, 64, 512)])
def test_fused_equals_oracle_and_unfused(oracle, N, A, D, K):
    from lipvq_vae_amd import ops
    p, model = _setup(N + D, A, D, K, oracle)
    assert ops.tokenize_supported(A, 64, 128, D, K)
    x = O.make_inputs(N, N, A)
    xt = torch.from_numpy(x).cuda()
    ze_ref = oracle.llfq_encode(p, x)
    idx_ref, zq_ref, usage_ref = oracle.nearest(ze_ref, p["quantizer.codebook"])
    packed, _, Wn = model._packed_encoder()
    w0, b0, w1, b1, _, b2, _ = (t.detach() for t in model._enc_params())
    raw = (w0, b0, w1, b1, Wn, b2)
    cb = model.quantizer.codebook.detach()
    prep = ops.nearest_prepare(cb)
    usage = torch.zeros(K, dtype=torch.int64, device="cuda")
    idx, zq, ze, ws = ops.tokenize(xt, packed, raw, cb, prep, usage=usage, want_ze=True)
    # without ze_out: z_e is never stored, uncertified rows are re-encoded by the exact kernel -- same answers
    usage_b = torch.zeros(K, dtype=torch.int64, device="cuda")
    idx_b, zq_b, ze_b, _ = ops.tokenize(xt, packed, raw, cb, prep, usage=usage_b)
    assert ze_b is None and torch.equal(idx_b, idx) and torch.equal(zq_b, zq) and torch.equal(usage_b, usage)
    assert np.array_equal(ze.cpu().numpy(), ze_ref)
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(zq.cpu().numpy(), zq_ref)
    assert np.array_equal(usage.cpu().numpy(), usage_ref)
    assert 0 <= int(ws[0]) <= N
    # module level: tokenize() takes the fused launch, encode()+_quantize() the unfused one
    idx2, zq2 = model.tokenize(xt, count_usage=False)
    idx3, zq3 = model._quantize(model.encode(xt), None)
    assert torch.equal(idx2, idx) and torch.equal(idx3, idx) and torch.equal(zq2, zq) and torch.equal(zq3, zq)


def test_fused_without_optional_outputs(oracle):
    from lipvq_vae_amd import ops
    p, model = _setup(3, 7, 64, 256, oracle)
    x = torch.from_numpy(O.make_inputs(3, 999, 7)).cuda()
    packed, _, Wn = model._packed_encoder()
    w0, b0, w1, b1, _, b2, _ = (t.detach() for t in model._enc_params())
    raw = (w0, b0, w1, b1, Wn, b2)
    cb = model.quantizer.codebook.detach()
    prep = ops.nearest_prepare(cb)
    idx, zq, ze, _ = ops.tokenize(x, packed, raw, cb, prep, want_zq=False)
    assert zq is None and ze is None
    idx2, _, _, _ = ops.tokenize(x, packed, raw, cb, prep)
    assert torch.equal(idx, idx2)


def test_config3_shape_matches_oracle(oracle):
    """BASELINE config 3 (K=8192, D=128): the fused launch at the full 524 288-row batch, checked against the
    multi-threaded oracle on a strided 16 384-row sample and against size-independent properties on all rows."""
    from lipvq_vae_amd import ops
    A, D, K, N = 7, 128, 8192, 4096 * 128
    p, model = _setup(303, A, D, K, oracle)
    x = O.make_inputs(303, N, A)
    xt = torch.from_numpy(x).cuda()
    idx, zq = model.tokenize(xt)
    assert int(model.code_usage.sum()) == N
    cb = model.quantizer.codebook.detach()
    assert torch.equal(zq, cb[idx])                                  # z_latent is exactly the selected code
    sel = np.arange(0, N, 32)
    ze_ref = oracle.llfq_encode(p, x[sel])
    idx_ref, _, _ = oracle.nearest(ze_ref, p["quantizer.codebook"])
    assert np.array_equal(idx.cpu().numpy()[sel], idx_ref)
    # the unfused exact path agrees on every row
    idx2, _, _ = ops.nearest(model.encode(xt), cb)
    assert torch.equal(idx2, idx)
    # rows left to the exact stage: a fraction of a percent with the three-product screen, a fifth with the one-product screen
    # this shape runs by default (include/lipvq.h: lipvq_screen_is_coarse)
    assert int(model.last_exact_rows[0]) < (N // 3 if ops.screen_is_coarse(K, D) else N // 20)


def test_skewed_and_degenerate_code_distributions(oracle):
    """Every row -> one code (the reference's default init) and a 4-code codebook: the usage histogram must be exact
    (it is aggregated per wave / per workgroup precisely because such distributions hammer a few counters)."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    for regime, K in (("default", 256), ("trained", 4)):
        p = O.make_params(61, 7, 64, K, regime=regime, oracle=oracle)
        model = LLFQVAE_V4(7, 64, num_codes=K).cuda()
        model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
        x = O.make_inputs(61, 20000, 7)
        ref = oracle.llfq_forward(p, x)
        xt = torch.from_numpy(x).cuda()
        idx, _ = model.tokenize(xt)                                # fused launch (LDS histogram)
        assert np.array_equal(idx.cpu().numpy(), ref["indices"])
        assert np.array_equal(model.code_usage.cpu().numpy(), ref["usage"])
        model.reset_usage()
        idx2, _ = model._quantize(model.encode(xt), model.code_usage)      # stand-alone screen + exact rows (wave aggregation)
        assert torch.equal(idx2, idx) and np.array_equal(model.code_usage.cpu().numpy(), ref["usage"])


def test_two_streams_concurrently(oracle):
    """The library is stateless: two tokenizers driven from two HIP streams at the same time (each with its own
    workspace) give the serial answers."""
    p, m1 = _setup(41, 7, 64, 1024, oracle)
    _, m2 = _setup(41, 7, 64, 1024, oracle)
    xs = [torch.from_numpy(O.make_inputs(50 + i, 40000 + 77 * i, 7)).cuda() for i in range(4)]
    serial = [m1.tokenize(x, count_usage=False)[0].clone() for x in xs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = [None] * 4
    for rep in range(3):
        with torch.cuda.stream(s1):
            out[0] = m1.tokenize(xs[0], count_usage=False)[0].clone()
            out[2] = m1.tokenize(xs[2], count_usage=False)[0].clone()
        with torch.cuda.stream(s2):
            out[1] = m2.tokenize(xs[1], count_usage=False)[0].clone()
            out[3] = m2.tokenize(xs[3], count_usage=False)[0].clone()
        torch.cuda.synchronize()
        for a, b in zip(out, serial):
            assert torch.equal(a, b)


def test_no_grad_forward_takes_the_fused_launch_and_agrees(oracle):
    """forward() without autograd (N above the small-batch threshold) runs encode + quantize as the fused launch; z_latent,
    loss and indices equal the autograd path's bit for bit."""
    p, model = _setup(61, 7, 64, 1024, oracle)
    x = torch.from_numpy(O.make_inputs(61, 5000, 7)).cuda()
    z_a, loss_a = model(x)
    idx_a = model.last_indices.clone()
    assert loss_a.requires_grad
    with torch.no_grad():
        z_b, loss_b = model(x)
    assert model.last_exact_rows is not None                      # the fused kernel's workspace: that path ran
    assert torch.equal(z_a, z_b) and torch.equal(idx_a, model.last_indices) and loss_a.item() == loss_b.item()
    f = oracle.llfq_forward(p, x.cpu().numpy())
    assert np.array_equal(z_b.cpu().numpy(), f["z_q"]) and abs(loss_b.item() - f["loss"]) <= 1e-5 * abs(f["loss"])


@pytest.mark.parametrize("screen", ["fine", "coarse"])
@pytest.mark.parametrize("N,A,D,K", [(5000, 7, 64, 1024), (2100, 12, 208, 1024), (3000, 7, 32, 256), (2500, 12, 128, 1000)])
def test_training_forward_launch_equals_unfused(oracle, monkeypatch, N, A, D, K, screen):
    """lipvq_tokenize_train_f32 (encoder + quantizer + everything autograd saves, one launch) against lipvq_mlp3_f32 with saved
    pre-activations + the stand-alone quantizer: z_e, the three pre-activations, indices and z_q bit for bit; and the module's
    gradients at a batch that takes this route equal the ones the unfused route gives."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.autograd import _ENC_ACTS
    monkeypatch.setenv("LIPVQ_SCREEN_MODE", screen)          # both screens exist for the training instance too
    monkeypatch.setenv("LIPVQ_SCREEN_MONITOR", "0")
    p, model = _setup(N + D, A, D, K, oracle)
    x = O.make_inputs(N + 1, N, A)
    xt = torch.from_numpy(x).cuda()
    idx, zq, ze, pre = model._tokenize_fused(xt, None, want_pre=True)
    ze_u, pre_u = ops.mlp3(xt, model._packed_encoder()[0], _ENC_ACTS, save_pre=True)
    idx_u, zq_u = model._quantize(ze_u, None)
    assert torch.equal(ze, ze_u) and torch.equal(idx, idx_u) and torch.equal(zq, zq_u)
    for a, b in zip(pre, pre_u):
        assert a.shape == b.shape and torch.equal(a, b)
    assert np.array_equal(ze.cpu().numpy(), oracle.llfq_encode(p, x))
    # gradients through the module (N > EXACT_ROWS_MAX: the fused training forward) vs the same step forced onto the unfused route
    assert N > model.EXACT_ROWS_MAX
    _, loss = model(xt)
    loss.backward()
    g_fused = {k: v.grad.clone() for k, v in model.named_parameters()}
    model.zero_grad()
    old = model.EXACT_ROWS_MAX
    try:
        type(model).EXACT_ROWS_MAX = 1 << 30            # (> N: forward() takes mlp3 + exact rows kernel)
        _, loss2 = model(xt)
        loss2.backward()
    finally:
        type(model).EXACT_ROWS_MAX = old
    assert abs(loss.item() - loss2.item()) <= 1e-6 * abs(loss2.item())
    for k, v in model.named_parameters():
        assert torch.allclose(g_fused[k], v.grad, rtol=0, atol=1e-5 * max(1e-12, float(v.grad.abs().max()))), k


@pytest.mark.parametrize("shape", ["w8rg2", "w4rg2", "w4rg1"])
@pytest.mark.parametrize("N,A,D,K", [(256 * 300 + 5, 7, 64, 1024), (9000, 7, 32, 256), (4100, 7, 128, 2048), (3000, 12, 208, 1024)])
def test_other_kernel_shapes_give_the_same_results(oracle, monkeypatch, shape, N, A, D, K):
    """tokenize_kernel exists in four (waves per workgroup, 32-row groups per wave) shapes (lipvq_fused.hip: tok_shape); only one
    is the default, the others stay in the library as measured alternatives -- every one of them must return the oracle's
    indices / z_q / usage (the shape only changes which wave owns which rows).  LIPVQ_TOK_SHAPE is read per launch."""
    p, model = _setup(N + D + 7, A, D, K, oracle)
    x = O.make_inputs(N + 3, N, A)
    xt = torch.from_numpy(x).cuda()
    idx_ref, zq_ref, usage_ref = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    monkeypatch.setenv("LIPVQ_TOK_SHAPE", shape)
    model.code_usage.zero_()
    idx, zq = model.tokenize(xt)
    assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(zq.cpu().numpy(), zq_ref)
    assert np.array_equal(model.code_usage.cpu().numpy(), usage_ref)
