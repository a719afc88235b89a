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
def test_fused_equals_oracle_and_unfused(oracle, lipvq_option, N, A, D, K):
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
    # without ze_out: z_e goes to a scratch in the workspace for the exact stage (the default since round 3) -- same answers
    usage_b = torch.zeros(K, dtype=torch.int64, device="cuda")
    idx_b, zq_b, ze_b, _ = ops.tokenize(xt, packed, raw, cb, prep, usage=usage_b)
    assert ze_b is None and torch.equal(idx_b, idx) and torch.equal(zq_b, zq) and torch.equal(usage_b, usage)
    # ... and with z_e never stored: uncertified rows are re-encoded from x by the exact kernel (nearest_rows_encode_kernel: what
    # large fast-mode batches run; the tok_ze_rows option is read per launch; the one-product screen always stores)
    lipvq_option("tok_ze_rows", "0")
    lipvq_option("screen_mode", "fine")
    usage_c = torch.zeros(K, dtype=torch.int64, device="cuda")
    idx_c, zq_c, _, _ = ops.tokenize(xt, packed, raw, cb, prep, usage=usage_c)
    lipvq_option("tok_ze_rows", None)
    lipvq_option("screen_mode", None)
    assert torch.equal(idx_c, idx) and torch.equal(zq_c, zq) and torch.equal(usage_c, usage)
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
def test_training_forward_launch_equals_unfused(oracle, lipvq_option, no_screen_monitor, N, A, D, K, screen):
    """lipvq_tokenize_train_f32 (encoder + quantizer + everything autograd saves, one launch) against lipvq_mlp3_f32 with saved
    pre-activations + the stand-alone quantizer: z_e, the three pre-activations, indices and z_q bit for bit; and the module's
    gradients at a batch that takes this route equal the ones the unfused route gives."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.autograd import _ENC_ACTS
    lipvq_option("screen_mode", screen)          # both screens exist for the training instance too
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


@pytest.mark.parametrize("N,A,D,K", [(66000, 7, 64, 1024), (65537, 12, 128, 1000), (70001, 3, 32, 256), (65539, 12, 208, 1024)])
def test_decoder_launch_with_folded_loss_equals_separate_launches(oracle, N, A, D, K):
    """lipvq_mlp3_loss_f32 (the decoder stack summing both squared errors itself, from 65 536 rows on) against lipvq_mlp3_f32 +
    lipvq_mse_pair_loss_f32: x_rec and the saved pre-activations bit for bit; the two means and the loss -- double sums in a
    different order -- to 1e-7; and the loss the module returns against the oracle's."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.autograd import _DEC_ACTS
    p, model = _setup(N + D, A, D, K, oracle)
    x = O.make_inputs(N + 5, N, A)
    xt = torch.from_numpy(x).cuda()
    dec = model._packed_decoder()
    cb = model.quantizer.codebook.detach()
    assert ops.mlp3_loss_supported(N, dec)
    idx, z_q, z_e = model._tokenize_fused(xt, None, want_ze=True)
    y, pre, l3 = ops.mlp3_loss(cb, dec, _DEC_ACTS, idx, xt, z_e, 0.25, ops.LOSS_LLFQ, save_pre=True)
    y_u, pre_u = ops.mlp3(cb, dec, _DEC_ACTS, gather_idx=idx, save_pre=True)
    l3_u = ops.mse_pair_loss(y_u, xt, z_q, z_e, 0.25, ops.LOSS_LLFQ)
    assert torch.equal(y, y_u)
    for a, b in zip(pre, pre_u):
        assert torch.equal(a, b)
    assert torch.allclose(l3, l3_u, rtol=1e-7, atol=0), (l3, l3_u)
    y2, none, l3b = ops.mlp3_loss(cb, dec, _DEC_ACTS, idx, xt, z_e, 0.25, ops.LOSS_LLFQ)      # without saving: same sums
    assert none is None and torch.equal(y2, y) and torch.equal(l3b, l3)
    with torch.no_grad():
        _, loss = model(xt)
    assert loss.item() == l3[2].item()
    f = oracle.llfq_forward(p, x[:4096])                       # (the oracle's loss on a slice the CPU finishes quickly ...)
    with torch.no_grad():
        _, loss_s = model(xt[:4096])                           # (... which takes the separate launches: same numbers as before)
    assert abs(loss_s.item() - f["loss"]) <= 1e-5 * abs(f["loss"])


@pytest.mark.parametrize("screen", ["fine", "coarse"])
@pytest.mark.parametrize("N,A,D,K", [(5000, 7, 64, 128), (2100, 12, 208, 1024), (3000, 7, 32, 256), (2500, 12, 128, 1000)])
def test_vq_training_forward_launch_equals_unfused(monkeypatch, lipvq_option, no_screen_monitor, N, A, D, K, screen):
    """lipvq_vq_tokenize_train_f32 (the plain VQVAE's encoder + quantizer + saved pre-activations, one launch) against
    lipvq_mlp3_f32(relu x 3, saved pre-activations) + the stand-alone quantizer, bit for bit; and the module's gradients at a batch
    that takes this route equal the ones the unfused route gives."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.autograd import _RELU3
    from lipvq_vae_amd.tokenizer import VQVAE
    lipvq_option("screen_mode", screen)
    torch.manual_seed(N + D)
    model = VQVAE(A, D, num_embeddings=K).cuda()
    with torch.no_grad():
        model.embedding.weight.uniform_(0.0, 0.5)
    model.invalidate_caches()
    xt = torch.from_numpy(O.make_inputs(N + 1, N, A)).cuda()
    idx, zq, ze, pre = model._tokenize_fused(xt, None, want_pre=True)
    ze_u, pre_u = ops.mlp3(xt, model._packed_encoder(), _RELU3, save_pre=True)
    idx_u, zq_u = model._quantize(ze_u, None)
    assert torch.equal(ze, ze_u) and torch.equal(idx, idx_u) and torch.equal(zq, zq_u)
    for a, b in zip(pre, pre_u):
        assert a.shape == b.shape and torch.equal(a, b)
    assert N > model.EXACT_ROWS_MAX and K >= model.FUSED_MIN_CODES
    calls = []
    real = ops.lib.lipvq_vq_tokenize_train_f32
    monkeypatch.setattr(ops.lib, "lipvq_vq_tokenize_train_f32", lambda *a: (calls.append(1), real(*a))[1])
    z1, loss = model(xt)
    loss.backward()
    assert calls == [1]                                        # the module's training forward took the fused launch
    g_fused = {k: v.grad.clone() for k, v in model.named_parameters()}
    model.zero_grad()
    monkeypatch.setattr(type(model), "fused_shape", lambda self: False)
    z2, loss2 = model(xt)
    loss2.backward()
    assert calls == [1] and torch.equal(z1, z2) and loss.item() == loss2.item()
    for k, v in model.named_parameters():             # (same inputs to the same backward; the embedding's scatter uses fp32 atomics at this size)
        assert torch.allclose(g_fused[k], v.grad, rtol=0, atol=1e-5 * max(1e-12, float(v.grad.abs().max()))), k


@pytest.mark.parametrize("with_g", [True, False])
@pytest.mark.parametrize("N,D,K,det", [(70001, 64, 1024, False), (66000, 208, 1000, False), (40000, 64, 128, True), (131072, 32, 37, True)])
def test_scatter_with_rows_formed_in_kernel_equals_scaled_diff_then_scatter(N, D, K, det, with_g):
    """lipvq_scatter_add_sorted_vq_f32 against lipvq_scaled_diff_f32 + lipvq_scatter_add_sorted_f32, both orders (256-row segments /
    one sequential chain per code): bit for bit."""
    from lipvq_vae_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(N + D)
    ze = torch.rand(N, D, device="cuda", generator=gen)
    table = torch.rand(K, D, device="cuda", generator=gen)
    idx = torch.randint(0, K, (N,), device="cuda", generator=gen)
    idx[: N // 3] = 5                                          # one hot code (many segments / a long chain)
    g = torch.randn(N, D, device="cuda", generator=gen) * 1e-6 if with_g else None
    gs = torch.tensor([1.7], device="cuda")
    alpha = 0.5 / (N * D)
    got = ops.scatter_add_vq(g, ze, table, idx, alpha, gscale=gs, deterministic=det)
    rows = ops.scaled_diff(table[idx], ze, alpha, gscale=gs, c=g)
    want = ops.scatter_add(rows, idx, K, route="sequential_sorted" if det else "sorted")
    assert torch.equal(got, want)
    ref = torch.zeros(K, D, dtype=torch.float64, device="cuda").index_add_(0, idx, rows.double())
    assert torch.allclose(got.double(), ref, rtol=0, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("N,A,D,K", [(66001, 7, 64, 128), (65536, 12, 208, 1024), (70003, 3, 32, 64)])
def test_vq_decoder_launch_with_straight_through_and_loss_equals_separate_launches(N, A, D, K):
    """The plain VQVAE's decoder at large batches (lipvq_mlp3_loss_f32 with ste_out): the launch forms z_st = z_e + (z_q - z_e)
    from E[idx] and z_e, runs the stack on it, stores it and sums both squared errors -- against lipvq_ste_f32 + lipvq_mlp3_f32 +
    lipvq_mse_pair_loss_f32: z_st, x_rec and the saved pre-activations bit for bit, the loss to 1e-7; and through the module."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.autograd import _RELU3
    from lipvq_vae_amd.tokenizer import VQVAE
    torch.manual_seed(N)
    model = VQVAE(A, D, num_embeddings=K).cuda()
    with torch.no_grad():
        model.embedding.weight.uniform_(0.0, 0.5)
    model.invalidate_caches()
    xt = torch.from_numpy(O.make_inputs(N + 3, N, A)).cuda()
    E = model.embedding.weight.detach()
    dec = model._packed_decoder()
    assert ops.mlp3_loss_supported(N, dec)
    idx, z_q, z_e = model._tokenize_fused(xt, None)
    cc = float(model.commitment_cost)
    y, pre, l3, z_st = ops.mlp3_loss(E, dec, _RELU3, idx, xt, z_e, cc, ops.LOSS_VQ, save_pre=True, ste=True)
    z_st_u = ops.ste(z_e, z_q)
    y_u, pre_u = ops.mlp3(z_st_u, dec, _RELU3, save_pre=True)
    l3_u = ops.mse_pair_loss(y_u, xt, z_q, z_e, cc, ops.LOSS_VQ)
    assert torch.equal(z_st, z_st_u) and torch.equal(y, y_u)
    for a, b in zip(pre, pre_u):
        assert torch.equal(a, b)
    assert torch.allclose(l3, l3_u, rtol=1e-7, atol=0), (l3, l3_u)
    with torch.no_grad():
        z_m, loss_m = model(xt)
    assert torch.equal(z_m, z_st) and loss_m.item() == l3[2].item()


def oracle_grads_cpu(model, xt, kind, gscale):
    """Parameter gradients of gscale * loss by torch autograd on the CPU in float64 (stock ops, the reference's forward as
    oracle/lipvq_oracle.py restates it; float64 so that the comparison sees the launches' rounding only -- a sequential fp32
    index_add_ over the thousands of rows of one code is itself 1e-5 off), with the code indices taken from the launch (the [N, K, D] distance tensor of the
    reference's argmin is 17 GB at these batches; index parity is what the other tests are for)."""
    import torch.nn.functional as F
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.named_parameters()}
    x = xt.cpu().double()
    idx = model.last_indices.cpu()
    if kind == "llfq":
        z_e = O.torch_llfq_encode(p, x)
        z_q = p["quantizer.codebook"][idx]
        h = F.gelu(F.linear(z_q, p["decoder.0.weight"], p["decoder.0.bias"]))
        h = F.gelu(F.linear(h, p["decoder.2.weight"], p["decoder.2.bias"]))
        x_rec = F.linear(h, p["to_output.weight"], p["to_output.bias"])
        loss = F.mse_loss(x_rec, x) + 0.25 * F.mse_loss(z_q.detach(), z_e) + 0.25 * F.mse_loss(z_q, z_e.detach())
    else:
        h = x
        for i in (0, 2, 4):
            h = F.relu(F.linear(h, p[f"encoder.{i}.weight"], p[f"encoder.{i}.bias"]))
        z_e = h
        z_q = F.embedding(idx, p["embedding.weight"])
        q_loss = F.mse_loss(z_q, z_e.detach()) + float(model.commitment_cost) * F.mse_loss(z_q.detach(), z_e)
        h = z_e + (z_q - z_e).detach()
        for i in (0, 2, 4):
            h = F.relu(F.linear(h, p[f"decoder.{i}.weight"], p[f"decoder.{i}.bias"]))
        loss = F.mse_loss(h, x) + q_loss
    (loss * gscale).backward()
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)).float() for k, v in p.items()}


@pytest.mark.parametrize("kind,N,A,D,K", [("llfq", 66000, 7, 64, 1024), ("llfq", 65537, 12, 128, 1000), ("llfq", 65600, 12, 208, 1024),
                                          ("vq", 66001, 7, 64, 128), ("vq", 65536, 12, 208, 1024)])
def test_backward_with_folded_loss_terms_equals_separate_launches(oracle, monkeypatch, kind, N, A, D, K):
    """From 65 536 rows on the latent-loss gradient terms ride in their consumers instead of three lipvq_scaled_diff_f32 streams:
    LipVQ's encoder chain computes its input alpha g (sigmoid(pre2) - codebook[idx]) itself (lipvq_mlp3_bwd_vq_f32), the plain
    VQVAE's decoder chain adds cc alpha g (z_e - E[idx]) to its last store, and the codebook rows alpha g (z_q - z_e) (+ the
    decoder's gradient) are formed inside the counting-sort scatter (lipvq_scatter_add_sorted_vq_f32).  Every parameter gradient
    must carry the same bits as the separate launches give (same operations in the same order, -ffp-contract=off), and agree
    with torch autograd on the CPU."""
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.tokenizer import VQVAE
    if kind == "llfq":
        p, model = _setup(N + D, A, D, K, oracle)
    else:
        torch.manual_seed(N)
        model = VQVAE(A, D, num_embeddings=K).cuda()
        with torch.no_grad():
            model.embedding.weight.uniform_(0.0, 0.4)
        model.invalidate_caches()
    xt = torch.from_numpy(O.make_inputs(N + 2, N, A)).cuda()
    calls = []
    real, real_s = ops.lib.lipvq_mlp3_bwd_vq_f32, ops.lib.lipvq_scatter_add_sorted_vq_f32
    monkeypatch.setattr(ops.lib, "lipvq_mlp3_bwd_vq_f32", lambda *a: (calls.append("chain"), real(*a))[1])
    monkeypatch.setattr(ops.lib, "lipvq_scatter_add_sorted_vq_f32", lambda *a: (calls.append("scatter"), real_s(*a))[1])
    _, loss = model(xt)
    (loss * 3.0).backward()                                    # (a gscale that is not 1)
    assert calls == ["chain", "scatter"] or calls == ["scatter", "chain"]          # the folded launches ran
    g_folded = {k: v.grad.clone() for k, v in model.named_parameters()}
    model.zero_grad()
    monkeypatch.setattr(ops, "mlp3_bwd_vq_supported", lambda N, pk: False)
    monkeypatch.setattr(ops, "scatter_add_vq", lambda g, ze, table, idx, alpha, gscale=None, zq=None, deterministic=None:
                        ops.scatter_add(ops.scaled_diff(zq, ze, alpha, gscale=gscale, c=g), idx, table.shape[0]))
    _, loss2 = model(xt)
    (loss2 * 3.0).backward()
    assert len(calls) == 2
    for k, v in model.named_parameters():
        assert torch.equal(g_folded[k], v.grad), k
    # and against torch autograd on the CPU (float64)
    ref = oracle_grads_cpu(model, xt, kind, 3.0)
    for k, v in model.named_parameters():
        assert torch.allclose(v.grad.cpu(), ref[k], rtol=0, atol=2e-5 * max(1e-12, float(ref[k].abs().max()))), k


@pytest.mark.parametrize("shape", ["w8rg2", "w4rg2", "w4rg1"])
@pytest.mark.parametrize("N,A,D,K", [(256 * 300 + 5, 7, 64, 1024), (9000, 7, 32, 256), (4100, 7, 128, 2048), (3000, 12, 208, 1024)])
def test_other_kernel_shapes_give_the_same_results(oracle, lipvq_option, shape, N, A, D, K):
    """tokenize_kernel exists in four (waves per workgroup, 32-row groups per wave) shapes (lipvq_fused.hip: tok_shape); only one
    is the default, the others stay in the library as measured alternatives -- every one of them must return the oracle's
    indices / z_q / usage (the shape only changes which wave owns which rows).  The tok_shape option is read per launch."""
    p, model = _setup(N + D + 7, A, D, K, oracle)
    x = O.make_inputs(N + 3, N, A)
    xt = torch.from_numpy(x).cuda()
    idx_ref, zq_ref, usage_ref = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    lipvq_option("tok_shape", shape)
    model.code_usage.zero_()
    idx, zq = model.tokenize(xt)
    assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(zq.cpu().numpy(), zq_ref)
    assert np.array_equal(model.code_usage.cpu().numpy(), usage_ref)


@pytest.mark.parametrize("N,A,D,K", [(65536, 7, 64, 1024), (40000, 7, 32, 256), (30011, 12, 208, 1024), (50000, 7, 128, 8192),
                                     (70000, 7, 64, 4096), (3000, 7, 64, 37)])
def test_workspace_header_is_kept_clean_by_the_launch_itself(oracle, N, A, D, K):
    """Round 4: nothing fills the workspace header per call.  The screening launch's last workgroup publishes the number of listed
    rows ([0], [1]) and zeroes the live word ([8], [9]); the slot-2 counter ([10]) is zeroed by the NEXT launch's first workgroup.
    Call after call on ONE workspace: same indices / z_q / usage as the all-pairs exact kernel on the same z_e, the same published
    count every time (three- and one-product screens; codebooks on both sides of LQ_LISTS_ALL_K = 2048, above which the scanning
    kernel runs behind the list kernel)."""
    from lipvq_vae_amd import ops
    p, model = _setup(N % 1000 + D, A, D, K, oracle)
    xt = torch.from_numpy(O.make_inputs(N % 977, N, A)).cuda()
    packed, _, Wn = model._packed_encoder()
    w0, b0, w1, b1, _, b2, _ = (t.detach() for t in model._enc_params())
    raw = (w0, b0, w1, b1, Wn, b2)
    cb = model.quantizer.codebook.detach()
    prep = ops.nearest_prepare(cb)
    ws = ops.tokenize_workspace(N, D, xt.device)
    assert int(ws[:16].abs().sum()) == 0                                       # lipvq_tokenize_workspace_init
    ref_i, ref_q, _ = ops.nearest(model.encode(xt), cb)
    ref_u = torch.bincount(ref_i, minlength=K)
    counts = []
    for call in range(4):
        usage = torch.zeros(K, dtype=torch.int64, device="cuda")
        idx, zq, _, ws_out = ops.tokenize(xt, packed, raw, cb, prep, usage=usage, workspace=ws)
        assert ws_out is ws and torch.equal(idx, ref_i), call
        assert torch.equal(zq, ref_q) and torch.equal(usage, ref_u)
        hdr = ws[:16].cpu()
        assert int(hdr[8]) == 0 and int(hdr[9]) == 0, (call, hdr.tolist())    # the live word is back at zero
        assert int(hdr[0]) == int(hdr[1])
        counts.append(int(hdr[0]))
    assert len(set(counts)) == 1 and 0 <= counts[0] <= N, counts              # the same rows are left to an exact decision
    if K >= 256:
        assert counts[0] > 0, "meant to exercise uncertified rows"


@pytest.mark.parametrize("N,A,D,K", [(300000, 7, 64, 1024), (20011, 7, 32, 256), (9000, 12, 64, 2048), (24000, 7, 128, 1024),
                                     (12345, 12, 208, 1024)])
def test_in_place_decisions_equal_the_list_kernel(oracle, lipvq_option, N, A, D, K):
    """Round 4: under the three-product screen with K <= 2048 the wave that screened a row decides it itself when the screen does not
    certify it (lq_screen_decide_inplace: the list kernel's own body on the row's stored z_e) -- the default at every batch size.
    Forced on, forced off (option tok_inplace) and default: same indices, z_q, usage and the same published count of rows
    decided exactly; both equal the all-pairs exact kernel."""
    from lipvq_vae_amd import ops
    p, model = _setup(N % 1000 + D + 1, A, D, K, oracle)
    xt = torch.from_numpy(O.make_inputs(N % 971, N, A)).cuda()
    packed, _, Wn = model._packed_encoder()
    w0, b0, w1, b1, _, b2, _ = (t.detach() for t in model._enc_params())
    raw = (w0, b0, w1, b1, Wn, b2)
    cb = model.quantizer.codebook.detach()
    prep = ops.nearest_prepare(cb)
    lipvq_option("screen_mode", "fine")
    ref_i, ref_q, _ = ops.nearest(model.encode(xt), cb)
    ref_u = torch.bincount(ref_i, minlength=K)
    seen = {}
    for setting in ("0", "1", None):
        lipvq_option("tok_inplace", setting)
        ws = ops.tokenize_workspace(N, D, xt.device)
        for call in range(2):
            usage = torch.zeros(K, dtype=torch.int64, device="cuda")
            idx, zq, _, _ = ops.tokenize(xt, packed, raw, cb, prep, usage=usage, workspace=ws)
            assert torch.equal(idx, ref_i) and torch.equal(zq, ref_q) and torch.equal(usage, ref_u), (setting, call)
            hdr = ws[:16].cpu()
            assert int(hdr[8]) == 0 and int(hdr[9]) == 0 and int(hdr[0]) == int(hdr[1]), (setting, hdr.tolist())
            seen.setdefault(setting, int(hdr[0]))
            assert seen[setting] == int(hdr[0])
    assert seen["0"] == seen["1"] == seen[None] and seen["0"] > 0, seen


def test_schedule_choices_give_the_same_results_and_tune_keeps_one(oracle, lipvq_option):
    """Round 4: the fused launch's two device-dependent schedule choices (options tok_defer_ze / tok_nt_ze; which combination is
    fastest depends on the MI355X device: profiles/r04_i_clock_ab.txt) change nothing but speed -- indices, z_q, z_e and usage are
    bit-identical under all four, and LLFQVAE_V4.tune (lipvq_tokenize_tune_f32) times them and keeps one."""
    N, A, D, K = 70000, 7, 64, 1024
    p, model = _setup(77, A, D, K, oracle)
    xt = torch.from_numpy(O.make_inputs(78, N, A)).cuda()
    ref = None
    for d in ("0", "1"):
        for n in ("0", "1"):
            lipvq_option("tok_defer_ze", d)
            lipvq_option("tok_nt_ze", n)
            model.code_usage.zero_()
            idx, zq, ze = model._tokenize_fused(xt, model.code_usage, want_ze=True)
            got = (idx.clone(), zq.clone(), ze.clone(), model.code_usage.clone())
            if ref is None:
                ref = got
                assert torch.equal(ze, model.encode(xt))
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (d, n)
    lipvq_option("tok_defer_ze", None)
    lipvq_option("tok_nt_ze", None)
    usage_before = model.code_usage.clone()
    t = model.tune(xt, launches=20)
    assert set(t["choice"]) == {"defer_ze", "nt_ze"} and all(v in (0, 1) for v in t["choice"].values())
    assert len(t["ms_per_launch"]) == 4 and all(0.0 < v < 50.0 for v in t["ms_per_launch"].values()), t
    assert torch.equal(model.code_usage, usage_before)                      # the tuner counts into a scratch of its own
    idx, _ = model.tokenize(xt, count_usage=False)
    assert torch.equal(idx, ref[0])
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        from lipvq_vae_amd import ops
        cb = model.quantizer.codebook.detach()
        packed, _, Wn = model._packed_encoder()
        w0, b0, w1, b1, _, b2, _ = (t_.detach() for t_ in model._enc_params())
        prep = ops.nearest_prepare(cb)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            with pytest.raises(RuntimeError, match="capturing"):
                ops.tokenize_tune(xt, packed, (w0, b0, w1, b1, Wn, b2), cb, prep, launches=2)


def test_tune_has_nothing_to_choose_for_wide_latents(oracle):
    """lipvq_tokenize_tune_f32 returns at once (choice -1 -> None) where the instances fix both schedule flags (D > 64)."""
    p, model = _setup(5, 12, 208, 1024, oracle)
    xt = torch.from_numpy(O.make_inputs(6, 4096, 12)).cuda()
    assert model.tune(xt, launches=5) is None
