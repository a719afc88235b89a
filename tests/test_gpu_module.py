"""GPU parity of the drop-in modules (LLFQVAE_V4 / VQVAE on the HIP library) against
 (a) the canonical oracle -- bit-exact on every forward tensor,
 (b) the golden vectors the reference itself produced -- exact indices, floats within 1e-5,
 (c) the reference's gradients and one AdamW(lr=1e-3, wd=1e-4) step (robomimic/algo/icl.py:887-889,
     968-970) -- tolerance 1e-5 relative to the gradient's scale (sums over rows are order dependent)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu

FLOAT_TOL = 1e-5          # north star: reconstructions within 1e-5 fp32


def _meta(g):
    return dict(eval(str(g["meta"])))


def _llfq_case(g, oracle):
    m = _meta(g)
    A, D, K, N = int(g["A"]), int(g["D"]), int(g["K"]), int(g["N"])
    p = O.make_params(int(g["seed"]), A, D, K, regime=m["regime"], oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"]), "parameter generator drifted from the fixture"
    x = O.make_inputs(int(g["seed"]), N, A, clamp=m.get("clamp", False))
    return p, x, (A, D, K, N)


def _model(cls, p, *args, **kw):
    model = cls(*args, **kw).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    return model


def _grad_close(got, ref, name):
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max()
    assert err <= FLOAT_TOL * scale + 1e-9, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


LLFQ_CASES = ["llfq_cfg1_trained", "llfq_v5main_trained", "llfq_real_k1024", "llfq_default_init",
              "llfq_cfg1_clamped", "llfq_cfg2_slice", "llfq_cfg3_slice", "llfq_ragged_n77", "llfq_n1",
              "llfq_odd_d37", "llfq_odd_d203"]        # latent widths that are not multiples of 8


@pytest.mark.parametrize("name", LLFQ_CASES)
def test_llfq_forward_vs_oracle_and_golden(name, oracle, golden_dir):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    g = np.load(golden_dir / f"{name}.npz")
    p, x, (A, D, K, N) = _llfq_case(g, oracle)
    model = _model(LLFQVAE_V4, p, A, D, num_codes=K)
    xt = torch.from_numpy(x).cuda()
    ref = oracle.llfq_forward(p, x)

    # the metric's path: encode + quantize
    z_e = model.encode(xt)
    idx, z_lat = model.tokenize(xt)
    assert np.array_equal(z_e.cpu().numpy(), ref["z_e"]), "z_e not bit-identical to the oracle"
    assert np.array_equal(idx.cpu().numpy(), ref["indices"]), "indices differ from the oracle"
    assert np.array_equal(z_lat.cpu().numpy(), ref["z_latent"])
    assert np.array_equal(model.code_usage.cpu().numpy(), ref["usage"])
    # against what the reference itself produced
    assert np.array_equal(idx.cpu().numpy(), g["indices"].astype(np.int64)), "indices differ from the reference"
    assert np.abs(z_e.cpu().numpy() - g["z_e"]).max() <= FLOAT_TOL
    assert abs(float(z_lat.double().sum()) - float(g["z_latent_sum"])) <= 1e-9 * max(1.0, abs(float(g["z_latent_sum"])))

    # the reference forward contract
    model.reset_usage()
    with torch.no_grad():
        z_latent, loss = model(xt)
    assert z_latent.shape == (N, D) and not z_latent.requires_grad and loss.dim() == 0
    assert torch.equal(z_latent, z_lat)
    assert abs(loss.item() - ref["loss"]) <= 1e-6 * abs(ref["loss"])
    if "loss" in g.files:
        assert abs(loss.item() - float(g["loss"])) <= FLOAT_TOL * abs(float(g["loss"]))
        xr = model.decode(idx).cpu().numpy()
        assert np.array_equal(xr, ref["x_recon"])
        assert np.abs(xr - g["x_recon"]).max() <= FLOAT_TOL


@pytest.mark.parametrize("name", ["llfq_cfg1_trained", "llfq_v5main_trained", "llfq_odd_d37"])
def test_llfq_training_step_vs_reference(name, oracle, golden_dir):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    g = np.load(golden_dir / f"{name}.npz")
    p, x, (A, D, K, N) = _llfq_case(g, oracle)
    model = _model(LLFQVAE_V4, p, A, D, num_codes=K)
    xt = torch.from_numpy(x).cuda()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)   # icl.py:887-889
    opt.zero_grad()                                                            # icl.py:913-914
    z_latent, loss = model(xt)
    assert loss.requires_grad and not z_latent.requires_grad
    loss.backward()                                                            # icl.py:968-969
    ograds = oracle.llfq_grads(p, x)
    for k, v in model.named_parameters():
        assert v.grad is not None, k
        got = v.grad.cpu().numpy()
        _grad_close(got, ograds[k], f"{k} vs oracle")
        _grad_close(got, g["grad/" + k], f"{k} vs reference")
    opt.step()                                                                 # icl.py:970
    for k, v in model.state_dict().items():
        ref = g["post/" + k]
        # one AdamW step moves every weight by ~lr; a gradient that differs in its last bits can flip
        # sign(m/sqrt(v)) only where the gradient is ~0, so compare with lr-scale tolerance there
        assert np.abs(v.cpu().numpy() - ref).max() <= 2.1e-3, k
        close = np.isclose(v.cpu().numpy(), ref, rtol=0, atol=2e-6)
        assert close.mean() > 0.999, f"{k}: only {close.mean():.4f} of the post-step weights match"
    # the packed-weight cache must notice the optimizer step
    z2, loss2 = model(xt)
    assert loss2.item() != loss.item()


def test_llfq_second_backward_and_grad_scale(oracle):
    """loss.backward() with a non-unit upstream gradient; gradients accumulate like autograd's."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(5, 7, 32, 128, oracle=oracle)
    x = O.make_inputs(5, 300, 7)
    model = _model(LLFQVAE_V4, p, 7, 32, num_codes=128)
    xt = torch.from_numpy(x).cuda()
    _, loss = model(xt)
    (3.0 * loss).backward()
    og = oracle.llfq_grads(p, x)
    for k, v in model.named_parameters():
        _grad_close(v.grad.cpu().numpy() / 3.0, og[k], k)


@pytest.mark.parametrize("name", ["vq_small_trained", "vq_main_trained", "vq_default_init", "vq_odd_d20"])
def test_vq_forward_backward(name, oracle, golden_dir):
    from lipvq_vae_amd.tokenizer import VQVAE
    g = np.load(golden_dir / f"{name}.npz")
    m = _meta(g)
    A, D, K, N = int(g["A"]), int(g["D"]), int(g["K"]), int(g["N"])
    p = O.make_params(int(g["seed"]), A, D, K, regime=m["regime"], variant="vq", oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"])
    x = O.make_inputs(int(g["seed"]), N, A)
    model = _model(VQVAE, p, A, D, num_embeddings=K)
    xt = torch.from_numpy(x).cuda()
    ref = oracle.vq_forward(p, x)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.zero_grad()
    z_latent, loss = model(xt)
    assert np.array_equal(model.last_indices.cpu().numpy(), ref["indices"])
    assert np.array_equal(model.last_indices.cpu().numpy(), g["indices"].astype(np.int64))
    assert np.array_equal(z_latent.cpu().numpy(), ref["z_latent"])
    assert np.abs(z_latent.cpu().numpy() - g["z_latent"]).max() <= FLOAT_TOL
    assert abs(loss.item() - float(g["loss"])) <= FLOAT_TOL * abs(float(g["loss"]))
    loss.backward()
    og = oracle.vq_grads(p, x)
    for k, v in model.named_parameters():
        got = v.grad.cpu().numpy()
        _grad_close(got, og[k], f"{k} vs oracle")
        _grad_close(got, g["grad/" + k], f"{k} vs reference")


@pytest.mark.parametrize("screen", ["fine", "coarse"])
@pytest.mark.parametrize("A,D,K,N", [(7, 64, 1024, 70000), (12, 208, 512, 9000), (7, 32, 256, 5000), (9, 128, 2048, 6000)])
def test_vq_fused_tokenize_equals_oracle(oracle, lipvq_option, no_screen_monitor, A, D, K, N, screen):
    """The plain VQVAE's encode + quantize in ONE launch (lipvq_vq_tokenize_f32: the fused kernel's ReLU instance with per-row fp16
    scales, either screen): indices, z_e (through the straight-through value), usage -- the oracle's, bit for bit; and it equals
    the unfused route (mlp3 + screened quantizer)."""
    from lipvq_vae_amd.tokenizer import VQVAE
    lipvq_option("screen_mode", screen)
    p = O.make_params(900 + D + K, A, D, K, variant="vq", oracle=oracle)
    x = O.make_inputs(901 + N, N, A)
    model = _model(VQVAE, p, A, D, num_embeddings=K)
    f = oracle.vq_forward(p, x)
    xt = torch.from_numpy(x).cuda()
    model.code_usage.zero_()
    assert model.fused_shape() and N > model.EXACT_ROWS_MAX
    idx, z_st = model.tokenize(xt)
    assert np.array_equal(idx.cpu().numpy(), f["indices"])
    assert np.array_equal(z_st.cpu().numpy(), f["z_latent"])
    assert np.array_equal(model.code_usage.cpu().numpy(), f["usage"])
    assert model.last_exact_rows is not None and int(model.last_exact_rows[0]) < N // 2
    idx_u, _ = model._quantize(model.encode(xt), None)
    assert torch.equal(idx_u, idx)
    with torch.no_grad():                               # the no-grad forward takes the same launch, then decoder + losses
        z_latent, loss = model(xt)
    assert np.array_equal(z_latent.cpu().numpy(), f["z_latent"]) and abs(loss.item() - f["loss"]) <= FLOAT_TOL * abs(f["loss"])
    z_latent, loss = model(xt)                          # with autograd: the unfused route (pre-activations are saved)
    assert np.array_equal(z_latent.detach().cpu().numpy(), f["z_latent"]) and abs(loss.item() - f["loss"]) <= FLOAT_TOL * abs(f["loss"])


@pytest.mark.parametrize("K,N", [(1024, 5000), (1024, 300), (128, 5000)])
def test_vq_module_routes(oracle, K, N):
    """VQVAE.tokenize / forward pick the screened quantizer (large codebook, large batch), the exact-rows kernel (large
    codebook, small batch) or the all-pairs kernel (small codebook): the same indices, straight-through values and loss as the
    oracle on every route."""
    from lipvq_vae_amd.tokenizer import VQVAE
    A, D = 7, 64
    p = O.make_params(700 + K, A, D, K, variant="vq", oracle=oracle)
    x = O.make_inputs(701 + N, N, A)
    model = _model(VQVAE, p, A, D, num_embeddings=K)
    f = oracle.vq_forward(p, x)
    model.code_usage.zero_()
    idx, z_st = model.tokenize(torch.from_numpy(x).cuda())
    assert np.array_equal(idx.cpu().numpy(), f["indices"])
    assert np.array_equal(z_st.cpu().numpy(), f["z_latent"])
    assert int(model.code_usage.sum()) == N
    z_latent, loss = model(torch.from_numpy(x).cuda())
    assert np.array_equal(z_latent.detach().cpu().numpy(), f["z_latent"])
    assert abs(loss.item() - f["loss"]) <= FLOAT_TOL * abs(f["loss"])
    if K >= 256 and N > 2048:
        assert model.last_exact_rows is not None and int(model.last_exact_rows[0]) < N // 2


def test_constructor_rng_matches_reference(golden_dir):
    """Same torch seed -> same initial weights as the reference constructors (drop-in property)."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE
    g = np.load(golden_dir / "init_seeded.npz")
    torch.manual_seed(1234)
    m = LLFQVAE_V4(12, 48, num_codes=96)
    sd = m.state_dict()
    keys = [k[5:] for k in g.files if k.startswith("llfq/")]
    assert sorted(sd.keys()) == sorted(keys)
    for k in keys:
        assert sd[k].dtype == torch.float32 and np.array_equal(sd[k].numpy(), g["llfq/" + k]), k
    torch.manual_seed(4321)
    m = VQVAE(12, 48, num_embeddings=64)
    sd = m.state_dict()
    keys = [k[3:] for k in g.files if k.startswith("vq/")]
    assert sorted(sd.keys()) == sorted(keys)
    for k in keys:
        assert np.array_equal(sd[k].numpy(), g["vq/" + k]), k


def test_api_errors_and_modes(oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(9, 7, 32, 64, oracle=oracle)
    model = _model(LLFQVAE_V4, p, 7, 32, num_codes=64)
    x = torch.randn(10, 7)
    with pytest.raises(RuntimeError):
        model(x)                                   # CPU tensor: no fallback
    with pytest.raises(TypeError):
        model(x.cuda().double())
    with pytest.raises(ValueError):
        model(x.cuda().reshape(2, 5, 7))           # must be flattened to [B*T, A] first
    model.eval()
    z, loss = model(x.cuda())                      # rollouts call the tokenizer in eval mode with grad on
    assert loss.requires_grad and z.shape == (10, 32)
    z0, loss0 = model(x.cuda()[:0])                # empty batch
    assert z0.shape == (0, 32)


def test_full_size_properties(oracle):
    """BASELINE config 2 at full size (524 288 rows): bit-identical to the multi-threaded oracle,
    plus size-independent properties (idempotence of quantisation, usage sums, determinism)."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, N = 7, 64, 1024, 4096 * 128
    p = O.make_params(77, A, D, K, oracle=oracle)
    x = O.make_inputs(77, N, A)
    model = _model(LLFQVAE_V4, p, A, D, num_codes=K)
    xt = torch.from_numpy(x).cuda()
    idx, z_lat = model.tokenize(xt)
    ze_ref = oracle.llfq_encode(p, x)
    idx_ref, zq_ref, usage_ref = oracle.nearest(ze_ref, p["quantizer.codebook"])
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(model.code_usage.cpu().numpy(), usage_ref)
    assert int(model.code_usage.sum()) == N
    # quantising a code vector returns that code (idempotence), for every code that is in use
    from lipvq_vae_amd import ops
    cb = model.quantizer.codebook.detach()
    idx2, _, _ = ops.nearest(cb, cb)
    assert torch.equal(idx2, torch.arange(K, device="cuda"))
    # determinism: run twice, compare bits
    idx3, z3 = model.tokenize(xt, count_usage=False)
    assert torch.equal(idx3, idx) and torch.equal(z3, z_lat)
    assert torch.equal(z_lat, cb[idx])


def test_degenerate_codebook_is_routed_around_the_screen(oracle):
    """A codebook in which every code exists twice (dead-code resets that copy live codes, a collapsed run): every row is an exact
    tie, the certified screen certifies nothing and the exact kernel -- built for a fraction of a percent of the rows, with
    candidate lists for at most N/8 of them -- would scan the whole codebook for the rest.  The tokenizer notices (the
    uncertified count of a call is read back without synchronising) and routes the following large batches through the
    all-pairs kernel; results are the oracle's on every call, whichever route ran -- and an ordinary model stays on the fused
    launch.  (The reference's default initialisation, although it maps every row to ONE code, is no such case: its rows are
    certified.)"""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, N = 7, 64, 1024, 12000
    p = O.make_params(41, A, D, K, oracle=oracle)
    p["quantizer.codebook"][K // 2:] = p["quantizer.codebook"][:K // 2]
    model = _model(LLFQVAE_V4, p, A, D, num_codes=K)
    x = O.make_inputs(42, N, A)
    xt = torch.from_numpy(x).cuda()
    idx_ref, zq_ref, _ = oracle.nearest(oracle.llfq_encode(p, x), p["quantizer.codebook"])
    assert idx_ref.max() < K // 2                      # first-minimum rule: the lower copy wins every tie
    routes = []
    for _ in range(4):
        idx, zq = model.tokenize(xt, count_usage=False)
        torch.cuda.synchronize()                       # lets the asynchronous read-back of the count land before the next call
        assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(zq.cpu().numpy(), zq_ref)
        routes.append(model.last_exact_rows is not None)
    assert routes[0] is True, "the first call has no history: it takes the screen"
    assert routes[1:] == [False, False, False], f"the screen was not bypassed after certifying nothing: {routes}"
    assert model._screen_monitor.last_fraction > 0.5
    z_latent, loss = model(xt)                         # the module forward follows the same routing
    f = oracle.llfq_forward(p, x)
    assert np.array_equal(z_latent.cpu().numpy(), f["z_q"]) and abs(loss.item() - f["loss"]) <= FLOAT_TOL * abs(f["loss"])
    # trained-like parameters: a fraction of a percent of the rows are uncertified, the fused launch stays
    p2 = O.make_params(43, A, D, K, oracle=oracle)
    m2 = _model(LLFQVAE_V4, p2, A, D, num_codes=K)
    for _ in range(3):
        m2.tokenize(xt, count_usage=False)
        torch.cuda.synchronize()
        assert m2.last_exact_rows is not None
    assert m2._screen_monitor.last_fraction < 0.05
