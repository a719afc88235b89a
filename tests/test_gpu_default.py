"""GPU: the default action branch (reference obs_nets.py:1244-1260) on the HIP library -- each new kernel against a plain
torch float64 evaluation of the same op, and DefaultActionNetwork against the fixtures the stock torch modules produced
(oracle/gen_golden.py::run_default_branch).  Tolerances: 1e-5 of a forward tensor's scale, 1e-4 of a gradient's L2 norm;
the ICRT-width fixture (D = 208) is itself only reproducible to its stored fp32 noise (~2e-4 against float64)."""
import hashlib
import warnings

import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import lipvq_vae_amd
    return lipvq_vae_amd.ops


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


@pytest.mark.parametrize("J,K", [(64, 12), (128, 64), (208, 128), (64, 7)])
@pytest.mark.parametrize("training", [False, True])
def test_spectral_norm_and_backward(ops, J, K, training):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(J + K)
    W = torch.randn(J, K, generator=g)
    u = F.normalize(torch.randn(J, generator=g), dim=0)
    v = F.normalize(torch.randn(K, generator=g), dim=0)
    gout = torch.randn(J, K, generator=g)
    Wd = W.double().requires_grad_(True)
    ud, vd = u.double(), v.double()
    if training:
        with torch.no_grad():
            vd = F.normalize(torch.mv(Wd.t(), ud), dim=0, eps=1e-12)
            ud = F.normalize(torch.mv(Wd, vd), dim=0, eps=1e-12)
    sigma = torch.dot(ud, torch.mv(Wd, vd))
    ref = Wd / sigma
    (ref * gout.double()).sum().backward()
    uc, vc = u.cuda(), v.cuda()
    Wsn, sg = ops.spectral_norm(W.cuda(), uc, vc, training)
    assert _rel(Wsn.cpu(), ref.detach()) <= 1e-5 and abs(float(sg) - float(sigma.detach())) <= 1e-5 * abs(float(sigma.detach()))
    assert _rel(uc.cpu(), ud) <= 1e-5 and _rel(vc.cpu(), vd) <= 1e-5          # written back in training mode, untouched otherwise
    gW = ops.spectral_norm_bwd(gout.cuda(), Wsn, uc, vc, sg)
    assert _rel(gW.cpu(), Wd.grad) <= 1e-4


def _attention_ref(qkv, H, keep, keep_prob):
    S, D3 = qkv.shape
    D, dh = D3 // 3, D3 // 3 // H
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    outs = []
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        p = torch.softmax((q[:, sl] @ k[:, sl].t()) / np.sqrt(dh), dim=-1)
        if keep is not None:
            p = p * keep[h].double() / keep_prob
        outs.append(p @ v[:, sl])
    return torch.cat(outs, dim=1)


@pytest.mark.parametrize("S,D,H,drop", [(80, 208, 8, False), (203, 64, 8, True), (1, 64, 8, False), (17, 256, 8, True),
                                        (1000, 208, 8, False), (65, 32, 4, True),
                                        # one case per padded head width (8, 16, 32 columns) with heads narrower than the padding
                                        (130, 128, 8, True), (77, 24, 8, False), (300, 104, 8, True), (2100, 64, 8, False)])
def test_attention_forward_backward(ops, S, D, H, drop):
    g = torch.Generator().manual_seed(S + D)
    qkv = torch.randn(S, 3 * D, generator=g)
    gout = torch.randn(S, D, generator=g)
    keep = (torch.rand(H, S, S, generator=g) >= 0.1).to(torch.uint8) if drop else None
    kp = 0.9 if drop else 1.0
    qd = qkv.double().requires_grad_(True)
    ref = _attention_ref(qd, H, keep, kp)
    (ref * gout.double()).sum().backward()
    kc = keep.cuda() if drop else None
    out, lse = ops.attention(qkv.cuda(), H, kc, kp)
    assert _rel(out.cpu(), ref.detach()) <= 1e-5
    gq = ops.attention_bwd(qkv.cuda(), out, gout.cuda(), lse, H, kc, kp)
    assert _rel(gq.cpu(), qd.grad) <= 1e-4


@pytest.mark.parametrize("N,E,with_b", [(80, 208, True), (203, 64, True), (5, 256, False), (70000, 32, True), (3, 8, True)])
def test_add_layernorm_forward_backward(ops, N, E, with_b):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N + E)
    a, b = torch.randn(N, E, generator=g), torch.randn(N, E, generator=g)
    w, bias, gy = torch.randn(E, generator=g), torch.randn(E, generator=g), torch.randn(N, E, generator=g)
    ad, bd, wd, biasd = (t.double().requires_grad_(True) for t in (a, b, w, bias))
    ref = F.layer_norm(ad + bd if with_b else ad, (E,), wd, biasd, 1e-5)
    (ref * gy.double()).sum().backward()
    y, xhat, rstd = ops.add_layernorm(a.cuda(), b.cuda() if with_b else None, w.cuda(), bias.cuda(), 1e-5, save=True)
    assert _rel(y.cpu(), ref.detach()) <= 1e-5
    gx, gw, gb = ops.layernorm_bwd(gy.cuda(), xhat, rstd, w.cuda())
    assert _rel(gx.cpu(), ad.grad) <= 1e-4 and _rel(gw.cpu(), wd.grad) <= 1e-4 and _rel(gb.cpu(), biasd.grad) <= 1e-4


def _module(golden_dir, name):
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.default_branch import DefaultActionNetwork
    g = np.load(golden_dir / f"{name}.npz")
    seed, A, D, N = (int(g[k]) for k in ("seed", "A", "D", "N"))
    p = O.make_default_branch_params(seed, A, D)
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    assert h.hexdigest() == str(g["params_sha256"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DefaultActionNetwork(A, D)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return g, m.cuda()


@pytest.mark.parametrize("name", ["default_icrt", "default_a7_d64"])
def test_module_matches_the_stock_modules(golden_dir, name):
    g, m = _module(golden_dir, name)
    x = torch.from_numpy(g["x"]).cuda()
    tol = max(1e-5, 4.0 * float(g["fp32_noise"]))           # fp32 rounding alone moves the ICRT-width output by ~2e-4
    m.eval()
    with torch.no_grad():
        y = m(x)
    assert y.shape == g["y"].shape and _rel(y.cpu().numpy(), g["y"]) <= tol
    u_before = m[0].weight_u.clone()
    y2 = m(x)                                                # eval with autograd: same values, buffers untouched
    assert torch.equal(y2.detach(), y) and torch.equal(m[0].weight_u, u_before)
    # one training-mode call with every dropout probability at 0: output, spectral-norm buffers, parameter gradients
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    yt = m(x)
    assert _rel(yt.detach().cpu().numpy(), g["y_train"]) <= tol
    for i in (0, 2, 4):
        assert np.allclose(m[i].weight_u.cpu().numpy(), g[f"train_u/{i}"], rtol=0, atol=2e-6)
        assert np.allclose(m[i].weight_v.cpu().numpy(), g[f"train_v/{i}"], rtol=0, atol=2e-6)
    (yt * torch.from_numpy(g["r"]).cuda()).sum().backward()
    gtol = max(1e-4, 40.0 * float(g["fp32_noise"]))
    for k, t in m.named_parameters():
        want, got = g["gdig/" + k], O.grad_digest(t.grad.cpu().numpy())
        assert np.abs(got - want).max() <= gtol * max(1e-6, want[1]), (k, np.abs(got - want).max(), want[1])


def test_training_mode_dropout_and_branch_shim(golden_dir):
    """p = 0.1 (the reference's): finite, different from eval, gradients reach every parameter; the group-encoder shim builds
    this branch when neither tokenizer switch is set (obs_nets.py:1244) and returns [B*T, D] without a tokenizer loss."""
    g, m = _module(golden_dir, "default_a7_d64")
    x = torch.from_numpy(g["x"]).cuda()
    m.train()
    torch.manual_seed(3)
    y = m(x)
    assert torch.isfinite(y).all() and _rel(y.detach().cpu().numpy(), g["y"]) > 1e-3
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    from lipvq_vae_amd.icl import ICLActionBranch, time_distributed
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        br = ICLActionBranch(7, 64, vq_vae_enabled=False).cuda().eval()
    out = time_distributed(torch.randn(3, 10, 7, device="cuda"), br)
    assert out.shape == (3, 10, 64) and br._vq_vae_loss is None


def test_graphed_eval_forward_equals_eager(golden_dir):
    from lipvq_vae_amd.default_branch import GraphedDefaultBranch
    g, m = _module(golden_dir, "default_icrt")
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    with torch.no_grad():
        want = m(x).clone()
    gr = GraphedDefaultBranch(m, x)
    assert torch.equal(gr(x), want)
    x2 = x.flip(0).contiguous()
    with torch.no_grad():
        want2 = m(x2).clone()
    assert torch.equal(gr(x2), want2)                       # new inputs through the same graph
    with torch.no_grad():
        m[6].bias.add_(1.0)                                 # an in-place parameter update is seen by the next replay
    assert torch.allclose(gr(x2), want2 + 1.0, rtol=0, atol=1e-5)
    with pytest.raises(ValueError):
        gr(x[:10])
