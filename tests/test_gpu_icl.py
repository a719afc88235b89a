"""GPU: the ICRT-side glue -- several tokenizer-optimizer steps track the torch-CPU restatement of the
reference run with the same choreography (icl.py:885-889, 913-914, 968-970)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [128, 1024])
def test_action_branch_training_loop_tracks_reference(oracle, K):
    from lipvq_vae_amd.icl import ICLActionBranch, VQTokenizerTrainer, time_distributed
    A, D, B, T = 12, 208, 8, 10                   # the real ICRT step shape (image mode: 8 x 10 prompt actions)
    p = O.make_params(31, A, D, K, oracle=oracle)
    branch = ICLActionBranch(A, D).cuda()
    # the branch builds the reference's default K = 1024 tokenizer (v5:52); K = 128 is the authors' __main__ width (v5:92)
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    if K != 1024:
        branch.action_network = LLFQVAE_V4(A, D, num_codes=K).cuda()
    assert branch.action_network.num_codes == K
    branch.action_network.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    trainer = VQTokenizerTrainer(branch.action_network)
    tp = {k: v.requires_grad_(True) for k, v in O.to_torch(p).items()}
    ref_opt = torch.optim.AdamW(list(tp.values()), lr=1e-3, weight_decay=1e-4)
    torch.set_num_threads(4)
    for step in range(4):
        x = O.make_inputs(100 + step, B * T, A).reshape(B, T, A)
        xt = torch.from_numpy(x).cuda()
        ctx = time_distributed(xt, branch)                       # [B, T, D], loss stashed on the branch
        assert ctx.shape == (B, T, D) and not ctx.requires_grad
        loss_fwd = branch._vq_vae_loss
        _, loss = trainer.train_on_actions(xt.reshape(B * T, A))
        ref_opt.zero_grad()
        zl, ref_loss, _ = O.torch_llfq_forward(tp, torch.from_numpy(x.reshape(B * T, A)))
        ref_loss.backward()
        ref_opt.step()
        assert abs(loss_fwd.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item()), step
        assert abs(loss.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item()), step
        assert np.abs(ctx.reshape(B * T, D).cpu().numpy() - zl.numpy()).max() <= 1e-4   # codebook drifts by ~lr per step
    sd = branch.action_network.state_dict()
    for k in O.LLFQ_KEYS:
        got, ref = sd[k].cpu().numpy(), tp[k].detach().numpy()
        # 4 AdamW steps move each weight by <= 4e-3; trajectories agree to a small fraction of that
        close = np.isclose(got, ref, rtol=0, atol=2e-5).mean()
        assert close > 0.995, (k, close)


def test_engine_free_forward_backward_equals_autograd(oracle):
    """autograd.forward_backward (no torch.autograd involved) returns the gradients loss.backward() produces."""
    from lipvq_vae_amd.autograd import forward_backward
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE
    for cls, variant, kw in ((LLFQVAE_V4, "llfq", dict(num_codes=256)), (VQVAE, "vq", dict(num_embeddings=256))):
        A, D, N = 12, 64, 300
        p = O.make_params(41, A, D, 256, variant=variant, oracle=oracle)
        m = cls(A, D, **kw).cuda()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        x = torch.from_numpy(O.make_inputs(41, N, A)).cuda()
        z, loss, params, grads = forward_backward(m, x)
        z2, loss2 = m(x)
        loss2.backward()
        assert torch.equal(z, z2) and torch.equal(loss, loss2.detach())
        for prm, g in zip(params, grads):
            scale = float(prm.grad.abs().max()) + 1e-12
            assert float((prm.grad - g).abs().max()) <= 1e-5 * scale      # codebook scatter-add uses float atomics


def test_bin_branch_feeds_the_embedding_stage(oracle, golden_dir):
    """bin_enabled branch (obs_nets.py:1214-1217) -> [B, T, D] context actions -> dense route of the input embedding:
    trains end to end (policy-side gradients reach the bin tokenizer's parameters) and matches the oracle chain."""
    from lipvq_vae_amd.embedding import ICLInputEmbedding
    from lipvq_vae_amd.icl import ICLActionBranch, time_distributed
    from test_oracle_bin import load_bin
    g, meta, bp = load_bin(golden_dir, "bin_icrt")
    A, D, B, T, E = meta["A"], meta["D"], 8, 10, 512
    branch = ICLActionBranch(A, D, bin_enabled=True).cuda()
    assert branch.bin_enabled and not branch.vq_vae_enabled
    sd = {k: torch.from_numpy(v.copy()) for k, v in bp.items()}
    sd["running_min"], sd["running_max"] = branch.action_network.running_min.cpu(), branch.action_network.running_max.cpu()
    branch.action_network.load_state_dict(sd)
    ep = O.make_embed_params(5, D, E, T, "parameter")
    emb = ICLInputEmbedding(D, E, T, emb_dropout=0.0).cuda()
    emb.load_state_dict({("params." if k == "embed_timestep" else "nets.") + k: torch.from_numpy(v.copy()) for k, v in ep.items()})
    x = g["x0"].reshape(B, T, A)
    rng = np.random.default_rng(0)
    obs, cobs = rng.standard_normal((B, T, D)).astype(np.float32), rng.standard_normal((B, T, D)).astype(np.float32)
    ctx = time_distributed(torch.from_numpy(x).cuda(), branch)
    assert ctx.shape == (B, T, D) and ctx.requires_grad and branch._vq_vae_loss is None
    out = emb(torch.from_numpy(obs).cuda(), torch.from_numpy(cobs).cuda(), ctx)
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in branch.action_network.parameters())
    # oracle chain: bin tokenizer -> dense Linear -> LayerNorm into the 2t+1 slots
    r = oracle.bin_forward(bp, g["x0"], np.full(A, np.inf, np.float32), np.full(A, -np.inf, np.float32))
    want = np.zeros((B, 3 * T, E), np.float32)
    W, b = ep["embed_encoder.weight"], ep["embed_encoder.bias"]
    oracle.embed_rows(oracle.linear(r["out"], W, b), None, ep["embed_timestep"][0], ep["embed_ln.weight"], ep["embed_ln.bias"],
                      1e-5, want, T, 3 * T * E, 2 * E, E)
    assert np.array_equal(out.detach().cpu().numpy()[:, 1:2 * T:2], want[:, 1:2 * T:2])


def test_icrt_training_steps_k1024_vs_reference_fixture(oracle, golden_dir):
    """BASELINE config 5 at the REAL shape (A = 12, D = 208, K = 1024, N = 8 x 10): three tokenizer steps through the product
    (ICLActionBranch + VQTokenizerTrainer: icl.py:913-914, 968-970) against what the REFERENCE module produced
    (tests/golden/llfq_icrt_train_k1024.npz): losses, indices, first-step gradients, final parameters."""
    from lipvq_vae_amd.icl import ICLActionBranch, VQTokenizerTrainer
    g = np.load(golden_dir / "llfq_icrt_train_k1024.npz")
    A, D, K, N, steps, seed = (int(g[k]) for k in ("A", "D", "K", "N", "steps", "seed"))
    p = O.make_params(seed, A, D, K, regime="trained", oracle=oracle)
    assert O.params_digest(p) == str(g["params_sha256"])
    branch = ICLActionBranch(A, D).cuda()
    net = branch.action_network
    assert net.num_codes == K
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    trainer = VQTokenizerTrainer(net)
    for st in range(steps):
        xt = torch.from_numpy(O.make_inputs(seed + st, N, A)).cuda()
        if st == 0:                                                    # first step by hand, to look at its gradients
            trainer.vq_optimizer.zero_grad()
            _, loss = net(xt)
            loss.backward()
            for k, v in net.named_parameters():
                ref = g["grad0/" + k]
                got = v.grad.cpu().numpy()
                if k == "quantizer.codebook":
                    assert abs(np.abs(got.astype(np.float64)).sum() - float(g["grad0_codebook_abs_sum"])) <= 1e-5 * float(g["grad0_codebook_abs_sum"])
                    got = got[g["grad0_rows"]]
                assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-12), k
            trainer.vq_optimizer.step()
        else:
            _, loss = trainer.train_on_actions(xt)
        assert abs(loss.item() - float(g[f"loss{st}"])) <= 2e-5 * abs(float(g[f"loss{st}"])), st
        assert np.array_equal(net.last_indices.cpu().numpy(), g[f"indices{st}"].astype(np.int64)), st
    sd = net.state_dict()
    for k in O.LLFQ_KEYS:
        got = sd[k].cpu().numpy()
        if k == "quantizer.codebook":
            assert abs(got.astype(np.float64).sum() - float(g["post_codebook_sum"])) <= 1e-6 * abs(float(g["post_codebook_sum"]))
            got = got[g["post_rows"]]
        ref = g["post/" + k]
        assert np.abs(got - ref).max() <= 3.2e-3, k                   # three AdamW steps move a weight by <= 3e-3
        assert np.isclose(got, ref, rtol=0, atol=2e-5).mean() > 0.995, k


def test_graphed_training_step_after_eager_training_equals_eager_trajectory(oracle):
    """Round 1's HIP-graph attempt faulted on replay when the model had been trained eagerly before the capture.  Three eager
    steps, then capture (GraphedTokenizerStep: caches invalidated so that every derived buffer lives in the graph's pool), then
    three replays -- interleaved with eager calls that rebuild the Python-side caches, the pattern that used to free buffers the
    graph still referred to -- must follow the all-eager trajectory of a twin model."""
    import copy
    from lipvq_vae_amd.icl import GraphedTokenizerStep, VQTokenizerTrainer
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, N = 12, 208, 1024, 80
    p = O.make_params(77, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    xs = [torch.from_numpy(O.make_inputs(300 + i, N, A)).cuda() for i in range(8)]
    tr = VQTokenizerTrainer(model)
    for i in range(3):
        tr.train_on_actions(xs[i])
    twin = copy.deepcopy(model)
    twin.invalidate_caches()
    tw = VQTokenizerTrainer(twin)
    tw.vq_optimizer.load_state_dict(copy.deepcopy(tr.vq_optimizer.state_dict()))     # (load_state_dict would alias the moments)
    # capture: two warm-up steps on xs[3] (real steps whose effect on parameters and optimizer state is undone: constructing the
    # graphed step must not train the model), then the graph
    before = {k: v.clone() for k, v in model.state_dict().items()}
    usage_before, last_before = model.code_usage.clone(), model.last_indices
    g = GraphedTokenizerStep(model, xs[3], optimizer_state=tr.vq_optimizer.state_dict(), warmup=2)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), f"construction changed {k}"
    # ... nor count the warm-up batches into the usage histogram: buffers, last_indices and the screen monitor
    # are what they were before the construction
    assert torch.equal(model.code_usage, usage_before) and int(usage_before.sum()) == 3 * N
    assert model.last_indices is last_before
    assert model._screen_monitor._pending is None and model._screen_monitor.bypass_calls == 0
    losses = []
    for i in range(4, 7):
        _, loss = g.step(xs[i])
        losses.append(float(loss))
        # eager use of the SAME model between replays (rebuilds packed weights / prepared codebook in the ordinary allocator)
        idx_e, _ = model.tokenize(xs[7], count_usage=False)
        _, ref_loss = tw.train_on_actions(xs[i])
        assert abs(losses[-1] - float(ref_loss)) <= 2e-5 * abs(float(ref_loss)), i
        idx_t, _ = twin.tokenize(xs[7], count_usage=False)
        assert (idx_e != idx_t).float().mean().item() < 0.02          # same parameters up to fp32 noise -> same codes (near-ties aside)
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert torch.isfinite(a).all(), k
        assert float((a - b).abs().max()) <= 5e-5, (k, float((a - b).abs().max()))     # 8 AdamW steps of <= 1e-3 each
    with pytest.raises(ValueError):
        g.step(xs[0][:40])


def test_graphed_training_step_at_a_large_batch_equals_eager(oracle):
    """The same capture at a batch that takes the large-batch routes (fused training forward with the screened quantizer, the loss
    summed by the decoder launch, the loss-gradient terms formed inside the backward chain and the counting-sort scatter, workspace
    memsets as graph nodes): three replays leave the parameters bit-identical to three eager steps of a twin."""
    import copy
    from lipvq_vae_amd.icl import GraphedTokenizerStep, VQTokenizerTrainer
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, N = 7, 64, 1024, 66000
    p = O.make_params(78, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    twin = copy.deepcopy(model)
    twin.invalidate_caches()
    tr, tw = VQTokenizerTrainer(model), VQTokenizerTrainer(twin)
    xs = [torch.from_numpy(O.make_inputs(400 + i, N, A)).cuda() for i in range(4)]
    g = GraphedTokenizerStep(model, xs[0], optimizer_state=tr.vq_optimizer.state_dict(), warmup=2)
    for i in range(1, 4):
        _, loss = g.step(xs[i])
        _, ref_loss = tw.train_on_actions(xs[i])
        assert float(loss) == float(ref_loss), i
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(a, b), k


def test_backward_refuses_parameters_changed_since_forward(oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(5, 7, 32, 128, oracle=oracle)
    m = LLFQVAE_V4(7, 32, num_codes=128).cuda()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    x = torch.from_numpy(O.make_inputs(5, 64, 7)).cuda()
    _, loss = m(x)
    with torch.no_grad():
        m.decoder[0].weight.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        loss.backward()
    # a write behind the version counter's back is what invalidate_caches() is for
    idx0, _ = m.tokenize(x, count_usage=False)
    m.quantizer.codebook.data.add_(0.25)
    m.invalidate_caches()
    idx1, _ = m.tokenize(x, count_usage=False)
    ze = m.encode(x)
    ref, _, _ = __import__("lipvq_vae_amd").ops.nearest(ze, m.quantizer.codebook.detach())
    assert torch.equal(idx1, ref)


def test_fused_adamw_tracks_torch_adamw():
    """lipvq_vae_amd.optim.AdamW (two launches for the whole list) against torch.optim.AdamW on identical parameters and
    gradients over 5 steps, and its state_dict loads into the stock optimizer (and back)."""
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.optim import AdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 12), (64,), (208, 128), (1024, 208), (7,), (1,)]
    pa = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = AdamW(pa, lr=1e-3, weight_decay=1e-4)
    ob = torch.optim.AdamW(pb, lr=1e-3, weight_decay=1e-4)
    for step in range(5):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        if step == 3:
            pa[2].grad = None; pb[2].grad = None             # a parameter without a gradient is skipped, its step does not advance
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), step
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    ob.load_state_dict(sa)                                    # interchangeable layouts
    oa.load_state_dict(ob.state_dict())
    for a in pa:
        a.grad = torch.ones_like(a)
    oa.step()
    with pytest.raises(ValueError):
        AdamW(pa, amsgrad=True)
