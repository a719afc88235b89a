"""GPU: the ICRT-side glue -- several tokenizer-optimizer steps track the torch-CPU restatement of the
reference run with the same choreography (icl.py:885-889, 913-914, 968-970)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def test_action_branch_training_loop_tracks_reference(oracle):
    from lipvq_vae_amd.icl import ICLActionBranch, VQTokenizerTrainer, time_distributed
    A, D, K, B, T = 12, 208, 128, 8, 10           # the real ICRT step shape (image mode: 8 x 10 prompt actions)
    p = O.make_params(31, A, D, K, oracle=oracle)
    branch = ICLActionBranch(A, D).cuda()
    # the branch builds the default K=1024 tokenizer; swap in a K=128 one to keep the CPU side quick
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    branch.action_network = LLFQVAE_V4(A, D, num_codes=K).cuda()
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
