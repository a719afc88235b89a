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
