"""GPU: the opt-in EMA codebook update (extension; lipvq_ema_update_f32 + lipvq_vae_amd.ema.EMACodebook) against the
oracle's restatement of the standard rule and a plain torch spelling of it (tolerance 1e-5 relative: the per-code sums
come from fp32 atomics over up to a few hundred rows per code; the oracle sums them in double)."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def _torch_rule(cs, es, counts, dw, decay, eps):
    cs = decay * cs + (1 - decay) * counts.float()
    es = decay * es + (1 - decay) * dw
    n = cs.sum()
    sm = (cs + eps) / (n + cs.numel() * eps) * n
    return cs, es, es / sm[:, None]


@pytest.mark.parametrize("K,D,N", [(1024, 64, 20000), (37, 32, 500), (8192, 128, 3000)])
def test_ema_update_matches_oracle_and_torch(oracle, K, D, N):
    from lipvq_vae_amd.ema import EMACodebook
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A = 7
    p = O.make_params(K + D, A, D, K, oracle=oracle)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    ema = EMACodebook(model.quantizer.codebook, decay=0.97, eps=1e-5)
    cs, es = np.zeros(K, np.float32), p["quantizer.codebook"].copy()
    cb_ref = p["quantizer.codebook"].copy()
    for step in range(3):
        x = O.make_inputs(step, N, A)
        xt = torch.from_numpy(x).cuda()
        with torch.no_grad():
            ze = model.encode(xt)
            idx, _ = model._quantize(ze, None)
        ze_ref = oracle.mlp3(x, p["encoder.0.weight"], p["encoder.0.bias"], p["encoder.2.weight"], p["encoder.2.bias"],
                             oracle.lipschitz_scale(p["to_latent.W"], p["to_latent.ci"])[1], p["to_latent.b"],
                             (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID))
        idx_ref, _, counts = oracle.nearest(ze_ref, cb_ref)
        assert np.array_equal(idx.cpu().numpy(), idx_ref), step            # the tokenizer sees the UPDATED codebook
        dw = np.zeros((K, D), np.float64)
        np.add.at(dw, idx_ref, ze_ref.astype(np.float64))
        t_cs, t_es, t_cb = _torch_rule(torch.from_numpy(cs), torch.from_numpy(es), torch.from_numpy(counts),
                                       torch.from_numpy(dw.astype(np.float32)), 0.97, 1e-5)
        cs, es, cb_ref = oracle.ema_update(cs, es, counts, dw.astype(np.float32), 0.97, 1e-5)
        got_counts = ema.update(ze, idx)
        assert np.array_equal(got_counts.cpu().numpy(), counts)
        for got, want, tw in ((ema.cluster_size, cs, t_cs), (ema.embed_sum, es, t_es), (model.quantizer.codebook.detach(), cb_ref, t_cb)):
            g = got.cpu().numpy()
            assert np.all(np.abs(g - want) <= 1e-5 * (1 + np.abs(want)))
            assert np.all(np.abs(g - tw.numpy()) <= 1e-5 * (1 + np.abs(tw.numpy())))
        cb_ref = model.quantizer.codebook.detach().cpu().numpy().copy()    # follow the GPU's last bits for the next step


def test_ema_rejects_bad_arguments():
    from lipvq_vae_amd import ops
    from lipvq_vae_amd._capi import LipvqLibraryError
    K, D = 8, 4
    cs, es = torch.zeros(K, device="cuda"), torch.zeros((K, D), device="cuda")
    counts, dw, cb = torch.zeros(K, dtype=torch.int64, device="cuda"), torch.zeros((K, D), device="cuda"), torch.zeros((K, D), device="cuda")
    with pytest.raises(LipvqLibraryError):
        ops.ema_update(cs, es, counts, dw, cb, 1.5, 1e-5)
    with pytest.raises(ValueError):
        ops.ema_update(cs, es, counts.float(), dw, cb, 0.9, 1e-5)
