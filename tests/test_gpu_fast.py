"""GPU: the opt-in fast mode (fp16 encoder GEMMs, lipvq_tokenize_fast_f32).  It is NOT a parity path; what is asserted:
few indices differ from parity mode, every difference is between near-equidistant codes (measured with the oracle's exact
distances from the fp32 z_e), outputs are well-formed (z_q rows are codebook rows, usage sums to N)."""
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


@pytest.mark.parametrize("N,A,D,K", [(60000, 7, 64, 1024), (20000, 12, 32, 256), (9000, 7, 128, 2000), (77, 7, 64, 1024)])
def test_fast_mode_flips_only_near_ties(oracle, N, A, D, K):
    p, model = _setup(N + K, A, D, K, oracle)
    x = O.make_inputs(N, N, A)
    xt = torch.from_numpy(x).cuda()
    idx_p, zq_p = model.tokenize(xt, count_usage=False)
    model.code_usage.zero_()
    idx_f, zq_f = model.tokenize(xt, mode="fast")
    cb = model.quantizer.codebook.detach()
    assert torch.equal(zq_f, cb[idx_f]) and int(model.code_usage.sum()) == N
    assert int(idx_f.min()) >= 0 and int(idx_f.max()) < K
    diff = (idx_f != idx_p).cpu().numpy()
    assert diff.mean() <= 0.01, diff.mean()                         # fp16 operands: a fraction of a percent
    if diff.any():
        rows = np.nonzero(diff)[0]
        ze = oracle.llfq_encode(p, x[rows])
        d = oracle.distances(ze, p["quantizer.codebook"])           # exact distances from the fp32 z_e
        ip, jf = idx_p.cpu().numpy()[rows], idx_f.cpu().numpy()[rows]
        dp, df = d[np.arange(rows.size), ip], d[np.arange(rows.size), jf]
        assert np.all(df >= dp)                                     # parity picked the true minimum
        assert np.max((df - dp) / dp) <= 5e-3                       # the fast pick is within 0.5 % of it
    # parity mode is untouched by the fast call (separate weight caches)
    idx_p2, _ = model.tokenize(xt, count_usage=False)
    assert torch.equal(idx_p2, idx_p)


def test_fast_mode_rejects_what_it_cannot_do(oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    m = LLFQVAE_V4(12, 208, num_codes=128).cuda()                   # D = 208: no fused kernel
    with pytest.raises(RuntimeError):
        m.tokenize(torch.randn(10, 12, device="cuda"), mode="fast")
    with pytest.raises(ValueError):
        m.tokenize(torch.randn(10, 12, device="cuda"), mode="turbo")
