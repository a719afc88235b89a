"""GPU: full-size parity pinned to the REFERENCE, not to the builder's oracle.  tests/golden/llfq_*_big.npz hold the
reference module's own indices and best / second-best distances on 65 536 rows of BASELINE config 2's widths, 4 096 rows of
config 3's and 16 384 rows of the reference's own widths (A = 12, D = 208, K = 1024).  The product's tokenize() must
reproduce every index, except on rows whose REFERENCE top-2 relative distance gap is below 1e-6 (the end-to-end fp32 noise
of two different-but-valid fp32 encoders; such rows are counted and reported -- the committed fixtures contain none)."""
import numpy as np
import pytest
import torch

from test_oracle_golden import BIG, assert_indices_match_reference, big_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("screen", ["default", "fine", "coarse"])
@pytest.mark.parametrize("name", BIG)
def test_tokenize_reproduces_reference_indices_full_size(name, screen, oracle, lipvq_option, no_screen_monitor):
    """screen: the library's own choice for the shape, or one of the two screens forced (lipvq_set_option("screen_mode"), read per launch):
    the three-product and the one-product screen must both reproduce the REFERENCE's indices -- whichever a shape defaults to,
    the other stays covered."""
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    if screen != "default":
        lipvq_option("screen_mode", screen)
    p, x, ref, gap = big_fixture(name, oracle)
    K, D = p["quantizer.codebook"].shape
    A = x.shape[1]
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    xt = torch.from_numpy(x).cuda()
    idx, zq = model.tokenize(xt)
    mism, near = assert_indices_match_reference(idx.cpu().numpy(), ref, gap, name)
    print(f"{name}: {mism} mismatches vs the reference, {near} reference near-tie rows (< 1e-6) of {ref.size}")
    assert torch.equal(zq, model.quantizer.codebook.detach()[idx])
    assert int(model.code_usage.sum()) == ref.size
    # the unfused exact route (all-pairs kernel, no screen) gives the same answers
    from lipvq_vae_amd import ops
    idx2, _, _ = ops.nearest(model.encode(xt), model.quantizer.codebook.detach())
    assert torch.equal(idx2, idx)
