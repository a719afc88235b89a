"""CPU: host-side behaviour of the drop-in modules that needs no GPU -- constructor signature,
state_dict keys/shapes (checkpoint compatibility, reference algo.py:323-337), refusal to run on CPU."""
import inspect

import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O


def test_llfq_constructor_and_state_dict():
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    sig = inspect.signature(LLFQVAE_V4.__init__)
    assert list(sig.parameters)[1:] == ["feature_dim", "latent_dim", "num_codes", "hidden_dim"]
    assert sig.parameters["num_codes"].default == 1024 and sig.parameters["hidden_dim"].default == 128
    m = LLFQVAE_V4(12, 208)
    sd = m.state_dict()
    assert tuple(sd.keys()) == ("encoder.0.weight", "encoder.0.bias", "encoder.2.weight", "encoder.2.bias",
                                "to_latent.W", "to_latent.b", "to_latent.ci", "quantizer.codebook",
                                "decoder.0.weight", "decoder.0.bias", "decoder.2.weight", "decoder.2.bias",
                                "to_output.weight", "to_output.bias")
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    assert shapes["encoder.0.weight"] == (64, 12) and shapes["encoder.2.weight"] == (128, 64)
    assert shapes["to_latent.W"] == (208, 128) and shapes["to_latent.ci"] == (208,)
    assert shapes["quantizer.codebook"] == (1024, 208) and shapes["to_output.weight"] == (12, 128)
    assert all(v.dtype == torch.float32 for v in sd.values())
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())
    # AdamW over .parameters() is what icl.py:887-889 builds
    torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
    assert torch.all(m.to_latent.ci == 1) and torch.all(m.to_latent.b == 0)


def test_vq_constructor_and_state_dict():
    from lipvq_vae_amd.tokenizer import VQVAE
    sig = inspect.signature(VQVAE.__init__)
    assert list(sig.parameters)[1:] == ["feature_dim", "latent_dim", "num_embeddings", "commitment_cost"]
    m = VQVAE(7, 32)
    assert sorted(m.state_dict().keys()) == sorted(O.VQ_KEYS)
    assert m.embedding.weight.shape == (128, 32)
    assert float(m.embedding.weight.abs().max()) <= 1 / 128


def test_reference_state_dict_loads_strictly(oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    p = O.make_params(3, 7, 32, 64, oracle=oracle)
    m = LLFQVAE_V4(7, 32, num_codes=64)
    missing = m.load_state_dict(O.to_torch(p), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert np.array_equal(m.quantizer.codebook.detach().numpy(), p["quantizer.codebook"])


def test_cpu_tensor_is_refused():
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    m = LLFQVAE_V4(7, 32, num_codes=64)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.randn(4, 7))
    with pytest.raises(RuntimeError, match="GPU only"):
        m.tokenize(torch.randn(4, 7))


def test_unsupported_hidden_dim_is_an_error():
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    with pytest.raises(ValueError):
        LLFQVAE_V4(7, 32, hidden_dim=100)


def test_perplexity_host_math():
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    m = LLFQVAE_V4(7, 32, num_codes=8)
    m.code_usage[:4] = 5
    assert abs(m.perplexity() - 4.0) < 1e-9
    m.reset_usage()
    assert int(m.code_usage.sum()) == 0
