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


def test_input_embedding_module_keys_and_errors():
    """Host side of the embedding stage (no GPU): the reference's state_dict keys, the three time-embedding modes, and
    the errors the reference would raise."""
    from lipvq_vae_amd.embedding import ICLInputEmbedding, sinusoidal_table
    m = ICLInputEmbedding(64, 512, 10)
    assert set(m.state_dict()) == {"nets.embed_encoder.weight", "nets.embed_encoder.bias", "params.embed_timestep",
                                   "nets.embed_ln.weight", "nets.embed_ln.bias"}
    assert m.time_table(10).shape == (10, 512)
    with pytest.raises(ValueError):
        m.time_table(7)                               # nn.Parameter time embedding: T must equal context_length
    e = ICLInputEmbedding(64, 256, 10, nn_parameter_for_timesteps=False)
    assert "nets.embed_timestep.weight" in e.state_dict() and e.time_table(4).shape == (4, 256)
    with pytest.raises(IndexError):
        e.time_table(11)
    s = ICLInputEmbedding(64, 128, 10, sinusoidal_embedding=True)
    assert not any("embed_timestep" in k for k in s.state_dict())
    tab = sinusoidal_table(5, 128, "cpu")
    assert tab.shape == (5, 128) and torch.allclose(tab[0, 0::2], torch.zeros(64)) and torch.allclose(tab[0, 1::2], torch.ones(64))
    with pytest.raises(ValueError):
        ICLInputEmbedding(64, 514, 10)                # embed_dim must be a multiple of 4
    with pytest.raises(RuntimeError):
        m.input_embedding(torch.zeros(2, 10, 64))     # CPU tensors are refused: no fallback


def test_bin_tokenizer_constructor_matches_reference_layout():
    from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding
    torch.manual_seed(5)
    m = AdaptiveBinActionEmbedding(4, 24, num_bins=6, embedding_dim=8, num_step_stop=3)
    torch.manual_seed(5)
    embs = [torch.nn.Embedding(6, 8) for _ in range(4)]
    l0, l2 = torch.nn.Linear(32, 16), torch.nn.Linear(16, 24)
    assert all(torch.equal(a.weight, b.weight) for a, b in zip(m.embedding_layers, embs))
    assert torch.equal(m.output_layer[0].weight, l0.weight) and torch.equal(m.output_layer[2].weight, l2.weight)
    assert torch.isinf(m.running_min).all() and torch.isinf(m.running_max).all() and m._update_enabled
    assert list(m.state_dict())[:2] == ["running_min", "running_max"]
    with pytest.raises(ValueError):
        m(torch.zeros(3, 5))                          # wrong action width
    with pytest.raises(RuntimeError):
        m(torch.zeros(3, 4))                          # CPU tensor


def test_action_branch_switches():
    from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding
    from lipvq_vae_amd.icl import ICLActionBranch
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE
    assert isinstance(ICLActionBranch(7, 32).action_network, LLFQVAE_V4)
    assert isinstance(ICLActionBranch(7, 32, variant="vqvae").action_network, VQVAE)
    b = ICLActionBranch(7, 32, bin_enabled=True)
    assert isinstance(b.action_network, AdaptiveBinActionEmbedding) and not b.vq_vae_enabled     # the reference's elif order
    from lipvq_vae_amd.default_branch import DefaultActionNetwork
    d = ICLActionBranch(7, 32, vq_vae_enabled=False)                     # obs_nets.py:1244: the default branch
    assert isinstance(d.action_network, DefaultActionNetwork) and not d.vq_vae_enabled and not d.bin_enabled
