"""CPU: the host glue of SURVEY 8f row 1 -- the nested-dict `icl_time_distributed` and the group-encoder-shaped shim -- with
stand-in ops (no kernel runs here; the action network itself is covered on the GPU in tests/test_gpu_icl.py)."""
import importlib.util
from collections import OrderedDict
from pathlib import Path

import pytest
import torch
import torch.nn as nn

import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd import icl

REF_TU = Path("/root/reference/robomimic/utils/tensor_utils.py")


class _ObsEnc(nn.Module):
    """Stand-in for an ObservationEncoder: concatenates its modalities and applies one Linear."""
    def __init__(self, shapes, feat):
        super().__init__()
        self.keys = list(shapes)
        self.lin = nn.Linear(sum(shapes.values()), feat)
        self.feat = feat

    def forward(self, obs_dict):
        return self.lin(torch.cat([obs_dict[k] for k in self.keys], dim=-1))

    def output_shape(self):
        return [self.feat]


def _inputs(B, T):
    g = torch.Generator().manual_seed(3)
    return {"obs": OrderedDict(eef=torch.randn(B, T, 3, generator=g), joint=torch.randn(B, T, 7, generator=g)),
            "prompt": {"obs": OrderedDict(eef=torch.randn(B, T, 3, generator=g), joint=torch.randn(B, T, 7, generator=g)),
                       "action": torch.randn(B, T, 12, generator=g)}}


def test_icl_time_distributed_maps_nested_inputs_and_reshapes_the_triple():
    B, T = 4, 10
    inp = _inputs(B, T)
    seen = {}

    def op(obs=None, prompt=None, scale=1.0):
        seen["shapes"] = (obs["eef"].shape, prompt["obs"]["joint"].shape, prompt["action"].shape)
        return obs["eef"] * scale, {"a": prompt["obs"]["joint"], "b": None}, [prompt["action"][:, :5]]

    o, co, ca = icl.icl_time_distributed(inp, op, inputs_as_kwargs=True, scale=2.0, activation=torch.tanh)
    assert seen["shapes"] == (torch.Size([B * T, 3]), torch.Size([B * T, 7]), torch.Size([B * T, 12]))
    assert torch.equal(o, torch.tanh(2.0 * inp["obs"]["eef"]))
    assert isinstance(co, dict) and co["b"] is None and torch.equal(co["a"], torch.tanh(inp["prompt"]["obs"]["joint"]))
    assert isinstance(ca, list) and ca[0].shape == (B, T, 5)
    # args / plain forms
    o2, _, _ = icl.icl_time_distributed([inp["obs"]["eef"], inp["obs"]["joint"]], lambda a, b: (a, b, a), inputs_as_args=True)
    assert torch.equal(o2, inp["obs"]["eef"])
    o3, _, _ = icl.icl_time_distributed(inp["obs"], lambda d: (d["eef"], d["joint"], d["eef"]))
    assert torch.equal(o3, inp["obs"]["eef"])
    with pytest.raises(ValueError):
        icl.icl_time_distributed({}, op)


@pytest.mark.skipif(not REF_TU.exists(), reason="the reference tree is not on this machine")
def test_icl_time_distributed_equals_the_reference_function():
    spec = importlib.util.spec_from_file_location("_ref_tensor_utils", REF_TU)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    inp = _inputs(3, 10)

    def op(obs=None, prompt=None):
        return obs["joint"] + 1, prompt["obs"]["eef"] * 2, prompt["action"].sum(-1, keepdim=True)

    got = icl.icl_time_distributed(inp, op, inputs_as_kwargs=True)
    want = ref.icl_time_distributed(inp, op, inputs_as_kwargs=True)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_group_encoder_shim_routes_like_the_reference(monkeypatch):
    B, T, A = 2, 10, 12
    encs = OrderedDict(obs=_ObsEnc({"eef": 3, "joint": 7}, 16))
    enc = icl.ICLObservationGroupEncoder(encs, action_input_shape=A, vq_vae_enabled=True)
    assert enc.output_shape() == [16]
    assert type(enc.action_network).__name__ == "LLFQVAE_V4" and enc.action_network.latent_dim == 16      # obs_nets.py:1193,1225-1227
    assert enc.action_network.feature_dim == A

    class _Tok(nn.Module):                       # CPU stand-in with the tokenizer's (z_latent, loss) contract
        def forward(self, a):
            return a[:, :4].detach() * 3.0, a.pow(2).mean()

    monkeypatch.setattr(enc, "action_network", _Tok())
    inp = _inputs(B, T)
    obs, cobs, cact = icl.icl_time_distributed(inp, enc, inputs_as_kwargs=True)        # the call of obs_nets.py:2571
    assert obs.shape == (B, T, 16) and cobs.shape == (B, T, 16) and cact.shape == (B, T, 4)
    flat = lambda t: t.reshape(B * T, -1)
    assert torch.equal(flat(obs), encs["obs"]({k: flat(v) for k, v in inp["obs"].items()}))
    assert torch.equal(flat(cobs), encs["obs"]({k: flat(v) for k, v in inp["prompt"]["obs"].items()}))
    assert torch.equal(flat(cact), flat(inp["prompt"]["action"])[:, :4] * 3.0)
    assert torch.equal(enc._vq_vae_loss, inp["prompt"]["action"].pow(2).mean())
    with pytest.raises(AssertionError):
        enc(prompt=inp["prompt"])                # an observation group is missing
    with pytest.raises(NotImplementedError):
        icl.ICLObservationGroupEncoder(encs, action_input_shape=A, fast_enabled=True)
    # the other switches build the sibling branches (obs_nets.py:1214-1217, 1244-1260)
    assert type(icl.ICLObservationGroupEncoder(encs, A, bin_enabled=True).action_network).__name__ == "AdaptiveBinActionEmbedding"
    assert type(icl.ICLObservationGroupEncoder(encs, A).action_network).__name__ == "DefaultActionNetwork"


# The keys a reference checkpoint holds for the group encoder (algo.py:323-337 saves nets.state_dict(); the encoder's own
# attributes are `nets` and `action_network`, obs_nets.py:1182-1260).  Restated here key by key -- shapes for A = 12, D = 16.
_A, _D = 12, 16
REF_KEYS_LLFQ = {"encoder.0.weight": (64, _A), "encoder.0.bias": (64,), "encoder.2.weight": (128, 64), "encoder.2.bias": (128,),
                 "to_latent.W": (_D, 128), "to_latent.b": (_D,), "to_latent.ci": (_D,), "quantizer.codebook": (1024, _D),
                 "decoder.0.weight": (64, _D), "decoder.0.bias": (64,), "decoder.2.weight": (128, 64), "decoder.2.bias": (128,),
                 "to_output.weight": (_A, 128), "to_output.bias": (_A,)}
REF_KEYS_VQ = {"encoder.0.weight": (64, _A), "encoder.0.bias": (64,), "encoder.2.weight": (128, 64), "encoder.2.bias": (128,),
               "encoder.4.weight": (_D, 128), "encoder.4.bias": (_D,), "decoder.0.weight": (128, _D), "decoder.0.bias": (128,),
               "decoder.2.weight": (64, 128), "decoder.2.bias": (64,), "decoder.4.weight": (_A, 64), "decoder.4.bias": (_A,),
               "embedding.weight": (128, _D)}
REF_KEYS_BIN = {"running_min": (_A,), "running_max": (_A,), **{f"embedding_layers.{i}.weight": (20, 64) for i in range(_A)},
                "output_layer.0.weight": (32 * _A, 64 * _A), "output_layer.0.bias": (32 * _A,),
                "output_layer.2.weight": (_D, 32 * _A), "output_layer.2.bias": (_D,)}


def _ref_keys_default():
    k = {}
    for i, (o, n) in zip((0, 2, 4), ((64, _A), (128, 64), (_D, 128))):         # spectral_norm(Linear): obs_nets.py:1252-1256
        k.update({f"{i}.bias": (o,), f"{i}.weight_orig": (o, n), f"{i}.weight_u": (o,), f"{i}.weight_v": (n,)})
    for l in range(4):                                                          # TransformerEncoder(num_layers=4): :1257
        p = f"5.layers.{l}."
        k.update({p + "self_attn.in_proj_weight": (3 * _D, _D), p + "self_attn.in_proj_bias": (3 * _D,),
                  p + "self_attn.out_proj.weight": (_D, _D), p + "self_attn.out_proj.bias": (_D,),
                  p + "linear1.weight": (256, _D), p + "linear1.bias": (256,), p + "linear2.weight": (_D, 256),
                  p + "linear2.bias": (_D,), p + "norm1.weight": (_D,), p + "norm1.bias": (_D,), p + "norm2.weight": (_D,),
                  p + "norm2.bias": (_D,)})
    k.update({"6.weight": (_D, _D), "6.bias": (_D,)})                           # :1258
    return k


@pytest.mark.parametrize("switches, ref_keys", [
    (dict(vq_vae_enabled=True), REF_KEYS_LLFQ), (dict(vq_vae_enabled=True, variant="vqvae"), REF_KEYS_VQ),
    (dict(bin_enabled=True), REF_KEYS_BIN), (dict(), None)], ids=["lipvq", "vqvae", "bin", "default"])
def test_group_encoder_loads_a_reference_shaped_checkpoint(switches, ref_keys):
    """the shim's state_dict must be `nets.<group>.*` + `action_network.*` so that the encoder part of a
    reference checkpoint loads with strict=True, and `checkpoint.insert_tokenizer_state` round-trips through it."""
    from lipvq_vae_amd import checkpoint
    ref_keys = _ref_keys_default() if ref_keys is None else ref_keys
    encs = OrderedDict(obs=_ObsEnc({"eef": 3, "joint": 7}, _D))
    enc = icl.ICLObservationGroupEncoder(encs, action_input_shape=_A, **switches)
    g = torch.Generator().manual_seed(5)
    ckpt = OrderedDict((f"nets.obs.{k}", torch.randn(v.shape, generator=g)) for k, v in encs["obs"].state_dict().items())
    for k, shape in ref_keys.items():
        ckpt["action_network." + k] = torch.randn(shape, generator=g)
    assert list(enc.state_dict().keys()) == list(ckpt.keys())                  # same keys, same order
    enc.load_state_dict(ckpt, strict=True)
    for k, v in ckpt.items():
        assert torch.equal(enc.state_dict()[k], v), k
    if "vq_vae_enabled" in switches:                                            # the tokenizers: checkpoint.py's round trip
        full = OrderedDict(("policy.nets.encoder." + k, v.clone()) for k, v in ckpt.items())
        variant, state, prefix = checkpoint.extract_tokenizer_state({"model": full})
        assert prefix == "policy.nets.encoder.action_network." and variant == switches.get("variant", "lipvq")
        enc2 = icl.ICLObservationGroupEncoder(OrderedDict(obs=_ObsEnc({"eef": 3, "joint": 7}, _D)), _A, **switches)
        enc2.action_network.load_state_dict(state, strict=True)
        back = checkpoint.insert_tokenizer_state(OrderedDict(), enc2.action_network, prefix=prefix)
        assert list(back.keys()) == [k for k in full if k.startswith(prefix)]
        for k, v in back.items():
            assert torch.equal(v, full[k]), k
