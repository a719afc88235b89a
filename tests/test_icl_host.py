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

    monkeypatch.setattr(enc.action_branch, "action_network", _Tok())
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
