"""Checkpoint compatibility with the reference.

The reference saves ``torch.save({"model": algo.nets.state_dict(), "config": ..., ...})``
(robomimic/utils/train_utils.py:1186-1235, robomimic/algo/algo.py:323-337); inside ``model`` the tokenizer's
14 tensors live under ``policy.nets.encoder.action_network.*`` (the ``action_network`` attribute of
``ICLObservationGroupEncoder``, robomimic/models/obs_nets.py:1225).  These helpers find that sub-dict in a
checkpoint (whatever the exact prefix), build the MI355X tokenizer with the dimensions the tensors imply, and
write a tokenizer's parameters back under the same prefix so that the reference can load the result.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Mapping, Tuple

import torch

from .tokenizer import LLFQVAE_V4, VQVAE

_LLFQ_MARK = "quantizer.codebook"
_VQ_MARK = "embedding.weight"


def _model_dict(ckpt) -> Mapping[str, torch.Tensor]:
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    if isinstance(ckpt, Mapping) and "model" in ckpt and isinstance(ckpt["model"], Mapping):
        ckpt = ckpt["model"]
    flat = OrderedDict()
    for k, v in ckpt.items():                     # algo.serialize() may nest one level: {"policy": state_dict}
        if isinstance(v, Mapping):
            for k2, v2 in v.items():
                flat[f"{k}.{k2}"] = v2
        else:
            flat[k] = v
    return flat


def find_tokenizer_prefix(ckpt) -> Tuple[str, str]:
    """(prefix, variant) with variant in {"lipvq", "vqvae"}; prefix includes the trailing dot ('' if none)."""
    sd = _model_dict(ckpt)
    hits = [(k[: -len(_LLFQ_MARK)], "lipvq") for k in sd if k.endswith(_LLFQ_MARK)]
    hits += [(k[: -len(_VQ_MARK)], "vqvae") for k in sd if k.endswith(_VQ_MARK) and (k[: -len(_VQ_MARK)] + "encoder.4.weight") in sd]
    if len(hits) != 1:
        raise KeyError(f"expected exactly one action tokenizer in the checkpoint, found {len(hits)}: {[h[0] for h in hits]}")
    return hits[0]


def extract_tokenizer_state(ckpt):
    """(variant, state_dict with the reference module's own keys, prefix)."""
    sd = _model_dict(ckpt)
    prefix, variant = find_tokenizer_prefix(sd)
    keys = LLFQVAE_V4(1, 1, num_codes=1, hidden_dim=32).state_dict().keys() if variant == "lipvq" else VQVAE(1, 1, 1).state_dict().keys()
    out = OrderedDict()
    for k in keys:
        if prefix + k not in sd:
            raise KeyError(f"checkpoint lacks {prefix + k}")
        out[k] = sd[prefix + k].detach().to(torch.float32)
    return variant, out, prefix


def tokenizer_from_checkpoint(ckpt, device="cuda"):
    """Build LLFQVAE_V4 / VQVAE with the dimensions the checkpoint's tensors imply and load them (strict)."""
    variant, state, _ = extract_tokenizer_state(ckpt)
    if variant == "lipvq":
        model = LLFQVAE_V4(feature_dim=state["encoder.0.weight"].shape[1], latent_dim=state["to_latent.W"].shape[0],
                           num_codes=state["quantizer.codebook"].shape[0], hidden_dim=state["encoder.2.weight"].shape[0])
    else:
        model = VQVAE(feature_dim=state["encoder.0.weight"].shape[1], latent_dim=state["encoder.4.weight"].shape[0],
                      num_embeddings=state["embedding.weight"].shape[0])
    model.load_state_dict(state, strict=True)
    return model.to(device)


def insert_tokenizer_state(model_state: dict, tokenizer, prefix: str = "policy.nets.encoder.action_network.") -> dict:
    """Write the tokenizer's parameters into a reference ``model`` state dict under ``prefix`` (in place)."""
    for k, v in tokenizer.state_dict().items():
        model_state[prefix + k] = v.detach().cpu().clone()
    return model_state
