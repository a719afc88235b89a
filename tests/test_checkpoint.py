"""Checkpoint compatibility (reference keys under policy.nets.encoder.action_network.*) and the bulk tokenizer CLI."""
import subprocess
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

ROOT = Path(__file__).resolve().parent.parent
PREFIX = "policy.nets.encoder.action_network."


def _fake_reference_checkpoint(p, extra=True):
    model = OrderedDict()
    if extra:    # neighbours a real checkpoint has (obs encoders, GPT backbone): must be ignored
        model["policy.nets.encoder.nets.obs.obs_nets.robot0_eef_pos.weight"] = torch.zeros(3)
        model["policy.nets.transformer.nets.h.0.ln_1.weight"] = torch.ones(8)
    for k, v in p.items():
        model[PREFIX + k] = torch.from_numpy(v.copy())
    return {"model": model, "config": "{}", "algo_name": "icl", "env_metadata": {}, "shape_metadata": {}}


def test_extract_build_and_insert_roundtrip(oracle):
    from lipvq_vae_amd.checkpoint import extract_tokenizer_state, find_tokenizer_prefix, insert_tokenizer_state, tokenizer_from_checkpoint
    p = O.make_params(5, 12, 208, 128, oracle=oracle)
    ck = _fake_reference_checkpoint(p)
    assert find_tokenizer_prefix(ck) == (PREFIX, "lipvq")
    variant, state, prefix = extract_tokenizer_state(ck)
    assert variant == "lipvq" and tuple(state) == O.LLFQ_KEYS and prefix == PREFIX
    tok = tokenizer_from_checkpoint(ck, device="cpu")
    assert (tok.feature_dim, tok.latent_dim, tok.num_codes, tok.hidden_dim) == (12, 208, 128, 128)
    for k in O.LLFQ_KEYS:
        assert np.array_equal(tok.state_dict()[k].numpy(), p[k])
    with torch.no_grad():
        tok.to_latent.b.add_(1.0)
    back = insert_tokenizer_state(ck["model"], tok)
    assert np.allclose(back[PREFIX + "to_latent.b"].numpy(), p["to_latent.b"] + 1.0)
    # a VQVAE checkpoint is recognised by its embedding + third encoder layer
    pv = O.make_params(6, 7, 32, 64, variant="vq", oracle=oracle)
    ckv = {"model": OrderedDict((PREFIX + k, torch.from_numpy(v.copy())) for k, v in pv.items())}
    tokv = tokenizer_from_checkpoint(ckv, device="cpu")
    assert type(tokv).__name__ == "VQVAE" and tokv.num_embeddings == 64
    with pytest.raises(KeyError):
        find_tokenizer_prefix({"model": OrderedDict(a=torch.zeros(1))})


@pytest.mark.gpu
def test_bulk_tokenizer_cli(tmp_path, oracle):
    p = O.make_params(7, 7, 64, 256, oracle=oracle)
    torch.save(_fake_reference_checkpoint(p), tmp_path / "model.pth")
    acts = O.make_inputs(7, 40 * 25, 7).reshape(40, 25, 7)
    np.save(tmp_path / "actions.npy", acts)
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "tokenize_dataset.py"), "--ckpt", str(tmp_path / "model.pth"),
                        "--actions", str(tmp_path / "actions.npy"), "--out", str(tmp_path / "tok.npz"), "--latents", "--rows", "300"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.load(tmp_path / "tok.npz")
    ref = oracle.llfq_forward(p, acts.reshape(-1, 7))
    assert np.array_equal(out["actions/indices"].reshape(-1), ref["indices"])
    assert np.array_equal(out["actions/z_latent"].reshape(-1, 64), ref["z_latent"])
    assert np.array_equal(out["code_usage"], ref["usage"])
