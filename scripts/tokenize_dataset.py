#!/usr/bin/env python3
"""Offline bulk tokenizer: actions -> code indices with a reference checkpoint.

    python scripts/tokenize_dataset.py --ckpt model.pth --actions actions.npy --out tokens.npz [--latents]

--actions: .npy / .npz (key --key, default "actions") holding [T, A], [B, T, A] or [N, A] float32 actions
           (the reference's HDF5 layout is data/<demo>/actions [T, A], robomimic/utils/dataset.py:559-573; HDF5
           input is accepted when h5py is installed: --actions file.hdf5 tokenizes every demo).
Runs the fused MI355X launch (encode + quantize) in batches of --rows rows resident in HBM.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def load_actions(path, key):
    p = Path(path)
    if p.suffix == ".npy":
        return {"actions": np.load(p)}
    if p.suffix == ".npz":
        return {key: np.load(p)[key]}
    if p.suffix in (".hdf5", ".h5"):
        import h5py   # not part of this image; present in the reference's environment
        with h5py.File(p, "r") as f:
            return {d: f["data"][d]["actions"][()] for d in f["data"]}
    raise SystemExit(f"unsupported input {p.suffix}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--actions", required=True)
    ap.add_argument("--key", default="actions")
    ap.add_argument("--out", required=True)
    ap.add_argument("--rows", type=int, default=4096 * 128)
    ap.add_argument("--latents", action="store_true", help="also store z_latent")
    args = ap.parse_args()
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.checkpoint import tokenizer_from_checkpoint
    model = tokenizer_from_checkpoint(args.ckpt, "cuda").eval()
    out, n_total, t_total = {}, 0, 0.0
    for name, a in load_actions(args.actions, args.key).items():
        a = np.ascontiguousarray(a, dtype=np.float32)
        flat = a.reshape(-1, a.shape[-1])
        idx_parts, z_parts = [], []
        for s in range(0, flat.shape[0], args.rows):
            x = torch.from_numpy(flat[s:s + args.rows]).cuda()
            torch.cuda.synchronize(); t = time.perf_counter()
            idx, z = model.tokenize(x)
            torch.cuda.synchronize(); t_total += time.perf_counter() - t
            idx_parts.append(idx.cpu().numpy().astype(np.int32))
            if args.latents:
                z_parts.append(z.cpu().numpy())
        out[f"{name}/indices"] = np.concatenate(idx_parts).reshape(a.shape[:-1])
        if args.latents:
            out[f"{name}/z_latent"] = np.concatenate(z_parts).reshape(*a.shape[:-1], -1)
        n_total += flat.shape[0]
    out["code_usage"] = model.code_usage.cpu().numpy()
    np.savez_compressed(args.out, **out)
    print(f"tokenized {n_total} actions in {t_total * 1e3:.2f} ms of GPU time ({n_total / max(t_total, 1e-9) / 1e6:.1f} M actions/s), "
          f"perplexity {model.perplexity():.1f}, wrote {args.out}")


if __name__ == "__main__":
    main()
