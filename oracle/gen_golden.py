#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF in the build container.

The reference modules (robomimic/models/vq_vae/backbone_lfqvae_v5.py and backbone.py) import
only torch, so they load from /root/reference by file path.  Nothing of the reference is
copied: a fixture holds arrays only (seeds/shapes, reference outputs, gradients, post-AdamW
parameters).  Parameters are NOT stored when they can be re-drawn from the seed with
oracle.lipvq_oracle.make_params (numpy PCG64 + the canonical C oracle, identical on every
machine); their sha256 is stored so a drifted generator is detected, not silently accepted.

Also asserts, while generating, that oracle.lipvq_oracle.torch_* (the torch-CPU restatement
bench.py times) is bit-identical to the reference module.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import lipvq_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def load_ref(ref_root: Path, rel: str, name: str):
    spec = importlib.util.spec_from_file_location(name, ref_root / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def top2(dist: torch.Tensor):
    v, _ = torch.topk(dist, 2, dim=-1, largest=False)
    return v[:, 0].numpy().copy(), v[:, 1].numpy().copy()


def ref_llfq_distances(model, z_e):
    """Distances exactly as LFQQuantizer.forward forms them (v5:39-45), row-chunked."""
    cb = model.quantizer.codebook
    out = []
    for s in range(0, z_e.shape[0], 64):
        z = z_e[s:s + 64]
        m = torch.clamp((2 * torch.sign(z) + 1).unsqueeze(1), max=1)
        out.append(torch.norm(m * (z.unsqueeze(1) - cb.unsqueeze(0)), dim=-1))
    return torch.cat(out)


def meta_of(**kw):
    kw = dict(kw)
    kw["torch_version"] = torch.__version__
    kw["cpu_capability"] = torch.backends.cpu.get_cpu_capability()
    return np.array(repr(sorted(kw.items())))


def run_llfq(ref, name, seed, N, A, D, K, regime="trained", full=False, clamp=False, oracle=None,
             chunk=None):
    torch.manual_seed(0)
    p = O.make_params(seed, A, D, K, regime=regime, variant="llfq", oracle=oracle)
    x_np = O.make_inputs(seed, N, A, clamp=clamp)
    model = ref.LLFQVAE_V4(A, D, num_codes=K)
    model.load_state_dict(O.to_torch(p), strict=True)
    model = model.float()
    x = torch.from_numpy(x_np.copy())
    out = dict(meta=meta_of(name=name, seed=seed, N=N, A=A, D=D, K=K, regime=regime, clamp=clamp,
                            variant="llfq", full=full),
               seed=seed, N=N, A=A, D=D, K=K, params_sha256=np.array(O.params_digest(p)))
    tp = O.to_torch(p)
    if chunk is None:
        if full:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)  # icl.py:887-889
            opt.zero_grad()
        z_latent, loss = model(x)
        # the torch restatement must be the reference, bit for bit
        zl2, loss2, ex = O.torch_llfq_forward(tp, x)
        assert torch.equal(z_latent, zl2) and torch.equal(loss, loss2), name
        with torch.no_grad():
            h = model.encoder(x)
            z_e = model.to_latent(h)
            z_q, idx = model.quantizer(z_e)
            x_rec = model.to_output(model.decoder(z_q))
            dist = ref_llfq_distances(model, z_e)
            d1, d2 = top2(dist)
            assert torch.equal(z_q, z_latent)
            assert torch.equal(ex["indices"], idx) and torch.equal(ex["z_e"], z_e)
            recon = torch.nn.functional.mse_loss(x_rec, x)
            commit = torch.nn.functional.mse_loss(z_q, z_e)
        out.update(z_e=z_e.numpy(), indices=idx.numpy().astype(np.int32), x_recon=x_rec.numpy(),
                   d_best=d1, d_second=d2, recon_loss=np.float32(recon.item()),
                   commitment_loss=np.float32(commit.item()), loss=np.float32(loss.item()),
                   z_latent_sum=np.float64(z_latent.double().sum().item()))
        if full:
            loss.backward()
            for k, v in model.named_parameters():
                out["grad/" + k] = v.grad.numpy().copy()
            opt.step()
            for k, v in model.state_dict().items():
                out["post/" + k] = v.numpy().copy()
    else:
        # big shapes: tokenise in row chunks like bench.py's CPU leg does
        idx, z_lat = O.torch_llfq_tokenize(tp, x, chunk=chunk)
        with torch.no_grad():
            z_e = torch.cat([model.to_latent(model.encoder(x[s:s + chunk]))
                             for s in range(0, N, chunk)])
            idx_ref = torch.cat([model.quantizer(z_e[s:s + chunk])[1] for s in range(0, N, chunk)])
            d1s, d2s = [], []
            for s in range(0, N, chunk):
                a, b = top2(ref_llfq_distances(model, z_e[s:s + chunk]))
                d1s.append(a), d2s.append(b)
        assert torch.equal(idx, idx_ref), name
        out.update(z_e=z_e.numpy(), indices=idx.numpy().astype(np.int32),
                   d_best=np.concatenate(d1s), d_second=np.concatenate(d2s),
                   z_latent_sum=np.float64(z_lat.double().sum().item()))
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    used = len(np.unique(out["indices"]))
    print(f"{name}: N={N} A={A} D={D} K={K} codes used={used}  min rel top-2 gap="
          f"{np.min((out['d_second'] - out['d_best']) / np.maximum(out['d_second'], 1e-30)):.3e}")


def run_llfq_big(ref, name, seed, N, A, D, K, chunk, oracle=None):
    """Full-size pin of the quantizer decision: the REFERENCE MODULE's own sub-modules (encoder, to_latent, quantizer) run
    row-chunked over N rows (the [N, K, D] temporary of v5:41-45 does not fit otherwise); the fixture keeps only what a
    parity test needs -- indices (uint16), the reference's best and second-best distance per row (fp32) -- so 65 536 rows
    stay well under 1 MB.  Parameters and inputs are re-drawn from the seed by the tests (sha256 stored)."""
    p = O.make_params(seed, A, D, K, regime="trained", variant="llfq", oracle=oracle)
    x = torch.from_numpy(O.make_inputs(seed, N, A).copy())
    model = ref.LLFQVAE_V4(A, D, num_codes=K)
    model.load_state_dict(O.to_torch(p), strict=True)
    model = model.float()
    idxs, d1s, d2s = [], [], []
    with torch.no_grad():
        for s in range(0, N, chunk):
            z_e = model.to_latent(model.encoder(x[s:s + chunk]))
            _, idx = model.quantizer(z_e)
            a, b = top2(ref_llfq_distances(model, z_e))
            idxs.append(idx), d1s.append(a), d2s.append(b)
    idx = torch.cat(idxs).numpy()
    assert idx.max() < 65536
    d1, d2 = np.concatenate(d1s), np.concatenate(d2s)
    # the torch restatement that bench.py times must be the reference on a slice of this size too
    n_chk = min(N, 8 * chunk)
    idx2, _ = O.torch_llfq_tokenize(O.to_torch(p), x[:n_chk], chunk=chunk)
    assert np.array_equal(idx2.numpy(), idx[:n_chk]), name
    np.savez_compressed(GOLD / f"{name}.npz",
                        meta=meta_of(name=name, seed=seed, N=N, A=A, D=D, K=K, regime="trained", variant="llfq-big", chunk=chunk),
                        seed=seed, N=N, A=A, D=D, K=K, params_sha256=np.array(O.params_digest(p)),
                        indices=idx.astype(np.uint16), d_best=d1, d_second=d2)
    rel = (d2 - d1) / np.maximum(d2, 1e-30)
    print(f"{name}: N={N} A={A} D={D} K={K} codes used={len(np.unique(idx))} min rel top-2 gap={rel.min():.3e} "
          f"rows with gap < 1e-6: {int((rel < 1e-6).sum())}")


def run_nearties(ref, name, seed, N, K, D, chunk, case=None):
    """The reference quantizer (LFQQuantizer.forward, v5:37-48) on adversarial near-tie rows (oracle.make_neartie_case, or
    make_neartie3_case: a third code within the one-product screen's margin);
    inputs are re-drawn from the seed by the tests, the fixture keeps the reference's indices and top-2 distances."""
    z, cb = (case or O.make_neartie_case)(seed, N, K, D)
    q = ref.LFQQuantizer(K, D)
    idxs, d1s, d2s = [], [], []
    with torch.no_grad():
        q.codebook.copy_(torch.from_numpy(cb))
        for s in range(0, N, chunk):
            zt = torch.from_numpy(z[s:s + chunk])
            _, idx = q(zt)
            m = torch.clamp((2 * torch.sign(zt) + 1).unsqueeze(1), max=1)
            a, b = top2(torch.norm(m * (zt.unsqueeze(1) - q.codebook.unsqueeze(0)), dim=-1))
            idxs.append(idx), d1s.append(a), d2s.append(b)
    idx = torch.cat(idxs).numpy()
    d1, d2 = np.concatenate(d1s), np.concatenate(d2s)
    np.savez_compressed(GOLD / f"{name}.npz", meta=meta_of(name=name, seed=seed, N=N, K=K, D=D, variant="llfq-nearties"),
                        seed=seed, N=N, K=K, D=D, indices=idx.astype(np.uint16), d_best=d1, d_second=d2)
    print(f"{name}: N={N} K={K} D={D} exact fp32 ties: {int((d1 == d2).sum())}, rel gap < 1e-6: "
          f"{int(((d2 - d1) / np.maximum(d2, 1e-30) < 1e-6).sum())}")


def run_vq_nearties(ref, name, seed, N, K, D, chunk):
    """The plain VQVAE's quantizer (VQVAE.quantize, vq:55-63: `(z_e.unsqueeze(1) - E).pow(2).sum(-1)`, argmin) on the same
    adversarial near-tie rows: pins the `pow(2).sum(-1)` order -- cascade sum, scalar tail first -- at widths that are not
    multiples of 8 (and one that is)."""
    z, cb = O.make_neartie_case(seed, N, K, D)
    m = ref.VQVAE(3, D, num_embeddings=K)
    idxs, d1s, d2s = [], [], []
    with torch.no_grad():
        m.embedding.weight.copy_(torch.from_numpy(cb))
        for s in range(0, N, chunk):
            zt = torch.from_numpy(z[s:s + chunk])
            zq, _ = m.quantize(zt)
            dist = (zt.unsqueeze(1) - m.embedding.weight).pow(2).sum(-1)
            idx = torch.argmin(dist, dim=1)
            # quantize() returns z_e + (z_q - z_e).detach(): its argmin is recovered through the embedding row it picked
            assert torch.equal(zt + (m.embedding(idx) - zt), zq), name
            a, b = top2(dist)
            idxs.append(idx), d1s.append(a), d2s.append(b)
    idx = torch.cat(idxs).numpy()
    d1, d2 = np.concatenate(d1s), np.concatenate(d2s)
    np.savez_compressed(GOLD / f"{name}.npz", meta=meta_of(name=name, seed=seed, N=N, K=K, D=D, variant="vq-nearties"),
                        seed=seed, N=N, K=K, D=D, indices=idx.astype(np.uint16), d_best=d1, d_second=d2)
    print(f"{name}: N={N} K={K} D={D} exact fp32 ties: {int((d1 == d2).sum())}")


def run_odd_width_all(v5, vq):
    """Latent widths that are NOT multiples of 8 (the reference takes latent_dim from the observation encoder's width,
    obs_nets.py:1193,1225-1227, which is arbitrary in low-dim mode): tails of 7, 4, 5, 4 and 3 elements after 0, 2, 4, 12 and 25
    whole 8-vectors -- every branch of torch.norm's remainder handling (oracle/probe_torch_norm.py)."""
    for i, D in enumerate((7, 20, 37, 100, 203)):
        run_nearties(v5, f"llfq_nearties_d{D}_k512", 610 + i, 4096, 512, D, chunk=256)
    for i, D in enumerate((7, 20, 64, 100, 203)):
        run_vq_nearties(vq, f"vq_nearties_d{D}_k512", 620 + i, 2048, 512, D, chunk=256)
    # whole modules at such widths (forward + backward + AdamW by the reference)
    torch.set_num_threads(1)
    orc = O.CanonicalOracle()
    run_llfq(v5, "llfq_odd_d37", 631, 300, 7, 37, 256, full=True, oracle=orc)
    run_llfq(v5, "llfq_odd_d203", 632, 96, 12, 203, 128, oracle=orc)
    run_vq(vq, "vq_odd_d20", 633, 200, 7, 20, 64, oracle=orc)


def run_llfq_train_compact(ref, name, seed, N, A, D, K, steps=3, oracle=None):
    """BASELINE config 5's tokenizer step at the REAL shape (obs_nets.py:2411 A = 12, latent = 208, K = 1024, N = 8 x 10 prompt
    actions): `steps` iterations of the reference's choreography (icl.py:913-914 zero_grad, forward, :968-970 backward + AdamW).
    The codebook alone is 852 KB, so the fixture keeps, per step, the loss and the indices, and after the LAST step every
    parameter except the codebook in full, the codebook rows the batches touched (+ 32 untouched ones) and a float64 checksum of
    the whole codebook; the first step's gradients are kept the same way.  Inputs: make_inputs(seed + step)."""
    p = O.make_params(seed, A, D, K, regime="trained", variant="llfq", oracle=oracle)
    model = ref.LLFQVAE_V4(A, D, num_codes=K)
    model.load_state_dict(O.to_torch(p), strict=True)
    model = model.float()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)      # icl.py:887-889
    out = dict(meta=meta_of(name=name, seed=seed, N=N, A=A, D=D, K=K, regime="trained", variant="llfq-train-compact", steps=steps),
               seed=seed, N=N, A=A, D=D, K=K, steps=steps, params_sha256=np.array(O.params_digest(p)))
    touched = set()
    for st in range(steps):
        x = torch.from_numpy(O.make_inputs(seed + st, N, A).copy())
        opt.zero_grad()
        z_latent, loss = model(x)
        with torch.no_grad():
            _, idx = model.quantizer(model.to_latent(model.encoder(x)))
        loss.backward()
        out[f"loss{st}"] = np.float32(loss.item())
        out[f"indices{st}"] = idx.numpy().astype(np.uint16)
        touched |= set(idx.numpy().tolist())
        if st == 0:
            rows0 = np.array(sorted(set(idx.numpy().tolist())), dtype=np.int32)
            for k, v in model.named_parameters():
                g = v.grad.numpy()
                if k == "quantizer.codebook":
                    out["grad0_rows"] = rows0
                    out["grad0/" + k] = g[rows0].copy()
                    out["grad0_codebook_abs_sum"] = np.float64(np.abs(g.astype(np.float64)).sum())
                else:
                    out["grad0/" + k] = g.copy()
        opt.step()
    rows = np.array(sorted(touched | set(range(K - 32, K))), dtype=np.int32)
    for k, v in model.state_dict().items():
        if k == "quantizer.codebook":
            out["post_rows"] = rows
            out["post/" + k] = v.numpy()[rows].copy()
            out["post_codebook_sum"] = np.float64(v.numpy().astype(np.float64).sum())
        else:
            out["post/" + k] = v.numpy().copy()
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(f"{name}: N={N} A={A} D={D} K={K} steps={steps} codes touched={len(touched)} losses="
          f"{[float(out[f'loss{i}']) for i in range(steps)]}")


def run_big_all(v5, orc):
    run_llfq_train_compact(v5, "llfq_icrt_train_k1024", 701, 80, 12, 208, 1024, steps=3, oracle=orc)
    run_nearties(v5, "llfq_nearties_d128_k8192", 601, 1024, 8192, 128, chunk=32)
    run_nearties(v5, "llfq_nearties_d208_k1024", 602, 2048, 1024, 208, chunk=128)
    run_nearties(v5, "llfq_nearties_d64_k1024", 603, 2048, 1024, 64, chunk=256)
    torch.set_num_threads(1)
    run_llfq_big(v5, "llfq_cfg2_big", 501, 65536, 7, 64, 1024, chunk=256, oracle=orc)        # BASELINE config 2's widths
    run_llfq_big(v5, "llfq_cfg3_big", 502, 4096, 7, 128, 8192, chunk=32, oracle=orc)         # BASELINE config 3's widths
    run_llfq_big(v5, "llfq_icrt_big", 503, 16384, 12, 208, 1024, chunk=128, oracle=orc)      # the reference's own widths (v5:89-92)


def run_round4_all(v5, orc):
    """Round 4: BASELINE config 3's widths pinned to the reference at volume -- 32 768 rows instead of 4 096 --
    and near-tie fixtures built for the one-product screen (three candidates per row)."""
    run_nearties(v5, "llfq_neartri_d128_k8192", 641, 1024, 8192, 128, chunk=32, case=O.make_neartie3_case)
    run_nearties(v5, "llfq_neartri_d208_k1024", 642, 512, 1024, 208, chunk=128, case=O.make_neartie3_case)
    torch.set_num_threads(8)       # (the chunked distance is the same arithmetic at any thread count: reductions run along D only)
    run_llfq_big(v5, "llfq_cfg3_big", 502, 32768, 7, 128, 8192, chunk=32, oracle=orc)


def run_vq(ref, name, seed, N, A, D, K, regime="trained", oracle=None):
    p = O.make_params(seed, A, D, K, regime=regime, variant="vq", oracle=oracle)
    x_np = O.make_inputs(seed, N, A)
    model = ref.VQVAE(A, D, num_embeddings=K)
    model.load_state_dict(O.to_torch(p), strict=True)
    x = torch.from_numpy(x_np.copy())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.zero_grad()
    z_latent, loss = model(x)
    zl2, loss2, ex = O.torch_vq_forward(O.to_torch(p), x)
    assert torch.equal(z_latent, zl2) and torch.equal(loss, loss2), name
    with torch.no_grad():
        z_e = model.encoder(x)
        dist = (z_e.unsqueeze(1) - model.embedding.weight).pow(2).sum(-1)
        idx = torch.argmin(dist, dim=1)
        d1, d2 = top2(dist)
        x_rec = ex["x_recon"]
    out = dict(meta=meta_of(name=name, seed=seed, N=N, A=A, D=D, K=K, regime=regime, variant="vq"),
               seed=seed, N=N, A=A, D=D, K=K, params_sha256=np.array(O.params_digest(p)),
               z_e=z_e.numpy(), indices=idx.numpy().astype(np.int32), z_latent=z_latent.numpy(),
               x_recon=x_rec.detach().numpy(), d_best=d1, d_second=d2,
               recon_loss=np.float32(ex["recon_loss"].item()),
               quantization_loss=np.float32(ex["quantization_loss"].item()),
               loss=np.float32(loss.item()))
    loss.backward()
    for k, v in model.named_parameters():
        out["grad/" + k] = v.grad.numpy().copy()
    opt.step()
    for k, v in model.state_dict().items():
        out["post/" + k] = v.numpy().copy()
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(f"{name}: N={N} A={A} D={D} K={K} codes used={len(np.unique(out['indices']))}")


def run_nearest_edge(ref, name, D=64, K=96, N=160):
    """Hand-built quantizer inputs: exact ties (duplicate codes), rows equal to a code, a pair of
    squared distances that differ but share one fp32 square root, zeros, negative entries."""
    rng = np.random.Generator(np.random.PCG64(4242))
    cb = rng.uniform(0, 1, (K, D)).astype(np.float32)
    cb[40] = cb[7]                      # exact duplicate: lower index must win
    cb[41] = cb[7]
    z = rng.uniform(0, 1, (N, D)).astype(np.float32)
    z[0] = cb[7]                        # distance exactly 0 to codes 7, 40, 41
    z[1] = cb[40] + np.float32(1e-3)
    z[2] = 0.0
    z[3] = -z[3]                        # negative latent: exercises the sign mask (v5:39-40)
    z[4, ::2] = 0.0                     # sign(0) = 0 entries
    # rows 5..: midpoints of code pairs -> near ties in both orders
    for r in range(5, 69):
        a, b = rng.integers(0, K, 2)
        z[r] = (cb[a] * np.float32(0.5) + cb[b] * np.float32(0.5))
    # sqrt-merge: code 51 is strictly closer in SQUARED distance than code 50 (s vs nextafter(s)), but
    # both squares round to one fp32 square root, so torch.norm + argmin pick the LOWER index, 50.
    z[70] = 0.0
    cb[50] = 0.0
    cb[51] = 0.0
    cb[51, 0] = np.float32(0.6)                          # s1 = fl(0.36)
    cb[50, 0] = np.float32(0.6)
    cb[50, 1] = np.float32(1.7263e-4)                    # adds ~1 ulp(s1): s2 = nextafter(s1)
    q = ref.LFQQuantizer(K, D)
    with torch.no_grad():
        q.codebook.copy_(torch.from_numpy(cb))
        zt = torch.from_numpy(z)
        zq, idx = q(zt)
        m = torch.clamp((2 * torch.sign(zt) + 1).unsqueeze(1), max=1)
        dist = torch.norm(m * (zt.unsqueeze(1) - q.codebook.unsqueeze(0)), dim=-1)
    assert idx[0].item() == 7
    sq = ((zt[70].unsqueeze(0) - q.codebook[50:52]) ** 2).sum(-1)
    assert sq[0].item() > sq[1].item() and dist[70, 50].item() == dist[70, 51].item() and idx[70].item() == 50
    np.savez_compressed(GOLD / f"{name}.npz", meta=meta_of(name=name, variant="llfq-quantizer"),
                        z_e=z, codebook=cb, indices=idx.numpy().astype(np.int32), z_q=zq.numpy(),
                        distances=dist.numpy())
    print(f"{name}: N={N} K={K} D={D}")


def run_embed(name, seed, B, T, Din, E, K, mode):
    """Fixture for the step after the tokenizer (obs_nets.py:2525-2543, 2580-2596).  ICLTransformer itself cannot be
    imported here (its module needs torchvision/robosuite/...), so the fixture is produced by the same stock
    torch MODULES its constructor builds (obs_nets.py:2425-2450: nn.Linear, nn.Parameter | nn.Embedding | the
    sinusoidal formula of transformers.py:58-77, nn.LayerNorm, nn.Dropout in eval) run in the order of its
    input_embedding()/forward(); oracle.lipvq_oracle.torch_transformer_embeddings must agree bit for bit."""
    ep = O.make_embed_params(seed, Din, E, T, mode)
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    codebook = rng.uniform(0.0, 1.0, (K, Din)).astype(np.float32)
    idx = rng.integers(0, K, (B, T)).astype(np.int64)
    obs = rng.standard_normal((B, T, Din)).astype(np.float32)
    cobs = rng.standard_normal((B, T, Din)).astype(np.float32)
    nets, params = torch.nn.ModuleDict(), torch.nn.ParameterDict()
    nets["embed_encoder"] = torch.nn.Linear(Din, E)
    if mode == "parameter":
        params["embed_timestep"] = torch.nn.Parameter(torch.zeros(1, T, E))
    elif mode == "embedding":
        nets["embed_timestep"] = torch.nn.Embedding(T, E)
    nets["embed_ln"] = torch.nn.LayerNorm(E)
    nets["embed_drop"] = torch.nn.Dropout(0.1)
    nets.eval()
    tp = O.to_torch(ep)
    with torch.no_grad():
        nets["embed_encoder"].weight.copy_(tp["embed_encoder.weight"]); nets["embed_encoder"].bias.copy_(tp["embed_encoder.bias"])
        nets["embed_ln"].weight.copy_(tp["embed_ln.weight"]); nets["embed_ln"].bias.copy_(tp["embed_ln.bias"])
        if mode == "parameter":
            params["embed_timestep"].copy_(tp["embed_timestep"])
        elif mode == "embedding":
            nets["embed_timestep"].weight.copy_(tp["embed_timestep.weight"])

    def input_embedding(inputs):
        emb = nets["embed_encoder"](inputs)
        ts = torch.arange(0, emb.shape[1], dtype=emb.dtype).unsqueeze(0).repeat(emb.shape[0], 1)
        if mode == "parameter":
            te = params["embed_timestep"]
        elif mode == "embedding":
            te = nets["embed_timestep"](ts.long())
        else:
            te = O.torch_sinusoidal(ts, E)
        return nets["embed_drop"](nets["embed_ln"](emb + te))

    with torch.no_grad():
        actions = torch.from_numpy(codebook[idx])           # z_latent rows of the LipVQ tokenizer = codebook rows
        o, co, ca = input_embedding(torch.from_numpy(obs)), input_embedding(torch.from_numpy(cobs)), input_embedding(actions)
        inter = torch.stack([co, ca], dim=2).view(B, -1, E)
        out = torch.cat([inter, o], dim=1)
        mine = O.torch_transformer_embeddings(tp, torch.from_numpy(obs), torch.from_numpy(cobs), actions)
    assert torch.equal(out, mine), "oracle torch restatement drifted from the stock-module sequence"
    np.savez_compressed(GOLD / f"{name}.npz", meta=meta_of(name=name, seed=seed, B=B, T=T, Din=Din, E=E, K=K, mode=mode),
                        digest=np.array(O.params_digest(ep)), codebook=codebook, indices=idx.astype(np.int32), obs=obs,
                        context_obs=cobs, embeddings=out.numpy())
    print(f"{name}: B={B} T={T} Din={Din} E={E} mode={mode}")


def _zero_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def run_default_branch(name, seed, A, D, N):
    """Fixture of the DEFAULT action branch (obs_nets.py:1244-1260).  ICLObservationGroupEncoder cannot be imported here (its
    module needs torchvision/robosuite/...), so the fixture is produced by the same stock torch MODULES its
    constructor builds, in its order (oracle.build_default_branch_modules restates those sixteen lines), called as the
    reference calls them (obs_nets.py:1343-1344: the 2-D [B*T, A] tensor).  Kept: the eval-mode output `y`; from ONE
    training-mode call with every dropout probability set to 0 (the only way to make a training-mode call reproducible):
    its output `y_train`, the spectral-norm vectors it leaves behind, and a digest (sum, norm, 16 entries) of every parameter
    gradient of L = sum(y_train * r).  (Gradients are taken in training mode on purpose: that is where the reference takes
    them, and torch's EVAL-mode fp32 backward through this stack is off by 3-28 % against float64 on the first encoder
    layer and everything before it, while its training-mode backward agrees to 5e-7.)  Parameters are re-drawn from the
    seed by the tests (sha256 stored)."""
    import hashlib
    p = O.make_default_branch_params(seed, A, D)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    x = rng.uniform(-1, 1, (N, A)).astype(np.float32)
    r = rng.standard_normal((N, D)).astype(np.float32)
    net = O.build_default_branch_modules(A, D)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    net.eval()
    with torch.no_grad():
        y = net(torch.from_numpy(x))
    out = {"seed": seed, "A": A, "D": D, "N": N, "x": x, "r": r, "y": y.numpy()}
    # how far fp32 rounding alone moves this output: the same modules in float64 (the ICRT-width case is sharp: ~2e-4)
    net64 = O.build_default_branch_modules(A, D).double()
    net64.load_state_dict({k: torch.from_numpy(v).double() for k, v in p.items()})
    net64.eval()
    with torch.no_grad():
        y64 = net64(torch.from_numpy(x).double())
    out["fp32_noise"] = np.float64(float((y.double() - y64).abs().max()) / float(y64.abs().max()))
    y2, _, _ = O.torch_default_branch(p, x)
    assert float((y2.detach() - y).abs().max()) <= 2e-6 * float(y.abs().max()), "oracle restatement drifted (eval forward)"
    net2 = O.build_default_branch_modules(A, D)
    net2.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    net2.train()
    _zero_dropout(net2)
    yt = net2(torch.from_numpy(x))
    (yt * torch.from_numpy(r)).sum().backward()
    out["y_train"] = yt.detach().numpy()
    y3, uv, P3 = O.torch_default_branch(p, x, training=True)
    (y3 * torch.from_numpy(r)).sum().backward()
    assert float((y3.detach() - yt.detach()).abs().max()) <= 2e-6 * float(yt.detach().abs().max()), "oracle restatement drifted (train forward)"
    for k, t in net2.named_parameters():
        ga, gb = t.grad.numpy(), P3[k].grad.numpy()
        assert np.abs(ga - gb).max() <= 1e-5 * max(1e-6, np.abs(ga).max()), (k, np.abs(ga - gb).max(), np.abs(ga).max())
        out["gdig/" + k] = O.grad_digest(ga)
    for i in (0, 2, 4):
        out[f"train_u/{i}"] = getattr(net2[i], "weight_u").detach().numpy().copy()
        out[f"train_v/{i}"] = getattr(net2[i], "weight_v").detach().numpy().copy()
        assert np.allclose(out[f"train_u/{i}"], uv[i][0], rtol=0, atol=1e-6) and np.allclose(out[f"train_v/{i}"], uv[i][1], rtol=0, atol=1e-6)
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    out["params_sha256"] = np.array(h.hexdigest())
    out["meta"] = meta_of(kind="default_branch")
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    print(f"{name}: y scale {float(np.abs(out['y']).max()):.3f}, {len(p)} state tensors")


def run_default_all():
    run_default_branch("default_icrt", 801, 12, 208, 80)       # the ICRT step shape (obs_nets.py:2411)
    run_default_branch("default_a7_d64", 802, 7, 64, 203)      # BASELINE's action width, ragged N, head width 8


def run_embed_all():
    run_embed("embed_parameter", 301, 3, 10, 64, 512, 1024, "parameter")      # the reference defaults (icl_config.py:133-160)
    run_embed("embed_embedding", 302, 2, 7, 32, 256, 256, "embedding")
    run_embed("embed_sinusoidal", 303, 2, 5, 208, 384, 128, "sinusoidal")     # E not a multiple of 256, D = 208


def run_bin(binmod, name, seed, N, A, D, nb=20):
    """Fixture of the sibling tokenizer (AdaptiveBinActionEmbedding, bin_action/backbone.py), produced by the
    REFERENCE CLASS ITSELF (the module imports only torch): two forward calls that update the running statistics, a
    third with the statistics frozen on wider-range actions (out-of-range values clamp into the edge bins), and the
    parameter gradients of sum(out * R) of the third call."""
    bp = O.make_bin_params(seed, A, D, nb)
    m = binmod.AdaptiveBinActionEmbedding(A, D, num_bins=nb)
    sd = {k: torch.from_numpy(v.copy()) for k, v in bp.items()}
    sd["running_min"], sd["running_max"] = m.running_min.clone(), m.running_max.clone()
    m.load_state_dict(sd, strict=True)
    tp = O.to_torch(bp)
    xs = [O.make_inputs(seed + 1, N, A), O.make_inputs(seed + 2, N, A, clamp=True), 1.5 * O.make_inputs(seed + 3, N, A)]
    rmin, rmax = torch.full((A,), float("inf")), torch.full((A,), float("-inf"))
    out = {}
    for step, x in enumerate(xs):
        xt = torch.from_numpy(x)
        if step == 2:
            m._update_enabled = False                      # what num_step_stop does (bin:71-74)
        y = m(xt)
        idx = m.discretize(xt)
        mine, midx, rmin, rmax = O.torch_bin_forward(tp, xt, rmin, rmax, update=step < 2)
        assert torch.equal(y, mine) and torch.equal(idx, midx), "torch restatement drifted from the reference"
        assert torch.equal(rmin, m.running_min) and torch.equal(rmax, m.running_max)
        out[f"x{step}"], out[f"out{step}"], out[f"bins{step}"] = x, y.detach().numpy(), idx.numpy().astype(np.int32)
        out[f"rmin{step}"], out[f"rmax{step}"] = m.running_min.numpy().copy(), m.running_max.numpy().copy()
    R = np.random.Generator(np.random.PCG64(seed + 9)).standard_normal(out["out2"].shape).astype(np.float32)
    m.zero_grad()
    (m(torch.from_numpy(xs[2])) * torch.from_numpy(R)).sum().backward()
    grads = {"grad/" + k: v.grad.numpy().copy() for k, v in m.named_parameters()}
    np.savez_compressed(GOLD / f"{name}.npz", meta=meta_of(name=name, seed=seed, N=N, A=A, D=D, nb=nb),
                        digest=np.array(O.params_digest(bp)), R=R, **out, **grads)
    print(f"{name}: N={N} A={A} D={D} nb={nb}")


def run_bin_all(ref_root):
    binmod = load_ref(ref_root, "robomimic/models/bin_action/backbone.py", "_ref_bin")
    run_bin(binmod, "bin_icrt", 401, 80, 12, 208)           # the ICRT step shape (obs_nets.py:2411: A = 12)
    run_bin(binmod, "bin_a7", 402, 333, 7, 64)              # BASELINE's action width, ragged N
    run_bin(binmod, "bin_nb5", 403, 64, 3, 32, nb=5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only-init", action="store_true")
    ap.add_argument("--only-default", action="store_true", help="only the default action branch fixtures")
    ap.add_argument("--only-edge", action="store_true")
    ap.add_argument("--only-embed", action="store_true")
    ap.add_argument("--only-bin", action="store_true")
    ap.add_argument("--only-big", action="store_true")
    ap.add_argument("--only-odd", action="store_true", help="only the near-tie fixtures at latent widths that are not multiples of 8")
    ap.add_argument("--only-round4", action="store_true", help="only llfq_cfg3_big at 32 768 rows and the three-candidate near-tie fixtures")
    args = ap.parse_args()
    if args.only_embed:
        GOLD.mkdir(parents=True, exist_ok=True)
        return run_embed_all()
    ref_root = Path(args.ref)
    if args.only_default:
        GOLD.mkdir(parents=True, exist_ok=True)
        torch.set_num_threads(1)
        return run_default_all()
    if args.only_bin:
        GOLD.mkdir(parents=True, exist_ok=True)
        return run_bin_all(ref_root)
    v5 = load_ref(ref_root, "robomimic/models/vq_vae/backbone_lfqvae_v5.py", "_ref_v5")
    vq = load_ref(ref_root, "robomimic/models/vq_vae/backbone.py", "_ref_vq")
    GOLD.mkdir(parents=True, exist_ok=True)
    if args.only_init:
        return run_init(v5, vq)
    if args.only_edge:
        return run_nearest_edge(v5, "llfq_nearest_edge")
    if args.only_big:
        return run_big_all(v5, O.CanonicalOracle())
    if args.only_odd:
        return run_odd_width_all(v5, vq)
    if args.only_round4:
        return run_round4_all(v5, O.CanonicalOracle())
    torch.set_num_threads(1)       # what the reference's train() sets (scripts/train.py:57)
    orc = O.CanonicalOracle()
    # BASELINE config 1 (CPU plumbing case), full fwd + bwd + AdamW
    run_llfq(v5, "llfq_cfg1_trained", 101, 1024, 7, 32, 256, full=True, oracle=orc)
    # the authors' own smoke shape (v5:89-92): B=80, A=12, D=208, K=128
    run_llfq(v5, "llfq_v5main_trained", 102, 80, 12, 208, 128, full=True, oracle=orc)
    # the real ICRT step shape (obs_nets.py:2411, K default 1024), forward only
    run_llfq(v5, "llfq_real_k1024", 103, 80, 12, 208, 1024, oracle=orc)
    # default (degenerate) init: every row -> one code
    run_llfq(v5, "llfq_default_init", 104, 256, 7, 32, 256, regime="default", oracle=orc)
    # clamped actions in [-1,1] (RoboCasa deltas)
    run_llfq(v5, "llfq_cfg1_clamped", 105, 512, 7, 32, 256, clamp=True, oracle=orc)
    # slices of BASELINE configs 2 and 3
    run_llfq(v5, "llfq_cfg2_slice", 106, 2048, 7, 64, 1024, oracle=orc, chunk=256)
    run_llfq(v5, "llfq_cfg3_slice", 107, 192, 7, 128, 8192, oracle=orc, chunk=32)
    # ragged sizes (N not a multiple of any tile), tiny N
    run_llfq(v5, "llfq_ragged_n77", 108, 77, 7, 64, 1024, oracle=orc)
    run_llfq(v5, "llfq_n1", 109, 1, 7, 32, 256, oracle=orc)
    # plain VQVAE variant (STE)
    run_vq(vq, "vq_small_trained", 201, 512, 7, 32, 128, oracle=orc)
    run_vq(vq, "vq_main_trained", 202, 80, 12, 64, 512, oracle=orc)
    run_vq(vq, "vq_default_init", 203, 128, 7, 32, 128, regime="default", oracle=orc)
    run_nearest_edge(v5, "llfq_nearest_edge")
    run_init(v5, vq)
    run_embed_all()
    run_default_all()
    run_bin_all(ref_root)
    run_big_all(v5, orc)
    run_odd_width_all(v5, vq)


def run_init(v5, vq):
    """Initial parameters the reference constructors draw under a fixed torch seed (drop-in check:
    the replacement modules must consume the RNG identically)."""
    torch.manual_seed(1234)
    m = v5.LLFQVAE_V4(12, 48, num_codes=96)
    out = {"llfq/" + k: v.numpy().copy() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    m = vq.VQVAE(12, 48, num_embeddings=64)
    out.update({"vq/" + k: v.numpy().copy() for k, v in m.state_dict().items()})
    np.savez_compressed(GOLD / "init_seeded.npz", **out)
    print("init_seeded: done")


if __name__ == "__main__":
    main()
