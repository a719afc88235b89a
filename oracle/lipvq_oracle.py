"""CPU oracle for the LipVQ-VAE action-tokenizer path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (lipvq-vae_amd/) never does.

Two restatements live here:

1. ``CanonicalOracle`` -- ctypes bindings of oracle/lipvq_oracle.c, the plain-C restatement in
   *canonical fp32* arithmetic (see that file's header).  The gfx950 kernels are required to
   equal it bit for bit on every forward tensor; it is multi-threaded (OpenMP) so that the
   full BASELINE batch (524 288 rows x 1024 codes) is checked in seconds.

2. ``torch_*`` functions -- the same algorithm spelled with the very torch-CPU ops the
   reference uses (reference: robomimic/models/vq_vae/backbone_lfqvae_v5.py:6-84 and
   robomimic/models/vq_vae/backbone.py:38-76), operating on a plain ``dict`` of tensors keyed
   like the reference's ``state_dict``.  It is bit-identical to the reference module run in
   the same process (tests/golden was produced by the reference itself and
   tests/test_oracle_golden.py holds this restatement to it) and is what bench.py times as
   the "reference CPU path" (`cpu_baseline.kind = "port"`), because the reference's files
   cannot travel to the GPU box.

Parameter regimes: ``make_params`` draws a seeded parameter set.  ``regime="default"`` mimics
the reference constructors' initial distributions; ``regime="trained"`` is the *trained-like*
regime of SURVEY.md section 7/8d (ci = 40, b ~ N(0,1), codebook U(0,1) with the first K/2 rows
replaced by sampled z_e + 0.02 noise), because at default init every row maps to one code and
parity tests would test nothing.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liblipvq_oracle.so"

ACT_NONE, ACT_GELU, ACT_SIGMOID, ACT_RELU = 0, 1, 2, 3
DIST_NORM, DIST_SQSUM = 0, 1

LLFQ_KEYS = (
    "encoder.0.weight", "encoder.0.bias", "encoder.2.weight", "encoder.2.bias",
    "to_latent.W", "to_latent.b", "to_latent.ci", "quantizer.codebook",
    "decoder.0.weight", "decoder.0.bias", "decoder.2.weight", "decoder.2.bias",
    "to_output.weight", "to_output.bias",
)
VQ_KEYS = (
    "encoder.0.weight", "encoder.0.bias", "encoder.2.weight", "encoder.2.bias",
    "encoder.4.weight", "encoder.4.bias",
    "decoder.0.weight", "decoder.0.bias", "decoder.2.weight", "decoder.2.bias",
    "decoder.4.weight", "decoder.4.bias", "embedding.weight",
)


def build(force: bool = False) -> Path:
    """Compile oracle/lipvq_oracle.c (gcc) if the shared object is missing or stale."""
    src = _HERE / "lipvq_oracle.c"
    hdr = _HERE.parent / "lipvq-vae_amd" / "csrc" / "lipvq_math.h"
    stale = (not _LIB_PATH.exists()) or any(
        p.exists() and p.stat().st_mtime > _LIB_PATH.stat().st_mtime for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return _LIB_PATH


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ty=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class CanonicalOracle:
    """ctypes view of oracle/liblipvq_oracle.so (numpy in, numpy out)."""

    def __init__(self):
        self.lib = C.CDLL(str(build()))
        self.lib.lq_ref_abi_version.restype = C.c_int

    # -- primitives -------------------------------------------------------------------------
    def lipschitz_scale(self, W, ci):
        W, ci = _f32(W), _f32(ci)
        D, H = W.shape
        scale = np.empty(D, np.float32)
        Wn = np.empty_like(W)
        self.lib.lq_ref_lipschitz_scale(_p(W), _p(ci), _p(scale), _p(Wn), C.c_int(D), C.c_int(H))
        return scale, Wn

    def mlp3(self, x, W0, b0, W1, b1, W2, b2, acts, save_pre=False):
        x = _f32(x)
        W0, b0, W1, b1, W2, b2 = map(_f32, (W0, b0, W1, b1, W2, b2))
        N, K0 = x.shape
        J0, J1, J2 = W0.shape[0], W1.shape[0], W2.shape[0]
        assert W0.shape[1] == K0 and W1.shape[1] == J0 and W2.shape[1] == J1
        y = np.empty((N, J2), np.float32)
        pre = [np.empty((N, J), np.float32) if save_pre else None for J in (J0, J1, J2)]
        self.lib.lq_ref_mlp3(_p(x), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2), _p(b2), _p(y),
                             _p(pre[0]), _p(pre[1]), _p(pre[2]), C.c_int64(N), C.c_int(K0),
                             C.c_int(J0), C.c_int(J1), C.c_int(J2), C.c_int(acts[0]),
                             C.c_int(acts[1]), C.c_int(acts[2]))
        return (y, pre) if save_pre else y

    def nearest(self, z, codebook, dist=DIST_NORM, want_best=False):
        z, cb = _f32(z), _f32(codebook)
        N, D = z.shape
        K = cb.shape[0]
        assert cb.shape[1] == D
        idx = np.empty(N, np.int64)
        zq = np.empty((N, D), np.float32)
        usage = np.zeros(K, np.int64)
        best = np.empty(N, np.float32) if want_best else None
        self.lib.lq_ref_nearest(_p(z), _p(cb), _p(idx, C.c_int64), _p(zq), _p(usage, C.c_int64),
                                _p(best), C.c_int64(N), C.c_int(K), C.c_int(D), C.c_int(dist))
        return (idx, zq, usage, best) if want_best else (idx, zq, usage)

    def distances(self, z, codebook, dist=DIST_NORM):
        z, cb = _f32(z), _f32(codebook)
        out = np.empty((z.shape[0], cb.shape[0]), np.float32)
        self.lib.lq_ref_distances(_p(z), _p(cb), _p(out), C.c_int64(z.shape[0]),
                                  C.c_int(cb.shape[0]), C.c_int(z.shape[1]), C.c_int(dist))
        return out

    def ste(self, ze, zq):
        ze, zq = _f32(ze), _f32(zq)
        out = np.empty_like(ze)
        self.lib.lq_ref_ste(_p(ze), _p(zq), _p(out), C.c_int64(ze.size))
        return out

    def mse_pair(self, xr, x, zq, ze):
        xr, x, zq, ze = map(_f32, (xr, x, zq, ze))
        out = np.empty(2, np.float32)
        self.lib.lq_ref_mse_pair(_p(xr), _p(x), C.c_int64(x.size), _p(zq), _p(ze),
                                 C.c_int64(ze.size), _p(out))
        return float(out[0]), float(out[1])

    def mlp3_bwd(self, x, W0, W1, W2, pre, gy, acts, want_gx=True):
        x, W0, W1, W2, gy = map(_f32, (x, W0, W1, W2, gy))
        pre = [_f32(p) for p in pre]
        N, K0 = x.shape
        J0, J1, J2 = W0.shape[0], W1.shape[0], W2.shape[0]
        g = dict(W0=np.empty_like(W0), b0=np.empty(J0, np.float32), W1=np.empty_like(W1),
                 b1=np.empty(J1, np.float32), W2=np.empty_like(W2), b2=np.empty(J2, np.float32),
                 x=np.empty_like(x) if want_gx else None)
        self.lib.lq_ref_mlp3_bwd(_p(x), _p(W0), _p(W1), _p(W2), _p(pre[0]), _p(pre[1]), _p(pre[2]),
                                 _p(gy), _p(g["W0"]), _p(g["b0"]), _p(g["W1"]), _p(g["b1"]),
                                 _p(g["W2"]), _p(g["b2"]), _p(g["x"]), C.c_int64(N), C.c_int(K0),
                                 C.c_int(J0), C.c_int(J1), C.c_int(J2), C.c_int(acts[0]),
                                 C.c_int(acts[1]), C.c_int(acts[2]))
        return g

    def lipschitz_bwd(self, W, ci, gWn):
        W, ci, gWn = map(_f32, (W, ci, gWn))
        gW, gci = np.empty_like(W), np.empty_like(ci)
        self.lib.lq_ref_lipschitz_bwd(_p(W), _p(ci), _p(gWn), _p(gW), _p(gci), C.c_int(W.shape[0]),
                                      C.c_int(W.shape[1]))
        return gW, gci

    def scatter_add(self, g, idx, K):
        g = _f32(g)
        idx = np.ascontiguousarray(idx, np.int64)
        out = np.empty((K, g.shape[1]), np.float32)
        self.lib.lq_ref_scatter_add(_p(g), _p(idx, C.c_int64), _p(out), C.c_int64(g.shape[0]),
                                    C.c_int(K), C.c_int(g.shape[1]))
        return out

    def math_probe(self, x, fn):
        x = _f32(x).reshape(-1)
        out = np.empty_like(x)
        self.lib.lq_ref_math_probe(_p(x), _p(out), C.c_int64(x.size), C.c_int(fn))
        return out

    # -- whole-path forwards (canonical arithmetic) ------------------------------------------
    def llfq_encode(self, p, x, save_pre=False):
        """x -> z_e  (v5:71-72)."""
        _, Wn = self.lipschitz_scale(p["to_latent.W"], p["to_latent.ci"])
        return self.mlp3(x, p["encoder.0.weight"], p["encoder.0.bias"], p["encoder.2.weight"],
                         p["encoder.2.bias"], Wn, p["to_latent.b"],
                         (ACT_GELU, ACT_GELU, ACT_SIGMOID), save_pre=save_pre)

    def llfq_forward(self, p, x):
        """Everything LLFQVAE_V4.forward computes (v5:70-84), as a dict of numpy arrays."""
        ze, pre_e = self.llfq_encode(p, x, save_pre=True)
        idx, zq, usage = self.nearest(ze, p["quantizer.codebook"], DIST_NORM)
        xr, pre_d = self.mlp3(zq, p["decoder.0.weight"], p["decoder.0.bias"], p["decoder.2.weight"],
                              p["decoder.2.bias"], p["to_output.weight"], p["to_output.bias"],
                              (ACT_GELU, ACT_GELU, ACT_NONE), save_pre=True)
        recon, commit = self.mse_pair(xr, x, zq, ze)
        loss = np.float32(recon) + np.float32(0.25) * np.float32(commit) + np.float32(0.25) * np.float32(commit)
        return dict(z_e=ze, indices=idx, z_q=zq, z_latent=zq.copy(), x_recon=xr, usage=usage,
                    recon_loss=recon, commitment_loss=commit, codebook_loss=commit,
                    loss=float(loss), pre_enc=pre_e, pre_dec=pre_d)

    def vq_forward(self, p, x, commitment_cost=0.25):
        """Everything VQVAE.forward computes (vq:38-76)."""
        R = (ACT_RELU, ACT_RELU, ACT_RELU)
        ze, pre_e = self.mlp3(x, p["encoder.0.weight"], p["encoder.0.bias"], p["encoder.2.weight"],
                              p["encoder.2.bias"], p["encoder.4.weight"], p["encoder.4.bias"], R,
                              save_pre=True)
        idx, zq, usage = self.nearest(ze, p["embedding.weight"], DIST_SQSUM)
        zst = self.ste(ze, zq)
        xr, pre_d = self.mlp3(zst, p["decoder.0.weight"], p["decoder.0.bias"], p["decoder.2.weight"],
                              p["decoder.2.bias"], p["decoder.4.weight"], p["decoder.4.bias"], R,
                              save_pre=True)
        recon, emb = self.mse_pair(xr, x, zq, ze)
        loss = np.float32(recon) + (np.float32(emb) + np.float32(commitment_cost) * np.float32(emb))
        return dict(z_e=ze, indices=idx, z_q=zq, z_latent=zst, x_recon=xr, usage=usage,
                    recon_loss=recon, embedding_loss=emb, loss=float(loss), pre_enc=pre_e,
                    pre_dec=pre_d)

    # -- whole-path gradients (double accumulation; compared with a tolerance) ----------------
    def llfq_grads(self, p, x, fwd=None):
        """dict of the 14 parameter gradients of loss (v5:83) for upstream gradient 1."""
        f = fwd or self.llfq_forward(p, x)
        x = _f32(x)
        N, A = x.shape
        ze, zq, idx = f["z_e"], f["z_q"], f["indices"]
        D = ze.shape[1]
        g_xrec = (2.0 / (N * A)) * (f["x_recon"].astype(np.float64) - x)
        gd = self.mlp3_bwd(zq, p["decoder.0.weight"], p["decoder.2.weight"], p["to_output.weight"],
                           f["pre_dec"], g_xrec.astype(np.float32), (ACT_GELU, ACT_GELU, ACT_NONE))
        g_zq = gd["x"].astype(np.float64) + (0.5 / (N * D)) * (zq.astype(np.float64) - ze)
        g_cb = self.scatter_add(g_zq.astype(np.float32), idx, p["quantizer.codebook"].shape[0])
        g_ze = ((0.5 / (N * D)) * (ze.astype(np.float64) - zq)).astype(np.float32)
        _, Wn = self.lipschitz_scale(p["to_latent.W"], p["to_latent.ci"])
        ge = self.mlp3_bwd(x, p["encoder.0.weight"], p["encoder.2.weight"], Wn, f["pre_enc"], g_ze,
                           (ACT_GELU, ACT_GELU, ACT_SIGMOID), want_gx=False)
        gW, gci = self.lipschitz_bwd(p["to_latent.W"], p["to_latent.ci"], ge["W2"])
        return {"encoder.0.weight": ge["W0"], "encoder.0.bias": ge["b0"], "encoder.2.weight": ge["W1"],
                "encoder.2.bias": ge["b1"], "to_latent.W": gW, "to_latent.b": ge["b2"], "to_latent.ci": gci,
                "quantizer.codebook": g_cb, "decoder.0.weight": gd["W0"], "decoder.0.bias": gd["b0"],
                "decoder.2.weight": gd["W1"], "decoder.2.bias": gd["b1"], "to_output.weight": gd["W2"],
                "to_output.bias": gd["b2"]}

    # -- the step after the tokenizer: input embedding + interleave (obs_nets.py:2525-2543, 2580-2596) -----
    def linear(self, x, W, b=None):
        x, W = _f32(x), _f32(W)
        N, Kin = x.shape
        E = W.shape[0]
        assert W.shape[1] == Kin
        b = None if b is None else _f32(b)
        y = np.empty((N, E), np.float32)
        self.lib.lq_ref_linear(_p(x), _p(W), _p(b), _p(y), C.c_int64(N), C.c_int(Kin), C.c_int(E))
        return y

    def embed_rows(self, src, idx, pos, ln_w, ln_b, eps, out, T, bstride, tstride, offset, N=None, want_stats=False):
        """Writes into ``out`` (a float32 C-contiguous numpy array, addressed flat); returns stats or None."""
        src, ln_w, ln_b = _f32(src), _f32(ln_w), _f32(ln_b)
        E = src.shape[1]
        idx = None if idx is None else np.ascontiguousarray(idx, np.int64)
        pos = None if pos is None else _f32(pos).reshape(-1, E)
        if N is None:
            N = src.shape[0] if idx is None else idx.size
        assert out.dtype == np.float32 and out.flags.c_contiguous
        stats = np.empty((N, 2), np.float32) if want_stats else None
        self.lib.lq_ref_embed_rows(_p(src), _p(idx, C.c_int64), _p(pos), _p(ln_w), _p(ln_b), C.c_float(eps), _p(out),
                                   _p(stats), C.c_int64(N), C.c_int(T), C.c_int(E), C.c_int64(src.shape[0]),
                                   C.c_int64(bstride), C.c_int64(tstride), C.c_int64(offset))
        return stats

    def transformer_embeddings(self, ep, obs, context_obs, codebook, idx, eps=1e-5):
        """[B][3T][E] input of the transformer backbone: the observation streams through the dense Linear, the
        context actions through the codebook table (idx [B][T] = the tokenizer's indices)."""
        obs, context_obs = _f32(obs), _f32(context_obs)
        B, T, Din = obs.shape
        W, b = ep["embed_encoder.weight"], ep["embed_encoder.bias"]
        E = W.shape[0]
        pos = embed_time_table(ep, T)
        out = np.empty((B, 3 * T, E), np.float32)
        lw, lb = ep["embed_ln.weight"], ep["embed_ln.bias"]
        table = self.linear(codebook, W, b)
        self.embed_rows(self.linear(context_obs.reshape(B * T, Din), W, b), None, pos, lw, lb, eps, out, T, 3 * T * E, 2 * E, 0)
        self.embed_rows(table, np.asarray(idx).reshape(-1), pos, lw, lb, eps, out, T, 3 * T * E, 2 * E, E)
        self.embed_rows(self.linear(obs.reshape(B * T, Din), W, b), None, pos, lw, lb, eps, out, T, 3 * T * E, E, 2 * T * E)
        return out

    # -- AdaptiveBinActionEmbedding (reference robomimic/models/bin_action/backbone.py) ---------------------
    def linear_act(self, x, W, b, act, save_pre=False):
        x, W = _f32(x), _f32(W)
        N, Kin = x.shape
        E = W.shape[0]
        b = None if b is None else _f32(b)
        y = np.empty((N, E), np.float32)
        pre = np.empty((N, E), np.float32) if save_pre else None
        self.lib.lq_ref_linear_act(_p(x), _p(W), _p(b), _p(y), _p(pre), C.c_int64(N), C.c_int(Kin), C.c_int(E), C.c_int(act))
        return (y, pre) if save_pre else y

    def bin_minmax(self, actions, rmin, rmax):
        actions = _f32(actions)
        rmin, rmax = _f32(rmin).copy(), _f32(rmax).copy()
        self.lib.lq_ref_bin_minmax(_p(actions), _p(rmin), _p(rmax), C.c_int64(actions.shape[0]), C.c_int(actions.shape[1]))
        return rmin, rmax

    def bin_discretize(self, actions, rmin, rmax, nb, want_boundaries=False):
        actions, rmin, rmax = _f32(actions), _f32(rmin), _f32(rmax)
        N, A = actions.shape
        bins = np.empty((A, N), np.int64)
        bd = np.empty((A, nb + 1), np.float32) if want_boundaries else None
        self.lib.lq_ref_bin_discretize(_p(actions), _p(rmin), _p(rmax), _p(bins, C.c_int64), _p(bd), C.c_int64(N),
                                       C.c_int(A), C.c_int(nb))
        return (bins, bd) if want_boundaries else bins

    def bin_table(self, bp):
        """P [A][nb][H]: per-dimension product of the embedding table with its block of output_layer.0.weight."""
        W1 = _f32(bp["output_layer.0.weight"])
        A = sum(1 for k in bp if k.startswith("embedding_layers.") and k.endswith(".weight"))
        ed = bp["embedding_layers.0.weight"].shape[1]
        return np.stack([self.linear(bp[f"embedding_layers.{i}.weight"], np.ascontiguousarray(W1[:, ed * i:ed * (i + 1)]))
                         for i in range(A)])

    def bin_hidden(self, bins, P, b1, save_pre=False):
        bins = np.ascontiguousarray(bins, np.int64)
        P, b1 = _f32(P), _f32(b1)
        A, N = bins.shape
        nb, H = P.shape[1], P.shape[2]
        h = np.empty((N, H), np.float32)
        pre = np.empty((N, H), np.float32) if save_pre else None
        self.lib.lq_ref_bin_hidden(_p(bins, C.c_int64), _p(P), _p(b1), _p(h), _p(pre), C.c_int64(N), C.c_int(A),
                                   C.c_int(nb), C.c_int(H))
        return (h, pre) if save_pre else h

    def bin_forward(self, bp, actions, rmin, rmax, update=True):
        """AdaptiveBinActionEmbedding.forward (backbone.py:68-89): returns dict(out, bins [A][N], running_min/max)."""
        nb = bp["embedding_layers.0.weight"].shape[0]
        if update:
            rmin, rmax = self.bin_minmax(actions, rmin, rmax)
        bins = self.bin_discretize(actions, rmin, rmax, nb)
        h = self.bin_hidden(bins, self.bin_table(bp), bp["output_layer.0.bias"])
        out = self.linear_act(h, bp["output_layer.2.weight"], bp["output_layer.2.bias"], ACT_GELU)
        return {"out": out, "bins": bins, "running_min": rmin, "running_max": rmax, "hidden": h}

    def ema_update(self, cluster_size, embed_sum, counts, dw, decay=0.99, eps=1e-5):
        """Opt-in EMA codebook update (extension): returns (cluster_size, embed_sum, codebook)."""
        cs, es = _f32(cluster_size).copy(), _f32(embed_sum).copy()
        counts, dw = np.ascontiguousarray(counts, np.int64), _f32(dw)
        K, D = es.shape
        cb = np.empty_like(es)
        self.lib.lq_ref_ema_update(_p(cs), _p(es), _p(counts, C.c_int64), _p(dw), _p(cb), C.c_float(decay), C.c_float(eps),
                                   C.c_int(K), C.c_int(D))
        return cs, es, cb

    def vq_grads(self, p, x, commitment_cost=0.25, fwd=None):
        f = fwd or self.vq_forward(p, x, commitment_cost)
        x = _f32(x)
        N, A = x.shape
        ze, zq, idx = f["z_e"], f["z_q"], f["indices"]
        D = ze.shape[1]
        R = (ACT_RELU, ACT_RELU, ACT_RELU)
        g_xrec = ((2.0 / (N * A)) * (f["x_recon"].astype(np.float64) - x)).astype(np.float32)
        gd = self.mlp3_bwd(f["z_latent"], p["decoder.0.weight"], p["decoder.2.weight"], p["decoder.4.weight"],
                           f["pre_dec"], g_xrec, R)
        g_emb = self.scatter_add(((2.0 / (N * D)) * (zq.astype(np.float64) - ze)).astype(np.float32), idx,
                                 p["embedding.weight"].shape[0])
        g_ze = (gd["x"].astype(np.float64) + (commitment_cost * 2.0 / (N * D)) * (ze.astype(np.float64) - zq))
        ge = self.mlp3_bwd(x, p["encoder.0.weight"], p["encoder.2.weight"], p["encoder.4.weight"], f["pre_enc"],
                           g_ze.astype(np.float32), R, want_gx=False)
        out = {"embedding.weight": g_emb}
        for i, k in ((0, "0"), (1, "2"), (2, "4")):
            out[f"encoder.{k}.weight"], out[f"encoder.{k}.bias"] = ge[f"W{i}"], ge[f"b{i}"]
            out[f"decoder.{k}.weight"], out[f"decoder.{k}.bias"] = gd[f"W{i}"], gd[f"b{i}"]
        return out


# ---------------------------------------------------------------------------------------------
# torch-CPU restatement (same op sequence as the reference; bit-identical to it in-process)
# ---------------------------------------------------------------------------------------------

def torch_llfq_encode(p, x):
    """x[N,A] -> z_e[N,D]   (v5:54-59 encoder, v5:6-12 normalization, v5:22-24 Lipschitz layer)."""
    import torch
    import torch.nn.functional as F
    h = F.gelu(F.linear(x, p["encoder.0.weight"], p["encoder.0.bias"]))
    h = F.gelu(F.linear(h, p["encoder.2.weight"], p["encoder.2.bias"]))
    W, b, ci = p["to_latent.W"], p["to_latent.b"], p["to_latent.ci"]
    rowsum = torch.sum(torch.abs(W), dim=1, keepdim=True)
    scale = torch.minimum(torch.tensor(1.0), F.softplus(ci).unsqueeze(1) / rowsum)
    return torch.sigmoid(torch.matmul(h, (W * scale).T) + b)


def torch_llfq_quantize(codebook, z_e):
    """z_e[N,D] -> (z_q[N,D], idx[N])   (v5:37-48).  Materialises the [N,K,D] difference tensor
    exactly as the reference does, so the caller must chunk N (see torch_llfq_tokenize)."""
    import torch
    mask = torch.clamp((2 * torch.sign(z_e) + 1).unsqueeze(1), max=1)
    diff = mask * (z_e.unsqueeze(1) - codebook.unsqueeze(0))
    idx = torch.argmin(torch.norm(diff, dim=-1), dim=-1)
    return codebook[idx], idx


def torch_llfq_forward(p, x):
    """(z_latent, loss, extras)   (v5:70-84)."""
    import torch.nn.functional as F
    z_e = torch_llfq_encode(p, x)
    z_q, idx = torch_llfq_quantize(p["quantizer.codebook"], z_e)
    z_latent = z_q.clone().detach()
    h = F.gelu(F.linear(z_q, p["decoder.0.weight"], p["decoder.0.bias"]))
    h = F.gelu(F.linear(h, p["decoder.2.weight"], p["decoder.2.bias"]))
    x_rec = F.linear(h, p["to_output.weight"], p["to_output.bias"])
    recon = F.mse_loss(x_rec, x)
    commit = F.mse_loss(z_q.detach(), z_e)
    cbl = F.mse_loss(z_q, z_e.detach())
    loss = recon + 0.25 * commit + 0.25 * cbl
    return z_latent, loss, dict(z_e=z_e, indices=idx, x_recon=x_rec, recon_loss=recon,
                                commitment_loss=commit, codebook_loss=cbl)


def torch_llfq_tokenize(p, x, chunk=256):
    """encode + quantize over row chunks (the metric's path): returns (indices, z_latent).
    chunk=256 keeps the reference's [chunk,K,D] temporary at 67 MB for K=1024, D=64."""
    import torch
    out_i, out_z = [], []
    with torch.no_grad():
        for s in range(0, x.shape[0], chunk):
            z_e = torch_llfq_encode(p, x[s:s + chunk])
            z_q, idx = torch_llfq_quantize(p["quantizer.codebook"], z_e)
            out_i.append(idx)
            out_z.append(z_q)
    return torch.cat(out_i), torch.cat(out_z)


def torch_vq_forward(p, x, commitment_cost=0.25):
    """(z_latent, loss, extras) of the plain VQVAE variant   (vq:38-76)."""
    import torch
    import torch.nn.functional as F
    h = x
    for i in (0, 2, 4):
        h = F.relu(F.linear(h, p[f"encoder.{i}.weight"], p[f"encoder.{i}.bias"]))
    z_e = h
    E = p["embedding.weight"]
    dist = (z_e.unsqueeze(1) - E).pow(2).sum(-1)
    idx = torch.argmin(dist, dim=1)
    z_q = F.embedding(idx, E)
    q_loss = F.mse_loss(z_q, z_e.detach()) + commitment_cost * F.mse_loss(z_q.detach(), z_e)
    z_st = z_e + (z_q - z_e).detach()
    z_latent = z_st.clone().detach()
    h = z_st
    for i in (0, 2, 4):
        h = F.relu(F.linear(h, p[f"decoder.{i}.weight"], p[f"decoder.{i}.bias"]))
    recon = F.mse_loss(h, x)
    return z_latent, recon + q_loss, dict(z_e=z_e, indices=idx, x_recon=h, recon_loss=recon,
                                          quantization_loss=q_loss)


# --- the step after the tokenizer (reference robomimic/models/obs_nets.py = "ob") ---------------------
# The reference class (ICLTransformer, ob:2330-2640) cannot be imported in the build container (its module pulls
# torchvision / robosuite / ...), so these functions restate ob:2485-2543 and ob:2580-2596 with the same stock torch
# ops on a dict keyed like its state_dict ("nets.embed_encoder.weight" -> "embed_encoder.weight", ...).

EMBED_MODES = ("parameter", "embedding", "sinusoidal")     # ob:2431-2445: nn.Parameter | nn.Embedding | sinusoidal


def embed_time_table(ep, T):
    """[T][E] float32 rows the reference adds to the embeddings of timesteps 0..T-1 (ob:2485-2523)."""
    if "embed_timestep" in ep:                              # nn.Parameter [1][max_timestep][E], ob:2437-2439
        tab = np.asarray(ep["embed_timestep"], np.float32)[0]
        assert tab.shape[0] == T, "nn.Parameter time embeddings broadcast only when T == context_length"
        return tab
    if "embed_timestep.weight" in ep:                       # nn.Embedding(max_timestep, E), ob:2441-2443
        return np.asarray(ep["embed_timestep.weight"], np.float32)[:T]
    import torch
    E = ep["embed_encoder.weight"].shape[0]
    return torch_sinusoidal(torch.arange(T, dtype=torch.float32)[None], E)[0].numpy()


def torch_sinusoidal(timesteps, E):
    """PositionalEncoding.forward (reference robomimic/models/transformers.py:58-77) on float timesteps [B][T]."""
    import math
    import torch
    div = torch.exp(torch.arange(0, E, 2) * (-math.log(10000.0) / E))[None, None].repeat(timesteps.shape[0], timesteps.shape[1], 1)
    pe = torch.zeros((timesteps.shape[0], timesteps.shape[1], E))
    pe[:, :, 0::2] = torch.sin(timesteps.unsqueeze(-1) * div)
    pe[:, :, 1::2] = torch.cos(timesteps.unsqueeze(-1) * div)
    return pe.detach()


def torch_input_embedding(ep, inputs, eps=1e-5):
    """ob:2525-2543 in eval mode (embed_drop = identity): LayerNorm(Linear(inputs) + time_embeddings)."""
    import torch
    import torch.nn.functional as F
    emb = F.linear(inputs, ep["embed_encoder.weight"], ep["embed_encoder.bias"])          # ob:2536
    B, T, E = emb.shape
    timesteps = torch.arange(0, T, dtype=emb.dtype).unsqueeze(0).repeat(B, 1)             # ob:2493-2502
    if "embed_timestep" in ep:
        time_emb = ep["embed_timestep"]                                                     # ob:2509-2510
    elif "embed_timestep.weight" in ep:
        time_emb = F.embedding(timesteps.long(), ep["embed_timestep.weight"])              # ob:2507,2512
    else:
        time_emb = torch_sinusoidal(timesteps, E)
    emb = emb + time_emb                                                                    # ob:2538
    return F.layer_norm(emb, (E,), ep["embed_ln.weight"], ep["embed_ln.bias"], eps)         # ob:2539


def torch_transformer_embeddings(ep, obs, context_obs, context_actions):
    """ob:2580-2596: embed the three streams, interleave the context pairs, append the observations."""
    import torch
    o = torch_input_embedding(ep, obs)
    co = torch_input_embedding(ep, context_obs)
    ca = torch_input_embedding(ep, context_actions)
    bs, _, D = o.shape
    inter = torch.stack([co, ca], dim=2).view(bs, -1, D)
    return torch.cat([inter, o], dim=1)


def make_embed_params(seed, Din, E, T, mode="parameter"):
    """Seeded parameters of the embedding stage keyed like the reference state_dict (minus the nets./params. prefix).
    The reference initialises the nn.Parameter time embedding and LayerNorm trivially (zeros / ones); a trained-like
    draw is used instead so that every term of the formula is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    W, b = _linear_init(rng, E, Din)
    ep = {"embed_encoder.weight": W, "embed_encoder.bias": b,
          "embed_ln.weight": (1.0 + 0.1 * rng.standard_normal(E)).astype(np.float32),
          "embed_ln.bias": (0.1 * rng.standard_normal(E)).astype(np.float32)}
    if mode == "parameter":
        ep["embed_timestep"] = (0.02 * rng.standard_normal((1, T, E))).astype(np.float32)
    elif mode == "embedding":
        ep["embed_timestep.weight"] = rng.standard_normal((T, E)).astype(np.float32)
    elif mode != "sinusoidal":
        raise ValueError(mode)
    return ep


# --- AdaptiveBinActionEmbedding (reference robomimic/models/bin_action/backbone.py = "bin") -----------------

def torch_bin_forward(bp, actions, rmin, rmax, update=True):
    """bin:68-89 with the stock torch ops of the reference on a dict keyed like its state_dict; returns
    (out [N, D], bin_indices [N, A] int64, running_min, running_max)."""
    import torch
    import torch.nn.functional as F
    A = actions.shape[1]
    nb = bp["embedding_layers.0.weight"].shape[0]
    if update:                                                                        # bin:37-40
        rmin = torch.minimum(rmin, actions.min(dim=0)[0])
        rmax = torch.maximum(rmax, actions.max(dim=0)[0])
    idx = []
    for i in range(A):                                                                # bin:42-66
        boundaries = torch.linspace(rmin[i], rmax[i], nb + 1)
        idx.append(torch.clamp(torch.bucketize(actions[:, i], boundaries) - 1, 0, nb - 1))
    idx = torch.stack(idx, dim=1)
    emb = torch.cat([F.embedding(idx[:, i], bp[f"embedding_layers.{i}.weight"]) for i in range(A)], dim=-1)   # bin:77-83
    h = F.gelu(F.linear(emb, bp["output_layer.0.weight"], bp["output_layer.0.bias"]))                          # bin:26-31
    out = F.gelu(F.linear(h, bp["output_layer.2.weight"], bp["output_layer.2.bias"]))
    return out, idx, rmin, rmax


def make_bin_params(seed, A, D, nb=20, ed=64):
    """Seeded parameters keyed like AdaptiveBinActionEmbedding.state_dict() (minus the two buffers): nn.Embedding
    N(0,1), nn.Linear defaults."""
    rng = np.random.Generator(np.random.PCG64(seed + 32452843))
    bp = {f"embedding_layers.{i}.weight": rng.standard_normal((nb, ed)).astype(np.float32) for i in range(A)}
    bp["output_layer.0.weight"], bp["output_layer.0.bias"] = _linear_init(rng, ed * A // 2, ed * A)
    bp["output_layer.2.weight"], bp["output_layer.2.bias"] = _linear_init(rng, D, ed * A // 2)
    return bp


# ---------------------------------------------------------------------------------------------
# seeded synthetic parameters / inputs (numpy PCG64: identical on every machine)
# ---------------------------------------------------------------------------------------------

def _linear_init(rng, out_f, in_f):
    # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias
    bound = 1.0 / np.sqrt(in_f)
    return (rng.uniform(-bound, bound, (out_f, in_f)).astype(np.float32),
            rng.uniform(-bound, bound, (out_f,)).astype(np.float32))


def make_inputs(seed, N, A, clamp=False):
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    x = rng.standard_normal((N, A)).astype(np.float32)
    return np.clip(x, -1.0, 1.0) if clamp else x


def make_params(seed, A, D, K, hidden=128, regime="trained", variant="llfq", oracle=None):
    """Seeded parameter dict (numpy float32) keyed like the reference state_dict."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = {}
    if variant == "llfq":
        p["encoder.0.weight"], p["encoder.0.bias"] = _linear_init(rng, 64, A)
        p["encoder.2.weight"], p["encoder.2.bias"] = _linear_init(rng, hidden, 64)
        p["to_latent.W"] = rng.standard_normal((D, hidden)).astype(np.float32)      # v5:18
        p["to_latent.b"] = np.zeros(D, np.float32)                                   # v5:19
        p["to_latent.ci"] = np.ones(D, np.float32)                                   # v5:20
        bound = np.sqrt(6.0 / D)                                                     # kaiming_uniform_, v5:35
        p["quantizer.codebook"] = rng.uniform(-bound, bound, (K, D)).astype(np.float32)
        p["decoder.0.weight"], p["decoder.0.bias"] = _linear_init(rng, 64, D)
        p["decoder.2.weight"], p["decoder.2.bias"] = _linear_init(rng, hidden, 64)
        p["to_output.weight"], p["to_output.bias"] = _linear_init(rng, A, hidden)
        if regime == "trained":
            oracle = oracle or CanonicalOracle()
            p["to_latent.ci"] = np.full(D, 40.0, np.float32)
            p["to_latent.b"] = rng.standard_normal(D).astype(np.float32)
            cb = rng.uniform(0.0, 1.0, (K, D)).astype(np.float32)
            M = max(4 * K, 1024)
            xs = rng.standard_normal((M, A)).astype(np.float32)
            ze = oracle.llfq_encode(p, xs)
            pick = rng.permutation(M)[: K // 2]
            noise = (0.02 * rng.standard_normal((K // 2, D))).astype(np.float32)
            cb[: K // 2] = ze[pick] + noise
            p["quantizer.codebook"] = cb
    elif variant == "vq":
        p["encoder.0.weight"], p["encoder.0.bias"] = _linear_init(rng, 64, A)
        p["encoder.2.weight"], p["encoder.2.bias"] = _linear_init(rng, 128, 64)
        p["encoder.4.weight"], p["encoder.4.bias"] = _linear_init(rng, D, 128)
        p["decoder.0.weight"], p["decoder.0.bias"] = _linear_init(rng, 128, D)
        p["decoder.2.weight"], p["decoder.2.bias"] = _linear_init(rng, 64, 128)
        p["decoder.4.weight"], p["decoder.4.bias"] = _linear_init(rng, A, 64)
        p["embedding.weight"] = rng.uniform(-1.0 / K, 1.0 / K, (K, D)).astype(np.float32)  # vq:36
        if regime == "trained":
            oracle = oracle or CanonicalOracle()
            p["encoder.4.bias"] = (0.3 + 0.2 * rng.standard_normal(D)).astype(np.float32)
            M = max(4 * K, 1024)
            xs = rng.standard_normal((M, A)).astype(np.float32)
            R = (ACT_RELU, ACT_RELU, ACT_RELU)
            ze = oracle.mlp3(xs, p["encoder.0.weight"], p["encoder.0.bias"], p["encoder.2.weight"],
                             p["encoder.2.bias"], p["encoder.4.weight"], p["encoder.4.bias"], R)
            pick = rng.permutation(M)[: K // 2]
            cb = rng.uniform(0.0, 1.0, (K, D)).astype(np.float32)
            cb[: K // 2] = ze[pick] + (0.02 * rng.standard_normal((K // 2, D))).astype(np.float32)
            p["embedding.weight"] = cb
    else:
        raise ValueError(variant)
    return p


def make_neartie_case(seed, N, K, D):
    """Adversarial quantizer inputs: every row sits on (or within +-4e-8 of) the bisector of two random codes, with the
    codebook in the trained regime's range -- rows whose two best codes are (nearly) equidistant, which only the exact
    arithmetic of the reference decides (tests/golden/llfq_nearties_*.npz hold the reference's answers).
    numpy PCG64 only: identical on every machine."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cb = rng.uniform(0.0, 1.0, (K, D)).astype(np.float32)
    a = rng.integers(0, K, N)
    b = (a + 1 + rng.integers(0, K - 1, N)) % K
    t = (np.float32(0.5) + ((np.arange(N) % 9) - 4).astype(np.float32) * np.float32(1e-8)).astype(np.float32)[:, None]
    z = (cb[a] * t + cb[b] * (np.float32(1.0) - t)).astype(np.float32)
    return z, cb


def make_neartie3_case(seed, N, K, D):
    """Near-ties with a THIRD code close by (round 4, for the one-product screen): row i sits on the bisector of codes a_i, b_i as in
    make_neartie_case, and code c_i = K - 1 - i is moved onto the sphere around the row through a_i, off by a relative
    1e-5 ... 2e-3 in radius -- inside the one-product screen's margin (about 2^-9 of the cross term), mostly outside the
    three-product screen's.  So the coarse screen must list THREE candidates (its 2nd and 3rd both within its bound) and the
    exact stage must pick the reference's.  N <= K / 2; numpy PCG64 only."""
    assert 2 * N <= K
    rng = np.random.Generator(np.random.PCG64(seed))
    cb = rng.uniform(0.0, 1.0, (K, D)).astype(np.float32)
    a = rng.integers(0, K - N, N)
    b = (a + 1 + rng.integers(0, K - N - 1, N)) % (K - N)
    t = (np.float32(0.5) + ((np.arange(N) % 9) - 4).astype(np.float32) * np.float32(1e-8)).astype(np.float32)[:, None]
    z = (cb[a] * t + cb[b] * (np.float32(1.0) - t)).astype(np.float32)
    u = rng.standard_normal((N, D))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = np.linalg.norm(z.astype(np.float64) - cb[a].astype(np.float64), axis=1, keepdims=True)
    delta = (np.array([1e-5, -1e-5, 1e-4, -1e-4, 5e-4, -5e-4, 2e-3, -2e-3])[np.arange(N) % 8])[:, None]
    cb[K - 1 - np.arange(N)] = (z.astype(np.float64) + r * (1.0 + delta) * u).astype(np.float32)
    return z, cb


def params_digest(p):
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(k.encode())
        h.update(np.ascontiguousarray(p[k]).tobytes())
    return h.hexdigest()


def to_torch(p):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v).copy()) for k, v in p.items()}


# ---------------------------------------------------------------------------------------------------
# The DEFAULT action branch (reference robomimic/models/obs_nets.py:1244-1260): spectral-norm MLP + 4 post-norm
# nn.TransformerEncoder layers over the unbatched [B*T, D] sequence + Linear.  Test infrastructure only.
#   make_default_branch_params   seeded parameters (the stock constructors' draws, perturbed so that the four cloned
#                                layers differ and biases / LayerNorm affine terms are not the trivial 0 / 1)
#   torch_default_branch         restatement in explicit torch-CPU ops -- power iteration as torch/nn/utils/
#                                spectral_norm.py writes it, F.linear, per-head softmax attention, F.layer_norm -- with NO
#                                nn.TransformerEncoder / spectral_norm hook involved (those produce the fixtures:
#                                oracle/gen_golden.py::run_default_branch)
# ---------------------------------------------------------------------------------------------------
DEFAULT_NHEAD, DEFAULT_FF, DEFAULT_LAYERS = 8, 256, 4


def build_default_branch_modules(A, D):
    """The reference's constructor text (obs_nets.py:1245-1260) with stock torch modules."""
    import warnings
    import torch  # noqa: F401
    import torch.nn as nn
    from torch.nn.utils import spectral_norm
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=DEFAULT_NHEAD, dim_feedforward=DEFAULT_FF, activation="gelu")
        return nn.Sequential(
            spectral_norm(nn.Linear(A, 64)), nn.GELU(),
            spectral_norm(nn.Linear(64, 128)), nn.GELU(),
            spectral_norm(nn.Linear(128, D)),
            nn.TransformerEncoder(layer, num_layers=DEFAULT_LAYERS),
            nn.Linear(D, D))


def make_default_branch_params(seed, A, D):
    """state_dict (numpy) of the default branch: constructor draws under torch.manual_seed(seed), then every float tensor
    except the spectral-norm vectors is perturbed by 0.05 * N(0, 1) from a seeded numpy generator."""
    import torch
    torch.manual_seed(seed)
    net = build_default_branch_modules(A, D)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, v in net.state_dict().items():
        a = v.detach().numpy().astype(np.float32).copy()
        if not (k.endswith("weight_u") or k.endswith("weight_v")):
            a = (a + 0.05 * rng.standard_normal(a.shape)).astype(np.float32)
        out[k] = a
    return out


def torch_default_branch(params, x, training=False, eps_sn=1e-12):
    """(y [N, D] tensor with grad_fn, dict of updated (u, v) per spectral layer, leaf parameter tensors).
    params: state_dict as numpy; x: numpy [N, A].  Dropout is NOT applied (eval semantics for every dropout; `training`
    only switches the spectral-norm power iteration, as module.training does in torch's hook)."""
    import torch
    import torch.nn.functional as F
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(not (k.endswith("_u") or k.endswith("_v")))
         for k, v in params.items()}
    uv = {}

    def sn(i):
        W, u, v = P[f"{i}.weight_orig"], P[f"{i}.weight_u"].detach().clone(), P[f"{i}.weight_v"].detach().clone()
        if training:                               # spectral_norm.py compute_weight, n_power_iterations = 1
            with torch.no_grad():
                v = F.normalize(torch.mv(W.t(), u), dim=0, eps=eps_sn)
                u = F.normalize(torch.mv(W, v), dim=0, eps=eps_sn)
        uv[i] = (u.numpy().copy(), v.numpy().copy())
        sigma = torch.dot(u, torch.mv(W, v))
        return W / sigma

    h = torch.from_numpy(np.ascontiguousarray(x))
    h = F.gelu(F.linear(h, sn(0), P["0.bias"]))
    h = F.gelu(F.linear(h, sn(2), P["2.bias"]))
    h = F.linear(h, sn(4), P["4.bias"])
    S, D = h.shape
    H, dh = DEFAULT_NHEAD, D // DEFAULT_NHEAD
    for li in range(DEFAULT_LAYERS):
        pre = f"5.layers.{li}."
        qkv = F.linear(h, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"])
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        heads = []
        for hh in range(H):
            sl = slice(hh * dh, (hh + 1) * dh)
            sc = (q[:, sl] / float(np.sqrt(dh))) @ k[:, sl].t()
            heads.append(torch.softmax(sc, dim=-1) @ v[:, sl])
        a = F.linear(torch.cat(heads, dim=1), P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + a, (D,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-5)
        f = F.linear(F.gelu(F.linear(h, P[pre + "linear1.weight"], P[pre + "linear1.bias"])), P[pre + "linear2.weight"],
                     P[pre + "linear2.bias"])
        h = F.layer_norm(h + f, (D,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-5)
    y = F.linear(h, P["6.weight"], P["6.bias"])
    return y, uv, P


def grad_digest(g):
    """What a fixture keeps of a (possibly large) gradient: sum, L2 norm and 16 entries at fixed strided positions."""
    a = np.asarray(g, dtype=np.float64).ravel()
    pos = (np.arange(16) * max(1, a.size // 16)) % a.size
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[pos]]).astype(np.float64)
