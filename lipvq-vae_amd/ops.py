"""Tensor-level wrappers over the C ABI (include/lipvq.h): validation, output allocation and
stream plumbing only -- all arithmetic happens in the HIP library.

Every function takes CUDA(HIP) fp32 contiguous tensors on one device and enqueues work on
``torch.cuda.current_stream()``.  A CPU tensor is an error (there is no CPU path in the
product; the CPU restatement lives in oracle/ and is test infrastructure).
"""
from __future__ import annotations

import ctypes as _C

import torch

from . import _capi
from ._capi import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, DIST_NORM, DIST_SQSUM, check, get_option, lib, set_option)  # noqa: F401

__all__ = ["lipschitz_scale", "mlp3_pack", "mlp3", "nearest", "ste", "mse_pair", "ACT_NONE", "ACT_GELU",
           "ACT_SIGMOID", "ACT_RELU", "DIST_NORM", "DIST_SQSUM"]


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the LipVQ tokenizer path runs on the GPU only (got a {t.device} tensor)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Current HIP stream of the current device as an integer handle (the fast private accessor when torch has it)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _on:
    """`with _on(device):` -- torch.cuda.device(device), skipped when that device is already current (the usual case:
    one process per GPU), which saves a few microseconds on every small launch of a training step."""

    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def lipschitz_scale(W: torch.Tensor, ci: torch.Tensor):
    """(scale[D], Wn[D,H]) of the reference's normalization() (backbone_lfqvae_v5.py:6-12)."""
    W, ci = _chk(W, "W"), _chk(ci, "ci")
    D, H = W.shape
    scale = torch.empty(D, device=W.device, dtype=torch.float32)
    Wn = torch.empty_like(W)
    with _on(W.device):
        check(lib.lipvq_lipschitz_scale_f32(_ptr(W), _ptr(ci), _ptr(scale), _ptr(Wn), D, H, _stream()),
              "lipvq_lipschitz_scale_f32")
    return scale, Wn


class PackedMlp3:
    """Weights of one three-layer stack in MFMA A-operand order (see csrc/lipvq_mlp.hip)."""

    __slots__ = ("buf", "K0", "J0", "J1", "J2")

    def __init__(self, buf, K0, J0, J1, J2):
        self.buf, self.K0, self.J0, self.J1, self.J2 = buf, K0, J0, J1, J2


def mlp3_pack(W0, b0, W1, b1, W2, b2) -> PackedMlp3:
    W0, b0, W1, b1, W2, b2 = (_chk(t, n) for t, n in
                              ((W0, "W0"), (b0, "b0"), (W1, "W1"), (b1, "b1"), (W2, "W2"), (b2, "b2")))
    J0, K0 = W0.shape
    J1, J2 = W1.shape[0], W2.shape[0]
    if W1.shape[1] != J0 or W2.shape[1] != J1 or b0.numel() != J0 or b1.numel() != J1 or b2.numel() != J2:
        raise ValueError("mlp3_pack: inconsistent layer shapes")
    n = lib.lipvq_mlp3_packed_floats(K0, J0, J1, J2)
    buf = torch.empty(n, device=W0.device, dtype=torch.float32)
    with _on(W0.device):
        check(lib.lipvq_mlp3_pack_f32(_ptr(W0), _ptr(b0), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(buf),
                                      K0, J0, J1, J2, _stream()), "lipvq_mlp3_pack_f32")
    return PackedMlp3(buf, K0, J0, J1, J2)


def mlp3(x: torch.Tensor, packed: PackedMlp3, acts, gather_idx: torch.Tensor | None = None,
         N: int | None = None, save_pre: bool = False):
    """y[N,J2] (and the three pre-activation tensors when save_pre).  With gather_idx, row n of
    the input is x[gather_idx[n]] (x is then the table, e.g. the codebook)."""
    x = _chk(x, "x")
    if x.dim() != 2 or x.shape[1] != packed.K0:
        raise ValueError(f"mlp3: x must be [N,{packed.K0}], got {tuple(x.shape)}")
    if gather_idx is not None:
        gather_idx = _chk(gather_idx, "gather_idx", torch.int64)
        N = gather_idx.numel()
    else:
        N = x.shape[0]
    dev = x.device
    y = torch.empty((N, packed.J2), device=dev, dtype=torch.float32)
    pre = [torch.empty((N, J), device=dev, dtype=torch.float32) if save_pre else None
           for J in (packed.J0, packed.J1, packed.J2)]
    with _on(dev):
        check(lib.lipvq_mlp3_f32(_ptr(x), _ptr(gather_idx), _ptr(packed.buf), _ptr(y), _ptr(pre[0]),
                                 _ptr(pre[1]), _ptr(pre[2]), N, packed.K0, packed.J0, packed.J1, packed.J2,
                                 int(acts[0]), int(acts[1]), int(acts[2]), _stream()), "lipvq_mlp3_f32")
    return (y, pre) if save_pre else y


def mlp3_loss_supported(N, packed: PackedMlp3) -> bool:
    """Can lipvq_mlp3_loss_f32 (the forward stack with the two mean-squared errors folded in) take this batch and stack?"""
    return bool(lib.lipvq_mlp3_loss_supported(int(N), packed.K0, packed.J0, packed.J1, packed.J2))


def mlp3_loss(x, packed: PackedMlp3, acts, gather_idx, target, latent, w: float, form: int, save_pre: bool = False, ste: bool = False):
    """(y, pre or None, out3): mlp3() plus out3 = tensor([mean((y - target)^2), mean((input rows - latent)^2), loss]) as
    mse_pair_loss() forms it, summed by the stack's own launch (mlp3_loss_supported() must hold).  ste=True: the stack runs on
    latent + (input rows - latent), the plain VQVAE's straight-through value, which is returned as a fourth result."""
    x, target, latent = _chk(x, "x"), _chk(target, "target"), _chk(latent, "latent")
    if x.dim() != 2 or x.shape[1] != packed.K0:
        raise ValueError(f"mlp3_loss: x must be [N,{packed.K0}], got {tuple(x.shape)}")
    if gather_idx is not None:
        gather_idx = _chk(gather_idx, "gather_idx", torch.int64)
        N = gather_idx.numel()
    else:
        N = x.shape[0]
    if tuple(target.shape) != (N, packed.J2) or tuple(latent.shape) != (N, packed.K0):
        raise ValueError(f"mlp3_loss: target must be [{N},{packed.J2}] and latent [{N},{packed.K0}]")
    dev = x.device
    y = torch.empty((N, packed.J2), device=dev, dtype=torch.float32)
    pre = [torch.empty((N, J), device=dev, dtype=torch.float32) if save_pre else None
           for J in (packed.J0, packed.J1, packed.J2)]
    out = torch.empty(3, device=dev, dtype=torch.float32)
    zst = torch.empty((N, packed.K0), device=dev, dtype=torch.float32) if ste else None
    ws = torch.empty(lib.lipvq_mse_workspace_bytes(), device=dev, dtype=torch.uint8)
    with _on(dev):
        check(lib.lipvq_mlp3_loss_f32(_ptr(x), _ptr(gather_idx), _ptr(packed.buf), _ptr(y), _ptr(pre[0]), _ptr(pre[1]), _ptr(pre[2]),
                                      N, packed.K0, packed.J0, packed.J1, packed.J2, int(acts[0]), int(acts[1]), int(acts[2]),
                                      _ptr(target), _ptr(latent), _ptr(zst), _ptr(out), float(w), int(form), _ptr(ws), _stream()),
              "lipvq_mlp3_loss_f32")
    if ste:
        return y, (pre if save_pre else None), out, zst
    return y, (pre if save_pre else None), out


def nearest(z: torch.Tensor, codebook: torch.Tensor, dist: int = DIST_NORM, usage: torch.Tensor | None = None,
            want_zq: bool = True, want_best: bool = False):
    """(idx[N] int64, zq[N,D] or None, best[N] or None); usage[K] int64 is accumulated in place."""
    z, codebook = _chk(z, "z"), _chk(codebook, "codebook")
    N, D = z.shape
    K = codebook.shape[0]
    if codebook.shape[1] != D:
        raise ValueError(f"nearest: codebook must be [K,{D}], got {tuple(codebook.shape)}")
    if usage is not None:
        usage = _chk(usage, "usage", torch.int64)
        if usage.numel() != K:
            raise ValueError("nearest: usage must have K entries")
    idx = torch.empty(N, device=z.device, dtype=torch.int64)
    zq = torch.empty_like(z) if want_zq else None
    best = torch.empty(N, device=z.device, dtype=torch.float32) if want_best else None
    with _on(z.device):
        check(lib.lipvq_nearest_f32(_ptr(z), _ptr(codebook), _ptr(idx), _ptr(zq), _ptr(usage), _ptr(best),
                                    N, K, D, int(dist), _stream()), "lipvq_nearest_f32")
    return idx, zq, best


def ste(ze: torch.Tensor, zq: torch.Tensor) -> torch.Tensor:
    ze, zq = _chk(ze, "ze"), _chk(zq, "zq")
    out = torch.empty_like(ze)
    with _on(ze.device):
        check(lib.lipvq_ste_f32(_ptr(ze), _ptr(zq), _ptr(out), ze.numel(), _stream()), "lipvq_ste_f32")
    return out


def mse_pair(xr, x, zq, ze) -> torch.Tensor:
    """tensor([mean((xr-x)^2), mean((zq-ze)^2)]) on the device."""
    xr, x, zq, ze = _chk(xr, "xr"), _chk(x, "x"), _chk(zq, "zq"), _chk(ze, "ze")
    if x.numel() == 0 or ze.numel() == 0:        # F.mse_loss of an empty tensor is nan
        return torch.full((2,), float("nan"), device=x.device, dtype=torch.float32)
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.lipvq_mse_workspace_bytes(), device=x.device, dtype=torch.uint8)
    with _on(x.device):
        check(lib.lipvq_mse_pair_f32(_ptr(xr), _ptr(x), x.numel(), _ptr(zq), _ptr(ze), ze.numel(), _ptr(out),
                                     _ptr(ws), _stream()), "lipvq_mse_pair_f32")
    return out


LOSS_LLFQ, LOSS_VQ = 0, 1


def mse_pair_loss(xr, x, zq, ze, w: float, form: int) -> torch.Tensor:
    """tensor([mean((xr-x)^2), mean((zq-ze)^2), loss]): the two means and the tokenizer's loss built from them on the device
    (LOSS_LLFQ: (m0 + w m1) + w m1;  LOSS_VQ: m0 + (m1 + w m1)) -- the reference's fp32 association, no torch arithmetic."""
    xr, x, zq, ze = _chk(xr, "xr"), _chk(x, "x"), _chk(zq, "zq"), _chk(ze, "ze")
    if x.numel() == 0 or ze.numel() == 0:        # F.mse_loss of an empty tensor is nan
        return torch.full((3,), float("nan"), device=x.device, dtype=torch.float32)
    out = torch.empty(3, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.lipvq_mse_workspace_bytes(), device=x.device, dtype=torch.uint8)
    with _on(x.device):
        check(lib.lipvq_mse_pair_loss_f32(_ptr(xr), _ptr(x), x.numel(), _ptr(zq), _ptr(ze), ze.numel(), _ptr(out), float(w), int(form),
                                          _ptr(ws), _stream()), "lipvq_mse_pair_loss_f32")
    return out


# ---- backward ---------------------------------------------------------------------------------

def mlp3_pack_bwd(W0, W1, W2) -> PackedMlp3:
    """Transposed weights of a stack for the backward-data chain (J2 -> J1 -> J0 -> K0)."""
    W0, W1, W2 = _chk(W0, "W0"), _chk(W1, "W1"), _chk(W2, "W2")
    J0, K0 = W0.shape
    J1, J2 = W1.shape[0], W2.shape[0]
    n = lib.lipvq_mlp3_packed_bwd_floats(K0, J0, J1, J2)
    buf = torch.empty(n, device=W0.device, dtype=torch.float32)
    with _on(W0.device):
        check(lib.lipvq_mlp3_pack_bwd_f32(_ptr(W0), _ptr(W1), _ptr(W2), _ptr(buf), K0, J0, J1, J2, _stream()),
              "lipvq_mlp3_pack_bwd_f32")
    return PackedMlp3(buf, K0, J0, J1, J2)


def mlp3_pack_bwd2(a, b):
    """mlp3_pack_bwd of two stacks, a = (W0, W1, W2) and b = (W0, W1, W2), in one launch -> (PackedMlp3, PackedMlp3)."""
    (aW0, aW1, aW2), (bW0, bW1, bW2) = ([_chk(t, "W") for t in a], [_chk(t, "W") for t in b])
    dims = []
    bufs = []
    for W0, W1, W2 in ((aW0, aW1, aW2), (bW0, bW1, bW2)):
        J0, K0 = W0.shape
        J1, J2 = W1.shape[0], W2.shape[0]
        dims.append((K0, J0, J1, J2))
        bufs.append(torch.empty(lib.lipvq_mlp3_packed_bwd_floats(K0, J0, J1, J2), device=W0.device, dtype=torch.float32))
    with _on(aW0.device):
        check(lib.lipvq_mlp3_pack_bwd2_f32(_ptr(aW0), _ptr(aW1), _ptr(aW2), _ptr(bufs[0]), *dims[0],
                                           _ptr(bW0), _ptr(bW1), _ptr(bW2), _ptr(bufs[1]), *dims[1], _stream()), "lipvq_mlp3_pack_bwd2_f32")
    return PackedMlp3(bufs[0], *dims[0]), PackedMlp3(bufs[1], *dims[1])


def mlp3_bwd_vq_supported(N, packed_bwd: PackedMlp3) -> bool:
    """Can lipvq_mlp3_bwd_vq_f32 (the chain with the VQ losses' gradient terms folded in) take this batch and stack?"""
    p = packed_bwd
    return bool(lib.lipvq_mlp3_bwd_vq_supported(int(N), p.K0, p.J0, p.J1, p.J2))


def _diff_term(term, N, width, name):
    """(a, a_idx, b, b_idx, alpha) -> checked tensors; an operand is [N, width] rows or (with its index vector) a table of such rows."""
    if term is None:
        return None, None, None, None, 0.0
    a, ia, b, ib, alpha = term
    a = _chk(a, name + ".a") if a is not None else None           # (in_term only: None = the forward's output, act2(pre2))
    b = _chk(b, name + ".b")
    ia = _chk(ia, name + ".a_idx", torch.int64) if ia is not None else None
    ib = _chk(ib, name + ".b_idx", torch.int64) if ib is not None else None
    for t, i, nm in ((a, ia, "a"), (b, ib, "b")):
        if t is None:
            continue
        if t.dim() != 2 or t.shape[1] != width or (i is None and t.shape[0] != N) or (i is not None and i.numel() != N):
            raise ValueError(f"{name}.{nm}: expected [{N}, {width}] rows or a table with an index vector of {N} entries")
    return a, ia, b, ib, float(alpha)


def mlp3_bwd(gy, pre, packed_bwd: PackedMlp3, acts, want_gx=True, in_term=None, out_term=None, gscale=None):
    """(g2, g1, g0, gx): dL/d(pre-activation) of the three layers and dL/d(input).
    in_term / out_term = (a, a_idx, b, b_idx, alpha): gy := alpha * gscale * (A - B) (gy may then be None; a = None stands for
    the forward's own output act2(pre2)) / gx += alpha * gscale * (A - B), folded into the launch; one term per launch,
    mlp3_bwd_vq_supported() must hold."""
    pre0, pre1 = _chk(pre[0], "pre0"), _chk(pre[1], "pre1")
    pre2 = _chk(pre[2], "pre2") if pre[2] is not None else None
    p = packed_bwd
    N = pre0.shape[0]
    dev = pre0.device
    ident = int(acts[2]) == ACT_NONE
    fused = in_term is not None or out_term is not None
    if gy is not None:
        gy = _chk(gy, "gy")
    elif in_term is None:
        raise ValueError("mlp3_bwd: gy is required without an in_term")
    if in_term is not None or not ident:
        g2 = torch.empty((N, p.J2), device=dev, dtype=torch.float32)
    else:
        g2 = gy
    g1 = torch.empty((N, p.J1), device=dev, dtype=torch.float32)
    g0 = torch.empty((N, p.J0), device=dev, dtype=torch.float32)
    gx = torch.empty((N, p.K0), device=dev, dtype=torch.float32) if want_gx else None
    with _on(dev):
        if fused:
            if out_term is not None and not want_gx:
                raise ValueError("mlp3_bwd: an out_term needs want_gx")
            ia_, iai, ib_, ibi, ial = _diff_term(in_term, N, p.J2, "in_term")
            if ia_ is not None or iai is not None:
                raise ValueError("mlp3_bwd: in_term's A is the forward's own output act2(pre2): pass a = a_idx = None")
            oa_, oai, ob_, obi, oal = _diff_term(out_term, N, p.K0, "out_term")
            gs = _chk(gscale, "gscale") if gscale is not None else None
            check(lib.lipvq_mlp3_bwd_vq_f32(_ptr(gy), _ptr(pre0), _ptr(pre1), _ptr(pre2), _ptr(p.buf),
                                            _ptr(g2) if g2 is not gy else None, _ptr(g1), _ptr(g0), _ptr(gx), N, p.K0, p.J0,
                                            p.J1, p.J2, int(acts[0]), int(acts[1]), int(acts[2]),
                                            _ptr(ib_), _ptr(ibi), ial,
                                            _ptr(oa_), _ptr(oai), _ptr(ob_), _ptr(obi), oal, _ptr(gs), _stream()),
                  "lipvq_mlp3_bwd_vq_f32")
        else:
            check(lib.lipvq_mlp3_bwd_f32(_ptr(gy), _ptr(pre0), _ptr(pre1), _ptr(pre2), _ptr(p.buf),
                                         None if ident else _ptr(g2), _ptr(g1), _ptr(g0), _ptr(gx), N, p.K0, p.J0,
                                         p.J1, p.J2, int(acts[0]), int(acts[1]), int(acts[2]), _stream()),
                  "lipvq_mlp3_bwd_f32")
    return g2, g1, g0, gx


def wgrad(G, H, h_act=ACT_NONE, hidx=None, want_bias=True):
    """(gW[J,Kd], gb[J]) of one Linear layer; H is the layer input (saved pre-activation + h_act,
    a raw tensor, or -- with hidx -- a table whose rows hidx select)."""
    G, H = _chk(G, "G"), _chk(H, "H")
    N, J = G.shape
    Kd = H.shape[1]
    if hidx is not None:
        hidx = _chk(hidx, "hidx", torch.int64)
    elif H.shape[0] != N:
        raise ValueError("wgrad: G and H must have the same number of rows")
    dev = G.device
    gW = torch.empty((J, Kd), device=dev, dtype=torch.float32)
    gb = torch.empty(J, device=dev, dtype=torch.float32) if want_bias else None
    ws = torch.empty(lib.lipvq_wgrad_workspace_bytes(N, J, Kd), device=dev, dtype=torch.uint8)
    with _on(dev):
        check(lib.lipvq_wgrad_f32(_ptr(G), _ptr(H), _ptr(hidx), int(h_act), _ptr(gW), _ptr(gb), _ptr(ws), N, J, Kd,
                                  _stream()), "lipvq_wgrad_f32")
    return gW, gb


def scatter_add(g, idx, K, deterministic=None, route=None):
    """gC[k] = sum of the rows of g whose idx is k (the gather's backward / index_add_).  deterministic=None follows
    torch.are_deterministic_algorithms_enabled(): strictly ascending row order per code, bit-identical to a sequential fp32
    index_add_.  Large batches go through a stable counting sort of the rows by code (csrc/lipvq_scatter.hip; no
    floating-point atomics): deterministic from 32 768 rows on (2.6 ms -> 0.1 ms at N = 524 288), otherwise from 65 536 rows
    on in 256-row segments (reproducible run to run; 70 vs 185 us) -- smaller batches: the scanning kernel / fp32 atomics.
    route = "atomics" | "sorted" | "sequential_sorted" | "sequential_scan" forces one (tests, measurements)."""
    g, idx = _chk(g, "g"), _chk(idx, "idx", torch.int64)
    N, D = g.shape
    if deterministic is None:
        deterministic = torch.are_deterministic_algorithms_enabled()
    if route is None:
        sortable = bool(lib.lipvq_scatter_add_sorted_supported(N, K, D))
        if deterministic:
            route = "sequential_sorted" if sortable else "sequential_scan"
        else:
            route = "sorted" if (sortable and N >= 65536) else "atomics"
    gC = torch.zeros((K, D), device=g.device, dtype=torch.float32)
    with _on(g.device):
        if route == "sequential_scan":
            check(lib.lipvq_scatter_add_det_f32(_ptr(g), _ptr(idx), _ptr(gC), N, K, D, _stream()), "lipvq_scatter_add_det_f32")
        elif route in ("sorted", "sequential_sorted"):
            ws = torch.empty(lib.lipvq_scatter_add_sorted_workspace_bytes(N, K, D), device=g.device, dtype=torch.uint8)
            check(lib.lipvq_scatter_add_sorted_f32(_ptr(g), _ptr(idx), _ptr(gC), _ptr(ws), N, K, D,
                                                   1 if route == "sequential_sorted" else 0, _stream()), "lipvq_scatter_add_sorted_f32")
        elif route == "atomics":
            check(lib.lipvq_scatter_add_f32(_ptr(g), _ptr(idx), _ptr(gC), N, K, D, _stream()), "lipvq_scatter_add_f32")
        else:
            raise ValueError(f"scatter_add: unknown route {route!r}")
    return gC


def scatter_add_vq(g, ze, table, idx, alpha, gscale=None, zq=None, deterministic=None):
    """gC[k] = sum over the rows n with idx[n] = k of  alpha * gscale * (table[k] - ze[n]) (+ g[n] when g is given):
    the codebook gradient of a training step -- the codebook-loss term formed per row plus what the decoder sent back --
    without the [N, D] intermediate.  Batches the counting-sort scatter takes (scatter_add's rule) form the rows inside its
    summing kernel (lipvq_scatter_add_sorted_vq_f32); smaller ones run scaled_diff + scatter_add: the same numbers either way."""
    ze, table, idx = _chk(ze, "ze"), _chk(table, "table"), _chk(idx, "idx", torch.int64)
    if g is not None:
        g = _chk(g, "g")
    N, D = ze.shape
    K = table.shape[0]
    if gscale is not None:
        gscale = _chk(gscale.reshape(1), "gscale")
    if deterministic is None:
        deterministic = torch.are_deterministic_algorithms_enabled()
    sortable = bool(lib.lipvq_scatter_add_sorted_supported(N, K, D))
    if not sortable or (not deterministic and N < 65536):
        return scatter_add(scaled_diff(zq if zq is not None else table[idx], ze, alpha, gscale=gscale, c=g), idx, K,
                           deterministic=deterministic)
    gC = torch.zeros((K, D), device=ze.device, dtype=torch.float32)
    ws = torch.empty(lib.lipvq_scatter_add_sorted_workspace_bytes(N, K, D), device=ze.device, dtype=torch.uint8)
    with _on(ze.device):
        check(lib.lipvq_scatter_add_sorted_vq_f32(_ptr(g), _ptr(ze), _ptr(table), float(alpha), _ptr(gscale), _ptr(idx), _ptr(gC),
                                                  _ptr(ws), N, K, D, 1 if deterministic else 0, _stream()),
              "lipvq_scatter_add_sorted_vq_f32")
    return gC


def lipschitz_bwd(W, ci, gWn):
    W, ci, gWn = _chk(W, "W"), _chk(ci, "ci"), _chk(gWn, "gWn")
    gW, gci = torch.empty_like(W), torch.empty_like(ci)
    with _on(W.device):
        check(lib.lipvq_lipschitz_bwd_f32(_ptr(W), _ptr(ci), _ptr(gWn), _ptr(gW), _ptr(gci), W.shape[0], W.shape[1],
                                          _stream()), "lipvq_lipschitz_bwd_f32")
    return gW, gci


def scaled_diff(a, b, alpha, gscale=None, c=None):
    """alpha * gscale * (a - b) + c, gscale a 0-dim/1-element device tensor (the upstream dL/dloss)."""
    a, b = _chk(a, "a"), _chk(b, "b")
    if c is not None:
        c = _chk(c, "c")
    if gscale is not None:
        gscale = _chk(gscale.reshape(1), "gscale")
    out = torch.empty_like(a)
    with _on(a.device):
        check(lib.lipvq_scaled_diff_f32(_ptr(a), _ptr(b), _ptr(c), float(alpha), _ptr(gscale), _ptr(out), a.numel(),
                                        _stream()), "lipvq_scaled_diff_f32")
    return out


# ---- nearest code, screened fast path ------------------------------------------------------------

class PreparedCodebook:
    """Per-codebook data of the MFMA screen (lipvq_nearest_prepare_f32); rebuild when the codebook changes."""

    __slots__ = ("buf", "K", "D")

    def __init__(self, buf, K, D):
        self.buf, self.K, self.D = buf, K, D


def nearest_screen_supported(K: int, D: int) -> bool:
    return bool(lib.lipvq_nearest_screened_supported(int(K), int(D)))


def screen_is_coarse(K: int, D: int) -> bool:
    """True if the screened routes would run the one-product screen for this shape now (lipvq_screen_is_coarse)."""
    return bool(lib.lipvq_screen_is_coarse(int(K), int(D)))


def nearest_prepare(codebook: torch.Tensor) -> PreparedCodebook:
    codebook = _chk(codebook, "codebook")
    K, D = codebook.shape
    buf = torch.empty(lib.lipvq_nearest_prep_bytes(K, D), device=codebook.device, dtype=torch.uint8)
    with _on(codebook.device):
        check(lib.lipvq_nearest_prepare_f32(_ptr(codebook), _ptr(buf), K, D, _stream()), "lipvq_nearest_prepare_f32")
    return PreparedCodebook(buf, K, D)


_small_ws: dict = {}          # (device index, stream handle) -> zero-at-rest workspace of lipvq_nearest_small_f32


def _nearest_small_workspace(N, K, dev):
    """The chip-wide small-batch kernel meets a row's partial minima behind per-row-group counters that are zero between
    launches (the last workgroup resets them): one zero-filled buffer per (device, stream) serves every EAGER call -- no fill
    launch per call.  Under stream capture every call gets its own zeroed buffer (one memset node)."""
    need = lib.lipvq_nearest_small_workspace_bytes(N, K)
    if torch.cuda.is_current_stream_capturing():
        # Never shared across captures: a cached buffer's zero fill would be a node of the FIRST graph that used it only (a
        # second graph replayed first, or two graphs on two streams, would meet on uninitialised or shared counters).  A fresh
        # zeroed buffer per call puts the fill into this graph and the buffer into this graph's private pool.
        return torch.zeros(need, device=dev, dtype=torch.uint8)
    key = (dev.index, _stream())
    ws = _small_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _small_ws[key] = torch.zeros(max(need, 1 << 16), device=dev, dtype=torch.uint8)
    return ws


def nearest_rows(z, codebook, usage=None, want_zq=True, dist: int = DIST_NORM, route=None):
    """(idx, zq) exactly as nearest(z, codebook, dist), every row decided by an exact kernel (no prepared codebook): the
    route for small batches.  Up to 4 096 rows run lipvq_nearest_small_f32 (4 rows x 64 codes per workgroup, the whole chip
    busy at 80 rows), larger ones the row kernels (route = "small" | "rows" forces one: tests, measurements)."""
    z, codebook = _chk(z, "z"), _chk(codebook, "codebook")
    N, D = z.shape
    K = codebook.shape[0]
    if codebook.shape[1] != D:
        raise ValueError("nearest_rows: codebook width does not match")
    if usage is not None:
        usage = _chk(usage, "usage", torch.int64)
    if dist not in (DIST_NORM, DIST_SQSUM):
        raise ValueError(f"nearest_rows: unknown distance rule {dist}")
    idx = torch.empty(N, device=z.device, dtype=torch.int64)
    zq = torch.empty_like(z) if want_zq else None
    small = bool(lib.lipvq_nearest_small_supported(N, K, D)) if route is None else route == "small"
    if small and N > 0:
        with _on(z.device):
            ws = _nearest_small_workspace(N, K, z.device)
            check(lib.lipvq_nearest_small_f32(_ptr(z), _ptr(codebook), _ptr(idx), _ptr(zq), _ptr(usage), _ptr(ws), N, K, D, int(dist),
                                              _stream()), "lipvq_nearest_small_f32")
        return idx, zq
    fn, name = ((lib.lipvq_nearest_rows_f32, "lipvq_nearest_rows_f32") if dist == DIST_NORM else
                (lib.lipvq_vq_nearest_rows_f32, "lipvq_vq_nearest_rows_f32"))
    if dist not in (DIST_NORM, DIST_SQSUM):
        raise ValueError(f"nearest_rows: unknown distance rule {dist}")
    with _on(z.device):
        check(fn(_ptr(z), _ptr(codebook), _ptr(idx), _ptr(zq), _ptr(usage), N, K, D, _stream()), name)
    return idx, zq


def nearest_screened(z, codebook, prep: PreparedCodebook, usage=None, want_zq=True, return_workspace=False,
                     debug_gamma=None, dist: int = DIST_NORM):
    """Same results as nearest(z, codebook, dist), via MFMA screening + exact re-scoring of the
    rows the screen cannot certify.  With return_workspace the int32 workspace is returned too
    (element 0 = number of rows decided by the exact kernel).  debug_gamma: test hook, returns the
    approximate distance matrix as well."""
    z, codebook = _chk(z, "z"), _chk(codebook, "codebook")
    N, D = z.shape
    K = codebook.shape[0]
    if (prep.K, prep.D) != (K, D) or codebook.shape[1] != D:
        raise ValueError("nearest_screened: prepared codebook does not match")
    if usage is not None:
        usage = _chk(usage, "usage", torch.int64)
    dev = z.device
    idx = torch.empty(N, device=dev, dtype=torch.int64)
    zq = torch.empty_like(z) if want_zq else None
    ws = torch.empty(max(16, lib.lipvq_nearest_workspace_bytes(N) // 4), device=dev, dtype=torch.int32)
    dt = None
    if dist not in (DIST_NORM, DIST_SQSUM) or (dist != DIST_NORM and debug_gamma is not None):
        raise ValueError("nearest_screened: unknown distance rule (the debug hook runs the norm rule only)")
    with _on(dev):
        if debug_gamma is None and dist == DIST_SQSUM:        # the plain VQVAE's rule (vq:57-63)
            check(lib.lipvq_vq_nearest_screened_f32(_ptr(z), _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                                    _ptr(usage), _ptr(ws), N, K, D, _stream()),
                  "lipvq_vq_nearest_screened_f32")
        elif debug_gamma is None:
            check(lib.lipvq_nearest_screened_f32(_ptr(z), _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                                 _ptr(usage), _ptr(ws), N, K, D, _stream()),
                  "lipvq_nearest_screened_f32")
        else:
            kpad = (K + 31) // 32 * 32
            dt = torch.empty((N, kpad), device=dev, dtype=torch.float32)
            check(lib.lipvq_screen_debug_f32(_ptr(z), _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                             _ptr(usage), _ptr(ws), _ptr(dt), float(debug_gamma), N, K, D,
                                             _stream()), "lipvq_screen_debug_f32")
    out = (idx, zq)
    if return_workspace:
        out = out + (ws,)
    if debug_gamma is not None:
        out = out + (dt,)
    return out


# ---- fused encode + quantize ------------------------------------------------------------------------

def tokenize_supported(A, J0, J1, D, K) -> bool:
    return bool(lib.lipvq_tokenize_supported(int(A), int(J0), int(J1), int(D), int(K)))


def tokenize_fast_supported(A, J0, J1, D, K) -> bool:
    return bool(lib.lipvq_tokenize_fast_supported(int(A), int(J0), int(J1), int(D), int(K)))


def tokenize_workspace(N: int, D: int, device) -> torch.Tensor:
    """int32 workspace of lipvq_tokenize_f32 (row list + z_e scratch) with its header zeroed (lipvq_tokenize_workspace_init: the
    fused calls keep the header's counters at zero themselves); callers may keep and reuse it -- on one stream at a time."""
    ws = torch.empty(max(16, (lib.lipvq_tokenize_workspace_bytes(N, D) + 3) // 4), device=device, dtype=torch.int32)
    with _on(ws.device):
        check(lib.lipvq_tokenize_workspace_init(_ptr(ws), _stream()), "lipvq_tokenize_workspace_init")
    return ws


def mlp3_pack_f16(W0, W1, W2):
    """fp16 MFMA fragments of an encoder stack for the fast tokenize mode (uint8 buffer)."""
    W0, W1, W2 = _chk(W0, "W0"), _chk(W1, "W1"), _chk(W2, "W2")
    A, J0, J1, D = W0.shape[1], W0.shape[0], W1.shape[0], W2.shape[0]
    nbytes = lib.lipvq_mlp3_packed_f16_bytes(A, J0, J1, D)
    if not nbytes or W1.shape[1] != J0 or W2.shape[1] != J1:
        raise ValueError("mlp3_pack_f16: unsupported stack shape")
    buf = torch.empty(nbytes, device=W0.device, dtype=torch.uint8)
    with _on(W0.device):
        check(lib.lipvq_mlp3_pack_f16_f32(_ptr(W0), _ptr(W1), _ptr(W2), _ptr(buf), A, J0, J1, D, _stream()),
              "lipvq_mlp3_pack_f16_f32")
    return buf


def tokenize(x, packed: PackedMlp3, raw, codebook, prep: PreparedCodebook, usage=None, want_zq=True, want_ze=False,
             workspace=None, packed16=None, want_pre=False):
    """(idx, zq, ze, workspace) of the fused encode + quantize launch (lipvq_tokenize_f32).  raw = the encoder's six
    unpacked tensors (W0, b0, W1, b1, W2 normalised, b2): the exact kernel re-encodes uncertified rows with them.
    want_pre=True (training forward, lipvq_tokenize_train_f32): also returns the three pre-activations as a 5th element."""
    raw = tuple(_chk(t, f"raw[{i}]") for i, t in enumerate(raw))
    raw_arr = (_C.c_void_p * 6)(*[t.data_ptr() for t in raw])
    x, codebook = _chk(x, "x"), _chk(codebook, "codebook")
    N, A = x.shape
    K, D = codebook.shape
    if (packed.K0, packed.J2) != (A, D) or (prep.K, prep.D) != (K, D):
        raise ValueError("tokenize: packed encoder / prepared codebook do not match the inputs")
    if usage is not None:
        usage = _chk(usage, "usage", torch.int64)
    dev = x.device
    idx = torch.empty(N, device=dev, dtype=torch.int64)
    zq = torch.empty((N, D), device=dev, dtype=torch.float32) if want_zq else None
    ze = torch.empty((N, D), device=dev, dtype=torch.float32) if (want_ze or want_pre) else None
    ws = workspace if workspace is not None else tokenize_workspace(N, D, dev)
    if want_pre:
        if packed16 is not None:
            raise ValueError("tokenize: the fast mode has no training variant")
        pre = (torch.empty((N, packed.J0), device=dev, dtype=torch.float32), torch.empty((N, packed.J1), device=dev, dtype=torch.float32),
               torch.empty((N, D), device=dev, dtype=torch.float32))
        with _on(dev):
            check(lib.lipvq_tokenize_train_f32(_ptr(x), _ptr(packed.buf), raw_arr, _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                               _ptr(usage), _ptr(ze), _ptr(pre[0]), _ptr(pre[1]), _ptr(pre[2]), _ptr(ws), N, A,
                                               packed.J0, packed.J1, D, K, _stream()), "lipvq_tokenize_train_f32")
        return idx, zq, ze, ws, pre
    with _on(dev):
        if packed16 is not None:            # fast mode (fp16 encoder GEMMs): not bit-identical, see include/lipvq.h
            if want_ze:
                raise ValueError("tokenize: the fast mode does not return z_e")
            check(lib.lipvq_tokenize_fast_f32(_ptr(x), _ptr(packed.buf), _ptr(packed16), raw_arr, _ptr(codebook), _ptr(prep.buf),
                                              _ptr(idx), _ptr(zq), _ptr(usage), _ptr(ws), N, A, packed.J0, packed.J1, D, K,
                                              _stream()), "lipvq_tokenize_fast_f32")
        else:
            check(lib.lipvq_tokenize_f32(_ptr(x), _ptr(packed.buf), raw_arr, _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                         _ptr(usage), _ptr(ze), _ptr(ws), N, A, packed.J0, packed.J1, D, K, _stream()),
                  "lipvq_tokenize_f32")
    return idx, zq, ze, ws


def tokenize_tune(x, packed: PackedMlp3, raw, codebook, prep: PreparedCodebook, workspace=None, launches=150):
    """lipvq_tokenize_tune_f32: time the fused launch's four device-dependent schedule combinations (identical results) on this
    device, shape and data and keep the fastest for the process.  Synchronous.  Returns (choice, [ms per launch] * 4) with
    choice = defer_ze | nt_ze << 1."""
    raw = tuple(_chk(t, f"raw[{i}]") for i, t in enumerate(raw))
    raw_arr = (_C.c_void_p * 6)(*[t.data_ptr() for t in raw])
    x, codebook = _chk(x, "x"), _chk(codebook, "codebook")
    N, A = x.shape
    K, D = codebook.shape
    if (packed.K0, packed.J2) != (A, D) or (prep.K, prep.D) != (K, D):
        raise ValueError("tokenize_tune: packed encoder / prepared codebook do not match the inputs")
    dev = x.device
    idx = torch.empty(N, device=dev, dtype=torch.int64)
    zq = torch.empty((N, D), device=dev, dtype=torch.float32)
    usage = torch.zeros(K, device=dev, dtype=torch.int64)           # scratch: every timed launch accumulates into it
    ws = workspace if workspace is not None else tokenize_workspace(N, D, dev)
    choice = _C.c_int(-1)
    ms4 = (_C.c_float * 4)()
    with _on(dev):
        check(lib.lipvq_tokenize_tune_f32(_ptr(x), _ptr(packed.buf), raw_arr, _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                          _ptr(usage), None, _ptr(ws), N, A, packed.J0, packed.J1, D, K, _stream(), int(launches),
                                          _C.byref(choice), ms4), "lipvq_tokenize_tune_f32")
    return int(choice.value), [float(v) for v in ms4]


def vq_tokenize(x, packed: PackedMlp3, codebook, prep: PreparedCodebook, usage=None, workspace=None, want_pre=False):
    """(idx, zq, ze, workspace) of the plain VQVAE's fused encode + quantize launch (lipvq_vq_tokenize_f32: ReLU encoder,
    `pow(2).sum(-1)` argmin): the same results as mlp3(relu x 3) + nearest(DIST_SQSUM).  z_e is always returned (the
    straight-through value needs it).  want_pre: (idx, zq, ze, [pre0, pre1, pre2], workspace) -- the training forward,
    lipvq_vq_tokenize_train_f32."""
    x, codebook = _chk(x, "x"), _chk(codebook, "codebook")
    N, A = x.shape
    K, D = codebook.shape
    if (packed.K0, packed.J2) != (A, D) or (prep.K, prep.D) != (K, D):
        raise ValueError("vq_tokenize: packed encoder / prepared codebook do not match the inputs")
    if usage is not None:
        usage = _chk(usage, "usage", torch.int64)
    dev = x.device
    idx = torch.empty(N, device=dev, dtype=torch.int64)
    zq = torch.empty((N, D), device=dev, dtype=torch.float32)
    ze = torch.empty((N, D), device=dev, dtype=torch.float32)
    ws = workspace if workspace is not None else tokenize_workspace(N, D, dev)
    if want_pre:
        pre = [torch.empty((N, J), device=dev, dtype=torch.float32) for J in (packed.J0, packed.J1, D)]
        with _on(dev):
            check(lib.lipvq_vq_tokenize_train_f32(_ptr(x), _ptr(packed.buf), _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq),
                                                  _ptr(usage), _ptr(ze), _ptr(pre[0]), _ptr(pre[1]), _ptr(pre[2]), _ptr(ws), N, A,
                                                  packed.J0, packed.J1, D, K, _stream()), "lipvq_vq_tokenize_train_f32")
        return idx, zq, ze, pre, ws
    with _on(dev):
        check(lib.lipvq_vq_tokenize_f32(_ptr(x), _ptr(packed.buf), _ptr(codebook), _ptr(prep.buf), _ptr(idx), _ptr(zq), _ptr(usage),
                                        _ptr(ze), _ptr(ws), N, A, packed.J0, packed.J1, D, K, _stream()), "lipvq_vq_tokenize_f32")
    return idx, zq, ze, ws


# ---------------------------------------------------------------------------------------------------
# the step after the tokenizer: input embedding + interleave (csrc/lipvq_embed.hip)
# ---------------------------------------------------------------------------------------------------

def linear(x, W, b=None, act=ACT_NONE, save_pre=False):
    """y = act(x . W^T + b) (nn.Linear), canonical fp32 (reference obs_nets.py:2536 `embed_encoder`; with act = GELU the
    second half of AdaptiveBinActionEmbedding.output_layer).  save_pre=True returns (y, pre-activation)."""
    x, W = _chk(x, "x"), _chk(W, "W")
    if x.dim() != 2 or W.dim() != 2 or W.shape[1] != x.shape[1]:
        raise ValueError(f"linear: x {tuple(x.shape)} and W {tuple(W.shape)} do not match")
    if b is not None:
        b = _chk(b, "b")
        if b.shape != (W.shape[0],):
            raise ValueError("linear: bias shape")
    N, Kin = x.shape
    E = W.shape[0]
    y = torch.empty((N, E), device=x.device, dtype=torch.float32)
    pre = torch.empty_like(y) if save_pre else None
    with _on(x.device):
        check(lib.lipvq_linear_act_f32(_ptr(x), _ptr(W), _ptr(b), _ptr(y), _ptr(pre), N, Kin, E, int(act), _stream()),
              "lipvq_linear_act_f32")
    return (y, pre) if save_pre else y


def _embed_args(src, idx, pos, N, T, E):
    if src.dim() != 2 or src.shape[1] != E:
        raise ValueError(f"embed_rows: src {tuple(src.shape)} is not [rows, {E}]")
    if idx is not None:
        idx = _chk(idx, "idx", torch.int64).reshape(-1)
        if idx.numel() != N:
            raise ValueError("embed_rows: idx must hold one index per row")
    if pos is not None:
        pos = _chk(pos, "pos")
        if pos.shape != (T, E):
            raise ValueError(f"embed_rows: pos {tuple(pos.shape)} is not [{T}, {E}]")
    return idx, pos


def embed_rows(src, idx, pos, ln_w, ln_b, eps, out, N, T, bstride, tstride, offset, want_stats=False):
    """out[slot(n)] = LayerNorm(src[idx[n] | n] + pos[n % T]) * ln_w + ln_b, slot(b*T+t) = b*bstride + t*tstride + offset
    floats into ``out`` (reference obs_nets.py:2537-2540 and the interleave of :2584-2596).  Returns stats [N,2] or None."""
    src, ln_w, ln_b, out = _chk(src, "src"), _chk(ln_w, "ln_w"), _chk(ln_b, "ln_b"), _chk(out, "out")
    E = ln_w.numel()
    idx, pos = _embed_args(src, idx, pos, N, T, E)
    if N:
        B = (N + T - 1) // T
        last = (B - 1) * bstride + (min(T, N) - 1) * tstride + offset + E
        if last > out.numel():
            raise ValueError("embed_rows: the output slots do not fit in `out`")
    stats = torch.empty((N, 2), device=src.device, dtype=torch.float32) if want_stats else None
    with _on(src.device):
        check(lib.lipvq_embed_rows_f32(_ptr(src), _ptr(idx), _ptr(pos), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(out),
                                       _ptr(stats), N, T, E, src.shape[0], bstride, tstride, offset, _stream()),
              "lipvq_embed_rows_f32")
    return stats


def embed_rows_bwd(gout, src, idx, pos, stats, ln_w, g_src, g_pos, g_lnw, g_lnb, N, T, bstride, tstride, offset):
    """Backward of embed_rows; accumulates into g_src / g_pos / g_lnw / g_lnb (each may be None)."""
    gout, src, stats, ln_w = _chk(gout, "gout"), _chk(src, "src"), _chk(stats, "stats"), _chk(ln_w, "ln_w")
    E = ln_w.numel()
    idx, pos = _embed_args(src, idx, pos, N, T, E)
    for name, t, shape in (("g_src", g_src, tuple(src.shape)), ("g_pos", g_pos, (T, E)), ("g_lnw", g_lnw, (E,)),
                           ("g_lnb", g_lnb, (E,))):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32):
            raise ValueError(f"embed_rows_bwd: {name} must be a contiguous fp32 tensor of shape {shape}")
    with _on(src.device):
        big_dense = idx is None and N >= 32768 and T <= 1024
        if big_dense or (idx is not None and lib.lipvq_embed_rows_bwd_ws_supported(N, T, E, src.shape[0])):
            # large batches: no atomics on the table / time embedding (row gradients written once, counting-sort scatter; dense
            # rows: the row gradients are g_src)
            ws = None if big_dense else torch.empty(lib.lipvq_embed_rows_bwd_workspace_bytes(N, T, E, src.shape[0]),
                                                    device=src.device, dtype=torch.uint8)
            check(lib.lipvq_embed_rows_bwd_ws_f32(_ptr(gout), _ptr(src), _ptr(idx), _ptr(pos), _ptr(stats), _ptr(ln_w),
                                                  _ptr(g_src), _ptr(g_pos), _ptr(g_lnw), _ptr(g_lnb), _ptr(ws), N, T, E,
                                                  src.shape[0], bstride, tstride, offset, _stream()), "lipvq_embed_rows_bwd_ws_f32")
            return
        check(lib.lipvq_embed_rows_bwd_f32(_ptr(gout), _ptr(src), _ptr(idx), _ptr(pos), _ptr(stats), _ptr(ln_w),
                                           _ptr(g_src), _ptr(g_pos), _ptr(g_lnw), _ptr(g_lnb), N, T, E, src.shape[0],
                                           bstride, tstride, offset, _stream()), "lipvq_embed_rows_bwd_f32")


# ---------------------------------------------------------------------------------------------------
# AdaptiveBinActionEmbedding (csrc/lipvq_bin.hip; reference robomimic/models/bin_action/backbone.py)
# ---------------------------------------------------------------------------------------------------

def bin_minmax(actions, running_min, running_max):
    """In-place running min/max update (backbone.py:37-40)."""
    actions = _chk(actions, "actions")
    for name, t in (("running_min", running_min), ("running_max", running_max)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (actions.shape[1],)):
            raise ValueError(f"bin_minmax: {name} must be a contiguous fp32 CUDA tensor of shape [{actions.shape[1]}]")
    N, A = actions.shape
    with _on(actions.device):
        check(lib.lipvq_bin_minmax_f32(_ptr(actions), _ptr(running_min), _ptr(running_max), N, A, _stream()),
              "lipvq_bin_minmax_f32")


def bin_discretize(actions, running_min, running_max, num_bins):
    """bins [A, N] int64 (backbone.py:42-66; the reference's [N, A] stack is bins.t())."""
    actions, running_min, running_max = _chk(actions, "actions"), _chk(running_min, "running_min"), _chk(running_max, "running_max")
    N, A = actions.shape
    bins = torch.empty((A, N), device=actions.device, dtype=torch.int64)
    with _on(actions.device):
        check(lib.lipvq_bin_discretize_f32(_ptr(actions), _ptr(running_min), _ptr(running_max), _ptr(bins), N, A,
                                           int(num_bins), _stream()), "lipvq_bin_discretize_f32")
    return bins


def bin_boundaries(running_min, running_max, num_bins):
    """[A, num_bins + 1] bin boundaries (backbone.py:42-53)."""
    running_min, running_max = _chk(running_min, "running_min"), _chk(running_max, "running_max")
    A = running_min.numel()
    out = torch.empty((A, num_bins + 1), device=running_min.device, dtype=torch.float32)
    with _on(out.device):
        check(lib.lipvq_bin_boundaries_f32(_ptr(running_min), _ptr(running_max), _ptr(out), A, int(num_bins), _stream()),
              "lipvq_bin_boundaries_f32")
    return out


def bin_hidden(bins, P, b1, save_pre=False):
    """h = gelu(b1 + sum_i P[i][bins[i]]) (backbone.py:77-86 up to the first GELU).  P [A, num_bins, H]."""
    bins, P, b1 = _chk(bins, "bins", torch.int64), _chk(P, "P"), _chk(b1, "b1")
    A, N = bins.shape
    if P.dim() != 3 or P.shape[0] != A or b1.shape != (P.shape[2],):
        raise ValueError("bin_hidden: P must be [A, num_bins, H] and b1 [H]")
    nb, H = P.shape[1], P.shape[2]
    h = torch.empty((N, H), device=P.device, dtype=torch.float32)
    pre = torch.empty_like(h) if save_pre else None
    with _on(P.device):
        check(lib.lipvq_bin_hidden_f32(_ptr(bins), _ptr(P), _ptr(b1), _ptr(h), _ptr(pre), N, A, nb, H, _stream()),
              "lipvq_bin_hidden_f32")
    return (h, pre) if save_pre else h


def act_bwd(g, pre, act):
    """g * act'(pre), elementwise."""
    g, pre = _chk(g, "g"), _chk(pre, "pre")
    if g.shape != pre.shape:
        raise ValueError("act_bwd: shapes differ")
    out = torch.empty_like(g)
    with _on(g.device):
        check(lib.lipvq_act_bwd_f32(_ptr(g), _ptr(pre), _ptr(out), g.numel(), int(act), _stream()), "lipvq_act_bwd_f32")
    return out


def ema_update(cluster_size, embed_sum, counts, dw, codebook, decay, eps):
    """In-place EMA codebook update (opt-in extension; include/lipvq.h lipvq_ema_update_f32)."""
    K, D = codebook.shape
    for name, t, shape, dt in (("cluster_size", cluster_size, (K,), torch.float32), ("embed_sum", embed_sum, (K, D), torch.float32),
                               ("counts", counts, (K,), torch.int64), ("dw", dw, (K, D), torch.float32),
                               ("codebook", codebook, (K, D), torch.float32)):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"ema_update: {name} must be a contiguous CUDA {dt} tensor of shape {shape}")
    ws = torch.empty(1, device=codebook.device, dtype=torch.float64)
    with _on(codebook.device):
        check(lib.lipvq_ema_update_f32(_ptr(cluster_size), _ptr(embed_sum), _ptr(counts), _ptr(dw), _ptr(codebook),
                                       float(decay), float(eps), K, D, _ptr(ws), _stream()), "lipvq_ema_update_f32")


# ---------------------------------------------------------------------------------------------------
# the default action branch (obs_nets.py:1244-1260; csrc/lipvq_xf.hip)
# ---------------------------------------------------------------------------------------------------

def spectral_norm(W, u, v, do_power_iteration: bool, eps: float = 1e-12):
    """(W / sigma, sigma [1]) of torch.nn.utils.spectral_norm; with do_power_iteration u and v are UPDATED IN PLACE."""
    W, u, v = _chk(W, "W"), _chk(u, "u"), _chk(v, "v")
    if W.dim() != 2 or u.shape != (W.shape[0],) or v.shape != (W.shape[1],):
        raise ValueError(f"spectral_norm: W {tuple(W.shape)}, u {tuple(u.shape)}, v {tuple(v.shape)}")
    Wsn = torch.empty_like(W)
    sigma = torch.empty(1, device=W.device, dtype=torch.float32)
    with _on(W.device):
        check(lib.lipvq_spectral_norm_f32(_ptr(W), _ptr(u), _ptr(v), _ptr(Wsn), _ptr(sigma), W.shape[0], W.shape[1],
                                          int(bool(do_power_iteration)), float(eps), _stream()), "lipvq_spectral_norm_f32")
    return Wsn, sigma


def spectral_norm_bwd(gWsn, Wsn, u, v, sigma):
    gWsn, Wsn = _chk(gWsn, "gWsn"), _chk(Wsn, "Wsn")
    gW = torch.empty_like(Wsn)
    with _on(Wsn.device):
        check(lib.lipvq_spectral_norm_bwd_f32(_ptr(gWsn), _ptr(Wsn), _ptr(_chk(u, "u")), _ptr(_chk(v, "v")), _ptr(_chk(sigma, "sigma")),
                                              _ptr(gW), Wsn.shape[0], Wsn.shape[1], _stream()), "lipvq_spectral_norm_bwd_f32")
    return gW


def _chk_keep(keep, H, S, dev):
    if keep is None:
        return None
    if keep.dtype != torch.uint8 or keep.shape != (H, S, S) or not keep.is_contiguous() or keep.device != dev:
        raise ValueError("attention: keep must be a contiguous uint8 [H, S, S] tensor on the input's device")
    return keep


def attention(qkv, nhead: int, keep=None, keep_prob: float = 1.0):
    """(out [S, D], lse [H, S]) of multi-head self-attention over the unbatched sequence qkv [S, 3D] (q | k | v)."""
    qkv = _chk(qkv, "qkv")
    if qkv.dim() != 2 or qkv.shape[1] % 3 != 0:
        raise ValueError(f"attention: qkv {tuple(qkv.shape)}")
    S, D = qkv.shape[0], qkv.shape[1] // 3
    keep = _chk_keep(keep, nhead, S, qkv.device)
    out = torch.empty((S, D), device=qkv.device, dtype=torch.float32)
    lse = torch.empty((nhead, S), device=qkv.device, dtype=torch.float32)
    with _on(qkv.device):
        check(lib.lipvq_attention_f32(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(keep), float(keep_prob), S, D, int(nhead), _stream()),
              "lipvq_attention_f32")
    return out, lse


def attention_bwd(qkv, out, gout, lse, nhead: int, keep=None, keep_prob: float = 1.0):
    qkv, out, gout, lse = _chk(qkv, "qkv"), _chk(out, "out"), _chk(gout, "gout"), _chk(lse, "lse")
    S, D = out.shape
    keep = _chk_keep(keep, nhead, S, qkv.device)
    gqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    with _on(qkv.device):
        check(lib.lipvq_attention_bwd_f32(_ptr(qkv), _ptr(out), _ptr(gout), _ptr(lse), _ptr(gqkv), _ptr(delta), _ptr(keep),
                                          float(keep_prob), S, D, int(nhead), _stream()), "lipvq_attention_bwd_f32")
    return gqkv


def add_layernorm(a, b, w, bias, eps: float, save: bool = False):
    """LayerNorm(a + b) * w + bias over the last dimension (b may be None); save=True also returns (xhat, rstd)."""
    a = _chk(a, "a")
    if b is not None:
        b = _chk(b, "b")
        if b.shape != a.shape:
            raise ValueError("add_layernorm: shapes differ")
    N, E = a.shape
    y = torch.empty_like(a)
    xhat = torch.empty_like(a) if save else None
    rstd = torch.empty(N, device=a.device, dtype=torch.float32) if save else None
    with _on(a.device):
        check(lib.lipvq_add_layernorm_f32(_ptr(a), _ptr(b), _ptr(_chk(w, "w")), _ptr(_chk(bias, "bias")), float(eps), _ptr(y),
                                          _ptr(xhat), _ptr(rstd), N, E, _stream()), "lipvq_add_layernorm_f32")
    return (y, xhat, rstd) if save else y


def layernorm_bwd(gy, xhat, rstd, w):
    gy, xhat = _chk(gy, "gy"), _chk(xhat, "xhat")
    N, E = gy.shape
    gx = torch.empty_like(gy)
    gw = torch.zeros(E, device=gy.device, dtype=torch.float32)
    gb = torch.zeros(E, device=gy.device, dtype=torch.float32)
    with _on(gy.device):
        check(lib.lipvq_layernorm_bwd_f32(_ptr(gy), _ptr(xhat), _ptr(_chk(rstd, "rstd")), _ptr(_chk(w, "w")), _ptr(gx), _ptr(gw),
                                          _ptr(gb), N, E, _stream()), "lipvq_layernorm_bwd_f32")
    return gx, gw, gb
