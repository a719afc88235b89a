"""Opt-in EMA codebook update -- an EXTENSION, not reference behaviour.

The reference trains the codebook by gradient (backbone_lfqvae_v5.py:32-35,81) and has no EMA; BASELINE.json's north star
and SURVEY.md section 8e name it as an off-by-default extension whose statistics are what crosses GPUs.  The rule is the
standard one (van den Oord et al. 2017, appendix A.1): running ``cluster_size[K]`` and ``embed_sum[K, D]``, Laplace
smoothing, ``codebook = embed_sum / smoothed_cluster_size``.  Per batch the ranks exchange the int64 usage counts and the
``[K, D]`` per-code sums of z_e (one all-reduce each, RCCL over xGMI on GPUs); everything else is local.
"""
from __future__ import annotations

import torch

from . import ops, sharded

__all__ = ["EMACodebook"]


class EMACodebook:
    """Keeps the EMA statistics of one codebook tensor and rewrites it in place after each batch."""

    def __init__(self, codebook: torch.Tensor, decay: float = 0.99, eps: float = 1e-5, group=None):
        if codebook.dim() != 2:
            raise ValueError("codebook must be [K, D]")
        self.codebook = codebook
        self.decay, self.eps, self.group = float(decay), float(eps), group
        self.cluster_size = torch.zeros(codebook.shape[0], device=codebook.device, dtype=torch.float32)
        self.embed_sum = codebook.detach().clone().contiguous()

    @torch.no_grad()
    def update(self, z_e: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor | None = None):
        """z_e [N, D] (this rank's encoder outputs), indices [N] int64 (their codes); counts [K] int64 may be passed
        when the quantizer already produced this batch's usage histogram.  Returns the global counts."""
        K, D = self.codebook.shape
        idx = indices.reshape(-1)
        if counts is None:
            counts = torch.bincount(idx, minlength=K)
        counts = counts.to(torch.int64).clone()
        dw = ops.scatter_add(z_e.detach().reshape(-1, D), idx, K)
        sharded.all_reduce_ema_stats(counts, dw, self.group)
        cb = self.codebook.detach()
        ops.ema_update(self.cluster_size, self.embed_sum, counts, dw, cb, self.decay, self.eps)
        # the kernel wrote through a raw pointer: tell torch, so that version-keyed caches (the tokenizer's prepared
        # codebook, autograd's saved-tensor checks) see the change
        torch.autograd.graph.increment_version(self.codebook)
        return counts
