"""Multi-GPU use of the tokenizer: one process per GPU, rows sharded, parameters replicated.

The path is row-independent given the parameters (every op of reference backbone_lfqvae_v5.py:70-76 is
row-wise), so a batch [B, T, A] is split over the ranks along B with NO data-path collective.  What does
cross ranks (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests):

  * ``all_reduce_usage``      the per-code usage histogram [K] int64 -- 8 KiB at K=1024: latency bound,
                              one small all-reduce per batch;
  * ``all_reduce_gradients``  data-parallel training: the reference's losses are means over the GLOBAL
                              row count (v5:79-81), so each rank's gradients are weighted by
                              n_local / n_global and summed in ONE flat fp32 buffer (14 tensors,
                              ~375 KiB at config 2) before AdamW.step(); replicas stay identical.

  * ``all_reduce_ema_stats``  opt-in EMA extension: int64 counts [K] + fp32 per-code sums [K, D].

Nothing here computes on the data: it only shards, flattens and calls torch.distributed.
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous share of n items; the first n % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(actions: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's sequences of a [B, T, A] batch, flattened to [B_local * T, A] the way the reference
    flattens before the tokenizer (robomimic/utils/tensor_utils.py:1066-1067)."""
    if actions.dim() != 3:
        raise ValueError(f"expected [B, T, A], got {tuple(actions.shape)}")
    s, e = shard_bounds(actions.shape[0], rank, world)
    return actions[s:e].reshape(-1, actions.shape[2])


def all_reduce_usage(usage: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the code-usage histogram over the ranks, in place."""
    if usage.dtype != torch.int64:
        raise TypeError("usage histogram must be int64")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(usage, op=dist.ReduceOp.SUM, group=group)
    return usage


def all_reduce_ema_stats(counts: torch.Tensor, dw: torch.Tensor, group=None) -> None:
    """Sum the per-batch EMA statistics over the ranks, in place: counts [K] int64 and dw [K, D] fp32 (opt-in EMA
    extension, lipvq_vae_amd.ema; 8 KiB + 256 KiB at K = 1024, D = 64)."""
    if counts.dtype != torch.int64 or dw.dtype != torch.float32:
        raise TypeError("EMA statistics must be int64 counts and fp32 sums")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(dw, op=dist.ReduceOp.SUM, group=group)


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], n_local: int, n_global: int, group=None) -> None:
    """Replace every .grad by the gradient of the GLOBAL-mean loss: sum_r (n_r / n_global) * grad_r.
    One flat buffer, one all-reduce."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    w = float(n_local) / float(n_global)
    flat = torch.cat([p.grad.reshape(-1) for p in ps]).mul_(w)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


class ShardedTokenizer:
    """Tokenise this rank's share of a batch and keep the global code-usage statistics.

    ``tokenizer`` is anything with ``tokenize(x[N,A]) -> (indices[N], z_latent[N,D])`` and a
    ``code_usage`` int64 tensor -- normally ``lipvq_vae_amd.tokenizer.LLFQVAE_V4`` on this rank's GPU.
    """

    def __init__(self, tokenizer, group=None):
        self.tokenizer = tokenizer
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def tokenize(self, actions: torch.Tensor, reduce_usage: bool = True):
        """actions [B, T, A] (the full batch, or this rank's own [B_local, T, A] with pre_sharded=True
        semantics left to the caller) -> (indices [B_local, T], z_latent [B_local, T, D]) of this rank's rows."""
        s, e = shard_bounds(actions.shape[0], self.rank, self.world)
        local = actions[s:e]
        b, t = local.shape[0], local.shape[1]
        idx, z = self.tokenizer.tokenize(local.reshape(b * t, -1))
        if reduce_usage:
            all_reduce_usage(self.tokenizer.code_usage, self.group)
        return idx.reshape(b, t), z.reshape(b, t, -1)
