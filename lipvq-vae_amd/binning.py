"""AdaptiveBinActionEmbedding on the HIP path: the sibling tokenizer behind the reference's ``bin_enabled`` switch.

Host-side mirror of reference robomimic/models/bin_action/backbone.py:5-89 ("bin"), selected at
robomimic/models/obs_nets.py:1214-1217 and called like the LipVQ tokenizer (obs_nets.py:1330-1333, minus the loss):
same constructor arguments, same ``state_dict`` keys (``running_min``, ``running_max``, ``embedding_layers.{i}.weight``,
``output_layer.{0,2}.{weight,bias}``), same RNG consumption at construction, same methods (``update_running_stats``,
``compute_bins``, ``discretize``, ``forward``).

MI355X design (csrc/lipvq_bin.hip): the first Linear acts on a concatenation of per-dimension embedding rows, so it is
evaluated as ``b1 + sum_i P[i][bin_i]`` with ``P[i] = emb_i . W1[:, 64i:64i+64]^T`` ([A, num_bins, H], rebuilt only
when a parameter changes): A gathered adds per hidden unit instead of a [N, 64A] x [64A, 32A] GEMM, and the
``[N, 64A]`` concatenation never exists.  Bin indices are bit-exact (torch's linspace rounding and lower-bound search
restated in lipvq_math.h); floats agree with the reference to 1e-5.  Gradients come from the library's backward
kernels behind one torch.autograd.Function.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU
from .tokenizer import _PackCache

__all__ = ["AdaptiveBinActionEmbedding"]


def _code_table(embs, W1, ed):
    """P [A, nb, H]: row (i, bin) = W1[:, ed*i : ed*(i+1)] . emb_i[bin]  (canonical chains from 0)."""
    return torch.stack([ops.linear(e, W1[:, ed * i:ed * (i + 1)].contiguous()) for i, e in enumerate(embs)])


class _BinFn(torch.autograd.Function):
    """out = gelu(L2(gelu(b1 + sum_i P[i][bins[i]]))) with gradients to the embeddings and both Linears."""

    @staticmethod
    def forward(ctx, bins, ed, W1, b1, W2, b2, *embs):
        P = _code_table(embs, W1, ed)
        need = any(ctx.needs_input_grad)
        if need:
            h, pre1 = ops.bin_hidden(bins, P, b1, save_pre=True)
            out, pre2 = ops.linear(h, W2, b2, act=ACT_GELU, save_pre=True)
            ctx.save_for_backward(bins, W1, W2, pre1, pre2, *embs)
            ctx.ed = ed
        else:
            out = ops.linear(ops.bin_hidden(bins, P, b1), W2, b2, act=ACT_GELU)
        return out

    @staticmethod
    def backward(ctx, gy):
        bins, W1, W2, pre1, pre2, *embs = ctx.saved_tensors
        ed = ctx.ed
        nb = embs[0].shape[0]
        g2 = ops.act_bwd(gy.contiguous(), pre2, ACT_GELU)
        gW2, gb2 = ops.wgrad(g2, pre1, h_act=ACT_GELU)                 # layer input = gelu(pre1)
        g1 = ops.act_bwd(ops.linear(g2, W2.t().contiguous()), pre1, ACT_GELU)
        gW1 = torch.empty_like(W1)
        gb1 = None
        g_embs = []
        for i, emb in enumerate(embs):
            gP = ops.scatter_add(g1, bins[i], nb)                        # [nb, H]: rows of g1 by this dimension's bin
            blk = W1[:, ed * i:ed * (i + 1)]
            gWi, gb = ops.wgrad(gP, emb)                                 # gP^T . emb_i  and  sum over bins = sum over rows
            gW1[:, ed * i:ed * (i + 1)] = gWi
            if gb1 is None:
                gb1 = gb
            g_embs.append(ops.linear(gP, blk.t().contiguous()))          # gP . W1_blk
        return (None, None, gW1, gb1, gW2, gb2, *g_embs)


class AdaptiveBinActionEmbedding(nn.Module):
    """Drop-in for the reference class of the same name (bin:5-89)."""

    def __init__(self, action_dim, output_dim, num_bins=20, embedding_dim=64, num_step_stop=10000):
        super().__init__()
        self.action_dim = action_dim
        self.num_bins = num_bins
        self.embedding_dim = embedding_dim
        self.register_buffer("running_min", torch.full((action_dim,), float("inf")))
        self.register_buffer("running_max", torch.full((action_dim,), float("-inf")))
        self.embedding_layers = nn.ModuleList(
            [nn.Embedding(num_embeddings=num_bins, embedding_dim=embedding_dim) for _ in range(action_dim)])
        self.output_layer = nn.Sequential(
            nn.Linear(embedding_dim * action_dim, embedding_dim * action_dim // 2),
            nn.GELU(),
            nn.Linear(embedding_dim * action_dim // 2, output_dim),
            nn.GELU(),
        )
        self._num_step = 0
        self._num_step_stop = num_step_stop
        self._update_enabled = True
        self._table_cache = _PackCache()
        self.last_bins = None                  # [A, N] int64 of the most recent forward

    # -- the reference's methods ---------------------------------------------------------------------
    def update_running_stats(self, actions):
        """bin:37-40 (in place on the two buffers)."""
        ops.bin_minmax(self._rows(actions), self.running_min, self.running_max)

    def compute_bins(self):
        """bin:42-53: list of A boundary tensors [num_bins + 1]."""
        return list(ops.bin_boundaries(self.running_min, self.running_max, self.num_bins).unbind(0))

    def discretize(self, actions):
        """bin:55-66: [N, A] int64 bin indices."""
        return ops.bin_discretize(self._rows(actions), self.running_min, self.running_max, self.num_bins).t()

    def _rows(self, actions):
        if actions.dim() != 2 or actions.shape[1] != self.action_dim:
            raise ValueError(f"expected actions [N, {self.action_dim}], got {tuple(actions.shape)}")
        return actions

    def forward(self, actions):
        actions = self._rows(actions)
        if self._update_enabled:                                          # bin:70-74
            self.update_running_stats(actions.detach())
            self._num_step += 1
            if self._num_step >= self._num_step_stop:
                self._update_enabled = False
        bins = ops.bin_discretize(actions.detach(), self.running_min, self.running_max, self.num_bins)
        self.last_bins = bins
        l1, l2 = self.output_layer[0], self.output_layer[2]
        embs = [e.weight for e in self.embedding_layers]
        params = (l1.weight, l1.bias, l2.weight, l2.bias, *embs)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _BinFn.apply(bins, self.embedding_dim, *params)
        with torch.no_grad():
            P = self._table_cache.get((l1.weight, *embs), lambda: _code_table([e.detach() for e in embs], l1.weight.detach(),
                                                                              self.embedding_dim))
            h = ops.bin_hidden(bins, P, l1.bias.detach())
            return ops.linear(h, l2.weight.detach(), l2.bias.detach(), act=ACT_GELU)
