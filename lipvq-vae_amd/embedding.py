"""The step right after the tokenizer: the reference's input embedding + interleave, on the HIP path.

Host-side mirror of the embedding stage of ``ICLTransformer`` (reference robomimic/models/obs_nets.py = "ob"):

* parameters ob:2425-2450 -- ``nets["embed_encoder"] = nn.Linear(input_dim, embed_dim)``, the time embedding
  (``params["embed_timestep"]`` nn.Parameter [1,T,E] | ``nets["embed_timestep"]`` nn.Embedding | sinusoidal),
  ``nets["embed_ln"] = nn.LayerNorm(embed_dim)``, ``nets["embed_drop"] = nn.Dropout(p)``.  The state_dict keys are the
  reference's (``nets.embed_encoder.weight`` ...), so a slice of its checkpoint loads strictly;
* ``input_embedding(inputs)`` ob:2525-2543;
* ``forward(obs, context_obs, context_actions | action_indices)`` ob:2580-2596: the [B, 3T, E] transformer input with
  context_obs at 2t, context_actions at 2t+1 and obs at 2T+t.

MI355X design (csrc/lipvq_embed.hip): each stream is written by one launch directly into its interleaved slots (no
stack/view/cat copies), and for the LipVQ tokenizer -- whose output rows ARE codebook rows (v5:47,84) -- the Linear
collapses into a [K, E] table rebuilt only when a parameter changes; the per-action work is a gather + LayerNorm driven
by the tokenizer's int64 indices, so z_latent never round-trips HBM.  Gradients (embed_encoder, time embedding,
LayerNorm, and the dense inputs; NOT the codebook: z_latent is detached in the reference, v5:74) come from the
library's backward kernels wrapped in torch.autograd.Functions.  Dropout stays torch's (identity in eval).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .tokenizer import _PackCache

__all__ = ["ICLInputEmbedding", "sinusoidal_table"]


def sinusoidal_table(T: int, E: int, device) -> torch.Tensor:
    """[T, E] rows of PositionalEncoding (reference robomimic/models/transformers.py:58-77) for timesteps 0..T-1."""
    t = torch.arange(T, dtype=torch.float32, device=device).unsqueeze(-1)
    div = torch.exp(torch.arange(0, E, 2, device=device) * (-math.log(10000.0) / E)).unsqueeze(0)
    pe = torch.zeros((T, E), device=device)          # built on `device`, as the reference builds it on its own
    pe[:, 0::2] = torch.sin(t * div)
    pe[:, 1::2] = torch.cos(t * div)
    return pe


class _LinearFn(torch.autograd.Function):
    """y = x . W^T + b on lipvq_linear_f32; backward = one more Linear (gx) + the wgrad kernel (gW, gb)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return ops.linear(x, W, b)

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gy = gy.contiguous()
        gx = ops.linear(gy, W.t().contiguous()) if ctx.needs_input_grad[0] else None
        gW = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gW, gb = ops.wgrad(gy, x)
        return gx, gW, gb


class _EmbedStreamsFn(torch.autograd.Function):
    """LayerNorm(src_s[idx_s | n] + pos[t]) for every stream s, each written into its slots of ONE output tensor.

    slots: tuple of (t_stride, offset) in floats per stream; the batch stride is the whole [S, E] block."""

    @staticmethod
    def forward(ctx, pos, ln_w, ln_b, eps, B, T, S, slots, idxs, *srcs):
        E = ln_w.numel()
        out = torch.empty((B, S, E), device=ln_w.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)
        stats = []
        for (tstride, offset), idx, src in zip(slots, idxs, srcs):
            stats.append(ops.embed_rows(src, idx, pos, ln_w, ln_b, eps, out, B * T, T, S * E, tstride, offset,
                                        want_stats=need))
        ctx.meta = (B, T, S, slots, idxs)
        ctx.save_for_backward(pos, ln_w, *[s for s in stats if s is not None], *srcs)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, T, S, slots, idxs = ctx.meta
        n = len(slots)
        pos, ln_w, *rest = ctx.saved_tensors
        stats, srcs = rest[:n], rest[n:]
        E = ln_w.numel()
        gout = gout.contiguous()
        g_pos = torch.zeros((T, E), device=gout.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        g_lnw = torch.zeros(E, device=gout.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        g_lnb = torch.zeros(E, device=gout.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        g_srcs = []
        for i, ((tstride, offset), idx, src, st) in enumerate(zip(slots, idxs, srcs, stats)):
            want = ctx.needs_input_grad[9 + i]
            g_src = (torch.zeros_like(src) if idx is not None else torch.empty_like(src)) if want else None
            ops.embed_rows_bwd(gout, src, idx, pos, st, ln_w, g_src, g_pos, g_lnw, g_lnb, B * T, T, S * E, tstride, offset)
            g_srcs.append(g_src)
        return (g_pos, g_lnw, g_lnb, None, None, None, None, None, None, *g_srcs)


class ICLInputEmbedding(nn.Module):
    """Embedding stage of the reference's ICLTransformer (ob:2425-2450, 2485-2543, 2580-2596) on the HIP path."""

    def __init__(self, input_dim, embed_dim=512, context_length=10, emb_dropout=0.1, sinusoidal_embedding=False,
                 nn_parameter_for_timesteps=True):
        super().__init__()
        if embed_dim % 4 or embed_dim > 1024:
            raise ValueError("embed_dim must be a multiple of 4 and <= 1024")
        self.input_dim, self.embed_dim, self.context_length = int(input_dim), int(embed_dim), int(context_length)
        self.sinusoidal_embedding = bool(sinusoidal_embedding)
        self.nn_parameter_for_timesteps = bool(nn_parameter_for_timesteps) and not self.sinusoidal_embedding
        self.nets = nn.ModuleDict()
        self.params = nn.ParameterDict()
        self.nets["embed_encoder"] = nn.Linear(input_dim, embed_dim)                       # ob:2425-2427
        if self.sinusoidal_embedding:                                                       # ob:2431-2432
            pass
        elif self.nn_parameter_for_timesteps:                                               # ob:2433-2439
            self.params["embed_timestep"] = nn.Parameter(torch.zeros(1, context_length, embed_dim))
        else:                                                                               # ob:2441-2443
            self.nets["embed_timestep"] = nn.Embedding(context_length, embed_dim)
        self.nets["embed_ln"] = nn.LayerNorm(embed_dim)                                     # ob:2447
        self.nets["embed_drop"] = nn.Dropout(emb_dropout)                                   # ob:2450
        self._table_cache = _PackCache()
        self._sin_cache = {}

    # -- pieces ------------------------------------------------------------------------------------
    def time_table(self, T: int) -> torch.Tensor:
        """[T, E] rows added to timesteps 0..T-1 (ob:2485-2523)."""
        if self.sinusoidal_embedding:
            dev = self.nets["embed_encoder"].weight.device
            key = (T, dev)
            if key not in self._sin_cache:
                self._sin_cache[key] = sinusoidal_table(T, self.embed_dim, dev)
            return self._sin_cache[key]
        if self.nn_parameter_for_timesteps:
            if T != self.context_length:                    # the reference's broadcast add fails the same way
                raise ValueError(f"sequence length {T} != context_length {self.context_length} (nn.Parameter time embedding)")
            return self.params["embed_timestep"][0]
        if T > self.context_length:
            raise IndexError(f"timestep {T - 1} out of range of the time embedding ({self.context_length})")
        return self.nets["embed_timestep"].weight[:T]

    def code_table(self, codebook: torch.Tensor) -> torch.Tensor:
        """[K, E] = embed_encoder(codebook): what the Linear yields for every possible tokenizer output row."""
        lin = self.nets["embed_encoder"]
        codebook = codebook.detach()                        # z_latent is detached in the reference (v5:74)
        if torch.is_grad_enabled() and (lin.weight.requires_grad or lin.bias.requires_grad):
            return _LinearFn.apply(codebook, lin.weight, lin.bias)
        return self._table_cache.get((codebook, lin.weight, lin.bias),
                                     lambda: ops.linear(codebook, lin.weight.detach(), lin.bias.detach()))

    def _dense(self, x3: torch.Tensor) -> torch.Tensor:
        lin = self.nets["embed_encoder"]
        B, T, Din = x3.shape
        if Din != self.input_dim:
            raise ValueError(f"expected inputs [..., {self.input_dim}], got {tuple(x3.shape)}")
        return _LinearFn.apply(x3.reshape(B * T, Din), lin.weight, lin.bias)

    def _embed(self, B, T, S, slots, idxs, srcs):
        ln = self.nets["embed_ln"]
        out = _EmbedStreamsFn.apply(self.time_table(T), ln.weight, ln.bias, ln.eps, B, T, S, tuple(slots), tuple(idxs),
                                    *srcs)
        return self.nets["embed_drop"](out)                                                # ob:2541

    # -- the reference's entry points -----------------------------------------------------------------
    def input_embedding(self, inputs: torch.Tensor) -> torch.Tensor:
        """ob:2525-2543 on one stream: [B, T, input_dim] -> [B, T, E]."""
        B, T, _ = inputs.shape
        E = self.embed_dim
        return self._embed(B, T, T, [(E, 0)], [None], [self._dense(inputs)])

    def input_embedding_tokens(self, indices: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
        """input_embedding(codebook[indices]) without materialising codebook[indices]: indices [B, T] int64."""
        B, T = indices.shape
        E = self.embed_dim
        return self._embed(B, T, T, [(E, 0)], [indices.reshape(-1)], [self.code_table(codebook)])

    def forward(self, obs, context_obs, context_actions=None, *, action_indices=None, codebook=None):
        """ob:2580-2596: the [B, 3T, E] transformer input.  The context actions come either as dense rows
        ``context_actions`` [B, T, input_dim] (any tokenizer) or as ``action_indices`` [B, T] + ``codebook`` (LipVQ)."""
        B, T, _ = obs.shape
        E = self.embed_dim
        if context_obs.shape != obs.shape:
            raise ValueError("obs and context_obs must have the same shape")
        if (action_indices is None) == (context_actions is None):
            raise ValueError("pass exactly one of context_actions / action_indices")
        if action_indices is not None:
            if codebook is None:
                raise ValueError("action_indices needs the codebook")
            if tuple(action_indices.shape) != (B, T):
                raise ValueError(f"action_indices must be [{B}, {T}]")
            a_idx, a_src = action_indices.reshape(-1), self.code_table(codebook)
        else:
            if context_actions.shape != obs.shape:
                raise ValueError("context_actions must have the same shape as obs")
            a_idx, a_src = None, self._dense(context_actions)
        slots = [(2 * E, 0), (2 * E, E), (E, 2 * T * E)]           # context_obs, context_actions, obs
        return self._embed(B, T, 3 * T, slots, [None, a_idx, None], [self._dense(context_obs), a_src, self._dense(obs)])
