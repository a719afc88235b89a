"""The DEFAULT action branch of the group encoder -- what ``ICLObservationGroupEncoder`` builds when none of
``fast_enabled / bin_enabled / vq_vae_enabled / ln_act_enabled`` is set (reference robomimic/models/obs_nets.py:1244-1260)
and calls as ``context_actions = self.action_network(prompt_actions)`` (obs_nets.py:1343-1344) -- on the HIP library.

``DefaultActionNetwork`` IS an ``nn.Sequential`` with the reference's children in the reference's order, so parameter names
(``0.weight_orig``, ``0.weight_u``, ``5.layers.2.self_attn.in_proj_weight``, ``6.bias`` ...), shapes, initial values and RNG
consumption are the reference's by construction and its checkpoints load unchanged.  The children are parameter CONTAINERS
only: ``forward`` never calls them.  The compute is
    3 x lipvq_spectral_norm_f32            (power iteration in training mode, writes weight_u / weight_v like torch's hook)
    lipvq_linear_act_f32 x 3               (spectral-norm MLP, GELU epilogues)
    4 x [in_proj Linear -> lipvq_attention_f32 -> out_proj Linear -> lipvq_add_layernorm_f32
         -> Linear+GELU -> Linear -> lipvq_add_layernorm_f32]                (post-norm encoder layers)
    lipvq_linear_act_f32                   (the closing Linear)
with backward kernels behind ``torch.autograd.Function``s.  The input is 2-D ``[B*T, A]``, which ``nn.TransformerEncoder``
treats as ONE unbatched sequence: every action of the batch attends to every other (S = B*T; the reference's behaviour,
kept).  Dropout (p = 0.1, four places per layer) follows torch's semantics with torch's own RNG -- elementwise ones through
``F.dropout``, the attention-probability one as a keep-mask handed to the kernel -- so training-mode outputs match the
reference in distribution, eval-mode outputs to 1e-5 (tests/test_gpu_default.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import ops
from .ops import ACT_GELU, ACT_NONE


class _SpectralFn(torch.autograd.Function):
    """weight_orig -> weight_orig / sigma (u, v: buffers, updated in place in training mode, constants for the gradient)."""

    @staticmethod
    def forward(ctx, W, u, v, training, eps):
        Wsn, sigma = ops.spectral_norm(W.detach(), u, v, training, eps)
        ctx.save_for_backward(Wsn, u.clone(), v.clone(), sigma)          # torch clones u, v after the iteration as well
        return Wsn

    @staticmethod
    def backward(ctx, gWsn):
        Wsn, u, v, sigma = ctx.saved_tensors
        return ops.spectral_norm_bwd(gWsn.contiguous(), Wsn, u, v, sigma), None, None, None, None


class _LinearActFn(torch.autograd.Function):
    """act(x W^T + b); backward = act' (elementwise), one more Linear (gx), the wgrad kernel (gW, gb)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        ctx.act = act
        if act != ACT_NONE:
            y, pre = ops.linear(x, W, b, act=act, save_pre=True)
            ctx.save_for_backward(x, W, pre)
        else:
            y = ops.linear(x, W, b)
            ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        if ctx.act != ACT_NONE:
            x, W, pre = ctx.saved_tensors
            gy = ops.act_bwd(gy, pre, ctx.act)
        else:
            x, W = ctx.saved_tensors
        gx = ops.linear(gy, W.t().contiguous()) if ctx.needs_input_grad[0] else None
        gW = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gW, gb = ops.wgrad(gy, x)
        return gx, gW, gb, None


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, nhead, keep, keep_prob):
        out, lse = ops.attention(qkv, nhead, keep, keep_prob)
        ctx.nhead, ctx.keep, ctx.keep_prob = nhead, keep, keep_prob
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse = ctx.saved_tensors
        return ops.attention_bwd(qkv, out, gout.contiguous(), lse, ctx.nhead, ctx.keep, ctx.keep_prob), None, None, None


class _AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, w, bias, eps):
        y, xhat, rstd = ops.add_layernorm(a, b, w, bias, eps, save=True)
        ctx.save_for_backward(xhat, rstd, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        xhat, rstd, w = ctx.saved_tensors
        gx, gw, gb = ops.layernorm_bwd(gy.contiguous(), xhat, rstd, w)
        return gx, gx, gw, gb, None


class DefaultActionNetwork(nn.Sequential):
    """Drop-in for the reference's default ``action_network`` (obs_nets.py:1244-1260)."""

    NHEAD, FF = 8, 256

    def __init__(self, action_input_shape: int, action_output_shape: int, num_layers: int = 4):
        D = int(action_output_shape)
        if D % self.NHEAD != 0 or D // self.NHEAD > 32 or D > 256:
            raise ValueError(f"DefaultActionNetwork: action_output_shape={D} must be a multiple of {self.NHEAD} and <= 256")
        transformer_layer = nn.TransformerEncoderLayer(d_model=D, nhead=self.NHEAD, dim_feedforward=self.FF, activation="gelu")
        super().__init__(
            spectral_norm(nn.Linear(action_input_shape, 64)),
            nn.GELU(),
            spectral_norm(nn.Linear(64, 128)),
            nn.GELU(),
            spectral_norm(nn.Linear(128, D)),
            nn.TransformerEncoder(transformer_layer, num_layers=num_layers),
            nn.Linear(D, D),
        )
        self.latent_dim = D

    def _sn_weight(self, lin: nn.Linear) -> torch.Tensor:
        # torch's hook iterates only in training mode (spectral_norm.py: do_power_iteration=module.training)
        return _SpectralFn.apply(lin.weight_orig, lin.weight_u, lin.weight_v, self.training, 1e-12)

    def forward(self, prompt_actions: torch.Tensor) -> torch.Tensor:
        x = prompt_actions
        if x.dim() != 2:
            raise ValueError(f"DefaultActionNetwork expects the flattened [B*T, A] action tensor, got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("DefaultActionNetwork runs on the HIP library only (no CPU path)")
        x = x.contiguous().float()
        h = _LinearActFn.apply(x, self._sn_weight(self[0]), self[0].bias, ACT_GELU)
        h = _LinearActFn.apply(h, self._sn_weight(self[2]), self[2].bias, ACT_GELU)
        h = _LinearActFn.apply(h, self._sn_weight(self[4]), self[4].bias, ACT_NONE)
        S = h.shape[0]
        for layer in self[5].layers:
            attn = layer.self_attn
            p = float(attn.dropout) if self.training else 0.0
            keep, keep_prob = None, 1.0
            if p > 0.0:
                keep = (torch.rand((self.NHEAD, S, S), device=h.device) >= p).to(torch.uint8)
                keep_prob = 1.0 - p
            qkv = _LinearActFn.apply(h, attn.in_proj_weight, attn.in_proj_bias, ACT_NONE)
            a = _AttentionFn.apply(qkv, self.NHEAD, keep, keep_prob)
            a = _LinearActFn.apply(a, attn.out_proj.weight, attn.out_proj.bias, ACT_NONE)
            a = F.dropout(a, layer.dropout1.p, self.training)
            h = _AddLayerNormFn.apply(h, a, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            f = _LinearActFn.apply(h, layer.linear1.weight, layer.linear1.bias, ACT_GELU)
            f = F.dropout(f, layer.dropout.p, self.training)
            f = _LinearActFn.apply(f, layer.linear2.weight, layer.linear2.bias, ACT_NONE)
            f = F.dropout(f, layer.dropout2.p, self.training)
            h = _AddLayerNormFn.apply(h, f, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        return _LinearActFn.apply(h, self[6].weight, self[6].bias, ACT_NONE)


class GraphedDefaultBranch:
    """Eval-mode forward of a DefaultActionNetwork captured in ONE HIP graph for a fixed [N, A] shape.

    The branch is ~40 small launches (at the ICRT step shape N = 80 every one of them is microseconds of GPU work), so an
    eager call is bound by Python + ctypes issue (~1.5 ms); a graph replay costs the GPU time alone.  Rollouts call the
    action branch once per environment step with the same prompt shape (obs_nets.py:1343-1344 under algo.py:736 set_eval()),
    which is what this serves.  Parameters are read at replay time through their storage, so in-place updates
    (optimizer steps, load_state_dict) are seen; re-capture after anything that REPLACES a parameter tensor (.to(), .cuda())."""

    def __init__(self, net: DefaultActionNetwork, example_actions: torch.Tensor):
        if net.training:
            raise RuntimeError("GraphedDefaultBranch captures the eval-mode forward: call net.eval() first")
        self.net = net
        self._x = example_actions.detach().contiguous().float().clone()
        side = torch.cuda.Stream(device=self._x.device)
        side.wait_stream(torch.cuda.current_stream(self._x.device))
        with torch.no_grad(), torch.cuda.stream(side):            # first-use work (LDS reservations, lazy module state) stays out of the capture
            for _ in range(3):
                net(self._x)
        torch.cuda.current_stream(self._x.device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._y = net(self._x)

    def __call__(self, prompt_actions: torch.Tensor) -> torch.Tensor:
        if prompt_actions.shape != self._x.shape:
            raise ValueError(f"captured for {tuple(self._x.shape)}, got {tuple(prompt_actions.shape)}")
        self._x.copy_(prompt_actions)
        self._graph.replay()
        return self._y
