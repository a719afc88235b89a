"""torch.autograd.Function wrappers: the reference's forward()/backward() contract on the HIP path.

Forward value flow (LLFQVAE_V4, reference backbone_lfqvae_v5.py:70-84):
    x --mlp3(gelu,gelu,sigmoid; W2 = Lipschitz-normalised)--> z_e --nearest--> (idx, z_q)
    z_q --mlp3(gelu,gelu,none; input gathered from the codebook)--> x_recon
    loss = mse(x_recon,x) + 0.25*mse(z_q.detach(),z_e) + 0.25*mse(z_q,z_e.detach())
Gradient routing (no straight-through estimator in this variant):
    recon     -> to_output, decoder, codebook (through the gather)
    codebook  -> codebook
    commit    -> to_latent.{W,b,ci}, encoder
The two latent mse terms have the same VALUE; the library computes it once.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, DIST_NORM, DIST_SQSUM

_ENC_ACTS = (ACT_GELU, ACT_GELU, ACT_SIGMOID)
_DEC_ACTS = (ACT_GELU, ACT_GELU, ACT_NONE)
_RELU3 = (ACT_RELU, ACT_RELU, ACT_RELU)


def _check_versions(ctx, params):
    """The backward kernels read the module's LIVE weights; refuse to run if one was modified in place since the forward."""
    saved = getattr(ctx, "param_versions", None)
    if saved is None:
        return
    for i, (p, v) in enumerate(zip(params, saved)):
        if p._version != v:
            raise RuntimeError(f"lipvq: parameter #{i} (shape {tuple(p.shape)}) was modified in place between forward and "
                               f"backward (version {v} -> {p._version}); the gradient would be computed with the new weights")


class _LLFQFn(torch.autograd.Function):
    """(z_latent, loss) = f(x, 14 parameters).  z_latent is non-differentiable (v5:74)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        # (the engine would otherwise hand backward() a materialised [N, D] zero tensor for the non-differentiable z_latent: a 134 MB
        # fill per step at the metric's batch)
        ctx.set_materialize_grads(False)
        enc_packed, scale, Wn = module._packed_encoder()
        dec_packed = module._packed_decoder()
        codebook = module.quantizer.codebook.detach()
        need_grad = any(ctx.needs_input_grad[2:])
        screen = module.screen_pays(x.shape[0])          # one routing decision per call (tokenizer._ScreenMonitor)
        fused_ok = x.shape[0] > module.EXACT_ROWS_MAX and screen and module.fused_shape()
        if need_grad and fused_ok:
            # large training batches: encoder + quantizer + everything the backward needs in ONE launch (lipvq_tokenize_train_f32)
            # instead of mlp3 (saved pre-activations) + the stand-alone screen over z_e
            idx, z_q, z_e, pre_e = module._tokenize_fused(x, module.code_usage, want_pre=True)
        elif need_grad:
            z_e, pre_e = ops.mlp3(x, enc_packed, _ENC_ACTS, save_pre=True)
            idx, z_q = module._quantize(z_e, module.code_usage, screen=screen)
        elif fused_ok:
            # no autograd (rollouts under no_grad, bulk evaluation): the fused encode + quantize launch, z_e written for the loss
            idx, z_q, z_e = module._tokenize_fused(x, module.code_usage, want_ze=True)
            pre_e = None
        else:
            z_e, pre_e = ops.mlp3(x, enc_packed, _ENC_ACTS), None
            idx, z_q = module._quantize(z_e, module.code_usage, screen=screen)
        # recon + 0.25*commit + 0.25*codebook, left to right (v5:83): (m0 + q) + q with q = 0.25 m1, evaluated on the device
        if ops.mlp3_loss_supported(x.shape[0], dec_packed):
            # large batches: the decoder launch sums both squared errors itself (its input rows ARE z_q, its output x_rec)
            x_rec, pre_d, l3 = ops.mlp3_loss(codebook, dec_packed, _DEC_ACTS, idx, x, z_e, 0.25, ops.LOSS_LLFQ, save_pre=need_grad)
            loss = l3[2]
        else:
            if need_grad:
                x_rec, pre_d = ops.mlp3(codebook, dec_packed, _DEC_ACTS, gather_idx=idx, save_pre=True)
            else:
                x_rec, pre_d = ops.mlp3(codebook, dec_packed, _DEC_ACTS, gather_idx=idx), None
            loss = ops.mse_pair_loss(x_rec, x, z_q, z_e, 0.25, ops.LOSS_LLFQ)[2]
        module.last_indices = idx
        ctx.module = module
        if need_grad:
            ctx.save_for_backward(x, z_e, z_q, idx, x_rec, Wn, scale, *pre_e, *pre_d)
            # backward() re-reads the module's live weights (decoder, codebook, encoder, to_latent): remember their versions, so
            # that a parameter update between forward and backward raises, as torch's saved-tensor check would
            ctx.param_versions = tuple(p._version for p in params)
        ctx.mark_non_differentiable(z_q)
        return z_q, loss

    @staticmethod
    def backward(ctx, _g_latent, g_loss):
        from .backward import llfq_backward
        m = ctx.module
        _check_versions(ctx, (*m._enc_params(), m.quantizer.codebook, *m._dec_params()))
        if g_loss is None:                               # the loss took no part in what is being differentiated
            return (None,) * len(ctx.needs_input_grad)
        grads = llfq_backward(m, ctx.saved_tensors, g_loss)
        return (None, None) + tuple(grads)


def llfq_forward(module, x):
    params = (*module._enc_params(), module.quantizer.codebook, *module._dec_params())
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        z_latent, loss = _LLFQFn.apply(module, x, *params)
    else:
        with torch.no_grad():
            z_latent, loss = _LLFQFn.forward(_NoCtx(len(params)), module, x, *params)
    return z_latent, loss


class _NoCtx:
    """Stand-in for the autograd context on the no-grad path (rollouts, tokenisation)."""

    def __init__(self, n):
        self.needs_input_grad = (False,) * (n + 2)

    def save_for_backward(self, *a):
        pass

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, flag):
        pass


class _VQFn(torch.autograd.Function):
    """(z_latent, loss) of the plain VQVAE (reference backbone.py:38-76)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        ctx.set_materialize_grads(False)                 # (see _LLFQFn.forward)
        enc_packed = module._packed_encoder()
        dec_packed = module._packed_decoder()
        E = module.embedding.weight.detach()
        need_grad = any(ctx.needs_input_grad[2:])
        big = x.shape[0] > module.EXACT_ROWS_MAX and module.num_embeddings >= module.FUSED_MIN_CODES
        screen = module._screen_monitor.use_screen() if big else None          # one routing decision per call
        if need_grad and big and screen and module.fused_shape():
            # large training batches: encoder + quantizer + the saved pre-activations in ONE launch (lipvq_vq_tokenize_train_f32)
            idx, z_q, z_e, pre_e = module._tokenize_fused(x, module.code_usage, want_pre=True)
        elif need_grad:
            z_e, pre_e = ops.mlp3(x, enc_packed, _RELU3, save_pre=True)
            idx, z_q = module._quantize(z_e, module.code_usage, screen=screen)  # vq:57-66 (screened / exact rows / all-pairs: same results)
        elif big and screen and module.fused_shape():
            # no autograd (rollouts, bulk evaluation): encoder + quantizer in ONE launch (lipvq_vq_tokenize_f32)
            idx, z_q, z_e = module._tokenize_fused(x, module.code_usage)
            pre_e = None
        else:
            z_e, pre_e = ops.mlp3(x, enc_packed, _RELU3), None
            idx, z_q = module._quantize(z_e, module.code_usage, screen=screen)
        # q_loss = m1 + commitment_cost m1 (vq:69-71); loss = m0 + q_loss (vq:50-51): evaluated on the device
        if ops.mlp3_loss_supported(x.shape[0], dec_packed):
            # large batches: the decoder launch forms z_st = z_e + (z_q - z_e) (vq:74) from E[idx] and z_e itself, runs on it, stores
            # it, and sums both squared errors -- no separate straight-through and mse passes over the [N, D] operands
            x_rec, pre_d, l3, z_st = ops.mlp3_loss(E, dec_packed, _RELU3, idx, x, z_e, float(module.commitment_cost), ops.LOSS_VQ,
                                                   save_pre=need_grad, ste=True)
            loss = l3[2]
        else:
            z_st = ops.ste(z_e, z_q)                                   # vq:74
            if need_grad:
                x_rec, pre_d = ops.mlp3(z_st, dec_packed, _RELU3, save_pre=True)
            else:
                x_rec, pre_d = ops.mlp3(z_st, dec_packed, _RELU3), None
            loss = ops.mse_pair_loss(x_rec, x, z_q, z_e, float(module.commitment_cost), ops.LOSS_VQ)[2]
        module.last_indices = idx
        ctx.module = module
        if need_grad:
            ctx.save_for_backward(x, z_e, z_q, z_st, idx, x_rec, *pre_e, *pre_d)
            ctx.param_versions = tuple(p._version for p in params)
        ctx.mark_non_differentiable(z_st)
        return z_st, loss

    @staticmethod
    def backward(ctx, _g_latent, g_loss):
        from .backward import vq_backward
        m = ctx.module
        _check_versions(ctx, (*m._enc_params(), *m._dec_params(), m.embedding.weight))
        if g_loss is None:
            return (None,) * len(ctx.needs_input_grad)
        grads = vq_backward(m, ctx.saved_tensors, g_loss)
        return (None, None) + tuple(grads)


def vq_forward(module, x):
    params = (*module._enc_params(), *module._dec_params(), module.embedding.weight)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _VQFn.apply(module, x, *params)
    with torch.no_grad():
        return _VQFn.forward(_NoCtx(len(params)), module, x, *params)


# ---------------------------------------------------------------------------------------------------
# engine-free forward + backward (used under HIP-graph capture, where the autograd engine's per-parameter
# AccumulateGrad nodes -- bound to whatever stream first produced a gradient -- must stay out of the way)
# ---------------------------------------------------------------------------------------------------

class _ManualCtx:
    def __init__(self, n):
        self.needs_input_grad = (False, False) + (True,) * n
        self.saved_tensors = ()
        self.module = None
        self.param_versions = None

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, flag):
        pass


def forward_backward(module, x):
    """(z_latent, loss, params, grads) with grads = dloss/dparams computed by the library's backward kernels,
    without touching torch.autograd.  params are in the order of the returned gradients."""
    from .backward import llfq_backward, vq_backward
    from .tokenizer import LLFQVAE_V4
    with torch.no_grad():
        if isinstance(module, LLFQVAE_V4):
            params = (*module._enc_params(), module.quantizer.codebook, *module._dec_params())
            ctx = _ManualCtx(len(params))
            z, loss = _LLFQFn.forward(ctx, module, x, *params)
            grads = llfq_backward(module, ctx.saved_tensors, torch.ones((), device=x.device, dtype=x.dtype))
        else:
            params = (*module._enc_params(), *module._dec_params(), module.embedding.weight)
            ctx = _ManualCtx(len(params))
            z, loss = _VQFn.forward(ctx, module, x, *params)
            grads = vq_backward(module, ctx.saved_tensors, torch.ones((), device=x.device, dtype=x.dtype))
    return z, loss, params, grads
