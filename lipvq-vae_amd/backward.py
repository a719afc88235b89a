"""Backward orchestration: which HIP launches produce the 14 (LipVQ) / 13 (VQVAE) parameter
gradients that autograd derives from the reference forwards (backbone_lfqvae_v5.py:70-84,
backbone.py:38-76).  Only launches and buffer plumbing here; every number is computed by the
library (include/lipvq.h, "backward" section)."""
from __future__ import annotations

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID

_ENC_ACTS = (ACT_GELU, ACT_GELU, ACT_SIGMOID)
_DEC_ACTS = (ACT_GELU, ACT_GELU, ACT_NONE)
_RELU3 = (ACT_RELU, ACT_RELU, ACT_RELU)


def _stack_grads(x_in, pre, g2, g1, g0, acts, hidx=None):
    """Weight/bias gradients of one three-layer stack from the per-layer dL/d(pre-activation)."""
    gW2, gb2 = ops.wgrad(g2, pre[1], h_act=acts[1])
    gW1, gb1 = ops.wgrad(g1, pre[0], h_act=acts[0])
    gW0, gb0 = ops.wgrad(g0, x_in, h_act=ACT_NONE, hidx=hidx)
    return gW0, gb0, gW1, gb1, gW2, gb2


def llfq_backward(module, saved, g_loss):
    """Gradients in the order (enc0.w, enc0.b, enc2.w, enc2.b, to_latent.W, .b, .ci, codebook,
    dec0.w, dec0.b, dec2.w, dec2.b, to_output.w, to_output.b)."""
    x, z_e, z_q, idx, x_rec, Wn, scale, pe0, pe1, pe2, pd0, pd1, pd2 = saved
    N, A = x.shape
    D = z_e.shape[1]
    K = module.num_codes
    g = g_loss.detach().to(dtype=x.dtype).reshape(1).contiguous()
    cb = module.quantizer.codebook.detach()
    enc0, enc2 = module.encoder[0], module.encoder[2]
    dec0, dec2, outl = module.decoder[0], module.decoder[2], module.to_output

    # recon = mean((x_rec - x)^2)  ->  d/dx_rec
    g_xrec = ops.scaled_diff(x_rec, x, 2.0 / (N * A), gscale=g)
    # decoder + to_output (input = codebook[idx])
    pk, pk_enc = ops.mlp3_pack_bwd2((dec0.weight.detach(), dec2.weight.detach(), outl.weight.detach()),
                                    (enc0.weight.detach(), enc2.weight.detach(), Wn))          # both stacks, one launch
    alpha = 0.25 * 2.0 / (N * D)
    _, g1d, g0d, g_zq_dec = ops.mlp3_bwd(g_xrec, (pd0, pd1, None), pk, _DEC_ACTS, want_gx=True)
    gWd0, gbd0, gWd2, gbd2, gWo, gbo = _stack_grads(cb, (pd0, pd1), g_xrec, g1d, g0d, _DEC_ACTS, hidx=idx)
    # codebook: decoder path + 0.25 d mse(z_q, z_e.detach()); large batches form the rows inside the scatter (no [N, D] stream)
    g_cb = ops.scatter_add_vq(g_zq_dec, z_e, cb, idx, alpha, gscale=g, zq=z_q)
    # encoder side: 0.25 d mse(z_q.detach(), z_e) only (no straight-through estimator, v5:74-81)
    if ops.mlp3_bwd_vq_supported(N, pk_enc):
        # large batches: the encoder chain forms its own input gradient, alpha g (sigmoid(pe2) - codebook[idx])
        g2e, g1e, g0e, _ = ops.mlp3_bwd(None, (pe0, pe1, pe2), pk_enc, _ENC_ACTS, want_gx=False,
                                        in_term=(None, None, cb, idx, alpha), gscale=g)
    else:
        g_ze = ops.scaled_diff(z_e, z_q, alpha, gscale=g)
        g2e, g1e, g0e, _ = ops.mlp3_bwd(g_ze, (pe0, pe1, pe2), pk_enc, _ENC_ACTS, want_gx=False)
    gWe0, gbe0, gWe2, gbe2, gWn, gbl = _stack_grads(x, (pe0, pe1), g2e, g1e, g0e, _ENC_ACTS)
    gWl, gci = ops.lipschitz_bwd(module.to_latent.W.detach(), module.to_latent.ci.detach(), gWn)
    return (gWe0, gbe0, gWe2, gbe2, gWl, gbl, gci, g_cb, gWd0, gbd0, gWd2, gbd2, gWo, gbo)


def vq_backward(module, saved, g_loss):
    """Gradients in the order (enc0.w, enc0.b, enc2.w, enc2.b, enc4.w, enc4.b, dec0.w, dec0.b,
    dec2.w, dec2.b, dec4.w, dec4.b, embedding.weight)."""
    x, z_e, z_q, z_st, idx, x_rec, pe0, pe1, pe2, pd0, pd1, pd2 = saved
    N, A = x.shape
    D = z_e.shape[1]
    K = module.num_embeddings
    cc = float(module.commitment_cost)
    g = g_loss.detach().to(dtype=x.dtype).reshape(1).contiguous()
    e, d = module.encoder, module.decoder

    g_xrec = ops.scaled_diff(x_rec, x, 2.0 / (N * A), gscale=g)
    pk, pk_enc = ops.mlp3_pack_bwd2((d[0].weight.detach(), d[2].weight.detach(), d[4].weight.detach()),
                                    (e[0].weight.detach(), e[2].weight.detach(), e[4].weight.detach()))
    E = module.embedding.weight.detach()
    if ops.mlp3_bwd_vq_supported(N, pk):
        # z_e: straight-through decoder gradient + commitment term, stored by the decoder chain itself (z_q read as E[idx])
        g2d, g1d, g0d, g_ze = ops.mlp3_bwd(g_xrec, (pd0, pd1, pd2), pk, _RELU3, want_gx=True,
                                           out_term=(z_e, None, E, idx, cc * 2.0 / (N * D)), gscale=g)
    else:
        g2d, g1d, g0d, g_zst = ops.mlp3_bwd(g_xrec, (pd0, pd1, pd2), pk, _RELU3, want_gx=True)
        g_ze = ops.scaled_diff(z_e, z_q, cc * 2.0 / (N * D), gscale=g, c=g_zst)
    gWd0, gbd0, gWd2, gbd2, gWd4, gbd4 = _stack_grads(z_st, (pd0, pd1), g2d, g1d, g0d, _RELU3)
    # embedding loss mse(z_q, z_e.detach()) -> embedding rows only (the decoder sees z_e through the STE)
    g_emb = ops.scatter_add_vq(None, z_e, E, idx, 2.0 / (N * D), gscale=g, zq=z_q)
    g2e, g1e, g0e, _ = ops.mlp3_bwd(g_ze, (pe0, pe1, pe2), pk_enc, _RELU3, want_gx=False)
    gWe0, gbe0, gWe2, gbe2, gWe4, gbe4 = _stack_grads(x, (pe0, pe1), g2e, g1e, g0e, _RELU3)
    return (gWe0, gbe0, gWe2, gbe2, gWe4, gbe4, gWd0, gbd0, gWd2, gbd2, gWd4, gbd4, g_emb)
