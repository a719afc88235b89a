"""Host glue around the tokenizer, restating the few lines of the reference that drive it.

* ``ICLActionBranch``      -- the action branch of ``ICLObservationGroupEncoder`` when ``vq_vae_enabled``
                             (robomimic/models/obs_nets.py:1219-1227 construction, :1335-1337 call) or
                             ``bin_enabled`` (:1214-1217, :1343-1344): owns ``action_network``, stashes ``_vq_vae_loss``.
* ``time_distributed``     -- the [B, T, ...] <-> [B*T, ...] reshape of
                             ``TensorUtils.icl_time_distributed`` (robomimic/utils/tensor_utils.py:1045-1090)
                             for the action leaf.
* ``VQTokenizerTrainer``   -- the optimiser choreography of ``ICLTransformer_GMM``
                             (robomimic/algo/icl.py:885-889 AdamW(lr=1e-3, wd=1e-4); :913-914 zero_grad;
                             :968-970 loss.backward(), step()), optionally data parallel
                             (``sharded.all_reduce_gradients``).
Pure plumbing: every number comes from the HIP library through ``LLFQVAE_V4``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import sharded
from .binning import AdaptiveBinActionEmbedding
from .tokenizer import LLFQVAE_V4, VQVAE


class ICLActionBranch(nn.Module):
    def __init__(self, action_input_shape: int = 12, action_output_shape: int = 208, vq_vae_enabled: bool = True,
                 variant: str = "lipvq", bin_enabled: bool = False):
        super().__init__()
        self.bin_enabled = bool(bin_enabled)
        self.vq_vae_enabled = bool(vq_vae_enabled) and not self.bin_enabled      # the reference's elif order
        if self.bin_enabled:             # obs_nets.py:1214-1217: the binning tokenizer of the paper's ablation
            self.action_network = AdaptiveBinActionEmbedding(action_dim=action_input_shape, output_dim=action_output_shape)
        elif not vq_vae_enabled:
            raise NotImplementedError("only the vq_vae_enabled and bin_enabled branches of the group encoder are on this path")
        elif variant == "lipvq":         # obs_nets.py:1225: the paper's tokenizer
            self.action_network = LLFQVAE_V4(feature_dim=action_input_shape, latent_dim=action_output_shape)
        elif variant == "vqvae":         # obs_nets.py:1220-1222 (commented-out alternative)
            self.action_network = VQVAE(feature_dim=action_input_shape, latent_dim=action_output_shape)
        else:
            raise ValueError(variant)
        self._vq_vae_loss = None

    def forward(self, prompt_actions: torch.Tensor) -> torch.Tensor:
        if self.bin_enabled:
            return self.action_network(prompt_actions)                    # obs_nets.py:1343-1344 (no tokenizer loss)
        context_actions, loss = self.action_network(prompt_actions)      # obs_nets.py:1336
        self._vq_vae_loss = loss                                          # obs_nets.py:1337
        return context_actions


def time_distributed(actions: torch.Tensor, op) -> torch.Tensor:
    """Apply op to [B, T, A] actions flattened to [B*T, A]; reshape the result back to [B, T, D]."""
    b, t = actions.shape[:2]
    out = op(actions.reshape(b * t, *actions.shape[2:]))
    return out.reshape(b, t, *out.shape[1:])


class VQTokenizerTrainer:
    def __init__(self, vq_vae_model: nn.Module, lr: float = 1e-3, weight_decay: float = 1e-4, group=None):
        self.model = vq_vae_model
        self.vq_optimizer = torch.optim.AdamW(vq_vae_model.parameters(), lr=lr, weight_decay=weight_decay)
        self.group = group

    def train_on_actions(self, prompt_actions: torch.Tensor, n_global: int | None = None):
        """One step: returns (context_actions detached, loss value tensor)."""
        self.vq_optimizer.zero_grad()
        context_actions, loss = self.model(prompt_actions)
        loss.backward()
        if n_global is not None:
            sharded.all_reduce_gradients(self.model.parameters(), prompt_actions.shape[0], n_global, self.group)
        self.vq_optimizer.step()
        return context_actions, loss.detach()
