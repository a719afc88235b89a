"""Host glue around the tokenizer, restating the few lines of the reference that drive it.

* ``build_action_network`` / ``ICLActionBranch`` -- the action branch of ``ICLObservationGroupEncoder`` when ``vq_vae_enabled``
                             (robomimic/models/obs_nets.py:1219-1227 construction, :1335-1337 call) or
                             ``bin_enabled`` (:1214-1217, :1343-1344) or neither (:1244-1260, the default branch):
                             owns ``action_network``, stashes ``_vq_vae_loss``.
* ``time_distributed``     -- the [B, T, ...] <-> [B*T, ...] reshape of
                             ``TensorUtils.icl_time_distributed`` (robomimic/utils/tensor_utils.py:1045-1090)
                             for the action leaf.
* ``icl_time_distributed`` -- the same function in full: nested dict / list / tuple inputs, kwargs / args call forms,
                             the ``(obs, context_obs, context_actions)`` triple (tensor_utils.py:1045-1090) -- what
                             ``ICL_MIMO_Transformer.forward`` calls on the group encoder (obs_nets.py:2571).
* ``ICLObservationGroupEncoder`` -- the group encoder as a whole object (obs_nets.py:1120-1383): observation encoders are
                             handed in and passed through untouched (they are outside this path), ``prompt["action"]`` goes
                             to ``action_network`` -- registered under that name, so ``state_dict()`` has the reference's
                             keys; ``forward(**inputs)`` returns the reference's triple, ``_vq_vae_loss`` holds the loss.
* ``VQTokenizerTrainer``   -- the optimiser choreography of ``ICLTransformer_GMM``
                             (robomimic/algo/icl.py:885-889 AdamW(lr=1e-3, wd=1e-4); :913-914 zero_grad;
                             :968-970 loss.backward(), step()), optionally data parallel
                             (``sharded.all_reduce_gradients``).
Pure plumbing: every number comes from the HIP library through ``LLFQVAE_V4``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import optim, sharded
from .binning import AdaptiveBinActionEmbedding
from .default_branch import DefaultActionNetwork
from .tokenizer import LLFQVAE_V4, VQVAE


def build_action_network(action_input_shape: int, action_output_shape: int, vq_vae_enabled: bool = True,
                         variant: str = "lipvq", bin_enabled: bool = False) -> nn.Module:
    """The reference's ``if bin_enabled / elif vq_vae_enabled / else`` chain (obs_nets.py:1213-1260) for the branches this
    library builds; the result is what the reference registers as ``self.action_network``."""
    if bin_enabled:                  # obs_nets.py:1214-1217: the binning tokenizer of the paper's ablation
        return AdaptiveBinActionEmbedding(action_dim=action_input_shape, output_dim=action_output_shape)
    if not vq_vae_enabled:           # obs_nets.py:1244-1260: spectral-norm MLP + TransformerEncoder + Linear
        return DefaultActionNetwork(action_input_shape, action_output_shape)
    if variant == "lipvq":           # obs_nets.py:1225: the paper's tokenizer
        return LLFQVAE_V4(feature_dim=action_input_shape, latent_dim=action_output_shape)
    if variant == "vqvae":           # obs_nets.py:1220-1222 (commented-out alternative)
        return VQVAE(feature_dim=action_input_shape, latent_dim=action_output_shape)
    raise ValueError(variant)


def _encode_actions(owner, prompt_actions: torch.Tensor) -> torch.Tensor:
    """obs_nets.py:1335-1344 on ``owner.action_network``; the tokenizer's loss is stashed on the owner."""
    if owner.bin_enabled or not owner.vq_vae_enabled:
        return owner.action_network(prompt_actions)                       # obs_nets.py:1343-1344 (no tokenizer loss)
    context_actions, loss = owner.action_network(prompt_actions)         # obs_nets.py:1336
    owner._vq_vae_loss = loss                                             # obs_nets.py:1337
    return context_actions


class ICLActionBranch(nn.Module):
    """The action branch alone (no observation encoders): ``action_network`` + ``_vq_vae_loss``."""

    def __init__(self, action_input_shape: int = 12, action_output_shape: int = 208, vq_vae_enabled: bool = True,
                 variant: str = "lipvq", bin_enabled: bool = False):
        super().__init__()
        self.bin_enabled = bool(bin_enabled)
        self.vq_vae_enabled = bool(vq_vae_enabled) and not self.bin_enabled      # the reference's elif order
        self.action_network = build_action_network(action_input_shape, action_output_shape, vq_vae_enabled, variant, bin_enabled)
        self._vq_vae_loss = None

    def forward(self, prompt_actions: torch.Tensor) -> torch.Tensor:
        return _encode_actions(self, prompt_actions)


def time_distributed(actions: torch.Tensor, op) -> torch.Tensor:
    """Apply op to [B, T, A] actions flattened to [B*T, A]; reshape the result back to [B, T, D]."""
    b, t = actions.shape[:2]
    out = op(actions.reshape(b * t, *actions.shape[2:]))
    return out.reshape(b, t, *out.shape[1:])


def _map_leaves(x, fn):
    """New nested dict / list / tuple with fn applied to every tensor leaf (None stays None) -- the traversal
    tensor_utils.py's join_dimensions / reshape_dimensions perform."""
    if isinstance(x, torch.Tensor):
        return fn(x)
    if x is None:
        return None
    if isinstance(x, dict):
        return type(x)((k, _map_leaves(v, fn)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(_map_leaves(v, fn) for v in x)
    raise NotImplementedError(f"icl_time_distributed: cannot map a leaf of type {type(x).__name__}")


def _first_leaf(x):
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, dict):
        it = x.values()
    elif isinstance(x, (list, tuple)):
        it = x
    else:
        return None
    for v in it:
        t = _first_leaf(v)
        if t is not None:
            return t
    return None


def icl_time_distributed(inputs, op, activation=None, inputs_as_kwargs=False, inputs_as_args=False, **kwargs):
    """``TensorUtils.icl_time_distributed`` (tensor_utils.py:1045-1090): every tensor of the nested ``inputs`` ([B, T, ...])
    is flattened to [B*T, ...], ``op`` is called (``op(**inputs)``, ``op(*inputs)`` or ``op(inputs)``) and must return the
    triple ``(obs, context_obs, context_actions)``, whose tensors are reshaped back to [B, T, ...].  B and T are read off
    the first tensor leaf, as the reference does."""
    first = _first_leaf(inputs)
    if first is None or first.dim() < 2:
        raise ValueError("icl_time_distributed: inputs hold no [B, T, ...] tensor")
    batch_size, seq_len = first.shape[:2]
    flat = _map_leaves(inputs, lambda t: t.reshape(-1, *t.shape[2:]))
    if inputs_as_kwargs:
        obs, context_obs, context_actions = op(**flat, **kwargs)
    elif inputs_as_args:
        obs, context_obs, context_actions = op(*flat, **kwargs)
    else:
        obs, context_obs, context_actions = op(flat, **kwargs)
    outs = []
    for o in (obs, context_obs, context_actions):
        if activation is not None:
            o = _map_leaves(o, activation)
        outs.append(_map_leaves(o, lambda t: t.reshape(batch_size, seq_len, *t.shape[1:])))
    return tuple(outs)


class ICLObservationGroupEncoder(nn.Module):
    """``ICLObservationGroupEncoder`` (obs_nets.py:1120-1383) around the MI355X action branch.

    The reference builds one ``ObservationEncoder`` per observation group itself (``obs_encoder_factory``: CNNs, randomisers,
    ... -- outside this path); here they are handed in ready-made as ``obs_encoders`` (any modules with
    ``forward(obs_dict) -> [N, F]`` and ``output_shape() -> [F]``) and called exactly where the reference calls them.  The
    action branch is ``build_action_network`` with ``latent_dim = sum of the groups' feature widths`` (obs_nets.py:1193), selected by
    the reference's switches; ``fast_enabled`` / ``ln_act_enabled`` need the FAST tokenizer + CLIP / Mamba packages and are
    refused.  ``forward(**inputs)`` = obs_nets.py:1264-1345: returns ``(obs, context_obs, context_actions)`` on [B*T, ...] rows and
    stashes the tokenizer's loss in ``_vq_vae_loss``."""

    def __init__(self, obs_encoders, action_input_shape, fast_enabled=False, bin_enabled=False, vq_vae_enabled=False,
                 ln_act_enabled=False, variant: str = "lipvq"):
        super().__init__()
        if fast_enabled or ln_act_enabled:
            raise NotImplementedError("fast_enabled / ln_act_enabled need the FAST tokenizer + CLIP / Mamba packages (out of scope)")
        self.nets = nn.ModuleDict(obs_encoders)                          # deterministic order: insertion order, as the OrderedDict
        self.observation_group_shapes = {k: None for k in self.nets}     # the reference iterates this mapping's keys
        self.fast_enabled, self.ln_act_enabled = False, False
        self.bin_enabled, self.vq_vae_enabled = bool(bin_enabled), bool(vq_vae_enabled)
        # registered under the reference's own attribute name (obs_nets.py:1214-1260), so that ``state_dict()`` yields
        # ``nets.<group>.*`` + ``action_network.*`` -- the keys a reference checkpoint holds (algo.py:323-337)
        self.action_network = build_action_network(int(action_input_shape), self.output_shape()[0], vq_vae_enabled=vq_vae_enabled,
                                                   variant=variant, bin_enabled=bin_enabled)
        self.vq_vae_enabled = self.vq_vae_enabled and not self.bin_enabled       # the reference's elif order
        self._vq_vae_loss = None                                         # read by obs_nets.py:2576-2577

    def output_shape(self):
        return [sum(int(self.nets[g].output_shape()[0]) for g in self.nets)]      # obs_nets.py:1347-1355

    def forward(self, **inputs):
        prompt_obs = inputs["prompt"]["obs"]                             # obs_nets.py:1282-1283
        prompt_actions = inputs["prompt"]["action"]
        missing = set(self.nets.keys()) - set(inputs)
        assert not missing, f"{list(inputs.keys())} does not contain all observation groups {list(self.nets.keys())}"
        outputs = [self.nets[g](inputs[g]) for g in self.nets]           # :1293-1296
        obs = torch.cat(outputs, dim=-1)                                 # :1302
        context_obs = torch.cat([self.nets["obs"](prompt_obs)], dim=-1)  # :1303-1304
        context_actions = _encode_actions(self, prompt_actions)          # :1335-1344 (loss stashed on this module)
        return obs, context_obs, context_actions


class VQTokenizerTrainer:
    def __init__(self, vq_vae_model: nn.Module, lr: float = 1e-3, weight_decay: float = 1e-4, group=None):
        self.model = vq_vae_model
        # icl.py:885-889; the same AdamW in two launches for the whole parameter list (optim.py) instead of torch's foreach form
        self.vq_optimizer = optim.AdamW(vq_vae_model.parameters(), lr=lr, weight_decay=weight_decay)
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


class GraphedTokenizerStep:
    """The tokenizer's training step (icl.py:913-914 zero_grad, forward, :968-970 backward + AdamW) captured ONCE into a HIP
    graph and replayed: at the ICRT step shape (N = 80) the eager step is host bound (~60 launches of 7-10 us of Python +
    ctypes each), a replay is one call.

    What makes the capture safe (round 1's attempt faulted on replay "when the model had been trained eagerly before capture"):
    a captured graph refers to every buffer BY ADDRESS.  The eager steps had left derived buffers behind -- packed weights,
    the prepared codebook, the fused launch's workspace, all owned by Python-side caches keyed on the parameters' version --
    in the ordinary allocator; the capture recorded their addresses, and the next eager call (or the cache refresh that an
    optimizer step triggers) freed them, so a replay read and wrote memory the allocator had handed to someone else.  Here

      * ``invalidate_caches()`` runs before the capture, so that every derived buffer is rebuilt INSIDE it: the pack /
        prepare kernels become graph nodes (each replay re-packs the weights the previous replay's AdamW wrote) and their
        buffers live in the graph's private pool, at addresses nobody else is ever given;
      * the step is the engine-free ``autograd.forward_backward`` (no autograd-engine streams or AccumulateGrad nodes),
        gradients are assigned to ``p.grad`` once, at capture (graph-pool tensors that every replay rewrites), and AdamW is
        ``capturable`` (its step counter lives on the device);
      * warm-up steps (real training steps, on a side stream) run before the capture, so that lazy one-time work -- the
        per-device LDS reservations (hipFuncSetAttribute), AdamW's state -- is not issued while capturing;
      * ``invalidate_caches()`` runs again after the capture and after every replay: a replayed optimizer step does not
        bump ``Tensor._version``, so eager code must never trust caches that predate it.
    The input is copied into a static buffer; shapes are fixed at construction."""

    def __init__(self, vq_vae_model: nn.Module, example_actions: torch.Tensor, lr: float = 1e-3, weight_decay: float = 1e-4,
                 optimizer_state: dict | None = None, warmup: int = 2):
        from .autograd import forward_backward
        self.model = vq_vae_model
        self._fb = forward_backward
        self.static_x = example_actions.detach().clone().contiguous()
        # AdamW in two launches for the whole parameter list (optim.py: lipvq_adamw_f32; its step counters live on the device,
        # so it is capturable).  torch's foreach capturable AdamW is ~8 launches (~80 us of the 550 us step), its fused=True form
        # one launch but on this build its trajectory leaves the eager AdamW's after one step even without a graph
        # (scripts/dev/debug_fused.py: loss 0.99112 vs 0.98865).
        self.vq_optimizer = optim.AdamW(vq_vae_model.parameters(), lr=lr, weight_decay=weight_decay)
        if optimizer_state is not None:
            import copy
            # continue an eager run: moments and step count carry over.  Deep copy: Optimizer.load_state_dict keeps the very
            # tensors it is handed when they already have the right device and dtype -- two optimizers would share moments
            self.vq_optimizer.load_state_dict(copy.deepcopy(optimizer_state))
            for grp in self.vq_optimizer.param_groups:
                grp["capturable"] = True
                for prm in grp["params"]:                            # a stock eager AdamW keeps `step` on the host
                    st = self.vq_optimizer.state.get(prm)
                    if st and "step" in st:
                        st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=prm.device)
        self.warmup_steps = int(warmup)
        # The warm-up steps are REAL training steps (they have to be: they make the one-time work happen); construction must not
        # train the model, though (round 2's did: two extra updates on the example batch).  Parameters and optimizer state are
        # snapshotted here and put back IN PLACE after the warm-up -- in place, because the capture below records the state
        # tensors' addresses and must find them allocated: state created by the warm-up is zeroed (= a fresh AdamW), state that
        # came with `optimizer_state` gets its values back.  Replay k then equals eager step k.
        params = list(vq_vae_model.parameters())
        saved_params = [p.detach().clone() for p in params]
        saved_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.vq_optimizer.state.get(p, {}).items()}
                       for p in params}
        # ... and so is everything else a step writes on the module: its buffers (code_usage -- warm-up counts would skew
        # perplexity() and, sharded, every rank's all-reduced histogram), last_indices, the screen monitor's reading
        saved_buffers = [(b, b.detach().clone()) for b in vq_vae_model.buffers()]
        saved_last = getattr(vq_vae_model, "last_indices", None)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup_steps):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, sp in zip(params, saved_params):
                p.copy_(sp)
                for k, v in self.vq_optimizer.state.get(p, {}).items():
                    if torch.is_tensor(v):
                        old = saved_state[id(p)].get(k)
                        if old is not None:
                            v.copy_(old)
                        else:
                            v.zero_()
            for b, sb in saved_buffers:
                b.copy_(sb)
        if hasattr(vq_vae_model, "last_indices"):
            vq_vae_model.last_indices = saved_last
        mon = getattr(vq_vae_model, "_screen_monitor", None)
        if mon is not None:
            mon.__init__()
        torch.cuda.synchronize()
        self.model.invalidate_caches()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._loss = self._eager_step()
        self.model.invalidate_caches()
        # the capture ran the forward once more (recording only): its index tensor is the graph's static one -- what a replay
        # fills -- and becomes `last_indices` with the first step(); until then the module shows what it showed before
        self._static_last = getattr(vq_vae_model, "last_indices", None)
        if hasattr(vq_vae_model, "last_indices"):
            vq_vae_model.last_indices = saved_last

    def _eager_step(self):
        z, loss, params, grads = self._fb(self.model, self.static_x)
        for p, g in zip(params, grads):
            p.grad = g
        self.vq_optimizer.step()
        self._z = z
        return loss

    def step(self, prompt_actions: torch.Tensor):
        """One training step on a batch of the captured shape: (context_actions, loss) -- static tensors, valid until the
        next step."""
        if prompt_actions.shape != self.static_x.shape:
            raise ValueError(f"graphed step was captured for {tuple(self.static_x.shape)}, got {tuple(prompt_actions.shape)}")
        self.static_x.copy_(prompt_actions)
        self.graph.replay()
        self.model.invalidate_caches()
        if hasattr(self.model, "last_indices"):
            self.model.last_indices = self._static_last
        return self._z, self._loss
