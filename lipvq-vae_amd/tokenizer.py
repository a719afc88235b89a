"""Drop-in replacements for the reference's action tokenizers, running on the HIP library.

``LLFQVAE_V4``  mirrors robomimic/models/vq_vae/backbone_lfqvae_v5.py:51-84 (the paper's
                LipVQ-VAE): same constructor signature and defaults, same 14 ``state_dict``
                keys/shapes/dtypes, same RNG consumption at construction (so the same
                ``torch.manual_seed`` gives the same initial weights), and
                ``forward(x[N,A]) -> (z_latent[N,D] without grad, loss 0-dim with grad_fn)``.
``VQVAE``       mirrors robomimic/models/vq_vae/backbone.py:6-76 (ReLU stacks, squared-L2
                argmin, commitment 0.25, straight-through estimator).

The ``nn.Linear`` / container sub-modules exist only to own the parameters under the
reference's names; they are never called.  All arithmetic is issued through
``lipvq_vae_amd.ops`` (C ABI -> hand-written gfx950 kernels).  Gradients are produced by the
library's backward kernels through one ``torch.autograd.Function`` per variant, so
``loss.backward(); AdamW.step()`` in the ICRT loop (robomimic/algo/icl.py:885-889,968-970)
works unchanged.

Extras that the reference does not have (allowed by SURVEY.md section 8b): ``last_indices``,
``code_usage`` (int64 [K], accumulated over forwards), ``tokenize()`` (encode + quantize
only: the BASELINE metric's path) and ``perplexity()``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, DIST_NORM, DIST_SQSUM

__all__ = ["LLFQVAE_V4", "VQVAE", "LipschitzMLP", "LFQQuantizer"]


class _PackCache:
    """Re-lays weights out for the MFMA kernels only when a parameter changed (an optimizer
    step bumps ``Tensor._version``; ``.to()``/``load_state_dict`` change data_ptr/version).

    What the key CANNOT see: writes that bypass autograd's version counter -- ``p.data.add_(..)`` (the reference's own
    ``embedding.weight.data.uniform_()`` idiom), a captured optimizer step replayed from a HIP graph, a raw-pointer writer.
    After such a write call ``module.invalidate_caches()``; ``load_state_dict``, ``_apply`` (``.to()``, ``.cuda()``,
    ``.float()``) and ``train()`` / ``eval()`` do it by themselves."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self._key:
            self._val = build()
            self._key = key
        return self._val

    def invalidate(self):
        self._key = None
        self._val = None


class LipschitzMLP(nn.Module):
    """Parameter holder of the Lipschitz latent layer (reference v5:15-24): W ~ N(0,1), b = 0, ci = 1."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.W = nn.Parameter(torch.randn(out_dim, in_dim))
        self.b = nn.Parameter(torch.zeros(out_dim))
        self.ci = nn.Parameter(torch.ones(out_dim))

    def normalized_weight(self):
        """(scale[D], W_norm[D,H]) = normalization(W, ci) of v5:6-12, computed on the GPU."""
        return ops.lipschitz_scale(self.W.detach(), self.ci.detach())


class LFQQuantizer(nn.Module):
    """Parameter holder of the codebook (reference v5:27-35): randn then kaiming_uniform_."""

    def __init__(self, num_codes, code_dim):
        super().__init__()
        self.num_codes = num_codes
        self.code_dim = code_dim
        self.codebook = nn.Parameter(torch.randn(num_codes, code_dim))
        nn.init.kaiming_uniform_(self.codebook)

    def forward(self, z_e):
        """(z_q, indices) of v5:37-48 for callers that use the quantizer on its own (no grad)."""
        idx, zq, _ = ops.nearest(z_e.detach(), self.codebook.detach(), DIST_NORM)
        return zq, idx


class _ScreenMonitor:
    """Keeps the certified screen from being used where it does not pay, without ever synchronising.

    The screen hands the rows it cannot certify to the exact kernel.  With trained parameters that is a fraction of a percent;
    with a DEGENERATE codebook -- the reference's own default initialisation maps every row to one code, a
    collapsed training run does the same -- it is (almost) every row, and the exact kernel, built for a few thousand rows with
    short candidate lists, would scan the whole codebook for each of them: orders of magnitude slower than the all-pairs exact
    kernel (results are the same on every route).  So after each screened call the uncertified-row count (workspace[0]) is
    copied to pinned host memory asynchronously; the NEXT call looks at it only if the copy's event has already completed
    (``event.query()``: never a wait).  A fraction above ``THRESHOLD`` routes the following ``PROBE_EVERY`` large-batch calls
    through the all-pairs kernel, then the screen is tried again."""

    THRESHOLD = 1.0 / 16.0        # three-product screen: a fraction of a percent is normal
    THRESHOLD_COARSE = 0.6        # one-product screen: 10-40 % of the rows go to the exact stage BY DESIGN (lipvq_screen_is_coarse)
    PROBE_EVERY = 16

    def __init__(self):
        self._pending = None
        self.bypass_calls = 0
        self.last_fraction = None

    ENABLED = True                # tests / measurements: False = always the screen (class attribute; no environment variable)

    def use_screen(self) -> bool:
        if not _ScreenMonitor.ENABLED:
            return True
        if torch.cuda.is_current_stream_capturing():
            # a graph capture records ONE route and no host decision can run at replay: take the route the last eager reading
            # chose (a codebook that was sending every row to the exact stage keeps the all-pairs kernel in the graph too)
            return self.bypass_calls <= 0
        if self._pending is not None:
            ev, host, n, coarse = self._pending
            if ev.query():
                self.last_fraction = float(host[0]) / max(1, n)
                self._pending = None
                if self.last_fraction > (self.THRESHOLD_COARSE if coarse else self.THRESHOLD):
                    self.bypass_calls = self.PROBE_EVERY
        if self.bypass_calls > 0:
            self.bypass_calls -= 1
            return False
        return True

    def record(self, ws: torch.Tensor, n: int, coarse: bool = False) -> None:
        if torch.cuda.is_current_stream_capturing() or self._pending is not None:
            return
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(ws[:1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (ev, host, int(n), bool(coarse))


class _TokenizerBase(nn.Module):
    def _init_extras(self, K):
        self._screen_monitor = _ScreenMonitor()
        self.register_buffer("code_usage", torch.zeros(K, dtype=torch.int64), persistent=False)
        self.last_indices = None
        self._enc_cache = _PackCache()
        self._dec_cache = _PackCache()
        self._cb_cache = _PackCache()
        self._enc16_cache = _PackCache()
        self._tok_ws, self._tok_ws_key = None, None
        self.last_exact_rows = None      # int32 device tensor: element 0 = rows the screen could not certify

    def reset_usage(self):
        self.code_usage.zero_()

    # -- derived buffers (packed weights, prepared codebook, fused-launch workspace) -------------------
    def invalidate_caches(self):
        """Drop everything derived from the parameters; the next call rebuilds it.  Needed after a write the version
        counter does not see (see _PackCache); cheap (a few small launches on the next forward)."""
        for c in (self._enc_cache, self._dec_cache, self._cb_cache, self._enc16_cache):
            c.invalidate()
        self._tok_ws, self._tok_ws_key = None, None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_caches()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_enc_cache"):
            self.invalidate_caches()
        return out

    def train(self, mode: bool = True):
        out = super().train(mode)
        if hasattr(self, "_enc_cache"):
            self.invalidate_caches()
        return out

    def perplexity(self):
        """exp(entropy) of the accumulated code-usage histogram."""
        c = self.code_usage.to(torch.float64)
        p = c / c.sum().clamp_min(1)
        nz = p > 0
        return float(torch.exp(-(p[nz] * p[nz].log()).sum()))

    @staticmethod
    def _as_rows(x):
        if x.dim() != 2:
            raise ValueError(f"expected [N, feature_dim] (the reference flattens [B,T,A] to [B*T,A] "
                             f"before the tokenizer, tensor_utils.py:1066-1067); got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise TypeError(f"expected float32 actions, got {x.dtype}")
        return x.contiguous()


class LLFQVAE_V4(_TokenizerBase):
    def __init__(self, feature_dim, latent_dim, num_codes=1024, hidden_dim=128):
        super().__init__()
        if hidden_dim % 32 or not (32 <= hidden_dim <= 256):
            raise ValueError("hidden_dim must be a multiple of 32 in [32, 256] on the MI355X path (MFMA tiles of 32)")
        # construction order = the reference's (v5:54-68), so RNG draws line up
        self.encoder = nn.Sequential(nn.Linear(feature_dim, 64), nn.GELU(), nn.Linear(64, hidden_dim), nn.GELU())
        self.to_latent = LipschitzMLP(hidden_dim, latent_dim)
        self.quantizer = LFQQuantizer(num_codes, latent_dim)
        self.decoder = nn.Sequential(nn.Linear(latent_dim, 64), nn.GELU(), nn.Linear(64, hidden_dim), nn.GELU())
        self.to_output = nn.Linear(hidden_dim, feature_dim)
        self.feature_dim, self.latent_dim, self.num_codes, self.hidden_dim = feature_dim, latent_dim, num_codes, hidden_dim
        self._init_extras(num_codes)

    # -- packed weights ---------------------------------------------------------------------
    def _enc_params(self):
        return (self.encoder[0].weight, self.encoder[0].bias, self.encoder[2].weight, self.encoder[2].bias,
                self.to_latent.W, self.to_latent.b, self.to_latent.ci)

    def _dec_params(self):
        return (self.decoder[0].weight, self.decoder[0].bias, self.decoder[2].weight, self.decoder[2].bias,
                self.to_output.weight, self.to_output.bias)

    def _packed_encoder(self):
        def build():
            w0, b0, w1, b1, W, b, ci = (t.detach() for t in self._enc_params())
            scale, Wn = ops.lipschitz_scale(W, ci)
            return ops.mlp3_pack(w0, b0, w1, b1, Wn, b), scale, Wn
        return self._enc_cache.get(self._enc_params(), build)

    def _packed_decoder(self):
        def build():
            return ops.mlp3_pack(*(t.detach() for t in self._dec_params()))
        return self._dec_cache.get(self._dec_params(), build)

    # rows up to which _quantize skips the screen: measured (scripts/measure_quantize_small.py, K = 1024) exact rows vs
    # screen [+ codebook preparation]: D = 208: 36 vs 75 [+69] us at N = 80, 74 vs 122 [+69] at 2048, 143 vs 89 [+69] at 4096
    EXACT_ROWS_MAX = 2048

    def screen_pays(self, n_rows: int) -> bool:
        """ONE routing decision per call: small batches never consult the monitor (the exact kernel copes with all of their rows);
        large ones ask it whether the screen has been certifying its rows (see _ScreenMonitor)."""
        return n_rows <= self.EXACT_ROWS_MAX or self._screen_monitor.use_screen()

    def _quantize(self, z_e, usage, screen=None):
        """(idx, z_q) of v5:37-48: MFMA screen + exact re-scoring where the latent width has a
        screening instance, the all-pairs exact kernel otherwise.  Identical results either way."""
        cb = self.quantizer.codebook.detach()
        if screen is None:
            screen = self.screen_pays(z_e.shape[0])
        if ops.nearest_screen_supported(cb.shape[0], cb.shape[1]) and 0 < z_e.shape[0] <= self.EXACT_ROWS_MAX:
            # training-step batches: the exact kernel on every row beats preparing the codebook + one screen launch
            self.last_exact_rows = None
            return ops.nearest_rows(z_e, cb, usage=usage)
        if ops.nearest_screen_supported(cb.shape[0], cb.shape[1]) and z_e.shape[0] > 0 and screen:
            prep = self._cb_cache.get((self.quantizer.codebook,), lambda: ops.nearest_prepare(cb))
            idx, zq, ws = ops.nearest_screened(z_e, cb, prep, usage=usage, return_workspace=True)
            self.last_exact_rows = ws
            self._screen_monitor.record(ws, z_e.shape[0], ops.screen_is_coarse(cb.shape[0], cb.shape[1]))
            return idx, zq
        self.last_exact_rows = None
        idx, zq, _ = ops.nearest(z_e, cb, DIST_NORM, usage=usage)
        return idx, zq

    # -- the metric's path ------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x):
        """z_e = to_latent(encoder(x))   (v5:71-72)."""
        packed, _, _ = self._packed_encoder()
        return ops.mlp3(self._as_rows(x), packed, (ACT_GELU, ACT_GELU, ACT_SIGMOID))

    def _tokenize_fused(self, x, usage, want_ze=False, fast=False, want_pre=False):
        """(idx, z_q, z_e | None) from ONE persistent launch: z_e never leaves registers unless asked for
        (csrc/lipvq_fused.hip).  Caller checks ops.tokenize_supported()."""
        cb = self.quantizer.codebook.detach()
        packed, _, Wn = self._packed_encoder()
        w0, b0, w1, b1, _, b2, _ = (t.detach() for t in self._enc_params())
        prep = self._cb_cache.get((self.quantizer.codebook,), lambda: ops.nearest_prepare(cb))
        key = (x.shape[0], x.device)
        if self._tok_ws_key != key:                          # the scratch (row list) is reused across calls
            self._tok_ws, self._tok_ws_key = ops.tokenize_workspace(x.shape[0], self.latent_dim, x.device), key
        packed16 = None
        if fast:
            packed16 = self._enc16_cache.get((w0, w1, self.to_latent.W, self.to_latent.ci),
                                             lambda: ops.mlp3_pack_f16(w0, w1, Wn))
        if want_pre:                                         # training forward: z_e and the three pre-activations as well
            idx, zq, ze, ws, pre = ops.tokenize(x, packed, (w0, b0, w1, b1, Wn, b2), cb, prep, usage=usage, workspace=self._tok_ws,
                                                want_pre=True)
            self.last_exact_rows = ws
            self._screen_monitor.record(ws, x.shape[0], ops.screen_is_coarse(self.num_codes, self.latent_dim))
            return idx, zq, ze, pre
        idx, zq, ze, ws = ops.tokenize(x, packed, (w0, b0, w1, b1, Wn, b2), cb, prep, usage=usage, want_ze=want_ze,
                                       workspace=self._tok_ws, packed16=packed16)
        self.last_exact_rows = ws
        self._screen_monitor.record(ws, x.shape[0], (not fast) and ops.screen_is_coarse(self.num_codes, self.latent_dim))
        return idx, zq, ze

    def fused_shape(self) -> bool:
        return ops.tokenize_supported(self.feature_dim, 64, self.hidden_dim, self.latent_dim, self.num_codes)

    @torch.no_grad()
    def tune(self, x, launches=150):
        """Measure the fused launch's device-dependent schedule choices on THIS device with this batch and keep the fastest for the
        process (lipvq_tokenize_tune_f32; MI355X devices hold different clocks under the same kernel, and what wins on one loses on
        another: profiles/r04_i_clock_ab.txt).  Results are identical under every choice.  Synchronous: call it once, outside any
        timed or captured region.  Returns {"choice": {"defer_ze", "nt_ze"}, "ms_per_launch": {...}} or None (no fused launch here)."""
        x = self._as_rows(x)
        if x.shape[0] == 0 or not self.fused_shape():
            return None
        cb = self.quantizer.codebook.detach()
        packed, _, Wn = self._packed_encoder()
        w0, b0, w1, b1, _, b2, _ = (t.detach() for t in self._enc_params())
        prep = self._cb_cache.get((self.quantizer.codebook,), lambda: ops.nearest_prepare(cb))
        choice, ms = ops.tokenize_tune(x, packed, (w0, b0, w1, b1, Wn, b2), cb, prep, launches=launches)
        if choice < 0:                                       # latent widths above 64: nothing device-dependent to choose
            return None
        return {"choice": {"defer_ze": choice & 1, "nt_ze": (choice >> 1) & 1},
                "ms_per_launch": {f"defer_ze={c & 1},nt_ze={(c >> 1) & 1}": ms[c] for c in range(4)}}

    @torch.no_grad()
    def tokenize(self, x, count_usage=True, mode="parity"):
        """encode + quantize: (indices[N] int64, z_latent[N,D])   (v5:71-74).

        mode="parity" (default): fp32 everywhere, indices bit-identical to the CPU oracle / the reference.
        mode="fast": the encoder's GEMMs on fp16 MFMAs (fp32 accumulation) -- SURVEY section 7's throughput mode; a
        fraction of a percent of the indices differ from parity mode, always between near-equidistant codes."""
        if mode not in ("parity", "fast"):
            raise ValueError(f"unknown tokenize mode {mode!r}")
        x = self._as_rows(x)
        usage = self.code_usage if count_usage else None
        shape = (self.feature_dim, 64, self.hidden_dim, self.latent_dim, self.num_codes)
        if mode == "fast" and not ops.tokenize_fast_supported(*shape):
            raise RuntimeError("tokenize(mode='fast') needs the fast kernel's shapes (hidden 64/128, D in {32, 64, 128})")
        screen = self.screen_pays(x.shape[0])
        if x.shape[0] > 0 and screen and self.fused_shape():
            idx, zq, _ = self._tokenize_fused(x, usage, fast=(mode == "fast"))
        else:
            idx, zq = self._quantize(self.encode(x), usage, screen=screen)
        self.last_indices = idx
        return idx, zq

    @torch.no_grad()
    def decode(self, indices):
        """x_recon = to_output(decoder(codebook[indices]))   (v5:75-76)."""
        return ops.mlp3(self.quantizer.codebook.detach(), self._packed_decoder(), (ACT_GELU, ACT_GELU, ACT_NONE),
                        gather_idx=indices)

    # -- reference forward ------------------------------------------------------------------
    def forward(self, x):
        from .autograd import llfq_forward
        return llfq_forward(self, self._as_rows(x))


class VQVAE(_TokenizerBase):
    def __init__(self, feature_dim, latent_dim, num_embeddings=128, commitment_cost=0.25):
        super().__init__()
        self.feature_dim = feature_dim
        self.latent_dim = latent_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.encoder = nn.Sequential(nn.Linear(feature_dim, 64), nn.ReLU(), nn.Linear(64, 128), nn.ReLU(),
                                     nn.Linear(128, latent_dim), nn.ReLU())
        self.decoder = nn.Sequential(nn.Linear(latent_dim, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                     nn.Linear(64, feature_dim), nn.ReLU())
        self.embedding = nn.Embedding(num_embeddings, latent_dim)
        self.embedding.weight.data.uniform_(-1 / num_embeddings, 1 / num_embeddings)
        self._init_extras(num_embeddings)

    def _enc_params(self):
        e = self.encoder
        return (e[0].weight, e[0].bias, e[2].weight, e[2].bias, e[4].weight, e[4].bias)

    def _dec_params(self):
        d = self.decoder
        return (d[0].weight, d[0].bias, d[2].weight, d[2].bias, d[4].weight, d[4].bias)

    def _packed_encoder(self):
        return self._enc_cache.get(self._enc_params(),
                                   lambda: ops.mlp3_pack(*(t.detach() for t in self._enc_params())))

    def _packed_decoder(self):
        return self._dec_cache.get(self._dec_params(),
                                   lambda: ops.mlp3_pack(*(t.detach() for t in self._dec_params())))

    @torch.no_grad()
    def encode(self, x):
        return ops.mlp3(self._as_rows(x), self._packed_encoder(), (ACT_RELU, ACT_RELU, ACT_RELU))

    # rows up to which the exact re-scoring kernel decides every row (no codebook preparation), as in LLFQVAE_V4._quantize
    EXACT_ROWS_MAX = 2048
    # the all-pairs kernel stays the route for small codebooks: at K = 128 (the reference's default, vq:7) it runs at its VALU
    # bound in 0.4 ms per 524 288 rows, where a screen launch + its uncertified rows would gain nothing
    SCREEN_MIN_CODES = 256
    # ... but the FUSED launch (encoder + screen in one kernel, z_e never re-read) beats encoder + all-pairs from far fewer codes on
    FUSED_MIN_CODES = 64

    def _quantize(self, z_e, usage, screen=None):
        """(idx, z_q) of vq:57-66 (`pow(2).sum(-1)`, argmin, embedding lookup): MFMA screen + exact re-scoring where the latent
        width has a screening instance and the codebook is large enough to pay for it, the all-pairs exact kernel otherwise.
        Identical results on every route."""
        cb = self.embedding.weight.detach()
        K, D = cb.shape
        n = z_e.shape[0]
        if ops.nearest_screen_supported(K, D) and K >= self.SCREEN_MIN_CODES and n > 0:
            if n <= self.EXACT_ROWS_MAX:
                self.last_exact_rows = None
                return ops.nearest_rows(z_e, cb, usage=usage, dist=DIST_SQSUM)
            if screen is None:
                screen = self._screen_monitor.use_screen()
            if screen:
                prep = self._cb_cache.get((self.embedding.weight,), lambda: ops.nearest_prepare(cb))
                idx, zq, ws = ops.nearest_screened(z_e, cb, prep, usage=usage, return_workspace=True, dist=DIST_SQSUM)
                self.last_exact_rows = ws
                self._screen_monitor.record(ws, n, ops.screen_is_coarse(K, D))
                return idx, zq
        self.last_exact_rows = None
        idx, zq, _ = ops.nearest(z_e, cb, DIST_SQSUM, usage=usage)
        return idx, zq

    def fused_shape(self) -> bool:
        return ops.tokenize_supported(self.feature_dim, 64, 128, self.latent_dim, self.num_embeddings)

    def _tokenize_fused(self, x, usage, want_pre=False):
        """(idx, z_q, z_e) from ONE persistent launch (lipvq_vq_tokenize_f32: the fused kernel's ReLU instance, per-row fp16
        scales): encoder + screen, then the exact stage for the rows the screen leaves.  want_pre: (idx, z_q, z_e, pre) with the
        three pre-activations a training step saves (lipvq_vq_tokenize_train_f32)."""
        cb = self.embedding.weight.detach()
        prep = self._cb_cache.get((self.embedding.weight,), lambda: ops.nearest_prepare(cb))
        key = (x.shape[0], x.device)
        if self._tok_ws_key != key:
            self._tok_ws, self._tok_ws_key = ops.tokenize_workspace(x.shape[0], self.latent_dim, x.device), key
        pre = None
        if want_pre:
            idx, zq, ze, pre, ws = ops.vq_tokenize(x, self._packed_encoder(), cb, prep, usage=usage, workspace=self._tok_ws, want_pre=True)
        else:
            idx, zq, ze, ws = ops.vq_tokenize(x, self._packed_encoder(), cb, prep, usage=usage, workspace=self._tok_ws)
        self.last_exact_rows = ws
        self._screen_monitor.record(ws, x.shape[0], ops.screen_is_coarse(cb.shape[0], cb.shape[1]))
        return (idx, zq, ze, pre) if want_pre else (idx, zq, ze)

    @torch.no_grad()
    def tokenize(self, x, count_usage=True):
        x = self._as_rows(x)
        usage = self.code_usage if count_usage else None
        n = x.shape[0]
        # the fused launch from the codebook size on where a screen pays at all (below, the all-pairs kernel at its VALU bound);
        # ONE routing decision per call (see _ScreenMonitor)
        big = n > self.EXACT_ROWS_MAX and self.num_embeddings >= self.FUSED_MIN_CODES
        screen = self._screen_monitor.use_screen() if big else None
        if big and screen and self.fused_shape():
            idx, zq, z_e = self._tokenize_fused(x, usage)
        else:
            z_e = self.encode(x)
            idx, zq = self._quantize(z_e, usage, screen=screen)
        self.last_indices = idx
        return idx, ops.ste(z_e, zq)

    def forward(self, x):
        from .autograd import vq_forward
        return vq_forward(self, self._as_rows(x))
