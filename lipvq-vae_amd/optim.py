"""``AdamW`` with the reference's constructor (``robomimic/algo/icl.py:885-889``:
``optim.AdamW(self.vq_vae_model.parameters(), lr=1e-3, weight_decay=1e-4)``) whose ``step()`` is two HIP launches for the whole
parameter list (``lipvq_adamw_f32``) instead of torch's eight to ten foreach launches -- at the ICRT step shape the optimizer
was ~80 us of a 550 us step.  It subclasses ``torch.optim.AdamW`` and keeps torch's state layout (``step`` as a float32
device scalar per parameter -- the capturable layout --, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict()``
interchange with a stock ``AdamW(capturable=True)`` and the step can be captured in a HIP graph.  amsgrad / maximize are refused;
parameters that are not fp32 CUDA tensors fall back to nothing -- they raise."""
from __future__ import annotations

import ctypes as C

import torch

from ._capi import check, lib
from .ops import _on, _stream

_MAX = 32


class AdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False):
        if amsgrad or maximize:
            raise ValueError("lipvq_vae_amd.optim.AdamW implements the plain AdamW the reference uses (amsgrad=False, maximize=False)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True)
        self._ws = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise RuntimeError("lipvq AdamW: parameters and gradients must be contiguous fp32 CUDA tensors")
                st = self.state[p]
                if len(st) == 0:                               # torch's lazy state, capturable layout
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not (torch.is_tensor(st["step"]) and st["step"].is_cuda):      # state loaded from an eager AdamW
                    st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
                todo.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"]))
            b1, b2 = group["betas"]
            for s in range(0, len(todo), _MAX):
                chunk = todo[s:s + _MAX]
                dev = chunk[0][0].device
                ws = self._ws.get(dev)
                if ws is None:
                    ws = self._ws[dev] = torch.empty(lib.lipvq_adamw_workspace_bytes() // 4, dtype=torch.float32, device=dev)
                n = len(chunk)
                arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in chunk])
                numels = (C.c_int64 * n)(*[t[0].numel() for t in chunk])
                with _on(dev):
                    check(lib.lipvq_adamw_f32(arr(0), arr(1), arr(2), arr(3), arr(4), numels, n, float(group["lr"]), float(b1),
                                              float(b2), float(group["eps"]), float(group["weight_decay"]), ws.data_ptr(),
                                              _stream()), "lipvq_adamw_f32")
                # the kernel wrote the parameters behind autograd's back: bump their version counters, as an in-place torch op
                # would (the tokenizer's packed-weight / prepared-codebook caches and autograd's saved-tensor checks key on them)
                for t in chunk:
                    torch.autograd.graph.increment_version(t[0])
        return loss
