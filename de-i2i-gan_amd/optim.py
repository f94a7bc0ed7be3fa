"""Fused multi-tensor Adam on the HIP kernel (csrc/adam.hip): one launch per optimizer step.

A ``torch.optim.Optimizer`` subclass, so ``lr_scheduler``s, ``param_groups`` / ``add_param_group`` and
``torch.amp.GradScaler`` keep working the way the reference's trainers use them (trainers/base_trainer.py:68-126,
mae_trainer.py:28,139-158).  Semantics = torch.optim.Adam (no amsgrad), or torch.optim.AdamW with ``weight_decay > 0``
(decoupled: p *= 1 - lr*wd before the update): parameters whose ``grad`` is None are skipped and get no state."""
import ctypes
import math

import torch

from . import _lib as L
from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, weight_decay=0.0, decoupled=True):
        """``decoupled=False``: torch.optim.Adam's weight decay (L2: the gradient becomes g + wd * p before the moments -- what
        stargan-v2's optimizers use, core/solver.py:52-56) instead of AdamW's"""
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self.decoupled = bool(decoupled)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                ops._require_gpu(p, "FusedAdam")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise TypeError("FusedAdam expects fp32 parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                by_step.setdefault(st["step"], []).append(p)
            b1, b2 = group["betas"]
            wd = float(group.get("weight_decay", 0.0))
            for t, plist in by_step.items():
                if wd > 0.0 and not self.decoupled:          # coupled L2 decay: one multi-tensor launch on the gradients, then plain Adam
                    if self.grad_scale != 1.0:
                        raise NotImplementedError("coupled weight decay with a gradient scale")
                    torch._foreach_add_([p.grad for p in plist], [p.data for p in plist], alpha=wd)
                self._launch(plist, t, float(group["lr"]), b1, b2, group["eps"], wd if self.decoupled else 0.0)
        return loss

    def _launch(self, plist, t, lr, b1, b2, eps, weight_decay=0.0):
        dev = plist[0].device
        lib = ops._lib_for(plist[0])
        n = len(plist)
        rows = []
        max_n = 0
        for p in plist:
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if not p.is_contiguous():
                raise RuntimeError("FusedAdam: non-contiguous parameter")
            st = self.state[p]
            rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
            max_n = max(max_n, p.numel())
            p._dei2i_keep = g          # keep a possibly-copied grad alive until the launch is enqueued
        # the device pointer table is rebuilt and uploaded only when a pointer moved (in steady state the caching allocator hands
        # every gradient the address it had the step before)
        cache = self.__dict__.setdefault("_table_cache", {})
        key = (dev, n, rows[0][0])
        hit = cache.get(key)
        if hit is not None and hit[0] == rows:
            table_dev = hit[1]
        else:
            table = torch.empty((n, 5), dtype=torch.int64, pin_memory=True)
            table.copy_(torch.tensor(rows, dtype=torch.int64))
            table_dev = table.to(dev, non_blocking=True)
            cache[key] = (rows, table_dev)
        L.check(lib.dei2i_adam_step(ctypes.c_void_p(table_dev.data_ptr()), n, max_n, lr, b1, b2, eps, 1.0 - b1 ** t,
                                    math.sqrt(1.0 - b2 ** t), self.grad_scale, weight_decay, ops._stream()), "adam_step")
        for p in plist:
            p._dei2i_epoch = getattr(p, "_dei2i_epoch", 0) + 1      # raw-pointer update: invalidate packed copies
            p._dei2i_keep = None


class _FusedPlain(torch.optim.Optimizer):
    """torch.optim.SGD / torch.optim.RMSprop as trainers/base_trainer.py:71-74 constructs them (``optim_cls(params, lr=...)``: no
    momentum, no weight decay; RMSprop alpha 0.99, eps 1e-8, not centered) on csrc/adam.hip's second kernel: one launch per step,
    parameters whose ``grad`` is None skipped (no state), ``grad_scale`` like FusedAdam (the data-parallel 1/world)."""
    KIND = None

    def __init__(self, params, lr, grad_scale=1.0, **defaults):
        if lr < 0.0:
            raise ValueError("invalid learning rate")
        super().__init__(params, dict(lr=lr, **defaults))
        self.grad_scale = float(grad_scale)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            dev = plist[0].device
            lib = ops._lib_for(plist[0])
            rows, max_n = [], 0
            for p in plist:
                ops._require_gpu(p, type(self).__name__)
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError(f"{type(self).__name__} expects contiguous fp32 parameters and fp32 gradients")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if self.KIND == 1 and len(st) == 0:
                    st["step"] = 0
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self.KIND == 1:
                    st["step"] += 1
                aux = st["square_avg"].data_ptr() if self.KIND == 1 else 0
                rows.append((p.data_ptr(), g.data_ptr(), aux, 0, p.numel()))
                max_n = max(max_n, p.numel())
                p._dei2i_keep = g
            table = torch.tensor(rows, dtype=torch.int64).to(dev)
            L.check(lib.dei2i_sgd_rmsprop_step(ctypes.c_void_p(table.data_ptr()), len(rows), max_n, self.KIND, float(group["lr"]),
                                               float(group.get("alpha", 0.0)), float(group.get("eps", 0.0)), self.grad_scale,
                                               ops._stream()), "sgd_rmsprop_step")
            for p in plist:
                p._dei2i_epoch = getattr(p, "_dei2i_epoch", 0) + 1      # raw-pointer update: invalidate packed copies
                p._dei2i_keep = None
            self._table_keep = table           # (alive until the next step's launch is enqueued)
        return loss


class FusedSGD(_FusedPlain):
    KIND = 0

    def __init__(self, params, lr, grad_scale=1.0):
        super().__init__(params, lr, grad_scale)


class FusedRMSprop(_FusedPlain):
    KIND = 1

    def __init__(self, params, lr, alpha=0.99, eps=1e-8, grad_scale=1.0):
        super().__init__(params, lr, grad_scale, alpha=alpha, eps=eps)
