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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)

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
            for t, plist in by_step.items():
                self._launch(plist, t, float(group["lr"]), b1, b2, group["eps"], float(group.get("weight_decay", 0.0)))
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
