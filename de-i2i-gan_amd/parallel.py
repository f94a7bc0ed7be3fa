"""Data-parallel minibatch sharding: one process per GPU, gradient all-reduce over RCCL/xGMI overlapped with backward.

The reference is single-device; this is the build's addition (section 8e).  Design for the
MI355X node (8 GPUs, point-to-point xGMI, no switch):

* Each rank runs the full D step and G step on its shard of the batch; the only exchange is SUM all-reduce of the
  gradients that step produced.  The 1/world averaging is folded into the fused Adam kernel's ``grad_scale`` -- no
  extra pass over the gradients.
* Overlap: a ``post_accumulate_grad`` hook fires as soon as autograd has finished a parameter's gradient.  Large
  gradients (>= ``direct_bytes``) are all-reduced in place immediately, small ones are coalesced into flat buckets of
  ``bucket_bytes``.  Collectives are issued on a dedicated side stream that waits on an event recorded on the compute
  stream, so RCCL traffic runs under the remaining backward kernels.  Backward reaches the discriminator's 134 MB
  ``enc_blk.5`` weight gradient first, so the biggest message has the whole rest of backward to hide behind.
* Every rank sees the same autograd graph, so hooks fire in the same order on every rank and the collective
  sequence matches without negotiation.  Parameters that get no gradient in a step (the never-executed
  ``norm_s`` / ``conv_s``; all of D during the G step) simply never enter a collective.
* BatchNorm statistics stay local (per-shard), like torch DDP without SyncBN; ``broadcast_buffers`` copies rank 0's
  running stats to every rank (torch DDP ``broadcast_buffers`` semantics) -- parity definition in SURVEY.md 8e.
* ``attach_ddp`` broadcasts rank 0's parameters and buffers (spectral-norm u / v included) once, one flat message per
  network, like torch DDP does at construction: replicas start identical whatever each rank's seed or checkpoint was.
* ``measure=True`` brackets every collective with events on the side stream and marks the end of backward on the compute
  stream; ``overlap_report()`` then says how much of the exchange ran under backward kernels (bench.py prints it).
* ``comm_dtype="bf16"``: the messages travel as bf16 (half the bytes on the per-link-bound xGMI rings: D's step is 180 MB in fp32,
  134 MB of it one tensor): each gradient is rounded to bf16 on the side stream, summed by RCCL in bf16, and widened back into
  the fp32 gradient the optimizer reads; the 1/world stays in Adam.  One rounding of every addend and of every partial sum
  (2^-9 relative each): tests/test_ddp_gloo.py bounds the error against the fp32 exchange.  Default fp32 (bit-exact sums).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import time

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, process_group=None, bucket_bytes: int = 16 << 20, direct_bytes: int = 4 << 20, overlap: bool = True,
                 force_collectives: bool = False, measure: bool = False, comm_dtype: str = "fp32"):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_bytes, self.direct_bytes, self.overlap = bucket_bytes, direct_bytes, overlap
        self.active = self.world > 1 or force_collectives      # force: run the collective path on a 1-rank group (tests)
        self._attached = set()
        self._side: Dict[torch.device, torch.cuda.Stream] = {}
        self._bucket: List[torch.Tensor] = []
        self._bucket_nbytes = 0
        self._inflight = []          # (work, flat, [grads]) to finish in reduce()
        self.stats = {"collectives": 0, "bytes": 0}
        self.measure = measure
        if comm_dtype not in ("fp32", "bf16"):
            raise ValueError(f"comm_dtype [{comm_dtype}] is not supported (fp32 | bf16)")
        self.comm_dtype = torch.bfloat16 if comm_dtype == "bf16" else torch.float32
        self._spans = []             # measure: (start event, end event) of every collective, on the side stream
        self._bwd_done = []          # measure: compute-stream event at the entry of reduce() = backward's last kernel
        self._span_marks = []        # index into _spans at each reduce() call

    # ---- wiring ------------------------------------------------------------------------------------------
    def attach(self, net: torch.nn.Module) -> None:
        """Register the gradient-ready hooks (once per network)."""
        if id(net) in self._attached:
            return
        self._attached.add(id(net))
        if not self.overlap:
            return
        for p in net.parameters():
            p.register_post_accumulate_grad_hook(self._on_grad_ready)

    def _comm_stream(self, device):
        if device.type != "cuda":
            return None
        s = self._side.get(device)
        if s is None:
            s = torch.cuda.Stream(device=device)
            self._side[device] = s
        return s

    # ---- hook path (during backward) ---------------------------------------------------------------------
    def _on_grad_ready(self, p: torch.Tensor) -> None:
        g = p.grad
        if g is None or not self.active:
            return
        nbytes = g.numel() * g.element_size()
        if nbytes >= self.direct_bytes:
            self._launch([g], flat=None)
        else:
            self._bucket.append(g)
            self._bucket_nbytes += nbytes
            if self._bucket_nbytes >= self.bucket_bytes:
                self._flush_bucket()

    def _flush_bucket(self) -> None:
        if self._bucket:
            grads, self._bucket, self._bucket_nbytes = self._bucket, [], 0
            self._launch(grads, flat=True)

    def _launch(self, grads: List[torch.Tensor], flat) -> None:
        t_host = time.perf_counter()
        try:
            self._launch_inner(grads, flat)
        finally:
            self.stats["launch_host_ms"] = self.stats.get("launch_host_ms", 0.0) + 1e3 * (time.perf_counter() - t_host)

    def _launch_inner(self, grads: List[torch.Tensor], flat) -> None:
        dev = grads[0].device
        side = self._comm_stream(dev)
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))      # the gradient's producer kernels
            side.wait_event(ready)
            from . import ops
            wg = ops.wgrad_stream(dev)                        # ... and the weight gradients computed on ops' side stream
            if wg is not None:
                side.wait_stream(wg)
            ctx = torch.cuda.stream(side)
        else:
            ctx = _NullCtx()
        with ctx:
            if self.measure and side is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(side)
            narrow = self.comm_dtype != grads[0].dtype
            if flat:
                buf = torch.cat([g.reshape(-1) for g in grads])
                for g in grads:
                    if side is not None:
                        g.record_stream(side)
                if narrow:
                    buf = buf.to(self.comm_dtype)
            else:
                buf = grads[0]
                if side is not None:
                    buf.record_stream(side)
                if narrow:
                    buf, flat = buf.reshape(-1).to(self.comm_dtype), True       # (copied back into the fp32 gradient in reduce())
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if self.measure and side is not None:
                work.wait()                       # stream-ordered on the side stream (no host block for NCCL work)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(side)
                self._spans.append((e0, e1))
        self.stats["collectives"] += 1
        self.stats["bytes"] += buf.numel() * buf.element_size()
        self._inflight.append((work, buf if flat else None, grads))

    # ---- after backward ----------------------------------------------------------------------------------
    def reduce(self, net: torch.nn.Module) -> None:
        """Finish the step's gradient exchange: after this returns (stream-ordered), every ``p.grad`` holds the SUM
        over ranks.  Use ``FusedAdam.grad_scale = 1/world`` (set by ``attach_ddp``) for the average."""
        if not self.active:
            return
        t_host = time.perf_counter()
        if self.measure and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self._bwd_done.append(ev)
        if id(net) not in self._attached or not self.overlap:
            for p in net.parameters():                      # no-overlap path: same bucketing, issued now
                if p.grad is not None:
                    self._on_grad_ready(p)
        self._flush_bucket()
        for work, flat, grads in self._inflight:
            dev = grads[0].device
            side = self._comm_stream(dev)
            ctx = torch.cuda.stream(side) if side is not None else _NullCtx()
            with ctx:
                work.wait()
                if flat is not None:                    # bucket (or narrowed message) -> the fp32 gradients, ONE multi-tensor launch
                    views, off = [], 0
                    for g in grads:
                        n = g.numel()
                        views.append(flat[off:off + n].view_as(g))
                        off += n
                    torch._foreach_copy_(grads, views)
        for dev, side in self._side.items():
            torch.cuda.current_stream(dev).wait_stream(side)    # optimizer kernels run after the reduced grads land
        self._inflight = []
        self.stats["reduce_host_ms"] = self.stats.get("reduce_host_ms", 0.0) + 1e3 * (time.perf_counter() - t_host)
        if self.measure:
            self._span_marks.append(len(self._spans))

    def overlap_report(self):
        """(measure=True; call after a device synchronise)  Per optimizer step: time the collectives occupied the side
        stream, and the part of it that ran AFTER backward's last kernel (exposed: the optimizer waits for it)."""
        if not self._bwd_done:
            return None
        comm = exposed = 0.0
        lo = 0
        for ev, hi in zip(self._bwd_done, self._span_marks):
            for e0, e1 in self._spans[lo:hi]:
                comm += e0.elapsed_time(e1)
            if hi > lo:
                exposed += max(0.0, ev.elapsed_time(self._spans[hi - 1][1]))
            lo = hi
        n = len(self._bwd_done)
        self._spans, self._bwd_done, self._span_marks = [], [], []
        return {"backward_passes": n, "comm_ms_per_pass": comm / n, "exposed_ms_per_pass": exposed / n,
                "overlap_frac": (1.0 - exposed / comm) if comm > 0 else None}

    def broadcast_parameters(self, net: torch.nn.Module, src: int = 0) -> None:
        """Rank ``src``'s parameters and buffers to every rank, one flat message per dtype (torch DDP does this when it
        wraps a module).  Parameters are written through ``.data`` and stamped so that packed weight copies refresh."""
        if self.world <= 1:
            return
        with torch.no_grad():
            tensors = [p.data for p in net.parameters()] + [b for b in net.buffers()]
            for dt in sorted({t.dtype for t in tensors}, key=str):
                group = [t for t in tensors if t.dtype == dt]
                flat = torch.cat([t.reshape(-1) for t in group])
                dist.broadcast(flat, src=src, group=self.pg)
                off = 0
                for t in group:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n
            for p in list(net.parameters()) + list(net.buffers()):
                p._dei2i_epoch = getattr(p, "_dei2i_epoch", 0) + 1

    def broadcast_buffers(self, net: torch.nn.Module, src: int = 0) -> None:
        bufs = [b for b in net.buffers() if b.is_floating_point()]
        if not self.active or not bufs:
            return
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.broadcast(flat, src=src, group=self.pg)
        views, off = [], 0
        for b in bufs:
            n = b.numel()
            views.append(flat[off:off + n].view_as(b))
            off += n
        with torch.no_grad():
            torch._foreach_copy_(bufs, views)              # one multi-tensor launch (18 buffers in the generator)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def attach_ddp(trainer, process_group=None, **kw) -> GradReducer:
    """Make a DefectGanTrainer data-parallel: gradients are summed across ranks after each backward and averaged
    inside the fused Adam kernel; generator BatchNorm buffers follow rank 0."""
    red = GradReducer(process_group, **kw)
    for name, net in trainer.model.networks.items():
        red.attach(net)
        red.broadcast_parameters(net)
        trainer.optimizers[name].grad_scale = 1.0 / red.world
    mask_token = getattr(trainer.model, "mask_token", None)         # MAE stage: trained by the generator's optimizer
    if isinstance(mask_token, torch.nn.Module):
        red.attach(mask_token)
        red.broadcast_parameters(mask_token)
    trainer.reducer = red
    return red
