"""Autograd bindings of the HIP kernels (libdei2i_hip.so).  One ``torch.autograd.Function`` per fused op.

Internal activations are NHWC tensors of the compute dtype (bf16, or f32 in parity mode) whose last dimension is the
channel count padded to a 16-byte vector.  NCHW fp32 appears only at the module boundary (inputs, outputs,
parameters, gradients, state_dict), as in the reference.  Every op requires a GPU tensor; there is no fallback.
"""
from __future__ import annotations

import ctypes
import weakref
from ctypes import byref, c_int, c_void_p
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L

ACT = {"none": L.ACT_NONE, None: L.ACT_NONE, "relu": L.ACT_RELU, "leaky_relu": L.ACT_LRELU}


@dataclass(frozen=True)
class Precision:
    name: str
    code: int
    dtype: torch.dtype
    vec: int

    def pad(self, c: int) -> int:
        return (c + self.vec - 1) // self.vec * self.vec


BF16 = Precision("bf16", L.BF16, torch.bfloat16, 8)
F32 = Precision("f32", L.F32, torch.float32, 4)


# ---- fp8 forward mode (BASELINE.json configs[4]) ----------------------------------------------------------------
# compute_dtype "fp8": tensors, backward and every other kernel are those of the bf16 mode; inside an fp8_forward(True)
# scope the FORWARD GEMM of the stride-1 3x3 convolutions the fp8 kernel takes runs on e4m3 operands (activations
# quantised with the fixed scale FP8_ACT_SCALE -- they follow a normalisation layer, |x| = O(1) -- weights with
# 448 / amax|w| computed on the device).  Layers the fp8 kernel does not take run in bf16 as usual (that is the bf16
# product path, not a fallback of a failed fp8 call: eligibility is asked first).
FP8_ACT_SCALE = 16.0
_fp8_forward = False


class fp8_forward:
    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global _fp8_forward
        self.prev, _fp8_forward = _fp8_forward, self.on
        return self

    def __exit__(self, *exc):
        global _fp8_forward
        _fp8_forward = self.prev
        return False


# fused conv + norm + act (halo-resident kernels).  Run-time switches for same-box A/B timing and for the parity tests of
# fused against unfused:
#   fuse_norm : statistics of a conv's / affine kernel's output from its own epilogue (no separate moments pass)  -- ON
#   fuse_pro  : BatchNorm / SPADE apply on the CONSUMER conv's operand path (the normalised tensor is never written) -- OFF:
#               measured at 256x256 batch 16 (profiles/r02_b_*): the in-LDS transform costs +33 us on a 99 us conv and
#               +34 us on a 91 us wgrad (it is repeated per output-channel tile and again in the wgrad, in kernels whose
#               VALU slots compete with the MFMA issue) against ~39 us of statistics + modulate kernels saved: +1.4 ms/step.
#   fuse_ring : SPADE -> nearest x2 upsample -> conv keeps the normalised tensor at the SOURCE resolution (+ the logical
#               image's 2-pixel frame in a compact ring tensor): the 4x larger upsampled tensor is never written or read -- ON
#   fuse_bwd  : the backward REDUCTIONS of a SPADE / BatchNorm layer (sum dxhat, sum dxhat*xhat, ... per channel) from the epilogue
#               of the input-gradient launch of the conv behind it, while the dz tile is on chip -- the streaming pass that read
#               dz and x again (spade_bwd_partial / bn_bwd_partial) is not launched -- ON
fuse_norm = True
fuse_pro = False
fuse_ring = True
fuse_bwd = True

_fp8_stash = []          # e4m3 copy produced by the last normalisation kernel, handed to its output tensor by the wrapper


def _fp8_copy_wanted(prec, c: int) -> bool:
    return _fp8_forward and prec is BF16 and c % 128 == 0


def _attach_fp8(out):
    if _fp8_stash:
        out._dei2i_fp8 = _fp8_stash.pop()
    return out


def wants_fp8(name) -> bool:
    return name in ("fp8", "fp8_e4m3", "e4m3")


def get_precision(name) -> Precision:
    if isinstance(name, Precision):
        return name
    if name in (None, "bf16", "bfloat16") or wants_fp8(name):
        return BF16
    if name in ("f32", "fp32", "float32"):
        return F32
    raise ValueError(f"compute dtype [{name}] is not supported (bf16 | f32)")


def precision_of(t: torch.Tensor) -> Precision:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"de-i2i-gan_amd.{what}: tensor is on {t.device}; the HIP kernels need a GPU tensor "
                           "(there is no CPU fallback)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


# The current stream's handle straight from the C layer: ``torch.cuda.current_stream()`` builds a Stream object per call (~8 us of host
# time, several hundred calls per step -- the short steps (MAE stage, recipe options) are host-bound).
_raw_stream_of = torch._C._cuda_getCurrentRawStream
_current_device = torch._C._cuda_getDevice


def _stream_id(device=None) -> int:
    """hipStream_t (as an int) of the current stream of ``device`` (None / index-less: the current device)"""
    idx = getattr(device, "index", device)
    return _raw_stream_of(_current_device() if idx is None else idx)


def _stream():
    return c_void_p(_raw_stream_of(_current_device()))


_initialised = set()


def _lib_for(t: torch.Tensor):
    lib = L.load()
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if dev not in _initialised:
        L.check(lib.dei2i_init(dev), "init")
        _initialised.add(dev)
    return lib


_workspaces = {}


def _workspace(device, nbytes: int, slot: str = "splitk") -> torch.Tensor:
    """Scratch of one kernel launch (split-K slabs, the padded dgrad frame, ...), reused launch after launch ON ONE STREAM: every
    stream that runs ops has its own set (two chains of a forked pass -- ``forked_chains`` -- must not share slabs)."""
    key = (torch.device(device), slot)
    sid = _stream_id(torch.device(device))
    if sid != _default_stream_id(key[0]):
        key = key + (sid,)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


_default_stream_ids = {}


def _default_stream_id(device):
    sid = _default_stream_ids.get(device)
    if sid is None:
        sid = _default_stream_ids[device] = torch.cuda.default_stream(device).cuda_stream
    return sid


# ---- in-place accumulation of parameter gradients within one backward pass --------------------------------------
# The G step applies every generator layer four times (defectgan_model.py:185-190), so autograd would receive four
# gradients per parameter and sum them with one add kernel each (171 launches per step).  Instead the first backward
# node of a parameter in a pass returns a fresh gradient tensor and remembers it under the pass's graph-task id; the
# later nodes of the same pass add into that memory inside their own kernels (wgrad reduce, BatchNorm finalize) and
# return None -- autograd treats an undefined gradient as "no contribution" and still runs the parameter's
# AccumulateGrad (and its post-accumulate hooks: the data-parallel reducer) once, after its last node.
# Only a WEAK reference is kept: the tensor lives exactly as long as autograd's input buffer holds it, so
# AccumulateGrad can still steal it (no copy), and a later node that finds the reference dead -- the engine dropped
# the gradient (torch.autograd.grad(inputs=[activation]): the accumulator is not executed) or replaced it by an
# out-of-place sum with another op's contribution -- simply returns a fresh tensor of its own, which is always correct.
# Non-leaf weights (concatenated gamma|beta) are not tracked.
_grad_slots = {}


def _grad_target(param, shape, device, stream=None):
    """-> (tensor to return to autograd or None, its address, accumulate flag); ``stream``: the stream the caller's kernels run on
    when that is not the current one (the weight-gradient side stream)"""
    tid = torch._C._current_graph_task_id()
    key = param.data_ptr() if (param.is_leaf and tid >= 0) else None
    # (in-place accumulation is a read-modify-write ordered by ONE stream: a node that runs on another stream -- the two chains of
    #  a forked pass -- returns its own tensor and autograd sums them, with its own stream synchronisation)
    sid = 0
    if key is not None and torch.device(device).type == "cuda":
        sid = stream.cuda_stream if stream is not None else _stream_id(torch.device(device))
    if key is not None:
        slot = _grad_slots.get(key)
        if slot is not None and slot[0] == tid and slot[2] == tuple(shape) and slot[3] == sid:
            first = slot[1]()
            if first is not None:                       # still the tensor in autograd's input buffer
                return None, first.data_ptr(), 1
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if key is not None:
        _grad_slots[key] = (tid, weakref.ref(t), tuple(shape), sid)
    return t, t.data_ptr(), 0


def _wants_grad(ctx, idx: int) -> bool:
    """needs_input_grad[idx], and the engine will actually consume that gradient in this pass (False for a parameter
    under torch.autograd.grad(inputs=[something else]): its wgrad would be computed and dropped)."""
    if not ctx.needs_input_grad[idx]:
        return False
    node = ctx.next_functions[idx][0]
    if node is None:
        return True
    try:
        return bool(torch._C._will_engine_execute_node(node))
    except RuntimeError:
        # torch refuses the query for a LEAF that torch.autograd.grad(inputs=[that leaf]) captures -- its gradient is wanted
        return True


# ---- spectral normalisation of a conv weight (csrc/spectral.hip) --------------------------------------------------
class _SpectralWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight_orig, u, v, iterate: bool):
        _require_gpu(weight_orig, "spectral_weight")
        w = weight_orig.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise TypeError("spectral_weight expects a contiguous fp32 weight")
        cout, k = w.shape[0], w[0].numel()
        lib = _lib_for(w)
        dev = w.device
        scratch = torch.empty(lib.dei2i_spectral_scratch_floats(cout, k), dtype=torch.float32, device=dev)
        u_used = torch.empty(cout, dtype=torch.float32, device=dev)
        v_used = torch.empty(k, dtype=torch.float32, device=dev)
        scal = torch.empty(4, dtype=torch.float32, device=dev)
        w_eff = torch.empty_like(w)
        L.check(lib.dei2i_spectral_fwd(cout, k, _p(w), _p(u), _p(v), 1 if iterate else 0, _p(scratch), _p(u_used), _p(v_used),
                                       _p(scal), _p(w_eff), _stream()), "spectral_fwd")
        if iterate:                      # the buffers were updated in place through raw pointers: bump their cache stamps
            u._dei2i_epoch = getattr(u, "_dei2i_epoch", 0) + 1
            v._dei2i_epoch = getattr(v, "_dei2i_epoch", 0) + 1
        ctx.save_for_backward(w_eff, u_used, v_used, scal)
        ctx.param = weight_orig
        return w_eff

    @staticmethod
    def backward(ctx, g):
        w_eff, u_used, v_used, scal = ctx.saved_tensors
        g = g.contiguous()
        cout, k = w_eff.shape[0], w_eff[0].numel()
        lib = _lib_for(w_eff)
        scratch = torch.empty(1024, dtype=torch.float32, device=g.device)
        # the same parameter's later forwards of this backward pass add into the first one's tensor (see _grad_target)
        dw, dw_ptr, accumulate = _grad_target(ctx.param, w_eff.shape, g.device)
        L.check(lib.dei2i_spectral_bwd(cout, k, _p(g), _p(w_eff), _p(u_used), _p(v_used), _p(scal), _p(scratch), c_void_p(dw_ptr),
                                       accumulate, _stream()), "spectral_bwd")
        return dw, None, None, None


def spectral_weight(weight_orig, u, v, iterate: bool):
    """weight_orig / sigma(weight_orig, u, v) with torch.nn.utils.spectral_norm's semantics; ``iterate`` runs one power
    iteration on the (u, v) buffers in place first (training mode)."""
    return _SpectralWeight.apply(weight_orig, u, v, bool(iterate))


# ---- eval-mode BatchNorm folded into the conv in front of it ------------------------------------------------------
# The generator passes of the D step (defectgan_model.py:251-262) and inference run BatchNorm on its RUNNING statistics: a fixed
# per-channel affine.  Folded into the conv's weights (and a bias), BN + LeakyReLU happen in the conv's own epilogue and the
# BatchNorm-apply pass over the conv's output -- a read and a write of the largest tensors of the pass -- is not run.
fold_eval_bn = True


def fold_bn_weight(weight, bn_weight, bn_bias, running_mean, running_var, eps: float):
    """-> (w_eff = a[co] * weight[co], b_eff = bn_bias - running_mean * a) with a = bn_weight * rsqrt(running_var + eps); no
    autograd (eval / no-grad passes only).  ``w_eff`` is marked as derived per call: its packed copy travels with the call."""
    _require_gpu(weight, "fold_bn_weight")
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        raise TypeError("fold_bn_weight expects a contiguous fp32 weight")
    cout, k = w.shape[0], w[0].numel()
    w_eff, b_eff = torch.empty_like(w), torch.empty(cout, dtype=torch.float32, device=w.device)
    L.check(_lib_for(w).dei2i_fold_bn_weight(cout, k, _p(w), _p(bn_weight.detach()), _p(bn_bias.detach()), _p(running_mean), _p(running_var),
                                             float(eps), _p(w_eff), _p(b_eff), _stream()), "fold_bn_weight")
    w_eff._dei2i_per_call = True
    return w_eff, b_eff


# ---- NoiseInjection's draw (architecture.py:385-389): N(0,1) on the activations' device; tests install a provider ----
noise_source = None


def draw_noise(shape, device):
    if noise_source is not None:
        return noise_source(tuple(shape))
    return torch.randn(shape, device=device)


class _NoiseInject(torch.autograd.Function):
    """x + weight * noise on an NHWC activation (one noise value per pixel, one scalar weight): one HIP launch forward;
    backward hands dy through and reduces dweight = sum(noise * rowsum(dy)) through ordered block partials."""

    @staticmethod
    def forward(ctx, x, weight, noise):
        _require_gpu(x, "noise_inject")
        prec = precision_of(x)
        x = x.contiguous()
        rows, c = x.numel() // x.shape[-1], x.shape[-1]
        noise = noise.detach().to(device=x.device, dtype=torch.float32).contiguous()
        if noise.numel() != rows:
            raise ValueError(f"noise_inject: {noise.numel()} noise values for {rows} pixels")
        w = weight.detach().reshape(1)
        out = torch.empty_like(x)
        lib = _lib_for(x)
        L.check(lib.dei2i_noise_fwd(prec.code, rows, c, _p(x), _p(noise), _p(w), _p(out), _stream()), "noise_fwd")
        ctx.prec, ctx.wshape, ctx.param = prec, weight.shape, weight
        ctx.save_for_backward(noise)
        return out

    @staticmethod
    def backward(ctx, dy):
        (noise,) = ctx.saved_tensors
        dw = None
        if _wants_grad(ctx, 1):
            dy = dy.contiguous()
            rows, c = noise.numel(), dy.shape[-1]
            part = torch.empty(1024, dtype=torch.float32, device=dy.device)
            dw, dw_ptr, accumulate = _grad_target(ctx.param, ctx.wshape, dy.device)
            L.check(_lib_for(dy).dei2i_noise_bwd(ctx.prec.code, rows, c, _p(dy), _p(noise), _p(part), c_void_p(dw_ptr), accumulate,
                                                 _stream()), "noise_bwd")
        return (dy if ctx.needs_input_grad[0] else None), dw, None


def noise_inject(x, weight, noise):
    """NoiseInjection on an NHWC activation: ``noise`` holds one value per pixel ((N,1,H,W) as the reference draws it)."""
    return _NoiseInject.apply(x, weight, noise)


_const_vecs = {}


def _const_vec(device, n: int, value: float) -> torch.Tensor:
    key = (device, n, value)
    v = _const_vecs.get(key)
    if v is None:
        v = torch.full((n,), value, dtype=torch.float32, device=device)
        _const_vecs[key] = v
    return v


# --------------------------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConvGeom:
    """nn.Conv2d geometry of one reference conv (architecture.py:51-56,95-100,228-233; normalization.py:17-22)."""
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    reflect: bool = False
    up: bool = False      # nearest x2 upsample fused in front of the conv (architecture.py:203)


_descs = {}


def _desc(prec: Precision, g: ConvGeom, n: int, h: int, w: int, cins: int, couts: int) -> L.ConvDesc:
    """The conv descriptor of (geometry, shape): made once (a ctypes structure costs ~3 us to build, a step asks ~500 times; the
    library only reads it)."""
    key = (prec.code, g, n, h, w, cins, couts)
    d = _descs.get(key)
    if d is None:
        if len(_descs) > 4096:
            _descs.clear()
        d = _descs[key] = L.ConvDesc(prec.code, n, h, w, g.cin, g.cout, cins, couts, g.k, g.k, g.stride, g.pad,
                                     L.PAD_REFLECT if g.reflect else L.PAD_ZERO, 1 if g.up else 0)
    return d


class PackedWeights:
    """Kernel-layout copies of one conv weight: forward [Cout][k*k][CinS] and dgrad (per stride-parity class
    [Cin][taps][CoutS]).  Re-packed lazily when any source parameter changed (autograd version counter, or the
    ``_dei2i_epoch`` stamp the fused Adam sets because it updates parameters through raw pointers)."""

    def __init__(self):
        self._key = None
        self.fwd = None
        self.dgrad = None
        self.fp8 = None            # (packed e4m3 forward weights, dequant scalar) of the fp8 forward mode
        self._packed_on = None     # (stream id, event) of the last pack launch: another stream waits for it before reading

    def _mark_packed(self):
        # (the event object is made once per cache: constructing one per pack launch -- every conv weight after every optimizer step --
        #  was ~20 us of host time each; re-recording moves it to the newest pack launch, which is the one a reader has to wait for)
        ev = self._packed_on[1] if self._packed_on is not None else torch.cuda.Event()
        ev.record()
        self._packed_on = (_stream_id(), ev)

    def _await_pack(self):
        if self._packed_on is not None and self._packed_on[0] != _stream_id():
            torch.cuda.current_stream().wait_event(self._packed_on[1])

    @staticmethod
    def _stamp(t: torch.Tensor):
        return (t.data_ptr(), t._version, getattr(t, "_dei2i_epoch", 0))

    def get(self, weight: torch.Tensor, sources, prec: Precision, geom: ConvGeom, cins: int, couts: int, need_dgrad: bool,
            need_fwd: bool = True, per_call: bool = False):
        # per_call: the weight is DERIVED from the sources anew on every forward and differs between calls whose sources
        # carry the same stamps by the time backward runs (spectral norm: u, v are iterated in place by later forwards)
        # -- its own address identifies the call, and the sources' stamps (moved by those later forwards) must not
        key = ((), prec.code, cins, couts, self._stamp(weight)) if per_call else \
            (tuple(self._stamp(s) for s in sources), prec.code, cins, couts, None)
        if key == self._key and (self.fwd is not None or not need_fwd) and (self.dgrad is not None or not need_dgrad):
            self._await_pack()        # the hit: nothing to build (this is the call of every conv of every pass)
            return self.fwd, self.dgrad
        lib = _lib_for(weight)
        if key != self._key:
            self._key, self.fwd, self.dgrad, self.fp8 = key, None, None, None
        d = _desc(prec, geom, 1, max(geom.k, 4), max(geom.k, 4), cins, couts)
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        self._await_pack()            # (copies packed by a launch on ANOTHER stream: the two chains of a forked pass share weights)
        packed = False
        if need_fwd and need_dgrad and self.fwd is None and self.dgrad is None:       # the training path: one launch
            self.fwd = torch.empty(lib.dei2i_packed_fwd_elems(byref(d)), dtype=prec.dtype, device=weight.device)
            self.dgrad = torch.empty(lib.dei2i_packed_dgrad_elems(byref(d)), dtype=prec.dtype, device=weight.device)
            L.check(lib.dei2i_pack_weight_both(byref(d), _p(w), _p(self.fwd), _p(self.dgrad), _stream()), "pack_weight_both")
            packed = True
        if need_fwd and self.fwd is None:
            self.fwd = torch.empty(lib.dei2i_packed_fwd_elems(byref(d)), dtype=prec.dtype, device=weight.device)
            L.check(lib.dei2i_pack_weight_fwd(byref(d), _p(w), _p(self.fwd), _stream()), "pack_weight_fwd")
            packed = True
        if need_dgrad and self.dgrad is None:
            self.dgrad = torch.empty(lib.dei2i_packed_dgrad_elems(byref(d)), dtype=prec.dtype, device=weight.device)
            L.check(lib.dei2i_pack_weight_dgrad(byref(d), _p(w), _p(self.dgrad), _stream()), "pack_weight_dgrad")
            packed = True
        if packed:
            self._mark_packed()
        return self.fwd, self.dgrad

    def get_fp8(self, weight: torch.Tensor, sources, prec: Precision, geom: ConvGeom, cins: int, couts: int):
        """e4m3 forward weights (per-tensor scale 448 / amax|w|, computed on the device) and the dequant scalar
        1 / (activation scale * weight scale); cached with the same stamps as the bf16 copies."""
        key = (tuple(self._stamp(s) for s in sources), prec.code, cins, couts, None)
        if key != self._key:
            self._key, self.fwd, self.dgrad, self.fp8 = key, None, None, None
        if self.fp8 is None:
            lib = _lib_for(weight)
            d = _desc(prec, geom, 1, 8, 32, cins, couts)
            w = weight.detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.float().contiguous()
            amax = torch.linalg.vector_norm(w, ord=float("inf")).reshape(1)
            wq = torch.empty(geom.cout * geom.k * geom.k * cins, dtype=torch.uint8, device=weight.device)
            dequant = torch.empty(1, dtype=torch.float32, device=weight.device)
            L.check(lib.dei2i_pack_weight_fwd_fp8(byref(d), _p(w), _p(amax), FP8_ACT_SCALE, _p(wq), _p(dequant), _stream()),
                    "pack_weight_fwd_fp8")
            self.fp8 = (wq, dequant)
        return self.fp8


_stats_stash = []        # (partial records, records per image) of the last producer, handed to its output tensor by the wrapper


def _attach_stats(out):
    if _stats_stash:
        out._dei2i_stats = _stats_stash.pop()
    return out


def _stats_of(t, n, hw, c):
    """The statistics records a producer kernel left for tensor ``t`` ((N, chunks, 2, C) fp32, chunks), or None."""
    st = getattr(t, "_dei2i_stats", None)
    if st is None:
        return None
    partial, chunks = st
    if partial.shape != (n, chunks, 2, c) or partial.device != t.device:
        return None
    return partial, chunks


class _NormBwdHint:
    """What the input-gradient launch of a conv needs to take the backward reductions of the norm layer in FRONT of the conv in
    its epilogue (csrc/conv_halo16.hip EPIN; include/dei2i_hip.h: dei2i_epi_norm), and where it leaves them: the norm's forward
    makes one, hangs it on its output tensor (``_dei2i_bwd_hint``) and keeps it; the conv that reads that tensor picks it up;
    in backward the conv's dgrad fills ``partial`` and the norm's backward -- handed that very dz tensor -- skips its own
    streaming pass.  kind 1: SPADE class mode + ReLU (x, gb, mean, rstd, up); kind 2: BatchNorm + act (x = y, a, b, mean, rstd)."""
    __slots__ = ("kind", "x", "gb", "mean", "rstd", "a", "b", "act", "up", "group_images", "partial", "chunks", "dz", "dz_version")

    def __init__(self, kind, x, mean, rstd, gb=None, a=None, b=None, act=0, up=False, group_images=0):
        self.kind, self.x, self.mean, self.rstd, self.gb, self.a, self.b, self.act, self.up = kind, x, mean, rstd, gb, a, b, act, up
        self.group_images = group_images                  # kind 2: a / b / mean / rstd are (groups, C) -- images per group, 0: (C,)
        self.partial = self.chunks = self.dz = self.dz_version = None

    def give(self, partial, chunks, dz):
        # the dz tensor itself is held until the norm's backward has looked at it: while this reference exists autograd's input
        # buffer cannot add a second consumer's gradient INTO dz (it accumulates in place only into a buffer nobody else holds),
        # so "same storage, same version" below means "exactly what the dgrad kernel reduced"
        self.partial, self.chunks, self.dz, self.dz_version = partial, chunks, dz, dz._version

    def drop(self):
        self.partial = self.chunks = self.dz = self.dz_version = None

    def take(self, dz):
        """-> (partial, records per image) when ``dz`` is the tensor the conv's dgrad wrote them for, else None (another
        consumer's gradient was added to it, or the dgrad took a kernel without that epilogue)."""
        partial, chunks, mine, version = self.partial, self.chunks, self.dz, self.dz_version
        self.drop()
        if (partial is None or not dz.is_contiguous() or dz.data_ptr() != mine.data_ptr() or dz.shape != mine.shape
                or dz._version != version):
            return None
        bwd_fused_counts["taken"] += 1
        return partial, chunks


bwd_fused_counts = {"epilogue": 0, "taken": 0}   # dgrad launches that took the reductions / norm backwards that used them (tests)
_hint_stash = []         # the hint of the last norm forward, handed to its output tensor by the wrapper


def _attach_hint(out):
    if _hint_stash:
        out._dei2i_bwd_hint = _hint_stash.pop()
    return out


def _conv_dgrad(lib, prec, geom, x_shape, couts, g, weight, cache, sources, per_call, dtype, device, hint=None):
    """Gradient w.r.t. the conv's physical input (N, H, W, CinS) from g = dL/dy (activation already folded in).  ``hint``: the
    norm layer that produced the input (_NormBwdHint) -- its backward reductions are taken in this launch when the kernel can."""
    n, h, w, cins = x_shape
    d = _desc(prec, geom, n, h, w, cins, couts)
    _, wd = cache.get(weight, sources, prec, geom, cins, couts, need_dgrad=True, need_fwd=False, per_call=per_call)
    if hint is not None:
        hint.drop()
        up = 1 if hint.up else 0
        if (fuse_bwd and prec is BF16 and tuple(hint.x.shape) == (n, h >> up, w >> up, cins) and hint.x.is_contiguous()
                and lib.dei2i_conv2d_dgrad_norm_supported(byref(d))):
            chunks = lib.dei2i_conv2d_dgrad_norm_chunks(byref(d))
            partial = torch.empty((n, chunks, 4 if hint.kind == 1 else 2, cins), dtype=torch.float32, device=device)
            dx = torch.empty(x_shape, dtype=dtype, device=device)
            en = L.EpiNormDesc(hint.kind, up, hint.act, hint.group_images, hint.x.data_ptr(), hint.mean.data_ptr(), hint.rstd.data_ptr(),
                               hint.gb.data_ptr() if hint.gb is not None else None,
                               hint.a.data_ptr() if hint.a is not None else None,
                               hint.b.data_ptr() if hint.b is not None else None, partial.data_ptr())
            L.check(lib.dei2i_conv2d_dgrad_input_norm(byref(d), _p(g), _p(wd), _p(dx), byref(en), _stream()), "conv2d_dgrad_input_norm")
            hint.give(partial, chunks, dx)
            bwd_fused_counts["epilogue"] += 1
            return dx
    ws = _workspace(device, lib.dei2i_conv2d_workspace_bytes(byref(d)))
    dx = torch.empty(x_shape, dtype=dtype, device=device)
    ext = None
    if (geom.reflect and geom.pad > 0) or geom.up:           # the dgrad frame differs from the input: scratch
        oh, ow = c_int(), c_int()
        lib.dei2i_conv2d_dgrad_shape(byref(d), byref(oh), byref(ow))
        ext = _workspace(device, n * oh.value * ow.value * cins * dx.element_size(), slot="dgrad_frame")
    L.check(lib.dei2i_conv2d_dgrad_input(byref(d), _p(g), _p(wd), _p(ext), _p(dx), _p(ws), ws.numel() * 4, _stream()),
            "conv2d_dgrad_input")
    return dx


def _wgrad_scratch(lib, d, device, slot="wgrad"):
    packed = lib.dei2i_wgrad_slab_elems(byref(d))
    # partial slabs: up to 64, or as many as fit 96 MB (a (co, ci, 9-tap) register block per CU is 256 x 295 KB)
    return _workspace(device, max(packed * 4, min(max(packed * 4 * 64, 96 << 20), 512 << 20)), slot=slot)


# ---- weight gradients on a side stream ---------------------------------------------------------------------------
# Nothing inside a backward pass reads the gradient of a LEAF weight: it goes into the tensor autograd's AccumulateGrad will
# hand to ``param.grad`` (see _grad_target), and the optimizer reads it after the pass.  So those wgrad launches (and their slab
# reduces) need not sit between the dgrad of their layer and the normalisation backward of the previous one on the one
# stream: they run on a second stream, behind an event on the main stream (their operands), with their own split-K scratch;
# the main stream waits for the side stream once, in a callback the engine runs at the end of the pass -- before anyone can
# look at ``param.grad``.  One-workgroup-per-CU MFMA kernels (the wgrads) then overlap the HBM-bound normalisation /
# activation backward kernels and the tails / prologues of the dgrads instead of queueing behind them.  Gradients of
# non-leaf weights (spectral norm's effective weight, concatenated heads / gamma|beta weights) ARE read inside the pass by
# the next autograd node and stay on the main stream.  The data-parallel reducer makes its all-reduce stream wait for the
# side stream as well (parallel.GradReducer._launch).
wgrad_side_stream = True
_wgrad_streams = {}
_wgrad_join_pending = {}         # device -> graph-task id of the pass whose end-of-pass join is queued


def wgrad_stream(device):
    """The side stream the leaf-weight gradients of the current / last backward pass were computed on (None: none yet)."""
    return _wgrad_streams.get(torch.device(device))


def _wgrad_join(device, tid):
    def join():
        if _wgrad_join_pending.get(device) == tid:
            del _wgrad_join_pending[device]
        side = _wgrad_streams[device]
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.default_stream(device).wait_stream(side)
        for st in _chain_streams.get(torch.device(device), ()):       # (whichever stream the caller goes on with)
            st.wait_stream(side)
    return join


def _conv_wgrad(lib, prec, geom, x, g, weight, pro=None, keep=()):
    """OIHW fp32 weight gradient (in-place accumulated across the nodes of one pass: _grad_target); ``pro``: the conv's
    input was normalised on the operand path, x is the un-normalised tensor; ``keep``: the tensors ``pro`` points into."""
    n, h, w, cins = x.shape
    d = _desc(prec, geom, n, h, w, cins, g.shape[-1])
    dev = x.device
    side = None
    tid = torch._C._current_graph_task_id()
    # only when AccumulateGrad will STEAL the tensor (param.grad undefined): with a defined .grad it launches `grad += dw` on the
    # main stream in the middle of the pass, before the end-of-pass join -- those gradients stay on the main stream
    if wgrad_side_stream and weight.is_leaf and tid >= 0 and weight.grad is None:
        side = _wgrad_streams.get(dev)
        if side is None:
            side = _wgrad_streams[dev] = torch.cuda.Stream(device=dev)
    if side is not None:
        # the side stream's split-K scratch is allocated (and re-grown) ON the side stream: the caching allocator then hands a
        # dropped buffer's block only to later side-stream allocations, which run behind the kernels that still use it
        with torch.cuda.stream(side):
            scratch = _wgrad_scratch(lib, d, dev, "wgrad_side")
    else:
        scratch = _wgrad_scratch(lib, d, dev, "wgrad")
    # (all leaf-weight gradients of a pass run on the ONE side stream, whichever stream their node belongs to: they accumulate in place
    #  across the chains of a forked pass too -- and no autograd add ever reads them before the end-of-pass join)
    dw, dw_ptr, accumulate = _grad_target(weight, weight.shape, dev, stream=side)

    def launch():
        if pro is None:
            L.check(lib.dei2i_conv2d_wgrad_oihw(byref(d), _p(x), _p(g), _p(scratch), scratch.numel(), c_void_p(dw_ptr), accumulate,
                                                _stream()), "conv2d_wgrad")
        else:
            L.check(lib.dei2i_conv2d_wgrad_oihw_pro(byref(d), _p(x), _p(g), _p(scratch), scratch.numel(), c_void_p(dw_ptr), accumulate,
                                                    byref(pro), _stream()), "conv2d_wgrad_pro")
    if side is None:
        launch()
        return dw
    side.wait_stream(torch.cuda.current_stream(dev))          # x, g (and an earlier node's contribution to dw) are ready
    with torch.cuda.stream(side):
        launch()
    for t in (x, g, dw) + tuple(keep):                        # the allocator must not hand their memory out before the side stream is done
        if t is not None:
            t.record_stream(side)
    if _wgrad_join_pending.get(dev) != tid:                   # (a pass that raised never ran its callback: its id is stale)
        _wgrad_join_pending[dev] = tid
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join(dev, tid))
    return dw


# ---- double backward (stargan-v2's R1 penalty, core/solver.py:573-583) ---------------------------------------------
# r1_reg differentiates d sum(D(x)) / dx with create_graph=True and back-propagates 0.5 * |that|^2 into D's parameters: the
# BACKWARD pass of D's input gradient is itself differentiated.  While such a graph is being recorded (grad mode on inside a
# backward: create_graph=True) the ops below compute their input gradients with autograd Functions of their own
# -- each a linear map whose transpose is again one of the kernels: the conv's input gradient (transpose: the forward conv;
# with respect to the weight: the wgrad kernel on (incoming, dy)), the activation mask, the average pool, the layout changes.
# Parameter gradients of that first-order pass are not needed (r1_reg asks for the input gradient only) and stay on the plain path.
def _second_order(g) -> bool:
    # grad mode is ON inside a backward pass exactly when it runs with create_graph=True; the incoming gradient itself need not
    # require grad (the seed of d sum(D(x)) is a constant) -- the result still depends on the weights
    return torch.is_grad_enabled() and g is not None


class _ActBwd(torch.autograd.Function):
    """g = dy * act'(z) with the mask taken from the activation's OUTPUT z (ReLU family): linear in dy, mask constant"""

    @staticmethod
    def forward(ctx, dy, z, act: int):
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        L.check(_lib_for(dy).dei2i_act_bwd(precision_of(dy).code, dy.numel(), _p(dy), _p(z), act, _p(g), _stream()), "act_bwd")
        ctx.act = act
        ctx.save_for_backward(z)
        return g

    @staticmethod
    def backward(ctx, gg):
        (z,) = ctx.saved_tensors
        return _ActBwd.apply(gg.contiguous(), z, ctx.act), None, None


class _ConvDgradFn(torch.autograd.Function):
    """dx = W^T (*) g: the conv's input gradient as a function of (g, W).  Its own gradients: with respect to g the FORWARD conv of
    the incoming gradient, with respect to W the weight-gradient kernel on (x := incoming, dy := g).  Plain geometry only (zero
    padding, no fused upsample): what the discriminators differentiated twice use."""

    @staticmethod
    def forward(ctx, g, weight, cache, sources, geom: ConvGeom, x_shape, prec):
        lib = _lib_for(g)
        g = g.contiguous()
        ctx.geom, ctx.cache, ctx.sources, ctx.prec = geom, cache, sources, prec
        ctx.save_for_backward(g, weight)
        return _conv_dgrad(lib, prec, geom, tuple(x_shape), g.shape[-1], g, weight, cache, sources, False, g.dtype, g.device)

    @staticmethod
    def backward(ctx, gdx):
        g, weight = ctx.saved_tensors
        gdx = gdx.contiguous()
        dg = dw = None
        if ctx.needs_input_grad[0]:
            dg = _Conv2d.apply(gdx, weight.detach(), None, ctx.cache, ctx.sources, ctx.geom, L.ACT_NONE, False)
        if _wants_grad(ctx, 1):
            dw = _conv_wgrad(_lib_for(g), ctx.prec, ctx.geom, gdx, g, weight)
        return dg, dw, None, None, None, None, None


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cache: PackedWeights, sources, geom: ConvGeom, act: int, want_stats: bool = False):
        _require_gpu(x, "conv2d")
        prec = precision_of(x)
        # the norm layer that wrote x left what this conv's dgrad needs to take its backward reductions (see _NormBwdHint)
        ctx.hint = getattr(x, "_dei2i_bwd_hint", None) if (x.is_contiguous() and not geom.up) else None
        x = x.contiguous()
        n, h, w, cins = x.shape
        couts = prec.pad(geom.cout)
        lib = _lib_for(x)
        d = _desc(prec, geom, n, h, w, cins, couts)
        per_call = bool(getattr(weight, "_dei2i_per_call", False))
        if per_call:
            # a weight derived anew for THIS call (spectral norm in training mode): its packed copies travel with the
            # call (ctx), not with the module -- the module's single slot would be overwritten by the next call before
            # this call's backward asks for the dgrad layout
            cache = PackedWeights()
        use_fp8 = bool(_fp8_forward and prec is BF16 and not per_call and lib.dei2i_conv2d_fp8_supported(byref(d)))
        # trainable weights: a backward pass of this optimizer step will want the dgrad layout too (also when THIS call
        # is the no-grad generator pass of the D step) -> both layouts in one pack launch
        # (a frozen weight behind an input that needs a gradient -- D inside the G step -- wants the dgrad layout just as well)
        want_dgrad = any(s_.requires_grad for s_ in sources) or bool(ctx.needs_input_grad[0])
        wf = None if use_fp8 else cache.get(weight, sources, prec, geom, cins, couts, need_dgrad=want_dgrad, per_call=per_call)[0]
        ho, wo = c_int(), c_int()
        lib.dei2i_conv2d_out_shape(byref(d), byref(ho), byref(wo))
        y = torch.empty((n, ho.value, wo.value, couts), dtype=prec.dtype, device=x.device)
        b32 = None
        if bias is not None:
            b32 = bias.detach().float().contiguous()
        # the epilogue of the halo-resident kernel can leave the per-channel statistics of y for the norm that follows
        stats_fused = bool(want_stats and fuse_norm and not use_fp8 and prec is BF16 and lib.dei2i_conv2d_fused_supported(byref(d), 0))
        if use_fp8:
            wq, dequant = cache.get_fp8(weight, sources, prec, geom, cins, couts)
            xq = getattr(x, "_dei2i_fp8", None)      # written by the producing normalisation kernel in the same pass
            if xq is not None and xq.numel() == x.numel():
                del x._dei2i_fp8                     # one consumer: release the copy after this launch
            else:
                xq = _workspace(x.device, x.numel(), slot="fp8_act")
                L.check(lib.dei2i_quantize_fp8(x.numel(), _p(x), FP8_ACT_SCALE, _p(xq), _stream()), "quantize_fp8")
            L.check(lib.dei2i_conv2d_fwd_fp8(byref(d), _p(xq), _p(wq), _p(b32), _p(dequant), act, _p(y), _stream()),
                    "conv2d_fwd_fp8")
        elif stats_fused:
            chunks = lib.dei2i_conv2d_stats_chunks(byref(d))
            partial = torch.empty((n, chunks, 2, couts), dtype=torch.float32, device=x.device)
            L.check(lib.dei2i_conv2d_fwd_fused(byref(d), _p(x), _p(wf), _p(b32), act, _p(y), None, _p(partial), _stream()),
                    "conv2d_fwd_fused")
            _stats_stash.append((partial, chunks))
        else:
            ws = _workspace(x.device, lib.dei2i_conv2d_workspace_bytes(byref(d)))
            L.check(lib.dei2i_conv2d_fwd(byref(d), _p(x), _p(wf), _p(b32), act, _p(y), _p(ws), ws.numel() * 4, _stream()),
                    "conv2d_fwd")
        ctx.geom, ctx.act, ctx.cache, ctx.sources, ctx.prec, ctx.per_call = geom, act, cache, sources, prec, per_call
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        geom, prec, act = ctx.geom, ctx.prec, ctx.act
        lib = _lib_for(x)
        dy = dy.contiguous()
        couts = dy.shape[-1]
        st = _stream()
        if _second_order(dy):                                # the input gradient as a differentiable function of dy (see _ConvDgradFn)
            if (geom.reflect and geom.pad > 0) or geom.up or ctx.per_call:
                raise NotImplementedError("double backward through a reflect-padded / upsample-fused / per-call-weight conv is not built")
            g2 = _ActBwd.apply(dy, y, act) if act != L.ACT_NONE else dy
            dx2 = _ConvDgradFn.apply(g2, weight, ctx.cache, ctx.sources, geom, tuple(x.shape), prec) if ctx.needs_input_grad[0] else None
            dw2 = _conv_wgrad(lib, prec, geom, x, g2.detach(), weight) if _wants_grad(ctx, 1) else None
            db2 = None
            if ctx.has_bias and _wants_grad(ctx, 2):
                db2 = g2.detach().float().sum(dim=(0, 1, 2))[:geom.cout]
            return dx2, dw2, db2, None, None, None, None, None
        if act != L.ACT_NONE:
            g = torch.empty_like(dy)
            L.check(lib.dei2i_act_bwd(prec.code, dy.numel(), _p(dy), _p(y), act, _p(g), st), "act_bwd")
        else:
            g = dy
        dx = dw = db = None
        if _wants_grad(ctx, 1):                              # (first: on its side stream it then runs beside the dgrad)
            dw = _conv_wgrad(lib, prec, geom, x, g, weight)
        if _wants_grad(ctx, 0):
            dx = _conv_dgrad(lib, prec, geom, tuple(x.shape), couts, g, weight, ctx.cache, ctx.sources, ctx.per_call, x.dtype, x.device,
                             hint=ctx.hint)
        if ctx.has_bias and _wants_grad(ctx, 2):
            dbf = torch.empty(couts, dtype=torch.float32, device=x.device)
            rows = g.numel() // couts
            part = torch.empty(lib.dei2i_colsum_blocks(rows) * couts, dtype=torch.float32, device=x.device)
            L.check(lib.dei2i_colsum(prec.code, rows, couts, _p(g), _p(part), _p(dbf), st), "colsum")
            db = dbf if couts == geom.cout else dbf[:geom.cout].clone()
        return dx, dw, db, None, None, None, None, None


def conv2d(x, weight, bias, cache: PackedWeights, geom: ConvGeom, act="none", sources=None, stats=False):
    """y = act(conv(x) + bias) on an NHWC activation; ``weight`` is the reference's OIHW fp32 parameter.  ``stats``: a
    BatchNorm / InstanceNorm follows -- leave the statistics records of y with it when the kernel can (see _stats_of)."""
    del _stats_stash[:]
    return _attach_stats(_Conv2d.apply(x, weight, bias, cache, tuple(sources) if sources is not None else (weight,), geom,
                                       ACT[act], bool(stats)))


# --------------------------------------------------------------------------------------------------------------
# layout at the module boundary
# --------------------------------------------------------------------------------------------------------------
class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prec: Precision, size):
        _require_gpu(x, "to_nhwc")
        x = x.detach().float().contiguous() if x.dtype != torch.float32 or not x.is_contiguous() else x
        n, c, hs, ws = x.shape
        h, w = (hs, ws) if size is None else size
        cs = prec.pad(c)
        out = torch.empty((n, h, w, cs), dtype=prec.dtype, device=x.device)
        lib = _lib_for(x)
        L.check(lib.dei2i_nchw_to_nhwc_resize(prec.code, n, c, hs, ws, h, w, cs, _p(x), _p(out), _stream()), "nchw_to_nhwc")
        ctx.shape, ctx.prec, ctx.resized = (n, c, hs, ws), prec, (h, w) != (hs, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.resized:
            raise RuntimeError("to_nhwc with nearest resize is only differentiable w.r.t. nothing (label maps)")
        n, c, h, w = ctx.shape
        if _second_order(g):                              # R1-style penalties differentiate this backward: the conversion is linear
            return _ToNCHW.apply(g, c), None, None
        g = g.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
        lib = _lib_for(g)
        L.check(lib.dei2i_nhwc_to_nchw(ctx.prec.code, n, c, h, w, g.shape[-1], _p(g), _p(dx), _stream()), "nhwc_to_nchw")
        return dx, None, None


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c: int):
        _require_gpu(x, "to_nchw")
        prec = precision_of(x)
        x = x.contiguous()
        n, h, w, cs = x.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        lib = _lib_for(x)
        L.check(lib.dei2i_nhwc_to_nchw(prec.code, n, c, h, w, cs, _p(x), _p(out), _stream()), "nhwc_to_nchw")
        ctx.prec, ctx.cs = prec, cs
        return out

    @staticmethod
    def backward(ctx, g):
        if _second_order(g):
            return _ToNHWC.apply(g, ctx.prec, None), None
        g = g.contiguous().float()
        n, c, h, w = g.shape
        dx = torch.empty((n, h, w, ctx.cs), dtype=ctx.prec.dtype, device=g.device)
        lib = _lib_for(g)
        L.check(lib.dei2i_nchw_to_nhwc(ctx.prec.code, n, c, h, w, ctx.cs, _p(g), _p(dx), _stream()), "nchw_to_nhwc")
        return dx, None


def to_nhwc(x_nchw, prec: Precision, size=None):
    """NCHW fp32 -> NHWC compute dtype (channels zero-padded); ``size`` fuses F.interpolate(mode='nearest')."""
    return _ToNHWC.apply(x_nchw, prec, size)


def to_nchw(x_nhwc, c: int):
    return _ToNCHW.apply(x_nhwc, c)


# --------------------------------------------------------------------------------------------------------------
# BatchNorm2d (+ LeakyReLU) (+ residual)
# --------------------------------------------------------------------------------------------------------------
forked_chains = True             # the G loss's two independent chains of generator passes on two streams (models/defectgan_model.py)
_chain_streams = {}


def chain_streams(device):
    """The two streams the chains of a forked pass run on (created once per device)."""
    dev = torch.device(device)
    st = _chain_streams.get(dev)
    if st is None:
        st = _chain_streams[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return st


# ---- passes of the same network that share nothing but the parameters, as ONE batch -----------------------------------------
# The G loss's chain heads (bg -> fake_defects, df -> fake_normals) and its chain tails (-> recover_normals, -> recover_defects)
# are pairs of independent passes of the same generator (defectgan_model.py:185-190).  Everything in the generator acts per
# sample except training-mode BatchNorm, whose statistics are per PASS: with ``bn_batch_groups(g)`` a training-mode BatchNorm
# takes its statistics (and its backward reductions) over each of g equal groups of the batch separately, so that one pass over
# the concatenated batch is the same function as the g passes -- with half the launches, half the per-launch tails and
# weight-gradient reductions, and every conv at twice the tile count.  The running statistics are updated per group through
# ``bn_running_deferred`` (pass_index = one index per group), which replays them in the reference's pass order.
label_path_batched = True        # SPADE's gamma | beta table convs of all modules in one launch per direction (csrc/label_path.hip)
paired_passes = True             # the G loss's four generator passes as two passes over 2 x batch (models/defectgan_model.py)
bn_groups = 1


class bn_batch_groups:
    def __init__(self, groups: int):
        self.groups = int(groups)

    def __enter__(self):
        global bn_groups
        self.prev, bn_groups = bn_groups, self.groups
        return self

    def __exit__(self, *exc):
        global bn_groups
        bn_groups = self.prev
        return False


# ---- two independent chains of generator passes on two streams ---------------------------------------------------
# The G loss runs the generator four times (defectgan_model.py:185-190): fake_defects -> recover_normals and fake_normals ->
# recover_defects are two chains that share nothing but the parameters.  On two streams the kernels of one chain fill the other
# chain's kernel tails and launch boundaries (a dependent kernel boundary is ~1.7 us and the step has ~1 350 of them) and its
# HBM-bound passes run beside the other's MFMA-bound ones.  What has to stay ordered is BatchNorm's running-statistics update
# (running = (1 - m) * running + m * batch, four times, in the reference's pass order): inside a ``bn_running_deferred`` scope a
# training-mode BatchNorm leaves m * batch of its pass in a buffer of its own, and ``apply()`` replays the updates in the order
# the passes were NUMBERED, on the joining stream.
class bn_running_deferred:
    current = None

    _arenas = {}                     # device -> flat fp32 zeros the passes' buffers are carved from (zeroed again by apply(): one fill per step)

    def __init__(self):
        self.updates = []            # (pass index, running_mean, running_var, num_batches_tracked, m * batch mean, m * batch var, momentum)
        self.pass_index = 0
        self._used = {}

    def __enter__(self):
        self.prev, bn_running_deferred.current = bn_running_deferred.current, self
        return self

    def __exit__(self, *exc):
        bn_running_deferred.current = self.prev
        return False

    def take(self, running_mean, running_var, num_batches_tracked, momentum, group=0):
        dev, n = running_mean.device, running_mean.numel()
        index = self.pass_index[group] if isinstance(self.pass_index, (tuple, list)) else self.pass_index
        arena, off = bn_running_deferred._arenas.get(dev), self._used.get(dev, 0)
        if arena is None or off + 2 * n > arena.numel() or running_mean.dtype != torch.float32:
            zm, zv = torch.zeros_like(running_mean), torch.zeros_like(running_var)          # (first step / an odd buffer: its own zeros)
            if running_mean.dtype == torch.float32:
                self._want = getattr(self, "_want", 0) + 2 * n
        else:
            zm, zv = arena[off:off + n], arena[off + n:off + 2 * n]
            self._used[dev] = off + 2 * n
        self.updates.append((index, running_mean, running_var, num_batches_tracked, zm, zv, momentum))
        return zm, zv

    def take_groups(self, running_mean, running_var, num_batches_tracked, momentum, groups):
        """take() for every group at once, the buffers evenly spaced in ONE block -- (groups, 2, n): group g's mean at row (g, 0), its
        variance at (g, 1) -- so that one finalize launch serves all groups (dei2i_bn_finalize_train_groups: stride 2 n)."""
        dev, n = running_mean.device, running_mean.numel()
        arena, off = bn_running_deferred._arenas.get(dev), self._used.get(dev, 0)
        if arena is None or off + 2 * n * groups > arena.numel() or running_mean.dtype != torch.float32:
            block = torch.zeros(groups * 2 * n, dtype=torch.float32, device=dev)
            self._want = getattr(self, "_want", 0) + 2 * n * groups
        else:
            block = arena[off:off + 2 * n * groups]
            self._used[dev] = off + 2 * n * groups
        block = block.view(groups, 2, n)
        for g in range(groups):
            index = self.pass_index[g] if isinstance(self.pass_index, (tuple, list)) else self.pass_index
            self.updates.append((index, running_mean, running_var, num_batches_tracked, block[g, 0], block[g, 1], momentum))
        return block

    def apply(self):
        """running <- (1 - m) * running + (m * batch) for every recorded update, passes in index order (on the current stream,
        which must already wait for the streams the passes ran on)"""
        for k in sorted({u[0] for u in self.updates}):
            ups = [u for u in self.updates if u[0] == k]
            for m in sorted({u[6] for u in ups}):
                sel = [u for u in ups if u[6] == m]
                run = [u[1] for u in sel] + [u[2] for u in sel]
                torch._foreach_mul_(run, 1.0 - m)
                torch._foreach_add_(run, [u[4] for u in sel] + [u[5] for u in sel])
            counters = [u[3] for u in ups if u[3] is not None]
            if counters:
                torch._foreach_add_(counters, 1)
        devs = {u[1].device for u in self.updates}
        self.updates = []
        for dev in devs:                         # the arena is all zeros again for the next scope; grown to what this one asked for
            need = self._used.get(dev, 0) + getattr(self, "_want", 0)
            arena = bn_running_deferred._arenas.get(dev)
            if arena is None or arena.numel() < need:
                bn_running_deferred._arenas[dev] = torch.zeros(max(need, 1 << 12), dtype=torch.float32, device=dev)
            elif self._used.get(dev, 0):
                arena[:self._used[dev]].zero_()
        self._used, self._want = {}, 0


def _bn_coefs_grouped(lib, y, prec, weight, bias, running_mean, running_var, momentum, eps, num_batches_tracked, groups):
    """Training-mode statistics + coefficients per group of the batch (see bn_batch_groups): a, b, mean, rstd of shape (groups, C)."""
    n, h, w, c = y.shape
    dev, st = y.device, _stream()
    w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    nf = w32.numel()
    deferred = bn_running_deferred.current
    if n % groups or nf > c or running_mean.numel() != nf or running_var.numel() != nf:
        raise ValueError(f"batchnorm_act: {groups} groups over a batch of {n} / {nf} features for a {c}-channel activation")
    if deferred is None or not isinstance(deferred.pass_index, (tuple, list)) or len(deferred.pass_index) != groups:
        raise RuntimeError("batchnorm_act: grouped batch statistics want a bn_running_deferred scope with one pass index per group")
    a, b, mean, rstd = (torch.empty((groups, c), dtype=torch.float32, device=dev) for _ in range(4))
    have = _stats_of(y, n, h * w, c) if fuse_norm else None
    if have is not None:
        partial, chunks = have
    else:
        chunks = lib.dei2i_moments_chunks(h * w)
        partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
        L.check(lib.dei2i_moments_partial(prec.code, n, h * w, c, _p(y), _p(partial), st), "moments_partial")
    ng = n // groups
    if nf == c and running_mean.dtype == torch.float32:
        block = deferred.take_groups(running_mean, running_var, num_batches_tracked, float(momentum), groups)       # (groups, 2, c)
        L.check(lib.dei2i_bn_finalize_train_groups(groups, ng, h * w, c, chunks, _p(partial), _p(w32), _p(b32), _p(block[0, 0]), _p(block[0, 1]),
                                                   2 * c, momentum, eps, _p(mean), _p(rstd), _p(a), _p(b), st), "bn_finalize_train_groups")
        return a, b, mean, rstd, nf
    if nf < c:                                   # padded channel stride (see _bn_coefs): c-sized vectors for the kernel
        w32, b32 = (torch.cat([v, v.new_zeros(c - nf)]) for v in (w32, b32))
    for g in range(groups):
        rm, rv = live = deferred.take(running_mean, running_var, num_batches_tracked, float(momentum), group=g)
        if nf < c:
            rm, rv = torch.zeros(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev)
        L.check(lib.dei2i_bn_finalize_train_chunks(ng, h * w, c, chunks, _p(partial[g * ng:]), _p(w32), _p(b32), _p(rm), _p(rv),
                                                   momentum, eps, _p(mean[g]), _p(rstd[g]), _p(a[g]), _p(b[g]), None, st),
                "bn_finalize_train")
        if nf < c:
            live[0].copy_(rm[:nf])
            live[1].copy_(rv[:nf])
    return a, b, mean, rstd, nf


def _bn_coefs(lib, y, prec, weight, bias, running_mean, running_var, training, momentum, eps, num_batches_tracked):
    """BatchNorm2d statistics + affine coefficients of an NHWC tensor: a[c] = weight * rstd, b[c] = bias - mean * a (batch
    statistics in training mode, running statistics otherwise; running buffers and the counter are updated in place).
    The statistics come from the records its producer left (conv epilogue / affine_act_stats) when there are any."""
    n, h, w, c = y.shape
    dev, st = y.device, _stream()
    a = torch.empty(c, dtype=torch.float32, device=dev)
    b = torch.empty(c, dtype=torch.float32, device=dev)
    w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    nf = w32.numel()
    if nf > c or running_mean.numel() != nf or running_var.numel() != nf:
        raise ValueError(f"batchnorm_act: {nf} features for a {c}-channel activation")
    rm, rv = running_mean, running_var
    deferred = bn_running_deferred.current if training else None
    if deferred is not None:                     # this pass's m * batch statistics go to buffers of their own (see bn_running_deferred)
        rm, rv = deferred.take(running_mean, running_var, num_batches_tracked, float(momentum))
        running_mean, running_var, num_batches_tracked = rm, rv, None
    if nf < c:
        # the channel stride is padded to a 16-byte vector (c > num_features: widths that are no multiple of 8 / 4): the
        # kernels index every per-channel vector up to c, so hand them padded copies -- weight = bias = 0 makes the
        # padded channels' a = b = 0 (their activations stay zero) -- and copy the live running statistics back
        def _padded(v, fill):
            o = torch.full((c,), fill, dtype=torch.float32, device=dev)
            o[:nf] = v.detach()
            return o
        w32, b32, rm, rv = _padded(w32, 0.0), _padded(b32, 0.0), _padded(running_mean, 0.0), _padded(running_var, 1.0)
    if training:
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        rstd = torch.empty(c, dtype=torch.float32, device=dev)
        have = _stats_of(y, n, h * w, c) if fuse_norm else None
        if have is not None:
            partial, chunks = have
        else:
            chunks = lib.dei2i_moments_chunks(h * w)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            L.check(lib.dei2i_moments_partial(prec.code, n, h * w, c, _p(y), _p(partial), st), "moments_partial")
        L.check(lib.dei2i_bn_finalize_train_chunks(n, h * w, c, chunks, _p(partial), _p(w32), _p(b32), _p(rm), _p(rv),
                                                   momentum, eps, _p(mean), _p(rstd), _p(a), _p(b), _p(num_batches_tracked), st),
                "bn_finalize_train")
        if nf < c:
            running_mean.copy_(rm[:nf])
            running_var.copy_(rv[:nf])
    else:
        L.check(lib.dei2i_bn_finalize_eval(c, _p(w32), _p(b32), _p(rm), _p(rv), eps, _p(a), _p(b), st),
                "bn_finalize_eval")
        mean = rm.detach().clone()
        rstd = torch.rsqrt(rv.detach() + eps)
    return a, b, mean, rstd, nf


def _bn_backward(lib, prec, dout, y, a, b, mean, rstd, act, training, weight, bias, nf, hint=None):
    """-> (dy, dweight, dbias) of z = act(a*y + b) given dL/dz (csrc/reduce.hip: bn_bwd_partial / bn_bwd_apply).  ``hint``: the
    dgrad of the conv behind the layer may have left the reduction records already (_NormBwdHint).  a, b, mean, rstd of shape
    (groups, C): statistics per group of the batch (bn_batch_groups) -- the reductions and the apply run per group, the
    parameter gradients are summed over the groups."""
    st = _stream()
    have = hint.take(dout) if hint is not None else None
    dout = dout.contiguous()
    n, h, w, c = y.shape
    groups = a.shape[0] if a.dim() == 2 else 1
    ng = n // groups
    pixels = ng * h * w
    a, b, mean, rstd = (t.view(groups, c) for t in (a, b, mean, rstd))
    if have is not None:
        partial, chunks = have[0], ng * have[1]          # (n, records per image, 2, c): a group's records are contiguous
    elif groups > 1:
        chunks = lib.dei2i_bn_bwd_chunks(pixels)
        partial = torch.empty((groups * chunks, 2, c), dtype=torch.float32, device=y.device)
        L.check(lib.dei2i_bn_bwd_partial_groups(prec.code, groups, pixels, c, _p(dout), _p(y), _p(a), _p(b), _p(mean), _p(rstd), act,
                                                _p(partial), st), "bn_bwd_partial_groups")
    else:
        chunks = lib.dei2i_bn_bwd_chunks(pixels)
        partial = torch.empty((groups * chunks, 2, c), dtype=torch.float32, device=y.device)
        for g in range(groups):
            L.check(lib.dei2i_bn_bwd_partial(prec.code, pixels, c, _p(dout[g * ng:]), _p(y[g * ng:]), _p(a[g]), _p(b[g]), _p(mean[g]),
                                             _p(rstd[g]), act, _p(partial[g * chunks:]), st), "bn_bwd_partial")
    padded = nf < c               # padded channel stride: c-sized scratch vectors, sliced to num_features below
    if padded:
        tmp_wb = torch.empty((2, c), dtype=torch.float32, device=y.device)
        dweight = dbias = None
        dw_ptr, db_ptr, acc_w, acc_b = tmp_wb.data_ptr(), tmp_wb.data_ptr() + 4 * c, 0, 0
    else:
        dweight, dw_ptr, acc_w = _grad_target(weight, (c,), y.device)
        dbias, db_ptr, acc_b = _grad_target(bias, (c,), y.device)
    acc_ptrs = (None, None)
    if acc_w or acc_b:            # the kernel needs this call's own sums as well: they go to scratch vectors
        if not (acc_w and acc_b):
            # one of the pair lost its first gradient tensor (see _grad_target): give both a fresh tensor
            dweight = torch.empty((c,), dtype=torch.float32, device=y.device)
            dbias = torch.empty((c,), dtype=torch.float32, device=y.device)
            dw_ptr, db_ptr, acc_w, acc_b = dweight.data_ptr(), dbias.data_ptr(), 0, 0
        else:
            acc_ptrs = (c_void_p(dw_ptr), c_void_p(db_ptr))
            tmp = torch.empty((2, c), dtype=torch.float32, device=y.device)
            dw_ptr, db_ptr = tmp.data_ptr(), tmp.data_ptr() + 4 * c
    dy = torch.empty_like(y)
    parts = partial.view(groups, -1)
    if groups > 1:
        # every group in two launches: the groups' own sums (read by their share of the apply launch) go to scratch, their total to the
        # parameters' gradient -- written, or added when an earlier use of the parameters in this pass holds the tensor already
        gsum = torch.empty((groups, 2, c), dtype=torch.float32, device=y.device)
        accumulate = acc_ptrs[0] is not None
        tgt_w, tgt_b = (acc_ptrs[0], acc_ptrs[1]) if accumulate else (c_void_p(dw_ptr), c_void_p(db_ptr))
        L.check(lib.dei2i_bn_bwd_apply_groups(prec.code, groups, pixels, c, _p(dout), _p(y), _p(a), _p(b), _p(mean), _p(rstd), act,
                                              1 if training else 0, _p(partial), chunks, _p(gsum), tgt_w, tgt_b, 1 if accumulate else 0,
                                              _p(dy), st), "bn_bwd_apply_groups")
    for g in range(groups if groups == 1 else 0):
        if g == 1:                # the later groups add into where the first one's sums went
            if acc_ptrs[0] is None:
                acc_ptrs = (c_void_p(dw_ptr), c_void_p(db_ptr))
            tmp = torch.empty((2, c), dtype=torch.float32, device=y.device)
            dw_ptr, db_ptr = tmp.data_ptr(), tmp.data_ptr() + 4 * c
        L.check(lib.dei2i_bn_bwd_apply(prec.code, pixels, c, _p(dout[g * ng:]), _p(y[g * ng:]), _p(a[g]), _p(b[g]), _p(mean[g]),
                                       _p(rstd[g]), act, 1 if training else 0, _p(parts[g]), chunks, c_void_p(dw_ptr), c_void_p(db_ptr),
                                       acc_ptrs[0], acc_ptrs[1], _p(dy[g * ng:]), st), "bn_bwd_apply")
    if padded:
        dweight, dbias = tmp_wb[0, :nf].clone(), tmp_wb[1, :nf].clone()
    return dy, dweight, dbias


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, weight, bias, res, running_mean, running_var, training: bool, momentum: float, eps: float, act: int,
                num_batches_tracked=None, want_stats: bool = False):
        _require_gpu(y, "batchnorm_act")
        prec = precision_of(y)
        y = y.contiguous()
        n, h, w, c = y.shape
        lib = _lib_for(y)
        st = _stream()
        dev = y.device
        groups = bn_groups if training else 1
        if groups > 1:
            a, b, mean, rstd, nf = _bn_coefs_grouped(lib, y, prec, weight, bias, running_mean, running_var, momentum, eps,
                                                     num_batches_tracked, groups)
        else:
            a, b, mean, rstd, nf = _bn_coefs(lib, y, prec, weight, bias, running_mean, running_var, training, momentum, eps,
                                             num_batches_tracked)
        out = torch.empty_like(y)
        if res is not None:
            res = res.contiguous()
        xq = torch.empty(out.numel(), dtype=torch.uint8, device=dev) if _fp8_copy_wanted(prec, c) else None
        ng = n // groups
        av, bv = a.view(groups, c), b.view(groups, c)
        partial = None
        if want_stats and fuse_norm and xq is None:
            # the statistics records of the output in the same pass (an InstanceNorm / BatchNorm reads this tensor next)
            chunks = lib.dei2i_moments_chunks(h * w)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            _stats_stash.append((partial, chunks))
        if groups > 1:                                    # every group in one launch (grid.y = group; coefficient rows (groups, c))
            if partial is not None:
                L.check(lib.dei2i_affine_act_stats_groups_fwd(prec.code, groups, n, h * w, c, _p(y), _p(av), _p(bv), _p(res), act, _p(out),
                                                              _p(partial), st), "affine_act_stats_groups")
            else:
                L.check(lib.dei2i_affine_act_groups_fwd(prec.code, groups, ng * h * w, c, _p(y), _p(av), _p(bv), _p(res), act, _p(out), _p(xq),
                                                        FP8_ACT_SCALE, st), "affine_act_groups")
        for g in range(groups if groups == 1 else 0):
            lo = g * ng
            rg = res[lo:] if res is not None else None
            if partial is not None:
                L.check(lib.dei2i_affine_act_stats_fwd(prec.code, ng, h * w, c, _p(y[lo:]), _p(av[g]), _p(bv[g]), _p(rg), act, _p(out[lo:]),
                                                       _p(partial[lo:]), st), "affine_act_stats")
            else:
                L.check(lib.dei2i_affine_act_fwd(prec.code, ng * h * w, c, _p(y[lo:]), _p(av[g]), _p(bv[g]), _p(rg), act, _p(out[lo:]),
                                                 _p(xq[lo * h * w * c:]) if xq is not None else None, FP8_ACT_SCALE, st), "affine_act")
        if xq is not None:
            _fp8_stash.append(xq)
        ctx.prec, ctx.act, ctx.training, ctx.has_res = prec, act, training, res is not None
        ctx.params, ctx.nf = (weight, bias), nf
        ctx.save_for_backward(y, a, b, mean, rstd)
        ctx.hint = None
        if fuse_bwd and prec is BF16 and ctx.needs_input_grad[0]:
            ctx.hint = _NormBwdHint(2, y, mean, rstd, a=a, b=b, act=act, group_images=ng if groups > 1 else 0)
            _hint_stash.append(ctx.hint)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, a, b, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.params
        dy, dweight, dbias = _bn_backward(_lib_for(y), ctx.prec, dout, y, a, b, mean, rstd, ctx.act, ctx.training, weight, bias, ctx.nf,
                                          hint=ctx.hint)
        return dy, dweight, dbias, (dout if ctx.has_res else None), None, None, None, None, None, None, None, None


def batchnorm_act(y, weight, bias, running_mean, running_var, training, act="none", res=None, momentum=0.1, eps=1e-5,
                  num_batches_tracked=None, stats=False):
    """num_batches_tracked: the module's int64 counter, incremented inside the statistics kernel in training mode.
    ``stats``: a norm layer reads the output next -- leave its statistics records with it (see _stats_of)."""
    del _fp8_stash[:]
    del _stats_stash[:]
    del _hint_stash[:]
    return _attach_hint(_attach_stats(_attach_fp8(_BatchNormAct.apply(y, weight, bias, res, running_mean, running_var, bool(training),
                                                                      float(momentum), float(eps), ACT[act], num_batches_tracked,
                                                                      bool(stats)))))


class _BnActConv(torch.autograd.Function):
    """conv(act(BatchNorm(y1))) with the BatchNorm apply + activation on the conv's operand path (architecture.py:116-118
    followed by the next block's conv, e.g. the two halves of a ResBlock, architecture.py:139-156): the normalised tensor is
    never written -- forward, and the weight gradient in backward, re-normalise y1's halo tiles in LDS (ConvPro, csrc/geom.h)."""

    @staticmethod
    def forward(ctx, y1, bn_w, bn_b, weight, running_mean, running_var, training, momentum, eps, act, num_batches_tracked,
                cache: PackedWeights, sources, geom: ConvGeom, want_stats):
        prec = precision_of(y1)
        y1 = y1.contiguous()
        n, h, w, c = y1.shape
        lib = _lib_for(y1)
        a, b, mean, rstd, nf = _bn_coefs(lib, y1, prec, bn_w, bn_b, running_mean, running_var, training, momentum, eps,
                                         num_batches_tracked)
        couts = prec.pad(geom.cout)
        d = _desc(prec, geom, n, h, w, c, couts)
        wf = cache.get(weight, sources, prec, geom, c, couts, need_dgrad=any(s_.requires_grad for s_ in sources))[0]
        ho, wo = c_int(), c_int()
        lib.dei2i_conv2d_out_shape(byref(d), byref(ho), byref(wo))
        y = torch.empty((n, ho.value, wo.value, couts), dtype=prec.dtype, device=y1.device)
        pro = L.ProDesc(a.data_ptr(), b.data_ptr(), 0, 0.2 if act == L.ACT_LRELU else (0.0 if act == L.ACT_RELU else 1.0), None)
        partial = None
        if want_stats:
            chunks = lib.dei2i_conv2d_stats_chunks(byref(d))
            partial = torch.empty((n, chunks, 2, couts), dtype=torch.float32, device=y1.device)
            _stats_stash.append((partial, chunks))
        L.check(lib.dei2i_conv2d_fwd_fused(byref(d), _p(y1), _p(wf), None, L.ACT_NONE, _p(y), byref(pro), _p(partial), _stream()),
                "conv2d_fwd_fused(bn)")
        ctx.prec, ctx.act, ctx.training, ctx.nf, ctx.geom = prec, act, training, nf, geom
        ctx.cache, ctx.sources, ctx.params, ctx.slope = cache, sources, (bn_w, bn_b), pro.slope
        ctx.save_for_backward(y1, a, b, mean, rstd, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        y1, a, b, mean, rstd, weight = ctx.saved_tensors
        prec, geom = ctx.prec, ctx.geom
        lib = _lib_for(y1)
        dy = dy.contiguous()
        dw = None
        if _wants_grad(ctx, 3):
            pro = L.ProDesc(a.data_ptr(), b.data_ptr(), 0, ctx.slope, None)
            dw = _conv_wgrad(lib, prec, geom, y1, dy, weight, pro, keep=(a, b))
        dy1 = dbw = dbb = None
        if _wants_grad(ctx, 0) or _wants_grad(ctx, 1) or _wants_grad(ctx, 2):
            hint = _NormBwdHint(2, y1, mean, rstd, a=a, b=b, act=ctx.act)
            dh = _conv_dgrad(lib, prec, geom, tuple(y1.shape), dy.shape[-1], dy, weight, ctx.cache, ctx.sources, False, y1.dtype, y1.device,
                             hint=hint)
            bn_w, bn_b = ctx.params
            dy1, dbw, dbb = _bn_backward(lib, prec, dh, y1, a, b, mean, rstd, ctx.act, ctx.training, bn_w, bn_b, ctx.nf, hint=hint)
        return (dy1, dbw, dbb, dw) + (None,) * 11


def bn_act_conv_supported(y1, bn_weight, weight, geom: ConvGeom, need_grad: bool) -> bool:
    """Can conv(act(BatchNorm(y1))) run with the norm on the conv's operand path?  (bf16, halo-resident forward kernel takes
    the shape, and -- when gradients are needed -- so does the halo-resident wgrad kernel; plain weights only)"""
    if not (fuse_norm and fuse_pro and y1.is_cuda and y1.dtype == torch.bfloat16 and not _fp8_forward) or bn_groups > 1:
        return False
    if getattr(weight, "_dei2i_per_call", False) or bn_weight.numel() != y1.shape[-1]:
        return False
    lib = _lib_for(y1)
    n, h, w, c = y1.shape
    d = _desc(BF16, geom, n, h, w, c, BF16.pad(geom.cout))
    if not lib.dei2i_conv2d_fused_supported(byref(d), 1):
        return False
    return bool(lib.dei2i_conv2d_wgrad_pro_supported(byref(d))) if need_grad else True


def bn_act_conv(y1, bn_weight, bn_bias, running_mean, running_var, training, act, weight, cache, geom, momentum=0.1, eps=1e-5,
                num_batches_tracked=None, sources=None, stats=False):
    """conv(act(BatchNorm2d(y1))), the norm + activation fused into the conv (ask bn_act_conv_supported first)."""
    del _stats_stash[:]
    return _attach_stats(_BnActConv.apply(y1, bn_weight, bn_bias, weight, running_mean, running_var, bool(training), float(momentum),
                                          float(eps), ACT[act], num_batches_tracked, cache,
                                          tuple(sources) if sources is not None else (weight,), geom, bool(stats)))


def add(x, res, stats=False):
    """x + res on NHWC activations (NormResBlock identity branch, architecture.py:350).  ``stats``: a norm layer reads the
    sum next -- leave its statistics records with it."""
    c = x.shape[-1]
    del _stats_stash[:]
    return _attach_stats(_AffineAdd.apply(x, res, _const_vec(x.device, c, 1.0), _const_vec(x.device, c, 0.0), bool(stats)))


class _AffineAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, ones, zeros, want_stats=False):
        _require_gpu(x, "add")
        prec = precision_of(x)
        x, res = x.contiguous(), res.contiguous()
        out = torch.empty_like(x)
        lib = _lib_for(x)
        c = x.shape[-1]
        if want_stats and fuse_norm and x.dim() == 4:
            n, hw = x.shape[0], x.shape[1] * x.shape[2]
            chunks = lib.dei2i_moments_chunks(hw)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=x.device)
            L.check(lib.dei2i_affine_act_stats_fwd(prec.code, n, hw, c, _p(x), _p(ones), _p(zeros), _p(res), L.ACT_NONE, _p(out),
                                                   _p(partial), _stream()), "add_stats")
            _stats_stash.append((partial, chunks))
        else:
            L.check(lib.dei2i_affine_act_fwd(prec.code, x.numel() // c, c, _p(x), _p(ones), _p(zeros), _p(res),
                                             L.ACT_NONE, _p(out), None, 1.0, _stream()), "add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, None, None, None


# --------------------------------------------------------------------------------------------------------------
# SPADE (InstanceNorm * (1+gamma) + beta) + ReLU, optional fused nearest x2 upsample of x
# --------------------------------------------------------------------------------------------------------------
def _spade_backward(lib, prec, dout, dskip, x, gb, mean, rstd, up, gb_mode, out_shape, hint=None):
    """-> (dx, dgb) of z = relu(IN(x)*(1+gamma)+beta) given dL/dz at the (upsampled) output resolution; dskip (optional) is
    added to dx inside the apply kernel (the res block's identity branch).  ``hint``: the dgrad of the conv behind the layer may
    have left the reduction records already (_NormBwdHint); the 24 border classes of the table's gradient are then all that is
    left of the first pass."""
    st = _stream()
    dev = x.device
    have = hint.take(dout) if (hint is not None and gb_mode == 1) else None
    dout = dout.contiguous()
    n, h, w, c = out_shape
    dgb = torch.empty_like(gb)         # dense (N,H,W,2C), or the (N,5,5,2C) class table -- both written in full
    if have is not None:
        partial, chunks = have
        L.check(lib.dei2i_spade_bwd_border(prec.code, n, h, w, c, 1 if up else 0, _p(dout), _p(x), _p(mean), _p(rstd), _p(gb),
                                           _p(dgb), st), "spade_bwd_border")
    else:
        chunks = lib.dei2i_moments_chunks(h * w)
        partial = torch.empty((n, chunks, 4, c), dtype=torch.float32, device=dev)
        L.check(lib.dei2i_spade_bwd_partial(prec.code, n, h, w, c, 1 if up else 0, _p(dout), _p(x), _p(mean), _p(rstd),
                                            _p(gb), gb_mode, _p(dgb), _p(partial), st), "spade_bwd_partial")
    coef = torch.empty((n, 2, c), dtype=torch.float32, device=dev)
    dx = torch.empty_like(x)
    L.check(lib.dei2i_spade_bwd_apply(prec.code, n, h, w, c, 1 if up else 0, _p(dout), _p(x), _p(mean), _p(rstd), _p(gb), gb_mode,
                                      _p(partial), chunks, _p(dgb) if gb_mode == 1 else None, _p(coef), _p(dskip), _p(dx), st),
            "spade_bwd_apply")
    return dx, dgb


class _SpadeRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gb, up: bool, gb_mode: int, eps: float, skip: bool = False):
        # skip: also hand x through as a second output (the res block's identity branch), so that the gradient arriving
        # on that branch is added inside the backward-apply kernel instead of by a separate autograd add launch
        _require_gpu(x, "spade_relu")
        prec = precision_of(x)
        x_in = x
        x, gb = x.contiguous(), gb.contiguous()
        if skip and (up or x is not x_in):
            raise ValueError("spade_relu(skip=True) wants a contiguous activation and no upsample")
        n, hs, ws, c = x.shape
        h, w = (hs * 2, ws * 2) if up else (hs, ws)
        lib = _lib_for(x)
        st = _stream()
        dev = x.device
        mean = torch.empty((n, c), dtype=torch.float32, device=dev)
        rstd = torch.empty((n, c), dtype=torch.float32, device=dev)
        # statistics of the upsampled tensor == statistics of the source tensor (every pixel replicated 4x); the records its
        # producer left (conv epilogue / affine kernel) when there are any
        have = _stats_of(x, n, hs * ws, c) if fuse_norm else None
        if have is not None:
            partial, chunks = have
        else:
            chunks = lib.dei2i_moments_chunks(hs * ws)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            L.check(lib.dei2i_moments_partial(prec.code, n, hs * ws, c, _p(x), _p(partial), st), "moments_partial")
        L.check(lib.dei2i_in_finalize_chunks(n, hs * ws, c, chunks, _p(partial), eps, _p(mean), _p(rstd), st), "in_finalize")
        out = torch.empty((n, h, w, c), dtype=prec.dtype, device=dev)
        xq = torch.empty(out.numel(), dtype=torch.uint8, device=dev) if _fp8_copy_wanted(prec, c) else None
        L.check(lib.dei2i_spade_act_fwd(prec.code, n, h, w, c, 1 if up else 0, _p(x), _p(mean), _p(rstd), _p(gb), gb_mode,
                                        _p(out), _p(xq), FP8_ACT_SCALE, st), "spade_act_fwd")
        if xq is not None:
            _fp8_stash.append(xq)
        ctx.prec, ctx.up, ctx.gb_mode = prec, up, gb_mode
        ctx.out_shape = (n, h, w, c)
        ctx.save_for_backward(x, gb, mean, rstd)        # backward recomputes the ReLU mask; the output is not kept
        ctx.hint = None
        if fuse_bwd and gb_mode == 1 and prec is BF16 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            ctx.hint = _NormBwdHint(1, x, mean, rstd, gb=gb, up=up)
            _hint_stash.append(ctx.hint)
        return (out, x_in) if skip else out

    @staticmethod
    def backward(ctx, dout, dskip=None):
        x, gb, mean, rstd = ctx.saved_tensors
        if dout is None:                                 # only the identity branch was used
            return dskip, None, None, None, None, None
        if dskip is not None:
            dskip = dskip.contiguous()
            if dskip.dtype != x.dtype or dskip.shape != x.shape:
                raise RuntimeError("spade_relu: identity-branch gradient does not match the activation")
        dx, dgb = _spade_backward(_lib_for(x), ctx.prec, dout, dskip, x, gb, mean, rstd, ctx.up, ctx.gb_mode, ctx.out_shape,
                                  hint=ctx.hint)
        return dx, dgb, None, None, None, None


def spade_relu(x, gb, up: bool, gb_mode: int, eps: float = 1e-5, skip: bool = False):
    """relu(IN(x) * (1 + gamma) + beta); with ``skip`` -> (that, x): x handed through for the res block's identity add."""
    del _fp8_stash[:]
    del _hint_stash[:]
    if skip:
        out, xs = _SpadeRelu.apply(x, gb, bool(up), int(gb_mode), float(eps), True)
        return _attach_hint(_attach_fp8(out)), xs
    return _attach_hint(_attach_fp8(_SpadeRelu.apply(x, gb, bool(up), int(gb_mode), float(eps))))


class _SpadeConv(torch.autograd.Function):
    """conv(relu(SPADE(up(x)))) -- architecture.py:241-245 (NormConvBlock) and :343-350 (NormResBlock) with
    normalization.py:24-37 -- for the constant-label-map case (class table, gb_mode 1), the InstanceNorm apply + modulate +
    ReLU (+ nearest x2 upsample) on the conv's operand path: one small preparation kernel (statistics finalize, interior
    coefficients, the 2-pixel frame's values) and the conv; the normalised tensor is never written.  Backward: the input
    gradient of the conv at the logical resolution, the weight gradient with the same operand-path transform, then the
    SPADE backward kernels of the unfused op."""

    @staticmethod
    def forward(ctx, x, gb, weight, up, eps, skip, cache: PackedWeights, sources, geom: ConvGeom, want_stats, ring_mode=False):
        # ring_mode (up only): instead of normalising on the conv's operand path, write z at the SOURCE resolution with the
        # interior-class coefficients (one elementwise pass over the small tensor) and let the conv read it through its fused
        # upsample, the logical frame's pixels from the ring tensor -- see fuse_ring
        prec = precision_of(x)
        x_in = x
        x, gb = x.contiguous(), gb.contiguous()
        if skip and (up or x is not x_in):
            raise ValueError("spade_conv(skip=True) wants a contiguous activation and no upsample")
        n, hs, ws, c = x.shape
        h, w = (hs * 2, ws * 2) if up else (hs, ws)
        lib = _lib_for(x)
        st = _stream()
        dev = x.device
        have = _stats_of(x, n, hs * ws, c)
        if have is not None:
            partial, chunks = have
        else:
            chunks = lib.dei2i_moments_chunks(hs * ws)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            L.check(lib.dei2i_moments_partial(prec.code, n, hs * ws, c, _p(x), _p(partial), st), "moments_partial")
        coefs = torch.empty((4, n, c), dtype=torch.float32, device=dev)          # mean | rstd | A | B
        ring = torch.empty((n, lib.dei2i_ring_pixels(h, w), c), dtype=prec.dtype, device=dev)
        L.check(lib.dei2i_spade_prep(prec.code, n, hs, ws, c, 1 if up else 0, _p(x), _p(partial), chunks, eps, _p(gb),
                                     _p(coefs[0]), _p(coefs[1]), _p(coefs[2]), _p(coefs[3]), _p(ring), st), "spade_prep")
        couts = prec.pad(geom.cout)
        d = _desc(prec, geom, n, hs, ws, c, couts)
        per_call = bool(getattr(weight, "_dei2i_per_call", False))
        if per_call:                                      # spectral norm in training mode: the packed copies travel with the call (see _Conv2d)
            cache = PackedWeights()
        wf = cache.get(weight, sources, prec, geom, c, couts, need_dgrad=any(s_.requires_grad for s_ in sources), per_call=per_call)[0]
        y = torch.empty((n, h, w, couts), dtype=prec.dtype, device=dev)
        out_partial = None
        if want_stats:
            ochunks = lib.dei2i_conv2d_stats_chunks(byref(d))
            out_partial = torch.empty((n, ochunks, 2, couts), dtype=torch.float32, device=dev)
            _stats_stash.append((out_partial, ochunks))
        z_src = None
        if ring_mode:
            z_src = torch.empty_like(x)
            L.check(lib.dei2i_affine_act_img_fwd(prec.code, n, hs * ws, c, _p(x), _p(coefs[2]), _p(coefs[3]), 0.0, _p(z_src), st),
                    "affine_act_img")
            L.check(lib.dei2i_conv2d_fwd_ring(byref(d), _p(z_src), _p(ring), _p(wf), None, L.ACT_NONE, _p(y), _p(out_partial), st),
                    "conv2d_fwd_ring")
        else:
            pro = L.ProDesc(coefs[2].data_ptr(), coefs[3].data_ptr(), c, 0.0, ring.data_ptr())
            L.check(lib.dei2i_conv2d_fwd_fused(byref(d), _p(x), _p(wf), None, L.ACT_NONE, _p(y), byref(pro), _p(out_partial), st),
                    "conv2d_fwd_fused(spade)")
        ctx.prec, ctx.up, ctx.geom, ctx.cache, ctx.sources, ctx.ring_mode, ctx.per_call = prec, up, geom, cache, sources, ring_mode, per_call
        ctx.out_shape = (n, h, w, c)
        ctx.save_for_backward(x, gb, coefs, ring, weight, z_src)
        return (y, x_in) if skip else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, gb, coefs, ring, weight, z_src = ctx.saved_tensors
        prec, geom, up = ctx.prec, ctx.geom, ctx.up
        if dy is None:                                   # only the identity branch was used
            return (dskip,) + (None,) * 10
        lib = _lib_for(x)
        dy = dy.contiguous()
        n, h, w, c = ctx.out_shape
        if dskip is not None:
            dskip = dskip.contiguous()
            if dskip.dtype != x.dtype or dskip.shape != x.shape:
                raise RuntimeError("spade_conv: identity-branch gradient does not match the activation")
        dw = None
        if _wants_grad(ctx, 2):
            if ctx.ring_mode:                            # the conv's input was z_src (+ ring): no transform in the wgrad
                dw = _conv_wgrad(lib, prec, geom, z_src, dy, weight, L.ProDesc(None, None, 0, 0.0, ring.data_ptr()), keep=(ring,))
            else:
                pro = L.ProDesc(coefs[2].data_ptr(), coefs[3].data_ptr(), c, 0.0, ring.data_ptr())
                dw = _conv_wgrad(lib, prec, geom, x, dy, weight, pro, keep=(coefs, ring))
        dx = dgb = None
        if _wants_grad(ctx, 0) or _wants_grad(ctx, 1):
            # dL/dz at the LOGICAL (upsampled) resolution: the conv seen as a plain conv on z (the SPADE backward sums the
            # 2x2 cells itself, and needs the per-logical-pixel mask and class)
            g_log = ConvGeom(geom.cin, geom.cout, geom.k, geom.stride, geom.pad, geom.reflect, False)
            hint = _NormBwdHint(1, x, coefs[0], coefs[1], gb=gb, up=up)
            dz = _conv_dgrad(lib, prec, g_log, (n, h, w, c), dy.shape[-1], dy, weight, ctx.cache, ctx.sources, ctx.per_call, x.dtype,
                             x.device, hint=hint)
            dx, dgb = _spade_backward(lib, prec, dz, dskip, x, gb, coefs[0], coefs[1], up, 1, ctx.out_shape, hint=hint)
        elif dskip is not None:
            dx = dskip
        return (dx, dgb, dw) + (None,) * 8


def spade_conv_supported(x, weight, geom: ConvGeom, need_grad: bool):
    """How can conv(relu(SPADE(up(x)))) run fused?  -> "pro" (the norm on the conv's operand path: fuse_pro), "ring" (upsampling
    blocks: z at the source resolution + the frame's ring tensor: fuse_ring) or None.  ``geom`` carries the upsample flag."""
    if not (fuse_norm and x.is_cuda and x.dtype == torch.bfloat16 and not _fp8_forward):
        return None
    lib = _lib_for(x)
    n, hs, ws, c = x.shape
    d = _desc(BF16, geom, n, hs, ws, c, BF16.pad(geom.cout))
    wg_ok = (not need_grad) or bool(lib.dei2i_conv2d_wgrad_pro_supported(byref(d)))
    if fuse_pro and wg_ok and lib.dei2i_conv2d_fused_supported(byref(d), 1):
        return "pro"
    if fuse_ring and geom.up and wg_ok and lib.dei2i_conv2d_ring_supported(byref(d)):
        return "ring"
    return None


def spade_conv(x, gb, weight, cache, geom: ConvGeom, eps: float = 1e-5, skip: bool = False, sources=None, stats=False, mode="pro"):
    """conv(relu(IN(up(x)) * (1 + gamma) + beta)) with the class table ``gb`` (N,5,5,2C); ``geom.up`` = nearest x2 upsample in
    front of the norm.  With ``skip`` -> (y, x).  ``mode``: what spade_conv_supported answered."""
    del _stats_stash[:]
    src = tuple(sources) if sources is not None else (weight,)
    ring_mode = mode == "ring"
    if skip:
        y, xs = _SpadeConv.apply(x, gb, weight, bool(geom.up), float(eps), True, cache, src, geom, bool(stats), ring_mode)
        return _attach_stats(y), xs
    return _attach_stats(_SpadeConv.apply(x, gb, weight, bool(geom.up), float(eps), False, cache, src, geom, bool(stats), ring_mode))


# --------------------------------------------------------------------------------------------------------------
# InstanceNorm2d(affine=False) + activation, AvgPool2d(2, 2): the blocks of the conv StyleExtractor (extractor.py:50-80)
# --------------------------------------------------------------------------------------------------------------
_zero_tables = {}


def _zero_table(device, dtype, n, c):
    key = (torch.device(device), dtype, n, c)
    t = _zero_tables.get(key)
    if t is None:
        if len(_zero_tables) > 16:
            _zero_tables.clear()
        t = _zero_tables[key] = torch.zeros((n, 5, 5, 2 * c), dtype=dtype, device=device)
    return t


class _InstanceNormAct(torch.autograd.Function):
    """z = act(InstanceNorm2d(x)), affine=False, eps 1e-5 (architecture.py:79-118 with norm_layer=nn.InstanceNorm2d): per-(n, c)
    statistics (the producer's records when it left any), then one affine + activation pass with per-image coefficients
    A = rstd, B = -mean * rstd.  ``slope``: 0.2 LeakyReLU, 0 ReLU, 1 none.  Backward: the SPADE backward kernels with
    gamma = beta = 0 (dei2i_in_act_bwd); ``res`` (optional) is added to the output and its gradient passed through."""

    @staticmethod
    def forward(ctx, x, slope: float, res, eps: float):
        _require_gpu(x, "instance_norm_act")
        prec = precision_of(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        lib = _lib_for(x)
        st = _stream()
        dev = x.device
        mean = torch.empty((n, c), dtype=torch.float32, device=dev)
        rstd = torch.empty((n, c), dtype=torch.float32, device=dev)
        have = _stats_of(x, n, h * w, c) if fuse_norm else None
        if have is not None:
            partial, chunks = have
        else:
            chunks = lib.dei2i_moments_chunks(h * w)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            L.check(lib.dei2i_moments_partial(prec.code, n, h * w, c, _p(x), _p(partial), st), "moments_partial")
        L.check(lib.dei2i_in_finalize_chunks(n, h * w, c, chunks, _p(partial), eps, _p(mean), _p(rstd), st), "in_finalize")
        a, b = rstd, -mean * rstd
        out = torch.empty_like(x)
        cv = c // (8 if prec is BF16 else 4)
        if 256 % cv == 0 or cv % 256 == 0:                # (the per-image kernel keeps one channel vector per thread)
            L.check(lib.dei2i_affine_act_img_fwd(prec.code, n, h * w, c, _p(x), _p(a), _p(b), slope, _p(out), st), "affine_act_img")
        else:                                             # odd channel counts: the per-channel kernel, image by image
            act = {1.0: L.ACT_NONE, 0.0: L.ACT_RELU, 0.2: L.ACT_LRELU}[slope]
            for i in range(n):
                L.check(lib.dei2i_affine_act_fwd(prec.code, h * w, c, _p(x[i]), _p(a[i]), _p(b[i]), None, act, _p(out[i]), None, 1.0, st),
                        "affine_act")
        if res is not None:
            out = out + res
        ctx.prec, ctx.slope, ctx.has_res = prec, slope, res is not None
        ctx.save_for_backward(x, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mean, rstd = ctx.saved_tensors
        prec = ctx.prec
        n, h, w, c = x.shape
        lib = _lib_for(x)
        dout = dout.contiguous()
        chunks = lib.dei2i_moments_chunks(h * w)
        partial = torch.empty((n, chunks, 4, c), dtype=torch.float32, device=x.device)
        coef = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        L.check(lib.dei2i_in_act_bwd(prec.code, n, h, w, c, _p(dout), _p(x), _p(mean), _p(rstd), ctx.slope,
                                     _p(_zero_table(x.device, x.dtype, n, c)), _p(partial), _p(coef), None, _p(dx), _stream()), "in_act_bwd")
        return dx, None, (dout if ctx.has_res else None), None


def instance_norm_act(x, act="none", res=None, eps=1e-5):
    """act(InstanceNorm2d(x)) (+ res) on an NHWC activation with H, W >= 4; act: "none" | "relu" | "leaky_relu" """
    slope = {"none": 1.0, "relu": 0.0, "leaky_relu": 0.2}[act or "none"]
    return _InstanceNormAct.apply(x, slope, res, float(eps))


# ---- SPADE's label path, second stage, batched over the modules of a generator (csrc/label_path.hip) ---------------------------
def label_gamma_beta_supported(actv, hidden: int, convs) -> bool:
    """Can ``label_gamma_beta`` take these modules?  (bf16, <= 16 modules, hidden a multiple of 32 up to 128, every norm_nc a multiple
    of 16, plain convs with a bias)"""
    if not (actv.is_cuda and actv.dtype == torch.bfloat16 and actv.dim() == 4 and tuple(actv.shape[1:3]) == (5, 5)):
        return False
    if not (1 <= len(convs) <= 16 and hidden % 32 == 0 and 32 <= hidden <= 128 and actv.shape[-1] == hidden * len(convs)):
        return False
    for g, b in convs:
        if g.bias is None or b.bias is None or g.weight.shape != b.weight.shape or tuple(g.weight.shape[1:]) != (hidden, 3, 3):
            return False
        if g.weight.shape[0] % 16 != 0 or g.weight.dtype != torch.float32 or getattr(g.weight, "_dei2i_per_call", False):
            return False
    return True


class _LabelGammaBeta(torch.autograd.Function):
    """(gamma | beta tables of every module) = conv3x3(actv slice; mlp_gamma | mlp_beta) + bias, all modules in one launch per direction
    (normalization.py:20-22,33-35 on the 5 x 5 class image).  actv: (N, 5, 5, modules * hidden), module i reads channels
    [i * hidden, (i + 1) * hidden).  params: (gamma.weight, gamma.bias, beta.weight, beta.bias) per module."""

    @staticmethod
    def forward(ctx, actv, hidden, cache, *params):
        _require_gpu(actv, "label_gamma_beta")
        actv = actv.contiguous()
        n_img, _, _, ctot = actv.shape
        nmod = len(params) // 4
        lib, st, dev = _lib_for(actv), _stream(), actv.device
        stamps = tuple(PackedWeights._stamp(params[4 * i + k]) for i in range(nmod) for k in (0, 2))
        if cache.get("stamps") != stamps:               # filters moved (an optimizer step): both bf16 layouts of every module, one launch
            cache["packed"] = [(torch.empty(lib.dei2i_label_gb_packed_elems(params[4 * i].shape[0], hidden), dtype=torch.bfloat16, device=dev),
                                torch.empty(lib.dei2i_label_gb_packed_elems(params[4 * i].shape[0], hidden), dtype=torch.bfloat16, device=dev))
                               for i in range(nmod)]
            cache["stamps"] = None
        packed = cache["packed"]
        mods = (L.LabelMod * nmod)()
        outs = []
        for i in range(nmod):
            gw, gbias, bw, bbias = (t.detach() for t in params[4 * i:4 * i + 4])
            c = gw.shape[0]
            gb = torch.empty((n_img, 5, 5, 2 * c), dtype=torch.bfloat16, device=dev)
            outs.append(gb)
            mods[i] = L.LabelMod(gw.data_ptr(), bw.data_ptr(), gbias.data_ptr(), bbias.data_ptr(), packed[i][0].data_ptr(),
                                 packed[i][1].data_ptr(), gb.data_ptr(), None, None, None, None, c, i * hidden, 1, 0)
        if cache["stamps"] is None:
            L.check(lib.dei2i_label_gb_pack(mods, nmod, hidden, st), "label_gb_pack")
            cache["stamps"] = stamps
        L.check(lib.dei2i_label_gb_fwd(mods, nmod, hidden, ctot, n_img, _p(actv), st), "label_gb_fwd")
        ctx.hidden, ctx.packed, ctx.nmod = hidden, packed, nmod
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(actv, *params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dgbs):
        actv, *params = ctx.saved_tensors
        hidden, nmod = ctx.hidden, ctx.nmod
        n_img, _, _, ctot = actv.shape
        lib, st, dev = _lib_for(actv), _stream(), actv.device
        mods = (L.LabelMod * nmod)()
        grads, keep = [], []
        for i in range(nmod):
            c = params[4 * i].shape[0]
            g = dgbs[i]
            if g is not None:
                g = g.contiguous()
                if g.dtype != torch.bfloat16 or tuple(g.shape) != (n_img, 5, 5, 2 * c):
                    raise RuntimeError("label_gamma_beta: the table gradient does not match the table")
                keep.append(g)
            dgw, dbw = torch.empty_like(params[4 * i]), torch.empty_like(params[4 * i + 2])
            dgbias, dbbias = torch.empty_like(params[4 * i + 1]), torch.empty_like(params[4 * i + 3])
            grads += [dgw, dgbias, dbw, dbbias]
            mods[i] = L.LabelMod(None, None, None, None, ctx.packed[i][0].data_ptr(), ctx.packed[i][1].data_ptr(),
                                 g.data_ptr() if g is not None else None, dgw.data_ptr(), dbw.data_ptr(), dgbias.data_ptr(), dbbias.data_ptr(),
                                 c, i * hidden, 1 if g is not None else 0, 0)
        dactv = None
        if ctx.needs_input_grad[0]:
            dactv = torch.empty_like(actv)
            L.check(lib.dei2i_label_gb_dgrad(mods, nmod, hidden, ctot, n_img, _p(dactv), st), "label_gb_dgrad")
        L.check(lib.dei2i_label_gb_wgrad(mods, nmod, hidden, ctot, n_img, _p(actv), st), "label_gb_wgrad")
        # (a module whose table got no gradient -- it did not run in this loss graph -- hands None to its filters, like the unfused graph)
        return (dactv, None, None) + tuple(gr if (ctx.needs_input_grad[3 + k] and dgbs[k // 4] is not None) else None
                                           for k, gr in enumerate(grads))


def label_gamma_beta(actv, hidden: int, convs, cache: dict):
    """-> [(N, 5, 5, 2 * norm_nc) gamma | beta table per module]; ``convs``: (mlp_gamma, mlp_beta) conv modules per SPADE module, in the
    order of their channel slices in ``actv``; ``cache``: a dict the caller keeps (the packed filters live there between steps)."""
    params = []
    for g, b in convs:
        params += [g.weight, g.bias, b.weight, b.bias]
    return list(_LabelGammaBeta.apply(actv, int(hidden), cache, *params))


class _SplitRows(torch.autograd.Function):
    """x -> consecutive row blocks of x (views, sizes given): what ``x[a:b]`` per block does, with ONE launch in backward (the
    concatenation of the blocks' gradients) instead of a zero fill + a copy per block and an add per extra block."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.set_materialize_grads(False)
        ctx.sizes, ctx.meta = sizes, (x.shape, x.dtype, x.device)
        outs, off = [], 0
        for n in sizes:
            outs.append(x.narrow(0, off, n))
            off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device = ctx.meta
        if all(g is None for g in grads):
            return (None,) + (None,) * len(ctx.sizes)
        parts = [g if g is not None else torch.zeros((n,) + tuple(shape[1:]), dtype=dtype, device=device)
                 for g, n in zip(grads, ctx.sizes)]
        return (torch.cat(parts, 0),) + (None,) * len(ctx.sizes)


def split_rows(x, sizes):
    """The consecutive row blocks of ``x`` (dim 0) as views; see _SplitRows."""
    if sum(sizes) != x.shape[0]:
        raise ValueError(f"split_rows: blocks {list(sizes)} do not cover {x.shape[0]} rows")
    if not x.requires_grad or not torch.is_grad_enabled():
        return tuple(x.split(list(sizes), 0))
    return _SplitRows.apply(x, *[int(n) for n in sizes])


class _Scale(torch.autograd.Function):
    """x * s on an NHWC activation (stargan-v2's "/ sqrt(2)" after every residual add, core/model.py:67,121)"""

    @staticmethod
    def forward(ctx, x, s: float):
        _require_gpu(x, "scale")
        x = x.contiguous()
        c = x.shape[-1]
        out = torch.empty_like(x)
        L.check(_lib_for(x).dei2i_affine_act_fwd(precision_of(x).code, x.numel() // c, c, _p(x), _p(_const_vec(x.device, c, float(s))),
                                                 _p(_const_vec(x.device, c, 0.0)), None, L.ACT_NONE, _p(out), None, 1.0, _stream()), "scale")
        ctx.s = s
        return out

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.s), None


def scale(x, s: float):
    return _Scale.apply(x, float(s))


class _Act(torch.autograd.Function):
    """a stand-alone activation of the ReLU family on an NHWC activation (stargan-v2 applies LeakyReLU(0.2) to a tensor whose
    un-activated value also feeds the block's shortcut, core/model.py:52-63)"""

    @staticmethod
    def forward(ctx, x, act: int):
        _require_gpu(x, "act")
        x = x.contiguous()
        c = x.shape[-1]
        out = torch.empty_like(x)
        L.check(_lib_for(x).dei2i_affine_act_fwd(precision_of(x).code, x.numel() // c, c, _p(x), _p(_const_vec(x.device, c, 1.0)),
                                                 _p(_const_vec(x.device, c, 0.0)), None, act, _p(out), None, 1.0, _stream()), "act")
        ctx.act = act
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return _ActBwd.apply(g.contiguous(), out, ctx.act), None


def leaky_relu(x):
    return _Act.apply(x, L.ACT_LRELU)


class _InAffineAct(torch.autograd.Function):
    """z = act(InstanceNorm2d(x) * (1 + gamma) + beta), gamma / beta (N, C) fp32 -- stargan-v2's AdaIN (core/model.py:69-80) and its
    InstanceNorm2d(affine=True) (gamma = weight - 1, beta = bias for every image) with the LeakyReLU(0.2) that follows both at every
    call site (model.py:53-61,104-112,333-334).  Forward: statistics -> per-image coefficients A = rstd (1 + gamma), B = beta - mean A
    -> one affine + activation pass.  Backward: dei2i_in_affine_act_bwd (the SPADE backward kernels in class mode with the
    activation's slope); the table's 25 classes are summed back to (N, C).  gamma / beta are rounded to the compute dtype first, so
    that forward and backward see the same values."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope: float, eps: float):
        _require_gpu(x, "in_affine_act")
        prec = precision_of(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        lib = _lib_for(x)
        st, dev = _stream(), x.device
        cl = gamma.shape[1]
        gb = torch.zeros((n, 2 * c), dtype=prec.dtype, device=dev)
        gb[:, :cl] = gamma.detach().to(prec.dtype)
        gb[:, c:c + cl] = beta.detach().to(prec.dtype)
        mean = torch.empty((n, c), dtype=torch.float32, device=dev)
        rstd = torch.empty((n, c), dtype=torch.float32, device=dev)
        have = _stats_of(x, n, h * w, c) if fuse_norm else None
        if have is not None:
            partial, chunks = have
        else:
            chunks = lib.dei2i_moments_chunks(h * w)
            partial = torch.empty((n, chunks, 2, c), dtype=torch.float32, device=dev)
            L.check(lib.dei2i_moments_partial(prec.code, n, h * w, c, _p(x), _p(partial), st), "moments_partial")
        L.check(lib.dei2i_in_finalize_chunks(n, h * w, c, chunks, _p(partial), eps, _p(mean), _p(rstd), st), "in_finalize")
        a = rstd * (1.0 + gb[:, :c].float())
        b = gb[:, c:].float() - mean * a
        out = torch.empty_like(x)
        cv = c // (8 if prec is BF16 else 4)
        if 256 % cv == 0 or cv % 256 == 0:
            L.check(lib.dei2i_affine_act_img_fwd(prec.code, n, h * w, c, _p(x), _p(a), _p(b), slope, _p(out), st), "affine_act_img")
        else:
            act = {1.0: L.ACT_NONE, 0.0: L.ACT_RELU, 0.2: L.ACT_LRELU}[slope]
            for i in range(n):
                L.check(lib.dei2i_affine_act_fwd(prec.code, h * w, c, _p(x[i]), _p(a[i].contiguous()), _p(b[i].contiguous()), None, act,
                                                 _p(out[i]), None, 1.0, st), "affine_act")
        ctx.prec, ctx.slope, ctx.cl = prec, slope, cl
        ctx.save_for_backward(x, mean, rstd, gb)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mean, rstd, gb = ctx.saved_tensors
        prec = ctx.prec
        n, h, w, c = x.shape
        if h < 4 or w < 4:
            raise NotImplementedError("in_affine_act backward needs H, W >= 4 (class-mode SPADE kernels)")
        lib = _lib_for(x)
        dout = dout.contiguous()
        table = gb.view(n, 1, 1, 2 * c).expand(n, 5, 5, 2 * c).contiguous()
        dtable = torch.empty_like(table)
        chunks = lib.dei2i_moments_chunks(h * w)
        partial = torch.empty((n, chunks, 4, c), dtype=torch.float32, device=x.device)
        coef = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        L.check(lib.dei2i_in_affine_act_bwd(prec.code, n, h, w, c, _p(dout), _p(x), _p(mean), _p(rstd), ctx.slope, _p(table), _p(dtable),
                                            _p(partial), _p(coef), None, _p(dx), _stream()), "in_affine_act_bwd")
        dgb = dtable.float().sum(dim=(1, 2))              # (N, 2C): the 25 classes hold the same (gamma | beta)
        return dx, dgb[:, :ctx.cl], dgb[:, c:c + ctx.cl], None, None


def in_affine_act(x, gamma, beta, act="leaky_relu", eps=1e-5):
    """act(InstanceNorm2d(x) * (1 + gamma) + beta) on an NHWC activation; gamma / beta: (N, C) fp32 (autograd flows into them)"""
    slope = {"none": 1.0, "relu": 0.0, "leaky_relu": 0.2}[act or "none"]
    return _InAffineAct.apply(x, gamma, beta, slope, float(eps))


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "avgpool2")
        prec = precision_of(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        out = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
        L.check(_lib_for(x).dei2i_avgpool2_fwd(prec.code, n, h, w, c, _p(x), _p(out), _stream()), "avgpool2_fwd")
        ctx.prec, ctx.shape = prec, (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        if _second_order(dout):
            return _AvgPool2Bwd.apply(dout, ctx.shape)
        n, h, w, c = ctx.shape
        dout = dout.contiguous()
        dx = torch.empty(ctx.shape, dtype=dout.dtype, device=dout.device)
        L.check(_lib_for(dout).dei2i_avgpool2_bwd(ctx.prec.code, n, h, w, c, _p(dout), _p(dx), _stream()), "avgpool2_bwd")
        return dx


class _AvgPool2Bwd(torch.autograd.Function):
    """the average pool's input gradient as a function of the output gradient (each value / 4 to its 2x2 cell); transpose: the pool"""

    @staticmethod
    def forward(ctx, dout, shape):
        n, h, w, c = shape
        dout = dout.contiguous()
        dx = torch.empty(shape, dtype=dout.dtype, device=dout.device)
        L.check(_lib_for(dout).dei2i_avgpool2_bwd(precision_of(dout).code, n, h, w, c, _p(dout), _p(dx), _stream()), "avgpool2_bwd")
        return dx

    @staticmethod
    def backward(ctx, gdx):
        return _AvgPool2.apply(gdx), None


def avgpool2(x):
    """nn.AvgPool2d(2, 2) on an NHWC activation (H, W even)"""
    return _AvgPool2.apply(x)


def upsample2(x):
    """nearest x2 upsample of an NHWC activation that no conv absorbs (stargan-v2's AdainResBlk shortcut when the channel count does not
    change, core/model.py:101-105): the average pool's adjoint spreads a value / 4 over its 2x2 cell, so this is 4 x that kernel"""
    n, h, w, c = x.shape
    return scale(_AvgPool2Bwd.apply(x, (n, 2 * h, 2 * w, c)), 4.0)


# --------------------------------------------------------------------------------------------------------------
# generator heads: tanh / sigmoid / compose  (generator.py:268-270), NaN guard (generator.py:266-267)
# --------------------------------------------------------------------------------------------------------------
class _Compose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, x_in):
        _require_gpu(raw, "compose")
        prec = precision_of(raw)
        raw = raw.contiguous()
        x32 = x_in.detach()
        if x32.dtype != torch.float32 or not x32.is_contiguous():
            x32 = x32.float().contiguous()
        n, h, w, cs = raw.shape
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=raw.device)
        prob = torch.empty((n, 1, h, w), dtype=torch.float32, device=raw.device)
        lib = _lib_for(raw)
        L.check(lib.dei2i_compose_fwd(prec.code, n, h, w, cs, _p(raw), _p(x32), _p(out), _p(prob), _stream()), "compose_fwd")
        ctx.prec = prec
        ctx.save_for_backward(raw, x32)
        return out, prob

    @staticmethod
    def backward(ctx, d_out, d_prob):
        raw, x32 = ctx.saved_tensors
        n, h, w, cs = raw.shape
        lib = _lib_for(raw)
        d_out = d_out.contiguous().float() if d_out is not None else None
        d_prob = d_prob.contiguous().float() if d_prob is not None else None
        d_raw = torch.empty_like(raw)
        d_x = torch.empty_like(x32) if ctx.needs_input_grad[1] else None
        if d_out is None and d_x is not None:
            d_x.zero_()
        L.check(lib.dei2i_compose_bwd(ctx.prec.code, n, h, w, cs, _p(raw), _p(x32), _p(d_out), _p(d_prob), _p(d_raw), _p(d_x),
                                      _stream()), "compose_bwd")
        return d_raw, d_x


def compose(raw, x_in):
    """(out, prob) = (x*(1-p) + tanh(raw[..., :3])*p, p = sigmoid(raw[..., 3])) as NCHW fp32."""
    return _Compose.apply(raw, x_in)


_nan_flags = {}


def nan_guard_(x):
    """In-place nan_to_num applied only if any NaN is present (reference semantics) -- no device->host sync."""
    _require_gpu(x, "nan_guard")
    prec = precision_of(x)
    flag = _nan_flags.get(x.device)
    if flag is None:
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        _nan_flags[x.device] = flag
    lib = _lib_for(x)
    L.check(lib.dei2i_nan_guard(prec.code, x.numel(), _p(x), _p(flag), _stream()), "nan_guard")
    return x


# --------------------------------------------------------------------------------------------------------------
# losses (models/base_model.py:68-80)
# --------------------------------------------------------------------------------------------------------------
_slabs = {}


def _zero_scalar(device) -> torch.Tensor:
    """0-dim fp32 zero carved from a pre-zeroed slab (one memset per 256 scalars instead of one per loss).  Slabs are
    never recycled, so a loss tensor stays valid for as long as anything references it."""
    slab, used = _slabs.get(device, (None, 0))
    if slab is None or used >= slab.numel():
        slab, used = torch.zeros(256, dtype=torch.float32, device=device), 0
    _slabs[device] = (slab, used + 1)
    return slab[used]


class _BceLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, tconst: float):
        _require_gpu(x, "bce_logits")
        x = x.contiguous().float()
        t = None if target is None else target.detach().contiguous().float().view(-1)
        if t is not None and t.numel() != x.numel():
            raise ValueError("bce_logits: target size mismatch")
        out = torch.empty((), dtype=torch.float32, device=x.device)      # written (not accumulated) by the finalize kernel
        lib = _lib_for(x)
        L.check(lib.dei2i_bce_logits_fwd(x.numel(), _p(x), _p(t), tconst, _p(out), _stream()), "bce_fwd")
        ctx.tconst = tconst
        ctx.save_for_backward(x, t)
        return out

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        lib = _lib_for(x)
        L.check(lib.dei2i_bce_logits_bwd(x.numel(), _p(x), _p(t), ctx.tconst, _p(g), _p(dx), _stream()), "bce_bwd")
        return dx, None, None


def bce_logits(x, target):
    """binary_cross_entropy_with_logits(x, target), mean.  ``target`` is a tensor or a python float constant."""
    if isinstance(target, (int, float)):
        return _BceLogits.apply(x, None, float(target))
    return _BceLogits.apply(x, target, 0.0)


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, "l1")
        a = a.contiguous().float()
        bb = None if b is None else b.contiguous().float()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        lib = _lib_for(a)
        L.check(lib.dei2i_l1_fwd(a.numel(), _p(a), _p(bb), _p(out), _stream()), "l1_fwd")
        ctx.save_for_backward(a, bb)
        return out

    @staticmethod
    def backward(ctx, g):
        a, bb = ctx.saved_tensors
        g = g.contiguous().float()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(bb) if (bb is not None and ctx.needs_input_grad[1]) else None
        if da is None and db is None:
            return None, None
        lib = _lib_for(a)
        L.check(lib.dei2i_l1_bwd(a.numel(), _p(a), _p(bb), _p(g), _p(da), _p(db), _stream()), "l1_bwd")
        return da, db


def l1(a, b=None):
    """l1_loss(a, b), mean; ``b=None`` means zeros (sd_con, defectgan_model.py:231-236)."""
    return _L1.apply(a, b)
