"""Micro-benchmark (not a test): the small / split-K gather GEMMs of the step (SPADE class tables, reflect-ring dgrad),
timed through the C ABI in a tight loop (hot caches) -- compare with their ~20 us in the step profile."""
import sys
from ctypes import byref, c_int
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib as L

DEV = "cuda:0"
lib = L.load()
prec = ops.get_precision("bf16")
p = ops._p
st = ops._stream()
# name, cin, cout, k, pad, reflect, H, W, N
CASES = [
    ("spade_gb_table 128>512 @5x5", 128, 512, 3, 1, False, 5, 5, 16),
    ("spade_gb_table 128>256 @5x5", 128, 256, 3, 1, False, 5, 5, 16),
    ("spade_shared 6>128 @5x5", 6, 128, 3, 1, False, 5, 5, 16),
    ("res 256>256 @64 (ring dgrad)", 256, 256, 3, 1, True, 64, 64, 16),
]


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for kv in sys.argv[1:]:
    name, val = kv.split("=")
    lib.dei2i_set_option(name.encode(), int(val))
for name, cin, cout, k, pad, refl, H, W, N in CASES:
    geom = ops.ConvGeom(cin, cout, k, 1, pad, refl, False)
    cins, couts = prec.pad(cin), prec.pad(cout)
    x = torch.randn(N, H, W, cins, device=DEV).to(prec.dtype)
    w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
    cache = ops.PackedWeights()
    wf, wd = cache.get(w, (w,), prec, geom, cins, couts, need_dgrad=True)
    d = ops._desc(prec, geom, N, H, W, cins, couts)
    y = torch.empty((N, H, W, couts), dtype=prec.dtype, device=DEV)
    ws = ops._workspace(x.device, lib.dei2i_conv2d_workspace_bytes(byref(d)))
    t_f = timeit(lambda: lib.dei2i_conv2d_fwd(byref(d), p(x), p(wf), None, 0, p(y), p(ws), ws.numel() * 4, st))
    dx = torch.empty_like(x)
    ho, wo = c_int(), c_int()
    lib.dei2i_conv2d_dgrad_shape(byref(d), byref(ho), byref(wo))
    ext = torch.empty(N * ho.value * wo.value * cins, dtype=prec.dtype, device=DEV) if refl else None
    t_d = timeit(lambda: lib.dei2i_conv2d_dgrad_input(byref(d), p(y), p(wd), p(ext), p(dx), p(ws), ws.numel() * 4, st))
    print(f"{name:34s} fwd {t_f:7.1f} us | dgrad_input {t_d:7.1f} us", flush=True)
