"""Micro-benchmark (not a test): the thin convs of the 256x256 step (stem 3->64 7x7 forward / its input gradient, heads 64->4 forward /
input gradient, D's first conv), kernel time from the library's events, operands warm / cold (640 MB written between launches)."""
import ctypes
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
lib = ops._lib_for(torch.zeros(1, device=DEV))
flush = torch.empty(640 << 20, dtype=torch.uint8, device=DEV)
PROF = 0


def timed(fn, cold):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.dei2i_prof_enable(PROF, 1)
    for _ in range(10):
        if cold:
            flush.fill_(1)
        fn()
    torch.cuda.synchronize()
    n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    L.check(lib.dei2i_prof_collect(PROF, byref(n), byref(ms), byref(fl)), "prof_collect")
    lib.dei2i_prof_enable(PROF, 0)
    return ms.value * 1e3 / max(n.value, 1)


# name, cin, cout, k, stride, pad, H, N, act
SHAPES = [("stem 3>64 7x7 @256 N32", 3, 64, 7, 1, 3, 256, 32, "none"), ("heads 64>4 3x3 @256 N16", 64, 4, 3, 1, 1, 256, 16, "none"),
          ("D0 3>64 4x4 s2 @256 N64", 3, 64, 4, 2, 1, 256, 64, "leaky_relu")]
print("%-28s %10s %18s %18s   (us per launch, warm / cold)" % ("", "", "forward", "input gradient"))
for name, cin, cout, k, s, pad, H, N, act in SHAPES:
    geom = ops.ConvGeom(cin, cout, k, s, pad, True, False)
    x = ops.to_nhwc(torch.randn(N, cin, H, H, device=DEV), ops.BF16).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, device=DEV) * 0.05).requires_grad_(True)
    cache = ops.PackedWeights()
    y = ops.conv2d(x, w, None, cache, geom, act)
    gy = torch.randn_like(y)
    L.launch_counts(reset=True)
    with torch.no_grad():
        f = lambda: ops.conv2d(x.detach(), w.detach(), None, cache, geom, act)      # noqa: E731
        tf = (timed(f, False), timed(f, True))
    ff = {k_: v for k_, v in L.launch_counts(reset=True).items() if v}

    def b():
        return torch.autograd.grad(y, [x], gy, retain_graph=True)
    tb = (timed(b, False), timed(b, True))
    fb = {k_: v for k_, v in L.launch_counts(reset=True).items() if v}
    print("%-28s %10s %8.1f /%7.1f  %8.1f /%7.1f   fwd %s | dgrad %s" % (name, "", tf[0], tf[1], tb[0], tb[1], sorted(ff), sorted(fb)), flush=True)
