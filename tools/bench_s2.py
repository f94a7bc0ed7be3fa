"""Micro-benchmark (not a test): forward of the 4x4 stride-2 convs of the 256x256 step on the halo kernel's stride-2 form
(conv_halo16.hip S2) against the LDS-DMA gather GEMM (option halo16_s2 = 0), same box, kernel time from the library's events."""
import ctypes
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
SHAPES = [("enc0 64>128 @256 N16", 64, 128, 256, 16), ("enc1 128>256 @128 N16", 128, 256, 128, 16), ("enc0 N32 (eval pass)", 64, 128, 256, 32),
          ("D1 64>128 @128 N64", 64, 128, 128, 64), ("D2 128>256 @64 N64", 128, 256, 64, 64), ("D3 256>512 @32 N64", 256, 512, 32, 64)]
lib = ops._lib_for(torch.zeros(1, device=DEV))
flush = torch.empty(640 << 20, dtype=torch.uint8, device=DEV)
PROF = 0    # PROF_GATHER_GEMM: both kernels report there


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
    return ms.value * 1e3 / max(n.value, 1), fl.value / max(n.value, 1)


print("%-26s %22s %22s   (us per launch warm / cold, TFLOP/s cold)" % ("", "halo S2", "gather GEMM v2"))
for name, cin, cout, H, N in SHAPES:
    geom = ops.ConvGeom(cin, cout, 4, 2, 1, True, False)
    x = torch.randn(N, H, H, cin, device=DEV).bfloat16()
    w = torch.randn(cout, cin, 4, 4, device=DEV) * 0.05
    cache = ops.PackedWeights()
    row, outs = [], []
    for opt in (1, 0):
        L.check(lib.dei2i_set_option(b"halo16_s2", opt), "opt")
        L.launch_counts(reset=True)
        y = ops.conv2d(x, w, None, cache, geom, "leaky_relu")
        fam = {k: v for k, v in L.launch_counts(reset=True).items() if v}
        outs.append(y)
        f = lambda: ops.conv2d(x, w, None, cache, geom, "leaky_relu")      # noqa: E731
        (tw, fl), (tc, _) = timed(f, False), timed(f, True)
        row.append("%6.1f /%6.1f  %5.0f %s" % (tw, tc, fl / tc / 1e6, list(fam)[0][:9]))
    L.check(lib.dei2i_set_option(b"halo16_s2", 1), "opt")
    d = (outs[0].float() - outs[1].float()).abs().max().item() / outs[1].float().abs().max().item()
    print("%-26s %30s %30s   maxrel %.1e" % (name, row[0], row[1], d), flush=True)
