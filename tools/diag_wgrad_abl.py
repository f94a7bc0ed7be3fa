"""Timing-only ablations of the halo-resident wgrad kernel (option wgrad_halo_abl; results are wrong in those builds): what does
each part of the loop cost on the res-block shape?  usage: diag_wgrad_abl.py [cin cout H N]"""
import ctypes
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
cin, cout, H, N = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 256, 64, 16)
lib = ops._lib_for(torch.zeros(1, device=DEV))
geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, False)
x = torch.randn(N, H, H, cin, device=DEV).bfloat16()
g = torch.randn(N, H, H, cout, device=DEV).bfloat16()
d = ops._desc(ops.BF16, geom, N, H, H, cin, cout)
scratch = ops._wgrad_scratch(lib, d, DEV)
dw = torch.empty(cout, cin, 3, 3, device=DEV)
PROF_WGRAD = 1
flops = 2.0 * N * H * H * 9 * cin * cout
names = {0: "full kernel", 1: "no LDS-DMA after the first half-tile", 2: "no fragment reads", 4: "no MFMAs", 8: "no slab stores", 3: "no DMA, no reads",
         6: "no reads, no MFMAs", 7: "no DMA / reads / MFMAs", 12: "no MFMAs, no stores", 15: "nothing (launch + prologue)"}
names[16] = "general-path address arithmetic, no DMA instruction"
for abl in (0, 8, 1, 16, 2, 4, 3, 6, 12, 7, 15, 0):
    lib.dei2i_set_option(b"wgrad_halo_abl", abl)
    for _ in range(3):
        L.check(lib.dei2i_conv2d_wgrad_oihw(byref(d), ops._p(x), ops._p(g), ops._p(scratch), scratch.numel(), ops._p(dw), 0, ops._stream()), "wgrad")
    torch.cuda.synchronize()
    lib.dei2i_prof_enable(PROF_WGRAD, 1)
    for _ in range(20):
        L.check(lib.dei2i_conv2d_wgrad_oihw(byref(d), ops._p(x), ops._p(g), ops._p(scratch), scratch.numel(), ops._p(dw), 0, ops._stream()), "wgrad")
    torch.cuda.synchronize()
    n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    L.check(lib.dei2i_prof_collect(PROF_WGRAD, byref(n), byref(ms), byref(fl)), "prof_collect")
    lib.dei2i_prof_enable(PROF_WGRAD, 0)
    us = ms.value * 1e3 / max(n.value, 1)
    print("abl %2d  %-40s %7.1f us/launch (%d launches)%s" % (abl, names[abl], us, n.value, "  %.0f TF/s" % (flops / us / 1e6) if abl == 0 else ""), flush=True)
lib.dei2i_set_option(b"wgrad_halo_abl", 0)
# in-kernel stamps of the full kernel (one launch)
dbg = torch.zeros(256 * 8 * 8 * 2, dtype=torch.int64, device=DEV)
lib.dei2i_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
L.check(lib.dei2i_conv2d_wgrad_oihw(byref(d), ops._p(x), ops._p(g), ops._p(scratch), scratch.numel(), ops._p(dw), 0, ops._stream()), "wgrad")
torch.cuda.synchronize()
lib.dei2i_set_debug_buffer(None)
r = dbg.view(-1, 8).cpu()
r = r[r[:, 4] > 0].double()
t0 = r[:, 0].min()
print("waves with stamps: %d" % r.shape[0])
print("entry skew (100 MHz ticks): min 0, median %.0f, max %.0f; exit: min %.0f median %.0f max %.0f  => kernel span %.1f us" % (
    (r[:, 0] - t0).median(), (r[:, 0] - t0).max(), (r[:, 4] - t0).min(), (r[:, 4] - t0).median(), (r[:, 4] - t0).max(), (r[:, 4] - t0).max() / 100.0))
for i, nm in ((1, "cycles to first compute"), (2, "cycles to end of loop"), (3, "cycles to stores complete"), (5, "waiting at the top of trips"),
              (6, "issuing LDS-DMA"), (7, "in compute()")):
    print("  %-30s mean %9.0f  min %9.0f  max %9.0f" % (nm, r[:, i].mean(), r[:, i].min(), r[:, i].max()))
print("  wave lifetime in 100 MHz ticks: mean %.0f (=> %.2f GHz from cycles / ticks)" % ((r[:, 4] - r[:, 0]).mean(), (r[:, 3] / ((r[:, 4] - r[:, 0]) * 10.0)).mean()))
