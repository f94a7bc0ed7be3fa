"""Diagnostic: per-phase cycle stamps of the 16x32-tile halo conv (conv_halo16.hip DIAG build, option v2_ablate = 6)."""
import sys, ctypes
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib
DEV = "cuda:0"
lib = _lib.load()
cin, cout, H, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 64, 16)))
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib.dei2i_set_option(b"halo16", mode)
abl = int(sys.argv[6]) if len(sys.argv) > 6 else 6
if abl > 10:        # timing-only builds without stamps: 10 + bit mask (1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs)
    lib.dei2i_set_option(b"v2_ablate", abl)
geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, False)
x = torch.randn(N, H, H, cin, device=DEV).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
cache = ops.PackedWeights()
for _ in range(200):
    y = ops.conv2d(x, w, None, cache, geom, "none")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    y = ops.conv2d(x, w, None, cache, geom, "none")
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
print("plain kernel wall %.1f us = %.0f TFLOP/s" % (us, 2.0 * N * H * H * cin * cout * 9 / us / 1e6))
if abl > 10 and mode != 3:
    sys.exit(0)
if mode == 3:       # the pipelined loop writes its cycle count whenever a debug buffer is set (every ablation build too)
    dbg = torch.zeros(4096 * 8 * 10, dtype=torch.int64, device=DEV)
    lib.dei2i_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    for _ in range(20):
        y = ops.conv2d(x, w, None, cache, geom, "none")
    torch.cuda.synchronize()
    d = dbg.view(-1, 10).cpu().double()
    d = d[d[:, 2] > 0]
    print("pipelined loop: median %.0f cycles per wave = %.0f per k-step; clock %.3f GHz" % (
        d[:, 0].median().item(), d[:, 0].median().item() / d[0, 2].item(), (d[:, 0] / d[:, 1]).median().item() * 0.1))
    sys.exit(0)
nwg = 4096
dbg = torch.zeros(nwg * 8 * 10, dtype=torch.int64, device=DEV)
lib.dei2i_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
lib.dei2i_set_option(b"v2_ablate", int(sys.argv[6]) if len(sys.argv) > 6 else 6)     # 7 / 8 / 9: timing-only ablations (no DMA / no fragment reads / no MFMA)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.conv2d(x, w, None, cache, geom, "none")
e1.record()
torch.cuda.synchronize()
print("diag kernel wall %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
raw = dbg.view(-1, 10).cpu()
pro = (raw[:, 3] >> 32).double()
raw[:, 3] &= 0xffffffff
d = raw.double()
if True:
    live = d[:, 2] > 0
    print("prologue %.0f cycles | epilogue %.0f cycles (medians over waves)" % (pro[live].median().item(), (d[live, 3] - d[live, 0] - pro[live]).median().item()))
nw = 8
d = d.view(-1, 8, 10)[:, :nw]
d = d[d[:, 0, 2] > 0]
nk = d[0, 0, 2].item()
print("workgroups", d.shape[0], "k-steps", nk)
print("loop cycles per wave: median %.0f -> %.0f per k-step; clock %.3f GHz; whole kernel %.0f cycles (prologue + epilogue %.0f)" % (
    d[..., 0].median().item(), d[..., 0].median().item() / nk, (d[..., 0] / d[..., 1]).median().item() * 0.1,
    d[..., 3].median().item(), d[..., 3].median().item() - d[..., 0].median().item()))
for grp, name in ((0, "group 0 (waves 0-3)"), (1, "group 1 (waves 4-7)")):
    sel = d[:, grp * 4:grp * 4 + 4, 4:].reshape(-1, 6)
    print(name, "cycles per k-step: M reads issued %.0f | M vmcnt wait %.0f | M lgkmcnt wait %.0f | M barrier %.0f | C mfma+dma %.0f | C barrier %.0f"
          % tuple((sel.median(0).values / nk).tolist()))
