"""Diagnostic: in-kernel cycles per k-step and the clock the halo conv kernel holds (s_memtime / s_memrealtime)."""
import sys, ctypes
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib
DEV = "cuda:0"
lib = _lib.load()
cin, cout, k, H, N = 256, 256, 3, 64, 16
geom = ops.ConvGeom(cin, cout, k, 1, 1, True, False)
x = torch.randn(N, H, H, cin, device=DEV).to(torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
cache = ops.PackedWeights()
for _ in range(200):
    y = ops.conv2d(x, w, None, cache, geom, "none")
dbg = torch.zeros(512 * 8 * 4 + 512 * 8 * 6, dtype=torch.int64, device=DEV)
lib.dei2i_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib.dei2i_set_option(b"v2_ablate", mode)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.conv2d(x, w, None, cache, geom, "none")
e1.record()
torch.cuda.synchronize()
print("mode", mode, "kernel wall %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
if mode not in (5, 6):
    sys.exit(0)
if mode == 6:
    e = dbg[512 * 8 * 4:].view(-1, 6).cpu().double()
    nk_ = 36.0
    for grp, name in ((0, "group 0 (waves 0-3)"), (1, "group 1 (waves 4-7)")):
        sel = e.view(512, 8, 6)[:, grp * 4:grp * 4 + 4].reshape(-1, 6)
        print(name, "cycles per k-step: issue %.0f | reads issued %.0f | vmcnt wait %.0f | lgkmcnt wait %.0f | barrier(M) %.0f | C phase + barrier %.0f" % tuple((sel.median(0).values / nk_).tolist()))
d = dbg[:512 * 8 * 4].view(-1, 4).cpu().double()
d = d[d[:, 2] > 0]
cyc, real, nk = d[:, 0], d[:, 1], d[:, 2]
print("waves", len(d), "k-steps", nk[0].item())
print("loop cycles per wave: median %.0f  -> %.0f cycles per k-step" % (cyc.median().item(), (cyc / nk).median().item()))
print("clock: median %.3f GHz" % ((cyc / real).median().item() * 0.1))
print("loop wall per tile: %.1f us" % (real.median().item() / 100.0))
if mode == 5:
    tot = dbg[512 * 8 * 4:].view(-1, 6)[:, 0].cpu().double()
    tot = tot[tot > 0]
    print("whole kernel per tile: median %.0f cycles -> prologue + epilogue %.0f cycles" % (tot.median().item(), tot.median().item() - cyc.median().item()))
