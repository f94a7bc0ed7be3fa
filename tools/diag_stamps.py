import sys, ctypes
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib
DEV = "cuda:0"
prec = ops.BF16
lib = _lib.load()
cin, cout, k, H, N = 256, 256, 3, 64, 16
geom = ops.ConvGeom(cin, cout, k, 1, 1, True, False)
x = torch.randn(N, H, H, cin, device=DEV).to(torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
cache = ops.PackedWeights()
for _ in range(3): y = ops.conv2d(x, w, None, cache, geom, "none")
dbg = torch.zeros(4096 * 4, dtype=torch.int64, device=DEV)
lib.dei2i_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
lib.dei2i_set_option(b"v2_ablate", 5)
torch.cuda.synchronize()
y = ops.conv2d(x, w, None, cache, geom, "none")
torch.cuda.synchronize()
d = dbg.view(-1, 4)[:512].cpu().double()
t0 = d[:, 0].min()
print("WG start spread (us @100MHz ticks?):", ((d[:, 0] - t0).max()).item())
print("prologue avg", (d[:, 1] - d[:, 0]).mean().item(), "loop avg", (d[:, 2] - d[:, 1]).mean().item(), "epilogue avg", (d[:, 3] - d[:, 2]).mean().item())
print("total span", (d[:, 3].max() - t0).item(), "per-WG total avg", (d[:, 3] - d[:, 0]).mean().item())
first = d[d[:, 0] < t0 + (d[:,3]-d[:,0]).mean() * 0.5]
print("n first-round WGs", len(first))
