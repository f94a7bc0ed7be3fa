"""What does the norm-backward epilogue of the reflect dgrad cost (conv_halo16.hip EPIN)?  The res-block dgrad alone, with the SPADE
reductions, and with timing-only switches (kind bits 8 / 9: no arithmetic / no reduction; results are wrong with those)."""
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
d = ops._desc(ops.BF16, geom, N, H, H, cin, cout)
g = torch.randn(N, H, H, cout, device=DEV).bfloat16()
x = torch.randn(N, H, H, cin, device=DEV).bfloat16()
w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.02
cache = ops.PackedWeights()
_, wd = cache.get(w, (w,), ops.BF16, geom, cin, cout, need_dgrad=True, need_fwd=False)
mean, rstd = torch.randn(N, cin, device=DEV), torch.rand(N, cin, device=DEV) + 0.5
gb = torch.randn(N, 5, 5, 2 * cin, device=DEV).bfloat16()
chunks = lib.dei2i_conv2d_dgrad_norm_chunks(byref(d))
partial = torch.empty(N, chunks, 4, cin, device=DEV)
dx = torch.empty(N, H, H, cin, device=DEV, dtype=torch.bfloat16)
ws = ops._workspace(DEV, lib.dei2i_conv2d_workspace_bytes(byref(d)))
PROF = 3    # PROF_HALO_FOLD
assert lib.dei2i_conv2d_dgrad_norm_supported(byref(d))


def run(kind):
    if kind is None:
        L.check(lib.dei2i_conv2d_dgrad_input(byref(d), ops._p(g), ops._p(wd), ops._p(dx), ops._p(dx), ops._p(ws), ws.numel() * 4, ops._stream()), "dgrad")
    else:
        en = L.EpiNormDesc(kind, 0, 0, 0, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gb.data_ptr(), None, None, partial.data_ptr())
        L.check(lib.dei2i_conv2d_dgrad_input_norm(byref(d), ops._p(g), ops._p(wd), ops._p(dx), byref(en), ops._stream()), "dgrad_norm")


for name, kind in (("dgrad alone", None), ("+ SPADE reductions", 1), ("  without the arithmetic", 1 | 0x100), ("  without the reduction", 1 | 0x200),
                   ("  without both (x loads only)", 1 | 0x300), ("dgrad alone", None), ("+ SPADE reductions", 1)):
    for _ in range(3):
        run(kind)
    torch.cuda.synchronize()
    lib.dei2i_prof_enable(PROF, 1)
    for _ in range(20):
        run(kind)
    torch.cuda.synchronize()
    n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    L.check(lib.dei2i_prof_collect(PROF, byref(n), byref(ms), byref(fl)), "prof_collect")
    lib.dei2i_prof_enable(PROF, 0)
    print("%-34s %7.1f us/launch (%d launches)" % (name, ms.value * 1e3 / max(n.value, 1), n.value), flush=True)
