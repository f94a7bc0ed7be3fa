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


# "cold": 640 MB are written between launches, so the operands come from HBM as they do inside the train step (warm: the 100 MB
# of g, x and dx stay in the 256 MB Infinity Cache from launch to launch)
flush = torch.empty(640 << 20, dtype=torch.uint8, device=DEV)
wf, _ = ops.PackedWeights().get(w, (w,), ops.BF16, geom, cin, cout, need_dgrad=False)
y = torch.empty(N, H, H, cout, device=DEV, dtype=torch.bfloat16)


def fwd():
    L.check(lib.dei2i_conv2d_fwd(byref(d), ops._p(x), ops._p(wf), None, 0, ops._p(y), ops._p(ws), ws.numel() * 4, ops._stream()), "fwd")


def timed(fn, fam, cold):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.dei2i_prof_enable(fam, 1)
    for _ in range(20):
        if cold:
            flush.fill_(1)
        fn()
    torch.cuda.synchronize()
    n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    L.check(lib.dei2i_prof_collect(fam, byref(n), byref(ms), byref(fl)), "prof_collect")
    lib.dei2i_prof_enable(fam, 0)
    return ms.value * 1e3 / max(n.value, 1)


print("%-34s %9s %9s   (us per launch, 20 launches each)" % ("", "warm", "cold"))
print("%-34s %9.1f %9.1f" % ("forward (pipelined loop)", timed(fwd, 2, False), timed(fwd, 2, True)), flush=True)
for name, kind in (("dgrad alone (FOLD)", None), ("+ SPADE reductions", 1), ("  without the arithmetic", 1 | 0x100), ("  without the reduction", 1 | 0x200),
                   ("  without both (x loads only)", 1 | 0x300), ("+ BatchNorm reductions", 2), ("dgrad alone (FOLD)", None), ("+ SPADE reductions", 1)):
    if kind == 2:
        a_, b_ = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV)
        m2, r2 = torch.randn(cin, device=DEV), torch.rand(cin, device=DEV) + 0.5
        p2 = torch.empty(N * chunks, 2, cin, device=DEV)

        def f(kind=kind):
            en = L.EpiNormDesc(2, 0, 2, 0, x.data_ptr(), m2.data_ptr(), r2.data_ptr(), None, a_.data_ptr(), b_.data_ptr(), p2.data_ptr())
            L.check(lib.dei2i_conv2d_dgrad_input_norm(byref(d), ops._p(g), ops._p(wd), ops._p(dx), byref(en), ops._stream()), "dgrad_norm")
    else:
        def f(kind=kind):
            run(kind)
    print("%-34s %9.1f %9.1f" % (name, timed(f, PROF, False), timed(f, PROF, True)), flush=True)
