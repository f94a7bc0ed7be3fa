"""Does the relative alignment of the streams of a 2-read + 1-write elementwise kernel matter?  bn_bwd_apply / spade_bwd_apply /
torch.add with the second input and the output shifted by `skew` bytes inside over-allocated buffers."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
lib = ops._lib_for(torch.zeros(1, device=DEV))
st = ops._stream()
p = ops._p
N = 16


def carve(shape, skew):
    n = 1
    for s in shape:
        n *= s
    buf = torch.empty(n + skew // 2 + 64, dtype=torch.bfloat16, device=DEV)
    t = buf[skew // 2: skew // 2 + n].view(shape)
    t.normal_()
    return t


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (H, W, C) in ((64, 64, 256), (128, 128, 128), (256, 256, 64)):
    pix = N * H * W
    T = pix * C * 2
    a = torch.rand(C, device=DEV) + 0.5
    b = torch.randn(C, device=DEV)
    mean = torch.randn(C, device=DEV)
    rstd = torch.rand(C, device=DEV) + 0.5
    bchunks = lib.dei2i_bn_bwd_chunks(pix)
    bpart = torch.zeros(bchunks, 2, C, device=DEV)
    dwt, dbs = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    for skew in (0, 256, 4096 + 256, 65536 + 2048 + 256, (1 << 20) + 8192 + 512):
        dz = carve((N, H, W, C), 0)
        y = carve((N, H, W, C), skew)
        out = carve((N, H, W, C), 2 * skew)
        t_bn = timeit(lambda: lib.dei2i_bn_bwd_apply(0, pix, C, p(dz), p(y), p(a), p(b), p(mean), p(rstd), L.ACT_LRELU, 1, p(bpart), bchunks,
                                                     p(dwt), p(dbs), None, None, p(out), st))
        t_add = timeit(lambda: torch.add(dz, y, out=out))
        print("%dx%dx%d skew %8d: bn_bwd_apply %6.1f us %5.2f TB/s | torch.add %6.1f us %5.2f TB/s | ptr mod 2MB: %x %x %x" % (
            H, W, C, skew, t_bn, 3 * T / t_bn / 1e6, t_add, 3 * T / t_add / 1e6, dz.data_ptr() % (2 << 20), y.data_ptr() % (2 << 20),
            out.data_ptr() % (2 << 20)), flush=True)
        del dz, y, out
