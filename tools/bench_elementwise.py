"""Micro-benchmark of the HBM-bound kernels at the 256x256 batch-16 (or, with --size 512, the 512x512 batch-4) activation shapes: microseconds per launch and the
achieved fraction of HBM bandwidth on ALGORITHMIC bytes (tensors each kernel must read / write once)."""
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
lib = ops._lib_for(torch.zeros(1, device=DEV))
st = ops._stream()
p = ops._p
BF = 0      # dtype code bf16
# usage: bench_elementwise.py [name filter] [--size 512]   (--size 512: the batch-4 512x512 generator's activation shapes)
SIZE = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 256
_args = [a for i, a in enumerate(sys.argv[1:], 1) if a != "--size" and sys.argv[i - 1] != "--size"]
N = 16 if SIZE == 256 else 4
only = _args[0] if _args else ""
SHAPES = ((256, 256, 64), (128, 128, 128), (64, 64, 256)) if SIZE == 256 else ((512, 512, 64), (256, 256, 128), (128, 128, 256), (64, 64, 512))


def timeit(name, fn, nbytes, reps=20):
    if only and only not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-44s %9.1f us  %7.1f MB  %6.2f TB/s  (%4.1f%% of 8 TB/s)" % (name, us, nbytes / 1e6, nbytes / us / 1e6, nbytes / us / 1e6 / 8 * 100), flush=True)


for (H, W, C) in SHAPES:
    tag = "%dx%dx%d" % (H, W, C)
    pix = N * H * W
    T = pix * C * 2                      # bytes of one bf16 activation
    x = torch.randn(N, H, W, C, device=DEV).bfloat16()
    y = torch.randn(N, H, W, C, device=DEV).bfloat16()
    dz = torch.randn(N, H, W, C, device=DEV).bfloat16()
    out = torch.empty_like(x)
    out2 = torch.empty_like(x)
    a = torch.rand(C, device=DEV) + 0.5
    b = torch.randn(C, device=DEV)
    mean = torch.randn(C, device=DEV)
    rstd = torch.rand(C, device=DEV) + 0.5
    chunks = lib.dei2i_moments_chunks(H * W)
    partial = torch.empty(N, chunks, 4, C, device=DEV)
    timeit(tag + " moments_partial", lambda: lib.dei2i_moments_partial(BF, N, H * W, C, p(x), p(partial), st), T)
    timeit(tag + " affine_act", lambda: lib.dei2i_affine_act_fwd(BF, pix, C, p(x), p(a), p(b), None, L.ACT_LRELU, p(out), None, 1.0, st), 2 * T)
    timeit(tag + " affine_act+res", lambda: lib.dei2i_affine_act_fwd(BF, pix, C, p(x), p(a), p(b), p(y), L.ACT_NONE, p(out), None, 1.0, st), 3 * T)
    timeit(tag + " act_bwd", lambda: lib.dei2i_act_bwd(BF, pix * C, p(dz), p(y), L.ACT_LRELU, p(out), st), 3 * T)
    bchunks = lib.dei2i_bn_bwd_chunks(pix)
    bpart = torch.empty(bchunks, 2, C, device=DEV)
    dwt, dbs = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    csp = torch.empty(lib.dei2i_colsum_blocks(pix) * C, device=DEV)
    timeit(tag + " bn_bwd_partial", lambda: lib.dei2i_bn_bwd_partial(BF, pix, C, p(dz), p(y), p(a), p(b), p(mean), p(rstd), L.ACT_LRELU, p(bpart), st), 2 * T)
    timeit(tag + " bn_bwd_apply", lambda: lib.dei2i_bn_bwd_apply(BF, pix, C, p(dz), p(y), p(a), p(b), p(mean), p(rstd), L.ACT_LRELU, 1, p(bpart), bchunks, p(dwt), p(dbs), None, None, p(out), st), 3 * T)
    ext = torch.randn(N, H + 2, W + 2, C, device=DEV).bfloat16()
    timeit(tag + " fold_pad reflect1", lambda: lib.dei2i_fold_pad(BF, N, H, W, C, 1, L.PAD_REFLECT, 0, p(ext), None, p(out), st), 2 * T)
    # SPADE (class-mode gamma/beta table), no upsample
    mean_n = torch.randn(N, C, device=DEV)
    rstd_n = torch.rand(N, C, device=DEV) + 0.5
    gb = torch.randn(N, 5, 5, 2 * C, device=DEV).bfloat16()
    timeit(tag + " spade_act", lambda: lib.dei2i_spade_act_fwd(BF, N, H, W, C, 0, p(x), p(mean_n), p(rstd_n), p(gb), 1, p(out), None, 1.0, st), 2 * T)
    dgb = torch.empty(N, 5, 5, 2 * C, device=DEV)
    coef = torch.empty(N, 2, C, device=DEV)
    z = torch.relu(torch.randn(N, H, W, C, device=DEV)).bfloat16()
    timeit(tag + " spade_bwd_partial", lambda: lib.dei2i_spade_bwd_partial(BF, N, H, W, C, 0, p(dz), p(x), p(mean_n), p(rstd_n), p(gb), 1, p(dgb), p(partial), st), 2 * T)
    timeit(tag + " spade_bwd_apply", lambda: lib.dei2i_spade_bwd_apply(BF, N, H, W, C, 0, p(dz), p(x), p(mean_n), p(rstd_n), p(gb), 1, p(partial), chunks, p(dgb), p(coef), None, p(out), st), 3 * T)
    timeit(tag + " colsum", lambda: lib.dei2i_colsum(BF, pix, C, p(dz), p(csp), p(dwt), st), T)
    timeit(tag + " torch add (reference point)", lambda: torch.add(x, y, out=out), 3 * T)
    del x, y, dz, out, out2, ext, z
