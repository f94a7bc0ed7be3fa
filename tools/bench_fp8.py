"""Micro-benchmark (not a test): fp8 (e4m3) forward conv against the bf16 kernel on the hot 3x3 shapes -- time per call
(C ABI, tight loop), including / excluding the activation quantisation pass, and the rms error against bf16."""
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib as L

DEV = "cuda:0"
lib = L.load()
p, st = ops._p, ops._stream()
CASES = [("res 256>256 @64 b16", 256, 256, 64, 16, False), ("dec0 256>128 @64 up b16", 256, 128, 64, 16, True),
         ("dec1 128>64 @128 up b16", 128, 64, 128, 16, True)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, cin, cout, h, n, up in CASES:
    geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, up)
    x = torch.randn(n, h, h, cin, device=DEV).relu_().to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.02
    cache = ops.PackedWeights()
    d = ops._desc(ops.BF16, geom, n, h, h, cin, cout)
    wf, _ = cache.get(w, (w,), ops.BF16, geom, cin, cout, need_dgrad=False)
    wq, dq = cache.get_fp8(w, (w,), ops.BF16, geom, cin, cout)
    ho = h * 2 if up else h
    y = torch.empty(n, ho, ho, cout, dtype=torch.bfloat16, device=DEV)
    y8 = torch.empty_like(y)
    xq = torch.empty(x.numel(), dtype=torch.uint8, device=DEV)
    ws = ops._workspace(x.device, lib.dei2i_conv2d_workspace_bytes(byref(d)))
    t_bf = timeit(lambda: lib.dei2i_conv2d_fwd(byref(d), p(x), p(wf), None, 0, p(y), p(ws), ws.numel() * 4, st))
    t_q = timeit(lambda: lib.dei2i_quantize_fp8(x.numel(), p(x), ops.FP8_ACT_SCALE, p(xq), st))
    t_8 = timeit(lambda: lib.dei2i_conv2d_fwd_fp8(byref(d), p(xq), p(wq), None, p(dq), 0, p(y8), st))
    err = float((y8.float() - y.float()).pow(2).mean().sqrt() / y.float().pow(2).mean().sqrt())
    fl = 2.0 * n * ho * ho * cout * cin * 9
    print(f"{name:26s} bf16 {t_bf:6.1f} us ({fl/t_bf/1e6:6.0f} TF/s) | fp8 conv {t_8:6.1f} us ({fl/t_8/1e6:6.0f} TF/s) + quantise {t_q:5.1f} us"
          f" | rms err vs bf16 {err*100:.2f} %", flush=True)
