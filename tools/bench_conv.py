"""Micro-benchmark (not a test): TFLOP/s of the conv kernels on the hot shapes of the 256x256 batch-16 step."""
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops

DEV = "cuda:0"
# name, cin, cout, k, stride, pad, reflect, up, H, N
SHAPES = [
    ("res3x3_256@64", 256, 256, 3, 1, 1, True, False, 64, 16),
    ("dec0_256>128@128", 256, 128, 3, 1, 1, True, False, 128, 16),
    ("dec1_128>64@256", 128, 64, 3, 1, 1, True, False, 256, 16),
    ("enc4x4s2_64>128@256", 64, 128, 4, 2, 1, True, False, 256, 16),
    ("enc4x4s2_128>256@128", 128, 256, 4, 2, 1, True, False, 128, 16),
    ("stem7x7_3>64@256", 3, 64, 7, 1, 3, True, False, 256, 16),
    ("D5_1024>2048@8", 1024, 2048, 4, 2, 1, True, False, 8, 16),
    ("D3_256>512@32", 256, 512, 4, 2, 1, True, False, 32, 16),
    ("heads_64>4@256", 64, 4, 3, 1, 1, True, False, 256, 16),
    ("dec0up_256>128@64", 256, 128, 3, 1, 1, True, True, 64, 16),
    ("dec1up_128>64@128", 128, 64, 3, 1, 1, True, True, 128, 16),
    ("D0_3>64@256", 3, 64, 4, 2, 1, True, False, 256, 16),
    ("D1_64>128@128", 64, 128, 4, 2, 1, True, False, 128, 16),
    ("D2_128>256@64", 128, 256, 4, 2, 1, True, False, 64, 16),
    ("D4_512>1024@16", 512, 1024, 4, 2, 1, True, False, 16, 16),
]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    prec = ops.get_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for kv in sys.argv[3:]:                      # library options for same-box A/B runs: name=int
        from de_i2i_gan_amd import _lib
        name, val = kv.split("=") if "=" in kv else ("v2_ablate", kv)
        _lib.load().dei2i_set_option(name.encode(), int(val))
    for name, cin, cout, k, s, pad, refl, up, H, N in SHAPES:
        if only and only not in name:
            continue
        geom = ops.ConvGeom(cin, cout, k, s, pad, refl, up)
        x = torch.randn(N, H, H, prec.pad(cin), device=DEV).to(prec.dtype).requires_grad_(True)
        w = (torch.randn(cout, cin, k, k, device=DEV) * 0.05).requires_grad_(True)
        cache = ops.PackedWeights()
        y = ops.conv2d(x, w, None, cache, geom, "none")
        gy = torch.randn_like(y)
        ho = y.shape[1]
        flops = 2.0 * N * ho * ho * cout * cin * k * k
        t_f = timeit(lambda: ops.conv2d(x.detach(), w.detach(), None, cache, geom, "none"))
        xd = x.detach().requires_grad_(True)
        yd = ops.conv2d(xd, w.detach(), None, cache, geom, "none")
        t_d = timeit(lambda: torch.autograd.grad(yd, xd, gy, retain_graph=True))
        wd = w.detach().requires_grad_(True)
        yw = ops.conv2d(x.detach(), wd, None, cache, geom, "none")
        t_w = timeit(lambda: torch.autograd.grad(yw, wd, gy, retain_graph=True))
        print(f"{name:24s} fwd {t_f*1e3:8.1f} us {flops/t_f/1e9:7.1f} TF/s | dgrad(+fold) {t_d*1e3:8.1f} us {flops/t_d/1e9:7.1f} TF/s | "
              f"wgrad(+unpack) {t_w*1e3:8.1f} us {flops/t_w/1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
