"""Diagnostic (not a test): are the halo conv's results bit-identical across weight-ring depths (options halo_stages) and
across repeated launches?  A race in the ring schedule shows as differing bits: a 3-deep ring (since removed) did exactly
that -- faster, and wrong in a fraction of the launches."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops, _lib

DEV = "cuda:0"
lib = _lib.load()
SHAPES = [("res", 256, 256, 64, 16, False), ("dec0", 256, 128, 128, 16, False), ("dec1", 128, 64, 256, 16, False),
          ("dec0up", 256, 128, 64, 16, True), ("res512", 512, 512, 64, 4, False), ("c64", 64, 64, 256, 8, False)]


def run(x, w, geom, gy):
    cache = ops.PackedWeights()
    xd = x.detach().requires_grad_(True)
    y = ops.conv2d(xd, w, None, cache, geom, "none")
    (dx,) = torch.autograd.grad(y, xd, gy)
    return y.detach().clone(), dx.detach().clone()


bad = 0
for name, cin, cout, H, N, up in SHAPES:
    torch.manual_seed(1)
    geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, up)
    x = torch.randn(N, H, H, cin, device=DEV).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    ho = H * 2 if up else H
    gy = torch.randn(N, ho, ho, cout, device=DEV).bfloat16()
    lib.dei2i_set_option(b"halo_bn", 0)
    lib.dei2i_set_option(b"halo_stages", 0)
    y4, d4 = run(x, w, geom, gy)
    for bn, stg in ((0, 0), (64, 4), (64, 6), (64, 8)):
        lib.dei2i_set_option(b"halo_bn", bn)
        lib.dei2i_set_option(b"halo_stages", stg)
        for rep in range(6):
            y, d = run(x, w, geom, gy)
            ny, nd = int((y.view(torch.int16) != y4.view(torch.int16)).sum()), int((d.view(torch.int16) != d4.view(torch.int16)).sum())
            if ny or nd:
                bad += 1
                print(f"{name} bn={bn} stages={stg} rep={rep}: {ny} fwd / {nd} dgrad elements differ from the shipped tile's "
                      f"first result (max |diff| {float((y.float()-y4.float()).abs().max()):.3e} / {float((d.float()-d4.float()).abs().max()):.3e})", flush=True)
    print(name, "done", flush=True)
lib.dei2i_set_option(b"halo_stages", 0)
lib.dei2i_set_option(b"halo_bn", 0)
print("differing runs:", bad)
