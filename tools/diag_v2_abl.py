"""Timing-only ablations of gather_gemm_v2 (option v2_ablate: 1 no LDS-DMA inside the loop, 2 no fragment reads / MFMAs, 3 neither; results
are wrong in those builds) on the stride-2 4x4 layers: which resource bounds the loop?"""
import ctypes
import sys
from ctypes import byref
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import _lib as L
from de_i2i_gan_amd import ops

DEV = "cuda:0"
lib = ops._lib_for(torch.zeros(1, device=DEV))
SHAPES = [("enc 64>128 @256 N16", 64, 128, 256, 16), ("enc 128>256 @128 N16", 128, 256, 128, 16), ("D1 64>128 @128 N64", 64, 128, 128, 64),
          ("D2 128>256 @64 N64", 128, 256, 64, 64), ("D3 256>512 @32 N64", 256, 512, 32, 64), ("D4 512>1024 @16 N64", 512, 1024, 16, 64),
          ("D5 1024>2048 @8 N64", 1024, 2048, 8, 64)]
PROF = 0    # PROF_GATHER_GEMM
for name, cin, cout, H, N in SHAPES:
    geom = ops.ConvGeom(cin, cout, 4, 2, 1, True, False)
    x = torch.randn(N, H, H, cin, device=DEV).bfloat16()
    w = torch.randn(cout, cin, 4, 4, device=DEV) * 0.02
    cache = ops.PackedWeights()
    flops = 2.0 * N * (H // 2) ** 2 * cout * cin * 16
    row = []
    for abl in (0, 1, 2, 3):
        lib.dei2i_set_option(b"v2_ablate", abl)
        for _ in range(2):
            ops.conv2d(x, w, None, cache, geom, "none")
        torch.cuda.synchronize()
        L.launch_counts(reset=True)
        lib.dei2i_prof_enable(PROF, 1)
        for _ in range(10):
            ops.conv2d(x, w, None, cache, geom, "none")
        torch.cuda.synchronize()
        n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        L.check(lib.dei2i_prof_collect(PROF, byref(n), byref(ms), byref(fl)), "prof_collect")
        lib.dei2i_prof_enable(PROF, 0)
        row.append(ms.value * 1e3 / max(n.value, 1))
        fam = {k: v for k, v in L.launch_counts(reset=True).items() if v}
    lib.dei2i_set_option(b"v2_ablate", 0)
    print("%-24s full %7.1f us (%4.0f TF/s) | no DMA %7.1f | no compute %7.1f | neither %7.1f   %s" % (
        name, row[0], flops / row[0] / 1e6, row[1], row[2], row[3], fam), flush=True)
