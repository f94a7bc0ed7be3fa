"""Calibration (not part of the product): what the vendor GEMM (hipBLASLt via torch.matmul) sustains on the GEMM
shapes equivalent to the hot convs -- an upper reference for the hand-written kernels' MFMA efficiency."""
import torch
DEV = "cuda:0"
def bench(M, N, K, iters=30):
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=DEV, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("M=%d N=%d K=%d: %.1f us  %.0f TF/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
bench(65536, 256, 2304)      # res-block 3x3 conv
bench(262144, 128, 2304)     # dec0
bench(1048576, 64, 1152)     # dec1
bench(8192, 8192, 8192)
bench(4096, 4096, 4096)
bench(16384, 256, 4096)
