import sys, math, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from de_i2i_gan_amd import ops, _lib
from oracle import defectgan_oracle as O
lib = _lib.load()
DEV = "cuda:0"
def rel(a, b): return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()
for mode in ([int(v) for v in sys.argv[1:]] or [3, 1]):
    lib.dei2i_set_option(b"halo16", mode)
    for (cin, cout, H, W, N, refl, up, act) in [(64, 128, 64, 64, 32, True, False, "none"), (32, 64, 64, 64, 32, True, False, "leaky_relu"),
                                                (96, 136, 64, 64, 16, False, False, "relu"), (64, 64, 32, 32, 32, True, True, "none"),
                                                (256, 256, 64, 64, 16, True, False, "none")]:
        torch.manual_seed(1)
        x = torch.randn(N, cin, H, W).bfloat16().float()
        w = (torch.randn(cout, cin, 3, 3) * math.sqrt(2.0 / (cin * 9))).bfloat16().float()
        y_ref = O.conv2d(O.upsample2x(x) if up else x, w, stride=1, pad=1, mode="reflect" if refl else "zeros")
        if act == "relu": y_ref = torch.relu(y_ref)
        if act == "leaky_relu": y_ref = O.leaky_relu(y_ref)
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
        geom = ops.ConvGeom(cin, cout, 3, 1, 1, refl, up)
        _lib.launch_counts(reset=True)
        y = ops.conv2d(xg, w.to(DEV), None, ops.PackedWeights(), geom, act, stats=True)
        torch.cuda.synchronize()
        cnt = {k: v for k, v in _lib.launch_counts(reset=True).items() if v}
        got = ops.to_nchw(y, cout)
        st = getattr(y, "_dei2i_stats", None)
        serr = None
        if st is not None:
            s0 = st[0].double().sum(1)[:, 0]
            serr = rel(s0, y.float().sum((1, 2)))
        print(mode, (cin, cout, H, W, N, refl, up, act), cnt, "rel_l2 %.2e" % rel(got, y_ref), "pad0", float(y[..., cout:].abs().max()) if y.shape[-1] > cout else 0.0, "stats", serr)
