"""Fused conv + norm + act (BASELINE.json configs[1]: "fused conv+InstanceNorm+act HIP kernels"):

  * statistics of a conv's output from the halo conv's epilogue (no separate moments pass);
  * BatchNorm apply + LeakyReLU (architecture.py:116-118) and SPADE's InstanceNorm * (1+gamma) + beta + ReLU (+ nearest x2
    upsample; normalization.py:24-37, architecture.py:241-245,343-350) on the NEXT conv's operand path, forward and in the
    weight gradient -- the normalised tensor is never written.

Checked (a) against the oracle's functions in float64 on bf16-rounded operands, and (b) against this build's own unfused
formulation (ops.fuse_norm = False) on the same inputs, where everything except the position of one bf16 rounding is
the same arithmetic."""
import math

import pytest
import torch

from helpers import make_opt
from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def ops():
    from de_i2i_gan_amd import ops as _ops
    keep = (_ops.fuse_norm, _ops.fuse_pro, _ops.fuse_ring)
    _ops.fuse_norm, _ops.fuse_pro, _ops.fuse_ring = True, True, True     # the operand-path fusion is off by default (ops.py: measured slower)
    yield _ops
    _ops.fuse_norm, _ops.fuse_pro, _ops.fuse_ring = keep


def _counts(reset=True, splitk=False):
    """launches per MFMA kernel family since the last call (``splitk``: keep the split-K finalize launches too)"""
    from de_i2i_gan_amd import _lib
    return {k: v for k, v in _lib.launch_counts(reset=reset).items() if v and (splitk or k != "splitk_finalize")}


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def nhwc(t, cs=None):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs or c, dtype=t.dtype)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out


@pytest.mark.parametrize("cin,cout,H,W,N,up,act", [(128, 128, 64, 64, 8, False, "none"), (64, 64, 64, 128, 8, False, "leaky_relu"),
                                                    (64, 136, 32, 64, 16, False, "none"), (128, 64, 32, 32, 16, True, "none"),
                                                    # 16 x 32 tile kernel (conv_halo16.hip): two 8 x 32 records per tile
                                                    (64, 128, 64, 64, 32, False, "none"), (128, 64, 64, 64, 16, True, "leaky_relu")])
def test_conv_epilogue_statistics_equal_the_stored_outputs_moments(ops, cin, cout, H, W, N, up, act):
    torch.manual_seed(3)
    x = torch.randn(N, H, W, cin, device=DEV).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=DEV) * math.sqrt(2.0 / (cin * 9))
    geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, up)
    _counts()
    y = ops.conv2d(x, w, None, ops.PackedWeights(), geom, act, stats=True)
    fam = _counts()
    big = N * ((H << up) // 16) * ((W << up) // 32) * max(1, (cout + 127) // 128 if cout >= 128 else 1) >= 224
    assert fam == ({"halo16_conv": 1} if big else {"halo_conv": 1}), fam
    partial, chunks = y._dei2i_stats
    assert partial.shape == (N, chunks, 2, y.shape[-1])
    yf = y.float()
    s0, s1 = yf.sum(dim=(1, 2)), (yf * yf).sum(dim=(1, 2))
    got = partial.double().sum(dim=1)
    assert maxrel(got[:, 0], s0) < 1e-4 and maxrel(got[:, 1], s1) < 1e-5
    ops.fuse_norm = False                      # and the output itself is the plain kernel's
    y0 = ops.conv2d(x, w, None, ops.PackedWeights(), geom, act, stats=True)
    assert not hasattr(y0, "_dei2i_stats") and torch.equal(y0, y)


@pytest.mark.parametrize("C,cout,H,W,N,training", [(128, 128, 64, 64, 8, True), (128, 64, 64, 64, 16, True), (256, 128, 32, 64, 16, False)])
def test_bn_act_conv_fused(ops, C, cout, H, W, N, training):
    """conv(LeakyReLU(BatchNorm(y1))): fused against the oracle (float64, rounded operands) and against the unfused kernels."""
    prec = ops.BF16
    torch.manual_seed(5)
    y1 = (torch.randn(N, C, H, W) * 1.5 + 0.3).bfloat16().float()
    w = (torch.randn(cout, C, 3, 3) * math.sqrt(2.0 / (C * 9))).bfloat16().float()
    bn_w, bn_b = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    rm0, rv0 = 0.1 * torch.randn(C), 1 + 0.1 * torch.rand(C)
    gy = torch.randn(N, cout, H, W).bfloat16().float()
    S = {"bn.weight": bn_w.double().requires_grad_(True), "bn.bias": bn_b.double().requires_grad_(True),
         "bn.running_mean": rm0.double().clone(), "bn.running_var": rv0.double().clone(),
         "bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    yr, wr = y1.double().requires_grad_(True), w.double().requires_grad_(True)
    h_ref = O.leaky_relu(O.batchnorm(S, "bn", yr, training))
    out_ref = O.conv2d(h_ref, wr, stride=1, pad=1, mode="reflect")
    g_ref = torch.autograd.grad(out_ref, [yr, wr, S["bn.weight"], S["bn.bias"]], gy.double())

    geom = ops.ConvGeom(C, cout, 3, 1, 1, True, False)
    res = {}
    for fused in (True, False):
        ops.fuse_norm = fused
        yg = nhwc(y1).to(DEV).bfloat16().requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        bw, bb = bn_w.to(DEV).requires_grad_(True), bn_b.to(DEV).requires_grad_(True)
        rm, rv = rm0.to(DEV), rv0.to(DEV)
        cache = ops.PackedWeights()
        _counts()
        if fused:
            assert ops.bn_act_conv_supported(yg, bw, wg, geom, True)
            out = ops.bn_act_conv(yg, bw, bb, rm, rv, training, "leaky_relu", wg, cache, geom, stats=True)
            assert hasattr(out, "_dei2i_stats")
        else:
            out = ops.conv2d(ops.batchnorm_act(yg, bw, bb, rm, rv, training, "leaky_relu"), wg, None, cache, geom)
        c_fwd = _counts()
        out.backward(nhwc(gy).to(DEV).bfloat16())
        c_bwd = _counts()
        res[fused] = (out.detach(), yg.grad, wg.grad, bw.grad, bb.grad, rm, rv, c_fwd, c_bwd)
    f, u = res[True], res[False]
    assert f[7] == {"halo_conv": 1} and u[7] == {"halo_conv": 1}
    assert f[8].get("wgrad_halo") == 1 and f[8].get("halo_conv") == 1
    # against the oracle: bf16 tolerances of the op tests (the intermediate h is rounded to bf16 once, like in the unfused path)
    assert maxrel(ops.to_nchw(f[0], cout), out_ref) < 1.5e-2 and rel_l2(ops.to_nchw(f[0], cout), out_ref) < 4e-3
    assert rel_l2(ops.to_nchw(f[1], C), g_ref[0]) < 1e-2                   # LeakyReLU kinks: relative L2
    assert rel_l2(f[2], g_ref[1]) < 5e-3 and rel_l2(f[3], g_ref[2]) < 1e-2 and rel_l2(f[4], g_ref[3]) < 1e-2
    if training:
        assert maxrel(f[5], S["bn.running_mean"]) < 5e-3 and maxrel(f[6], S["bn.running_var"]) < 5e-3
    # against the unfused kernels: the same arithmetic on the same rounded intermediate -> (near) bit-identical
    for a, b, nm in zip(f[:7], u[:7], ("y", "dy1", "dw", "dbn_w", "dbn_b", "running_mean", "running_var")):
        assert maxrel(a, b) < 1e-5, nm


@pytest.mark.parametrize("C,cout,hs,ws,N,up,skip,mode", [(128, 128, 64, 64, 8, False, True, "pro"), (256, 128, 32, 32, 16, True, False, "pro"),
                                                          (128, 64, 64, 64, 8, True, False, "pro"), (64, 128, 64, 64, 16, False, False, "pro"),
                                                          (128, 128, 16, 32, 64, False, False, "pro"),
                                                          # "ring": z at the source resolution + the logical frame's ring tensor (the
                                                          # default path of the upsampling decoder blocks, 16 x 32 tile conv kernel)
                                                          (256, 128, 64, 64, 16, True, False, "ring"), (128, 64, 64, 64, 16, True, False, "ring"),
                                                          (64, 128, 32, 64, 32, True, False, "ring")])
def test_spade_conv_fused(ops, C, cout, hs, ws, N, up, skip, mode):
    """conv(ReLU(SPADE(up(x)))) on a constant label map through the product modules: fused against the oracle (float64,
    rounded operands) and against the unfused kernels (same inputs)."""
    from de_i2i_gan_amd.networks.architecture import SPADE, Conv2d
    prec = ops.BF16
    torch.manual_seed(7)
    label_nc, hidden = 6, 32
    x = (torch.randn(N, C, hs, ws) * 1.3 + 0.2).bfloat16().float()
    seg = torch.zeros(N, label_nc, 1, 1)
    for i in range(N):
        seg[i, 1 + i % 5] = 1
    mod = SPADE(label_nc, C, hidden_nc=hidden, kernel_size=(3, 3), padding="same")
    conv = Conv2d(C, cout, 3, padding="same", padding_mode="reflect", bias=False)
    S = {}
    with torch.no_grad():
        for k_, p in mod.named_parameters():
            p.copy_(O.formula_tensor("spade." + k_, tuple(p.shape)) * (3.0 if p.dim() == 4 else 1.0))
            S["sp." + k_] = (p.detach().clone().bfloat16().float() if p.dim() == 4 else p.detach().clone()).double().requires_grad_(True)
        conv.weight.copy_((torch.randn(cout, C, 3, 3) * math.sqrt(2.0 / (C * 9))).bfloat16().float())
    mod, conv = mod.to(DEV), conv.to(DEV)
    h, w = (2 * hs, 2 * ws) if up else (hs, ws)
    gy = torch.randn(N, cout, h, w).bfloat16().float()
    gs = torch.randn(N, C, hs, ws).bfloat16().float()
    # ---- oracle ----
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    z_ref = torch.relu(O.spade(S, "sp", O.upsample2x(xr) if up else xr, seg.double()))
    out_ref = O.conv2d(z_ref, wr, stride=1, pad=1, mode="reflect")
    loss = (out_ref * gy.double()).sum() + ((xr * gs.double()).sum() if skip else 0.0)
    keys = list(S)
    g_ref = torch.autograd.grad(loss, [xr, wr] + [S[k_] for k_ in keys])
    # ---- product: fused and unfused ----
    res = {}
    for fused in (True, False):
        ops.fuse_norm = fused
        ops.fuse_pro = mode == "pro"
        for p in list(mod.parameters()) + list(conv.parameters()):
            p.grad = None
        mod._gb_cache.clear()
        xg = nhwc(x).to(DEV).bfloat16().requires_grad_(True)
        _counts()
        if skip:
            out, xs = mod.fused_conv(xg, seg.to(DEV), conv, up=up, skip=True)
            assert xs.data_ptr() == xg.data_ptr()
        else:
            out, xs = mod.fused_conv(xg, seg.to(DEV), conv, up=up), None
        c_fwd = _counts()
        if skip:
            torch.autograd.backward([out, xs], [nhwc(gy).to(DEV).bfloat16(), nhwc(gs).to(DEV).bfloat16()])
        else:
            out.backward(nhwc(gy).to(DEV).bfloat16())
        res[fused] = (out.detach(), xg.grad, conv.weight.grad.clone(), {k_: p.grad.clone() for k_, p in mod.named_parameters()}, c_fwd)
    f, u = res[True], res[False]
    assert f[4].get("halo_conv" if mode == "pro" else "halo16_conv") == 1, f[4]
    m = {"y_vs_oracle": (maxrel(ops.to_nchw(f[0], cout), out_ref), rel_l2(ops.to_nchw(f[0], cout), out_ref)),
         "dx_vs_oracle": rel_l2(ops.to_nchw(f[1], C), g_ref[0]), "dx_unfused_vs_oracle": rel_l2(ops.to_nchw(u[1], C), g_ref[0]),
         "dw_vs_oracle": rel_l2(f[2], g_ref[1]), "dw_unfused_vs_oracle": rel_l2(u[2], g_ref[1]),
         "params_vs_oracle": {k_: rel_l2(f[3][k_[3:]], gref) for k_, gref in zip(keys, g_ref[2:])},
         "y_vs_unfused": rel_l2(f[0], u[0]), "dx_vs_unfused": rel_l2(f[1], u[1]), "dw_vs_unfused": rel_l2(f[2], u[2]),
         "params_vs_unfused": {k_: rel_l2(f[3][k_], u[3][k_]) for k_ in f[3]}}
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/spade_conv_{mode}_{C}_{cout}_{hs}_{int(up)}.json", "w") as fh:
        json.dump(m, fh, indent=1)
    tol = 1.5e-2 * 3                               # the op tests' SPADE bound: gamma / beta / actv are bf16 intermediates too
    assert m["y_vs_oracle"][0] < tol and m["y_vs_oracle"][1] < 1.5e-2, m
    # InstanceNorm's backward subtracts means: the bf16 rounding of dz is amplified by the cancellation (the unfused kernels
    # sit at the same distance from the float64 reference -- recorded above -- and 1e-2 from the fused ones)
    assert m["dx_vs_oracle"] < 5e-2 and m["dx_vs_oracle"] < 1.3 * m["dx_unfused_vs_oracle"] + 2e-3, m
    assert m["dw_vs_oracle"] < 1.5e-2, m
    for k_, v in m["params_vs_oracle"].items():
        # the label path's intermediates (actv, the gamma | beta table and its gradient) are bf16 tensors: a few ReLU masks of
        # actv flip against the float64 reference (measured up to 9.4e-2 on mlp_shared.bias, the end of that chain, identically for the unfused kernels)
        assert v < 0.12, (k_, m)
    # against the unfused kernels: identical up to the position of one rounding (A*x + B vs ((x-m)*r)*(1+gamma) + beta)
    assert m["y_vs_unfused"] < 2e-3 and m["dx_vs_unfused"] < 1e-2 and m["dw_vs_unfused"] < 2e-3, m
    for k_, v in m["params_vs_unfused"].items():
        assert v < 2e-2, (k_, m)


def test_generator_and_both_loss_graphs_fused_equal_unfused_at_256_batch_4(ops):
    """The whole model at 256x256 (default widths), batch 4 -- the smallest batch at which the 64x64 res-block convs reach
    the halo-resident kernels: fused and unfused builds of the same step agree on the losses to 2e-4 and on every
    gradient to cosine 0.999 (one bf16 rounding moves in each fused SPADE)."""
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=256, batch=4, num_layers=5, ngf=64, ndf=64, hidden_nc=128)
    bg, labels, df = O.synthetic_batch(4, 256)
    res = {}
    for fused in (True, False):
        ops.fuse_norm = fused
        torch.manual_seed(123)
        tr = DefectGanTrainer(make_opt(c, DEV, "bf16"))
        G, D = tr.model.netG, tr.model.netD
        _counts()
        g1, c1 = tr.model("discriminator", bg, labels, df)
        (g1 + 2 * c1).backward()
        dgr = torch.cat([p.grad.double().flatten() for p in D.parameters()])
        for p in D.parameters():
            p.grad = None
        ls = tr.model("generator", bg, labels, df)
        (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
        ggr = {k: p.grad.double().flatten().cpu() for k, p in G.named_parameters() if p.grad is not None}
        bufs = {k: v.double().cpu() for k, v in G.state_dict().items() if "running" in k}
        res[fused] = ([float(g1.detach()), float(c1.detach())] + [float(v.detach()) for v in ls], dgr.cpu(), ggr, bufs)
        del tr
    f, u = res[True], res[False]
    full_f = torch.cat([f[2][k] for k in u[2]])
    full_u = torch.cat([u[2][k] for k in u[2]])
    cosv = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))     # noqa: E731
    per = {k: cosv(f[2][k], u[2][k]) for k in u[2] if float(u[2][k].norm()) > 1e-3 * float(full_u.norm())}
    m = {"losses_fused": f[0], "losses_unfused": u[0], "D_cos": cosv(f[1], u[1]), "G_cos": cosv(full_f, full_u),
         "G_worst_param_cos": min(per.values()), "running_stats_maxrel": max(maxrel(f[3][k], u[3][k]) for k in u[3])}
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fused_vs_unfused_256x4.json", "w") as fh:
        json.dump(m, fh, indent=1)
    # one bf16 rounding moves in each fused SPADE (A*x + B instead of ((x - m) * r) * (1 + gamma) + beta), i.e. a 2^-9 perturbation
    # of a few activations per layer; the gradients then differ like bf16 differs from f32 (tests/test_hot_shapes_gpu.py:
    # ReLU masks within rounding of 0 flip), the losses and forward statistics by 1e-4
    for a, b in zip(f[0], u[0]):
        assert abs(a - b) <= 5e-4 * max(abs(b), 1e-3), m
    assert m["D_cos"] > 0.9995 and m["G_cos"] > 0.999 and m["G_worst_param_cos"] > 0.95, m
    assert m["running_stats_maxrel"] < 2e-3, m


@pytest.mark.gpu
def test_weight_gradients_on_the_side_stream_are_bit_identical_and_joined():
    """ops.wgrad_side_stream: leaf-weight gradients are computed on a second stream and joined by an engine callback at the end
    of the backward pass -- reading ``param.grad`` on the current stream right after ``backward()`` must see the finished
    gradients, and they must be the bits of the single-stream run."""
    import numpy as np
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    from helpers import make_opt
    from oracle import defectgan_oracle as O
    c = dict(image_size=128, batch=4, num_layers=4, ngf=32, ndf=32, hidden_nc=64)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    grads = {}
    for side in (True, False):
        ops.wgrad_side_stream = side
        try:
            torch.manual_seed(11)
            tr = DefectGanTrainer(make_opt(c, "cuda:0", "bf16"))
            ls = tr.model("generator", bg, labels, df)
            (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
            # no torch.cuda.synchronize() here: the read below is ordered behind the side stream by the join alone
            grads[side] = [p.grad.clone() for p in tr.model.netG.parameters() if p.grad is not None]
            assert (ops.wgrad_stream("cuda:0") is not None) or not side
        finally:
            ops.wgrad_side_stream = True
    assert len(grads[True]) == len(grads[False]) > 50
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_side_stream_scratch_regrowth_at_default_widths_is_bit_identical():
    """The side stream's split-K scratch is sized per layer (96 MB for the small weights, ~151 MB for a 256x256x3x3 one, 512 MB for
    D's enc_blk.5), so at the DEFAULT widths (ngf = ndf = 64) it re-grows inside the first backward pass while earlier side-stream
    kernels may still be using the buffer it replaces.  It is allocated on the side stream (ops._conv_wgrad), so the dropped block can
    only be handed to later side-stream allocations.  First D and G backward of a trainer, side stream against single stream, with
    the workspaces dropped in between (cold path both times): every gradient bit for bit."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=128, batch=4, num_layers=5, ngf=64, ndf=64, hidden_nc=128)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    grads = {}
    for side in (True, False):
        ops.wgrad_side_stream = side
        for k_ in [k_ for k_ in ops._workspaces if k_[1] in ("wgrad_side", "wgrad")]:
            del ops._workspaces[k_]
        torch.cuda.empty_cache()
        try:
            torch.manual_seed(11)
            tr = DefectGanTrainer(make_opt(c, "cuda:0", "bf16"))
            ld = tr.model("discriminator", bg, labels, df)
            (ld[0] + 2 * ld[1]).backward()
            ls = tr.model("generator", bg, labels, df)
            (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
            grads[side] = [p.grad.clone() for net in (tr.model.netD, tr.model.netG) for p in net.parameters() if p.grad is not None]
            if side:      # the regrow path was actually taken: the slot ended larger than its first size
                ws = [v for k_, v in ops._workspaces.items() if k_[1] == "wgrad_side"]      # (keyed by device, slot and side stream)
                assert len(ws) == 1 and ws[0].numel() * 4 > (96 << 20), [w_.numel() for w_ in ws]
        finally:
            ops.wgrad_side_stream = True
    assert len(grads[True]) == len(grads[False]) > 58
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_side_stream_is_not_taken_when_grad_is_already_defined():
    """Two backward() calls without zero_grad (gradient accumulation): with ``param.grad`` defined AccumulateGrad launches
    ``grad += dw`` on the main stream in the MIDDLE of the pass, before the end-of-pass join, so those weight gradients must be
    computed on the main stream (ops._conv_wgrad takes the side stream only when ``weight.grad is None``).  Against
    wgrad_side_stream = False, bit for bit."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.architecture import Conv2d
    res = {}
    for side in (True, False):
        ops.wgrad_side_stream = side
        try:
            torch.manual_seed(5)
            conv = Conv2d(256, 256, 3, padding="same", padding_mode="reflect", bias=False).to(DEV)
            x = torch.randn(8, 64, 64, 256, device=DEV).bfloat16()
            g1 = torch.randn(8, 64, 64, 256, device=DEV).bfloat16()
            g2 = torch.randn(8, 64, 64, 256, device=DEV).bfloat16()
            conv(x).backward(g1)
            first = conv.weight.grad
            conv(x).backward(g2)                 # .grad defined: accumulated in place by AccumulateGrad
            assert conv.weight.grad is first
            res[side] = conv.weight.grad.clone()
        finally:
            ops.wgrad_side_stream = True
    assert torch.isfinite(res[True]).all() and torch.equal(res[True], res[False])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,C,cout,hs,ws,N,up,skip", [("spade", 256, 256, 64, 64, 16, False, True), ("spade", 128, 64, 64, 64, 8, True, False),
                                                         ("spade", 64, 64, 128, 128, 16, False, False), ("spade", 160, 128, 64, 64, 16, False, False),
                                                         ("bn", 256, 256, 64, 64, 16, False, False), ("bn", 64, 128, 128, 128, 16, False, False),
                                                         ("bn_eval", 256, 128, 64, 64, 16, False, False)])
def test_norm_backward_reductions_from_the_dgrad_epilogue(kind, C, cout, hs, ws, N, up, skip):
    """SURVEY.md Appendix B "their backward reductions from the dgrad epilogue": the per-channel sums of the SPADE / BatchNorm
    backward (normalization.py:24-37, architecture.py:116-118) taken by the input-gradient launch of the conv behind the
    layer (ops.fuse_bwd, conv_halo16.hip EPIN) against the streaming pass over dz and x (reduce.hip) on the same inputs:
    the same per-element arithmetic on the same bf16 dz, only the order of the fp32 additions differs.  The oracle parity
    of the default (epilogue) path is test_spade_conv_fused / test_bn_act_conv_fused / the model goldens."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.architecture import SPADE, Conv2d
    torch.manual_seed(11)
    x = (torch.randn(N, C, hs, ws) * 1.3 + 0.2).bfloat16()
    h, w = (2 * hs, 2 * ws) if up else (hs, ws)
    gy = nhwc(torch.randn(N, cout, h, w).bfloat16()).to(DEV)
    gs = nhwc(torch.randn(N, C, hs, ws).bfloat16()).to(DEV)
    conv = Conv2d(C, cout, 3, padding="same", padding_mode="reflect", bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, C, 3, 3) * math.sqrt(2.0 / (C * 9)))
    conv = conv.to(DEV)
    keep = (ops.fuse_bwd, ops.fuse_pro, ops.fuse_ring, ops.fuse_norm)
    res = {}
    try:
        ops.fuse_pro, ops.fuse_ring, ops.fuse_norm = False, True, True
        if kind == "spade":
            seg = torch.zeros(N, 6, 1, 1)
            for i in range(N):
                seg[i, 1 + i % 5] = 1
            mod = SPADE(6, C, hidden_nc=32, kernel_size=(3, 3), padding="same")
            with torch.no_grad():
                for k_, p in mod.named_parameters():
                    p.copy_(O.formula_tensor("spade." + k_, tuple(p.shape)) * (3.0 if p.dim() == 4 else 1.0))
            mod = mod.to(DEV)
        else:
            bn_w = (torch.rand(C) + 0.5).to(DEV).requires_grad_(True)
            bn_b = (torch.randn(C) * 0.1).to(DEV).requires_grad_(True)
        for on in (True, False):
            ops.fuse_bwd = on
            conv.weight.grad = None
            before = dict(ops.bwd_fused_counts)
            xg = nhwc(x).to(DEV).requires_grad_(True)
            if kind == "spade":
                for p in mod.parameters():
                    p.grad = None
                mod._gb_cache.clear()
                if skip:
                    out, xs = mod.fused_conv(xg, seg.to(DEV), conv, up=up, skip=True)
                    torch.autograd.backward([out, xs], [gy, gs])
                else:
                    mod.fused_conv(xg, seg.to(DEV), conv, up=up).backward(gy)
                params = {k_: p.grad.clone() for k_, p in mod.named_parameters()}
            else:
                bn_w.grad = bn_b.grad = None
                rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
                z = ops.batchnorm_act(xg, bn_w, bn_b, rm, rv, kind == "bn", "leaky_relu")
                conv(z).backward(gy)
                params = {"weight": bn_w.grad.clone(), "bias": bn_b.grad.clone()}
            torch.cuda.synchronize()
            took = {k_: ops.bwd_fused_counts[k_] - before[k_] for k_ in before}
            assert took == ({"epilogue": 1, "taken": 1} if on else {"epilogue": 0, "taken": 0}), (on, took)
            res[on] = (xg.grad.clone(), conv.weight.grad.clone(), params)
    finally:
        ops.fuse_bwd, ops.fuse_pro, ops.fuse_ring, ops.fuse_norm = keep
    e, s = res[True], res[False]
    assert torch.isfinite(e[0]).all()
    assert rel_l2(e[0], s[0]) < 2e-4 and maxrel(e[0], s[0]) < 1e-2, (rel_l2(e[0], s[0]), maxrel(e[0], s[0]))   # bf16 dx: an ulp here and there
    assert torch.equal(e[1], s[1])                                  # the weight gradient does not see the change at all
    for k_ in e[2]:
        assert rel_l2(e[2][k_], s[2][k_]) < 2e-4, (k_, rel_l2(e[2][k_], s[2][k_]))


@pytest.mark.gpu
@pytest.mark.parametrize("conv_last", [True, False])
def test_norm_backward_reductions_are_not_used_when_dz_has_a_second_consumer(conv_last):
    """The ResBlock case (architecture.py:139-156: x feeds the block's first conv AND its identity add): the BatchNorm output has two
    consumers, so dL/dz is the conv's input gradient PLUS the identity branch's gradient, summed by autograd's input buffer -- in
    place into whichever arrives first when nobody else holds that tensor.  The dgrad epilogue only saw its own share: its records
    must not be used (ops._NormBwdHint keeps the dgrad's tensor referenced until the norm's backward has looked at it, so the sum is
    a new tensor, and compares storage + version).  Both arrival orders; against the streaming pass bit for bit."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.architecture import Conv2d
    torch.manual_seed(5)
    N, C, H = 16, 256, 64
    x = nhwc((torch.randn(N, C, H, H) * 1.3 + 0.2).bfloat16()).to(DEV)
    gy = nhwc(torch.randn(N, C, H, H).bfloat16()).to(DEV)
    gs = nhwc(torch.randn(N, C, H, H).bfloat16()).to(DEV)
    conv = Conv2d(C, C, 3, padding="same", padding_mode="reflect", bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, C, 3, 3) * math.sqrt(2.0 / (C * 9)))
    conv = conv.to(DEV)
    bn_w = (torch.rand(C) + 0.5).to(DEV).requires_grad_(True)
    bn_b = (torch.randn(C) * 0.1).to(DEV).requires_grad_(True)
    keep = (ops.fuse_bwd, ops.fuse_pro)
    res = {}
    try:
        ops.fuse_pro = False
        for on in (True, False):
            ops.fuse_bwd = on
            conv.weight.grad = bn_w.grad = bn_b.grad = None
            before = dict(ops.bwd_fused_counts)
            xg = x.clone().requires_grad_(True)
            z = ops.batchnorm_act(xg, bn_w, bn_b, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, "leaky_relu")
            if conv_last:                                # the later node runs first in backward: the conv's gradient arrives first
                side = ops.add(z, z)
                y = conv(z)
            else:
                y = conv(z)
                side = ops.add(z, z)
            torch.autograd.backward([y, side], [gy, gs])
            torch.cuda.synchronize()
            took = {k_: ops.bwd_fused_counts[k_] - before[k_] for k_ in before}
            assert took["taken"] == 0, (on, took)        # the records of a shared dz are never used
            res[on] = (xg.grad.clone(), bn_w.grad.clone(), bn_b.grad.clone())
    finally:
        ops.fuse_bwd, ops.fuse_pro = keep
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_eval_mode_batchnorm_folded_into_the_conv_weights():
    """ops.fold_eval_bn: in the no-grad eval-mode generator pass (the D step's two passes, defectgan_model.py:251-262; inference) every
    conv -> BatchNorm(running statistics) -> LeakyReLU block is ONE conv launch on weights scaled per output channel (architecture.py:
    116-118 is then a fixed affine).  (a) the folded weights / bias against the formula; (b) the generator's eval output with the
    fold against the three-kernel formulation: exact-f32 mode 1e-5 (fp32 rounding of w * a), bf16 within the bf16 tolerances of the
    model tests -- and against the oracle's eval forward in f32 mode (1e-3)."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.generator import DefectGanGenerator
    from helpers import formula_fill
    torch.manual_seed(2)
    w = torch.randn(48, 24, 3, 3, device=DEV)
    bw, bb = torch.rand(48, device=DEV) + 0.5, torch.randn(48, device=DEV)
    rm, rv = torch.randn(48, device=DEV), torch.rand(48, device=DEV) + 0.1
    w_eff, b_eff = ops.fold_bn_weight(w, bw, bb, rm, rv, 1e-5)
    a = bw * torch.rsqrt(rv + 1e-5)
    assert maxrel(w_eff, w * a.view(-1, 1, 1, 1)) < 1e-6 and maxrel(b_eff, bb - rm * a) < 1e-6
    c = dict(image_size=64, batch=2, num_layers=4, ngf=16, ndf=16, hidden_nc=32)
    cfg = O.Cfg(image_size=64, ngf=16, ndf=16, num_layers=4, hidden_nc=32)
    bg, labels, _ = O.synthetic_batch(2, 64)
    SG = O.make_state(O.generator_state_shapes(cfg))
    with torch.no_grad():
        ref, ref_p = O.generator_forward(SG, bg, labels.reshape(2, 6, 1, 1), cfg, training=False)
    keep = ops.fold_eval_bn
    try:
        for pname in ("f32", "bf16"):
            G = DefectGanGenerator(make_opt(c, DEV, pname)).to(DEV)
            formula_fill(G)
            G.eval()
            outs = {}
            for on in (True, False):
                ops.fold_eval_bn = on
                with torch.no_grad():
                    outs[on] = G(bg.to(DEV), labels.to(DEV))
            for a_, b_ in zip(outs[True], outs[False]):
                if pname == "f32":
                    assert maxrel(a_, b_) < 1e-5
                else:
                    assert rel_l2(a_, b_) < 0.12
            if pname == "f32":
                assert maxrel(outs[True][0], ref) < 1e-3 and maxrel(outs[True][1], ref_p) < 1e-3
            # with autograd on (an eval-mode pass that is differentiated) nothing is folded: same numbers as the unfolded pass
            ops.fold_eval_bn = True
            out_g = G(bg.to(DEV), labels.to(DEV))
            assert torch.equal(out_g[0].detach(), outs[False][0])
    finally:
        ops.fold_eval_bn = keep


@pytest.mark.gpu
@pytest.mark.parametrize("pname", ["f32", "bf16"])
def test_generator_chains_on_two_streams_equal_the_one_stream_pass(pname):
    """ops.forked_chains: the G loss's passes bg -> fake_defects -> recover_normals and df -> fake_normals -> recover_defects
    (defectgan_model.py:185-190) on two streams, joined before D.  Same kernels on the same operands: the five losses are the
    one-stream pass's BITS; BatchNorm's running statistics after the four ordered updates (replayed after the join,
    ops.bn_running_deferred) and the counter agree to fp32 rounding of the update formula; the parameter gradients differ only by
    the order in which the two chains' contributions are summed (in-place within a chain, one autograd add across chains)."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=128, batch=4, num_layers=4, ngf=32, ndf=32, hidden_nc=64)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    res = {}
    keep, keep_paired = ops.forked_chains, ops.paired_passes
    ops.paired_passes = False
    try:
        for fork in (True, False):
            ops.forked_chains = fork
            torch.manual_seed(11)
            tr = DefectGanTrainer(make_opt(c, "cuda:0", pname))
            G = tr.model.netG
            assert tr.model._forks_generator_chains(bg.to(DEV), None) == fork
            for it in range(2):                      # the second pass reuses the chains' streams, workspaces and packed weights
                for p in G.parameters():
                    p.grad = None
                ls = tr.model("generator", bg, labels, df)
                (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
            torch.cuda.synchronize()
            res[fork] = ([float(v) for v in ls], {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in G.state_dict().items() if "running_" in k or "num_batches" in k})
    finally:
        ops.forked_chains, ops.paired_passes = keep, keep_paired
    (lf, gf, bf), (l1, g1, b1) = res[True], res[False]
    assert lf == l1, (lf, l1)
    assert gf.keys() == g1.keys() and len(gf) > 50
    for k in gf:
        assert rel_l2(gf[k], g1[k]) < (1e-5 if pname == "f32" else 2e-5) or float(g1[k].norm()) < 1e-6 * max(float(v.norm()) for v in g1.values()), k
    for k in bf:
        if "num_batches" in k:
            assert int(bf[k]) == int(b1[k]) == 8
        else:
            assert maxrel(bf[k], b1[k]) < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("pname", ["f32", "bf16"])
def test_paired_generator_passes_equal_the_four_passes(pname):
    """ops.paired_passes: the G loss's four generator passes (defectgan_model.py:185-190) as one pass over [bg | df] and one over
    [fake_defects | fake_normals], training-mode BatchNorm taking its statistics per half of the batch (ops.bn_batch_groups) and
    its four running-statistics updates replayed in the reference's order.  Against the four-pass form on one stream: the same
    function.  Losses, running statistics and counters agree to rounding.  The parameter gradients of this 4-pass chain are
    ill-conditioned (ReLU / |a - b| kinks; remainders of cancelling sums): the four-pass form itself, fed inputs one fp32 ulp
    away, moves them by up to 3e-3 of their norm in f32 and 20% in bf16 (tools/diag_pair.py) -- that run is the yardstick: per
    parameter and for the whole gradient, paired - four must stay within a small multiple of four' - four.  (Exactness of the
    paired form against the fp64 oracle on replayed kinks: test_model_gpu.py::test_step_gradients_match_oracle_fp64.)"""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=128, batch=4, num_layers=4, ngf=32, ndf=32, hidden_nc=64)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    res = {}
    keep, keep_fork = ops.paired_passes, ops.forked_chains
    ops.forked_chains = False
    try:
        for tag, paired, scale in (("paired", True, 1.0), ("four", False, 1.0), ("four'", False, 1.0 + 2.0 ** -22)):
            ops.paired_passes = paired
            torch.manual_seed(11)
            tr = DefectGanTrainer(make_opt(c, "cuda:0", pname))
            G = tr.model.netG
            assert tr.model._pairs_generator_passes(bg.to(DEV), df.to(DEV), None) == paired
            ops.bwd_fused_counts["taken"] = 0
            for it in range(2):
                for p in G.parameters():
                    p.grad = None
                ls = tr.model("generator", bg * scale, labels, df * scale)
                (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
            torch.cuda.synchronize()
            res[tag] = ([float(v.detach()) for v in ls], {k: p.grad.double() for k, p in G.named_parameters() if p.grad is not None},
                        {k: v.clone() for k, v in G.state_dict().items() if "running_" in k or "num_batches" in k},
                        ops.bwd_fused_counts["taken"])
    finally:
        ops.paired_passes, ops.forked_chains = keep, keep_fork
    (lp, gp, bp, tp), (l1, g1, b1, t1), (ln, gn, _, _) = res["paired"], res["four"], res["four'"]
    tol_l, tol_b = (1e-5, 1e-5) if pname == "f32" else (1e-3, 1e-2)
    print("paired vs four passes:", lp, l1, "epilogue reductions taken:", tp, t1)
    for a_, b_ in zip(lp, l1):
        assert abs(a_ - b_) <= tol_l * max(abs(b_), 1e-3), (lp, l1)
    assert gp.keys() == g1.keys() and len(gp) > 50
    gmax = max(float(v.norm()) for v in g1.values())
    sq = lambda d: sum(float(v.norm()) ** 2 for v in d.values()) ** 0.5
    whole_p = sq({k: gp[k] - g1[k] for k in g1}) / sq(g1)
    whole_n = sq({k: gn[k] - g1[k] for k in g1}) / sq(g1)
    print("whole gradient: paired - four %.3e, four' - four %.3e" % (whole_p, whole_n))
    assert whole_p < 3 * whole_n + 1e-6
    for k in g1:
        ep, en = float((gp[k] - g1[k]).norm()), float((gn[k] - g1[k]).norm())
        assert ep <= 5 * en + 2e-5 * gmax, (k, ep, en, float(g1[k].norm()), gmax)
    for k in bp:
        if "num_batches" in k:
            assert int(bp[k]) == int(b1[k]) == 8
        else:
            assert maxrel(bp[k], b1[k]) < tol_b, k
    if pname == "bf16":
        assert tp > 0              # the BatchNorm backward reductions still come out of the dgrad epilogue (per-group coefficients)


@pytest.mark.gpu
@pytest.mark.parametrize("pname,C,H,W,N,epilogue", [("bf16", 256, 64, 64, 32, True), ("bf16", 64, 32, 32, 8, False), ("f32", 48, 16, 16, 4, False),
                                                     ("f32", 6, 16, 16, 4, False)])
def test_batchnorm_per_group_of_the_batch_equals_one_call_per_group(pname, C, H, W, N, epilogue):
    """ops.bn_batch_groups(2): a training-mode BatchNorm + LeakyReLU (architecture.py:116-118) over a batch that carries two passes
    takes its statistics, its apply and its backward per half of the batch -- against the same op called once per half (the
    reference's form: one pass, one call): outputs, input gradients (behind a conv, so that on the halo-kernel shapes the backward
    reductions come from the dgrad epilogue with per-group coefficients) and the replayed running statistics are the same BITS;
    the parameter gradients are the same two-term fp32 sums.  C = 6 exercises the padded channel stride."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.architecture import Conv2d
    prec = ops.BF16 if pname == "bf16" else ops.F32
    torch.manual_seed(5)
    cs = prec.pad(C)
    x = nhwc((torch.randn(N, C, H, W) * 1.5 + 0.3).to(prec.dtype), cs).to(DEV)
    conv = Conv2d(C, 32, 3, padding="same", padding_mode="reflect", bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(32, C, 3, 3) * math.sqrt(2.0 / (C * 9)))
    conv = conv.to(DEV)
    gy = torch.randn(N, H, W, prec.pad(32), device=DEV).to(prec.dtype)
    res = {}
    for grouped in (True, False):
        bn_w = (torch.rand(C) + 0.5).to(DEV).requires_grad_(True)
        bn_b = (torch.randn(C) * 0.1).to(DEV).requires_grad_(True)
        torch.manual_seed(6)
        bn_w.data.copy_(torch.rand(C) + 0.5)
        bn_b.data.copy_(torch.randn(C) * 0.1)
        rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        conv.weight.grad = None
        xg = x.clone().requires_grad_(True)
        before = dict(ops.bwd_fused_counts)
        with ops.bn_running_deferred() as running:
            if grouped:
                with ops.bn_batch_groups(2):
                    running.pass_index = (0, 1)
                    z = ops.batchnorm_act(xg, bn_w, bn_b, rm, rv, True, "leaky_relu", num_batches_tracked=nbt)
                out = conv(z)
            else:
                zs = []
                for g in range(2):
                    running.pass_index = g
                    zs.append(ops.batchnorm_act(xg[g * (N // 2):(g + 1) * (N // 2)], bn_w, bn_b, rm, rv, True, "leaky_relu",
                                                num_batches_tracked=nbt))
                z = torch.cat(zs, 0)
                out = torch.cat([conv(t) for t in zs], 0)
            running.apply()
        out.backward(gy)
        torch.cuda.synchronize()
        took = ops.bwd_fused_counts["taken"] - before["taken"]
        res[grouped] = (z.detach().clone(), xg.grad.clone(), bn_w.grad.clone(), bn_b.grad.clone(), rm.clone(), rv.clone(), int(nbt), took)
    g_, s_ = res[True], res[False]
    assert torch.equal(g_[0], s_[0])                     # the normalised activations
    assert torch.equal(g_[1], s_[1])                     # dL/dx
    assert maxrel(g_[2], s_[2]) < 1e-6 and maxrel(g_[3], s_[3]) < 1e-6
    assert torch.equal(g_[4], s_[4]) and torch.equal(g_[5], s_[5]) and g_[6] == s_[6] == 2
    if epilogue:
        assert g_[7] == 1 and s_[7] == 2                 # one dgrad launch took both groups' reductions


@pytest.mark.gpu
@pytest.mark.parametrize("N,hidden,Cs,dead", [(64, 128, [256, 256, 128, 64], False), (5, 64, [32, 16, 48], True), (3, 32, [16], False)])
def test_label_path_batched_equals_the_per_module_convs(N, hidden, Cs, dead):
    """ops.label_gamma_beta (csrc/label_path.hip): the gamma | beta table convs of several SPADE modules (normalization.py:20-22,33-35 on
    the 5 x 5 class image), each on its channel slice of one activation tensor, in one launch per direction -- against one generic conv
    per module (SPADE._gamma_beta's form): the same bf16 operands with fp32 accumulation in another order.  ``dead``: the last module's
    table gets no gradient (a module whose table a loss graph does not use): its filters get no gradient, its slice of the activation
    gradient is zero."""
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.networks.architecture import Conv2d
    torch.manual_seed(3)
    nm = len(Cs)
    actv0 = torch.relu(torch.randn(N, 5, 5, nm * hidden)).bfloat16().to(DEV)
    convs = [(Conv2d(hidden, C, 3, padding="same").to(DEV), Conv2d(hidden, C, 3, padding="same").to(DEV)) for C in Cs]
    gys = [torch.randn(N, 5, 5, 2 * C, device=DEV).bfloat16() for C in Cs]
    live = nm - 1 if dead else nm
    assert ops.label_gamma_beta_supported(actv0, hidden, convs)
    params = [p for g, b in convs for p in (g.weight, g.bias, b.weight, b.bias)]

    a = actv0.clone().requires_grad_(True)
    cache = {}
    tabs = ops.label_gamma_beta(a, hidden, convs, cache)
    torch.autograd.backward(tabs[:live], gys[:live])
    got_tabs, got_da = [t.detach().clone() for t in tabs], a.grad.clone()
    got_p = [p.grad.clone() if p.grad is not None else None for p in params]
    for p in params:
        p.grad = None
    again = ops.label_gamma_beta(actv0, hidden, convs, cache)            # the packed filters are reused (same stamps): same bits
    assert all(torch.equal(x, y) for x, y in zip(again, got_tabs))

    a2 = actv0.clone().requires_grad_(True)
    refs = []
    for i, ((g, b), C) in enumerate(zip(convs, Cs)):
        geom = ops.ConvGeom(hidden, 2 * C, 3, 1, 1, False, False)
        refs.append(ops.conv2d(a2[..., i * hidden:(i + 1) * hidden], torch.cat([g.weight, b.weight], 0), torch.cat([g.bias, b.bias], 0),
                               ops.PackedWeights(), geom, "none", sources=(g.weight, b.weight)))
    torch.autograd.backward(refs[:live], gys[:live])
    torch.cuda.synchronize()
    for i in range(nm):
        assert rel_l2(got_tabs[i], refs[i]) < 3e-3 and maxrel(got_tabs[i], refs[i]) < 2e-2, (i, rel_l2(got_tabs[i], refs[i]))
    assert rel_l2(got_da, a2.grad) < 4e-3, rel_l2(got_da, a2.grad)
    if dead:
        assert float(got_da[..., (nm - 1) * hidden:].abs().max()) == 0.0
    for k, p in enumerate(params):
        if p.grad is None:                           # the dead module's filters: no gradient in either form
            assert k // 4 == nm - 1 and got_p[k] is None
        else:
            assert rel_l2(got_p[k], p.grad) < (2e-4 if p.dim() == 4 else 1e-5), (k, rel_l2(got_p[k], p.grad))
