"""GPU parity of every HIP op (called through the C ABI via de_i2i_gan_amd.ops) against the CPU oracle's maths.

f32 mode is the exact-f32 MFMA parity path (tolerance 2e-4 of the tensor's max); bf16 mode is compared with a
reference evaluated on bf16-rounded operands (tolerance 1.5e-2 of the max: bf16 output rounding is 2^-9)."""
import math
from ctypes import byref

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-4, "bf16": 1.5e-2}


@pytest.fixture(scope="module")
def ops():
    from de_i2i_gan_amd import ops as _ops
    return _ops


def dev():
    return torch.device("cuda:0")


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def rounded(t, prec):
    return t.to(prec.dtype).float() if prec.name == "bf16" else t


def nhwc_ref(t, cs):
    """NCHW cpu tensor -> (N,H,W,cs) zero-padded"""
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs, dtype=t.dtype)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out


# (cin, cout, k, stride, pad, reflect, up, H, W, N, bias, act)
CONV_CASES = [
    (3, 8, 7, 1, 3, True, False, 16, 20, 2, False, "none"),        # stem
    (8, 16, 4, 2, 1, True, False, 16, 16, 2, False, "leaky_relu"),  # D / encoder strided conv + fused LReLU
    (16, 16, 3, 1, 1, True, False, 12, 10, 3, False, "none"),      # res / decoder conv
    (16, 8, 3, 1, 1, True, True, 6, 8, 2, False, "none"),          # fused nearest upsample
    (6, 16, 3, 1, 1, False, False, 5, 5, 3, True, "relu"),         # SPADE mlp_shared (zero pad, bias, relu)
    (16, 32, 3, 1, 1, False, False, 9, 7, 2, True, "none"),        # SPADE gamma|beta conv
    (32, 6, 4, 1, 0, False, False, 4, 4, 4, False, "none"),        # cls_clf full-extent valid conv
    (32, 1, 3, 1, 1, True, False, 4, 4, 4, False, "none"),         # src_clf
    (64, 160, 3, 1, 1, True, False, 24, 24, 2, False, "none"),     # multi-tile M and N, K = 576
    (256, 64, 4, 2, 1, True, False, 8, 8, 2, False, "leaky_relu"),  # K = 4096, small M -> split-K
    (64, 4, 3, 1, 1, True, False, 32, 32, 2, False, "none"),       # heads: narrow N tile
    # shapes that take the LDS-DMA 256-row-tile kernel in bf16 (channel stride % 64 == 0, >= 128 tiles)
    (64, 128, 3, 1, 1, True, False, 64, 64, 8, False, "none"),     # v2 BN=128, reflect; dgrad -> v2 BN=64
    (128, 64, 3, 1, 1, True, True, 32, 32, 8, False, "none"),      # v2 BN=64 with fused upsample; dgrad -> v2 BN=128
    (64, 128, 4, 2, 1, True, False, 128, 128, 8, False, "leaky_relu"),   # stride 2: 4 dgrad parity classes in one launch
    (192, 128, 3, 1, 1, False, False, 32, 32, 4, False, "none"),    # wgrad v2 with a 128-column k-tile spanning taps (Cs = 192)
    (128, 192, 3, 1, 1, False, False, 30, 34, 32, True, "none"),   # zero padding through the zero page, ragged M, bias
    (64, 256, 3, 1, 1, True, False, 64, 60, 8, True, "relu"),      # 256x256-tile variant (2-stage ring), ragged M; wgrad v2
    # halo-resident kernel (bf16, stride-1 3x3, H % 8 == 0, W % 32 == 0, >= 128 tiles); the (64,128,...,64,64,8) and the
    # fused-upsample case above take it too
    (128, 128, 3, 1, 1, True, False, 64, 32, 16, False, "none"),   # two 64-channel slices (halo double buffer), BN=128
    (64, 136, 3, 1, 1, False, False, 32, 64, 16, True, "relu"),    # zero boundary via the zero page, bias, ragged N tile
    # halo-resident wgrad (bf16, stride-1 3x3, Cout >= 96, H % 4 == 0, W % 32 == 0, combos x splits >= 128)
    (128, 256, 3, 1, 1, True, False, 16, 32, 16, False, "none"),   # 2 co tiles x 2 ci slices, 32 pixel splits of 2 half-tiles
    (64, 160, 3, 1, 1, True, True, 8, 16, 16, False, "none"),      # fused upsample, ragged Cout (160 = 128 + 32)
    (128, 64, 3, 1, 1, True, False, 16, 32, 8, False, "none"),     # 64 co x 128 ci variant (forced: see FORCE_WGRAD_HALO)
    (256, 48, 3, 1, 1, False, False, 8, 32, 4, False, "none"),     # the same with ragged Cout and zero padding
]
CONV_CASES += [
    # thin-input kernel (bf16, 8-channel source vectors, <= 64 outputs, >= 256 tiles of 8x32 pixels)
    (3, 64, 7, 1, 3, True, False, 128, 128, 4, False, "none"),          # generator stem
    (3, 24, 7, 1, 3, True, False, 64, 64, 2, False, "none"),            # thin-input wgrad with a partial co block
    (3, 64, 4, 2, 1, True, False, 256, 128, 8, True, "leaky_relu"),     # discriminator's first conv: stride 2, bias, LReLU
    (64, 4, 3, 1, 1, True, False, 128, 128, 4, False, "none"),          # heads: the interior dgrad (dY has 8 channels)
    (64, 4, 3, 1, 1, True, False, 16, 32, 4, False, "none"),            # heads wgrad: 64 x 64 block, taps split over two wave groups (forced)
]
CONV_CASES += [
    # 16x32-tile halo kernel (conv_halo16.hip: bf16, stride-1 3x3, H % 16 == 0, W % 32 == 0, channel stride % 32 == 0, a grid of
    # >= 7/8 of the CUs); the interior dgrad of each case takes it as well
    (64, 128, 3, 1, 1, True, False, 64, 64, 32, False, "none"),        # two 32-channel slices, BN = 128
    (32, 64, 3, 1, 1, True, False, 64, 64, 32, False, "leaky_relu"),   # ONE slice (no halo re-issue), BN = 64, fused LReLU
    (96, 136, 3, 1, 1, False, False, 64, 64, 16, True, "relu"),        # three slices, zero padding, bias, ragged N tile
    (64, 64, 3, 1, 1, True, True, 32, 32, 32, False, "none"),          # fused nearest upsample
    # the pipelined loop's ring / wait protocol on short and odd k-loops, and tiles that do NOT touch the image border:
    (32, 128, 3, 1, 1, True, False, 64, 64, 32, False, "none"),        # ONE slice: 9 k-steps, the weight ring never wraps fully
    (128, 128, 3, 1, 1, True, False, 48, 96, 32, False, "none"),       # 3 x 3 tiles per image (one interior tile); the dgrad is a
                                                                       # FOLD launch of the 128-channel tile with rings on some tiles only
    (160, 128, 3, 1, 1, False, False, 32, 64, 32, True, "leaky_relu"), # five slices (odd), the smallest FOLD-eligible image, zero pad
]
FORCE_WGRAD_HALO = {(128, 64, 3, 1, 1, True, False, 16, 32, 8, False, "none"), (256, 48, 3, 1, 1, False, False, 8, 32, 4, False, "none"),
                    (64, 4, 3, 1, 1, True, False, 16, 32, 4, False, "none")}


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(ops, pname, case, request):
    cin, cout, k, s, pad, reflect, up, H, W, N, has_bias, act = case
    prec = ops.get_precision(pname)
    if case in FORCE_WGRAD_HALO:         # small shapes normally stay on wgrad_v2 / v1 (too few workgroups): force the kernel
        from de_i2i_gan_amd import _lib
        _lib.load().dei2i_set_option(b"wgrad_halo", 2)
        request.addfinalizer(lambda: _lib.load().dei2i_set_option(b"wgrad_halo", 1))
    torch.manual_seed(1 + CONV_CASES.index(case))        # (hash() of a tuple with strings changes from process to process)
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin, k, k) * math.sqrt(2.0 / (cin * k * k))
    b = torch.randn(cout) * 0.3 if has_bias else None
    xr, wr = rounded(x, prec).double().requires_grad_(True), rounded(w, prec).double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_bias else None
    xl = O.upsample2x(xr) if up else xr
    y_ref = O.conv2d(xl, wr, stride=s, pad=pad, mode="reflect" if reflect else "zeros", bias=br)
    if act == "relu":
        y_ref = torch.relu(y_ref)
    elif act == "leaky_relu":
        y_ref = O.leaky_relu(y_ref)
    gy = torch.randn(y_ref.shape)
    gy_r = rounded(gy, prec).double()
    grads = torch.autograd.grad(y_ref, [xr, wr] + ([br] if has_bias else []), gy_r)

    xg = x.to(dev()).requires_grad_(True)
    wg = w.to(dev()).requires_grad_(True)
    bg = b.to(dev()).requires_grad_(True) if has_bias else None
    geom = ops.ConvGeom(cin, cout, k, s, pad, reflect, up)
    cache = ops.PackedWeights()
    xh = ops.to_nhwc(xg, prec)
    y = ops.conv2d(xh, wg, bg, cache, geom, act)
    assert y.shape[-1] == prec.pad(cout)
    assert y[..., cout:].abs().max().item() == 0 if prec.pad(cout) > cout else True       # channel padding is zero
    tol = TOL[pname]
    assert maxrel(ops.to_nchw(y, cout), y_ref) < tol
    gyh = nhwc_ref(gy, prec.pad(cout)).to(dev()).to(prec.dtype)
    y.backward(gyh)
    # with a fused activation a pre-activation within rounding of 0 can land on the other side of the kink than in the
    # reference (a full-magnitude outlier in a handful of elements) -> judge those cases by relative L2
    err = (lambda a, b: ((a.detach().double().cpu() - b).norm() / b.norm()).item()) if act != "none" else maxrel
    gtol = max(tol, 4e-3) if act != "none" else tol           # a few kink flips among millions of outputs (measured up to 2.1e-3)
    assert err(xg.grad, grads[0]) < gtol, "dgrad"
    assert err(wg.grad, grads[1]) < gtol, "wgrad"
    if has_bias:
        assert err(bg.grad, grads[2]) < gtol, "bias grad"


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("training,act,with_res,C", [(True, "leaky_relu", False, 16), (True, "none", True, 16), (False, "leaky_relu", False, 16),
                                                     # num_features below the padded channel stride (6 -> 8 in both modes): the
                                                     # per-channel vectors are shorter than the activation's channel dimension
                                                     (True, "leaky_relu", False, 6), (False, "none", True, 6)])
def test_batchnorm_act(ops, pname, training, act, with_res, C):
    prec = ops.get_precision(pname)
    torch.manual_seed(3)
    N, H, W = 3, 10, 12
    y = torch.randn(N, C, H, W) * 1.7 + 0.4
    res = torch.randn(N, C, H, W) if with_res else None
    S = {"bn.weight": (1 + 0.2 * torch.randn(C)).double().requires_grad_(True), "bn.bias": (0.1 * torch.randn(C)).double().requires_grad_(True),
         "bn.running_mean": 0.1 * torch.randn(C).double(), "bn.running_var": (1 + 0.1 * torch.rand(C)).double(),
         "bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    rm0, rv0 = S["bn.running_mean"].clone(), S["bn.running_var"].clone()
    yr = rounded(y, prec).double().requires_grad_(True)
    rr = rounded(res, prec).double().requires_grad_(True) if with_res else None
    out_ref = O.batchnorm(S, "bn", yr, training)
    if act == "leaky_relu":
        out_ref = O.leaky_relu(out_ref)
    if with_res:
        out_ref = out_ref + rr
    g = torch.randn(N, C, H, W)
    gr = rounded(g, prec).double()
    grads = torch.autograd.grad(out_ref, [yr, S["bn.weight"], S["bn.bias"]] + ([rr] if with_res else []), gr)

    wg = S["bn.weight"].detach().float().to(dev()).requires_grad_(True)
    bg = S["bn.bias"].detach().float().to(dev()).requires_grad_(True)
    rm, rv = rm0.float().to(dev()), rv0.float().to(dev())
    yg = ops.to_nhwc(y.to(dev()), prec).requires_grad_(True)
    resg = ops.to_nhwc(res.to(dev()), prec).requires_grad_(True) if with_res else None
    out = ops.batchnorm_act(yg, wg, bg, rm, rv, training, act, resg)
    tol = TOL[pname]
    assert maxrel(ops.to_nchw(out, C), out_ref) < tol
    assert out.shape[-1] == prec.pad(C) and (prec.pad(C) == C or float(out[..., C:].abs().max()) == 0.0)     # padding stays zero
    out.backward(nhwc_ref(g, prec.pad(C)).to(dev()).to(prec.dtype))
    assert wg.grad.shape == (C,) and bg.grad.shape == (C,) and rm.shape == (C,)
    assert maxrel(ops.to_nchw(yg.grad, C), grads[0]) < tol * 2
    assert maxrel(wg.grad, grads[1]) < tol * 2
    assert maxrel(bg.grad, grads[2]) < tol * 2
    if with_res:
        assert maxrel(ops.to_nchw(resg.grad, C), grads[3]) < tol
    if training:      # running stats: momentum 0.1, unbiased variance
        assert maxrel(rm, S["bn.running_mean"]) < 1e-4 + (5e-3 if pname == "bf16" else 0)
        assert maxrel(rv, S["bn.running_var"]) < 1e-4 + (5e-3 if pname == "bf16" else 0)
    else:
        assert torch.equal(rm.cpu(), rm0.float())


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("up,class_mode", [(False, False), (True, False), (False, True), (True, True)])
def test_spade_relu(ops, pname, up, class_mode):
    """SPADE + ReLU through the product modules (architecture.SPADE) vs the oracle's spade()."""
    from de_i2i_gan_amd.networks.architecture import SPADE
    prec = ops.get_precision(pname)
    torch.manual_seed(11)
    N, C, Hs, Ws, label_nc, hidden = 2, 16, 6, 8, 6, 16
    x = torch.randn(N, C, Hs, Ws) * 1.3 + 0.2
    seg = torch.zeros(N, label_nc, 1, 1) if class_mode else torch.rand(N, label_nc, 3, 2)
    if class_mode:
        seg[0, 1], seg[1, 3] = 1, 1
    mod = SPADE(label_nc, C, hidden_nc=hidden, kernel_size=(3, 3), padding="same")
    S = {}
    with torch.no_grad():
        for k_, p in mod.named_parameters():
            p.copy_(O.formula_tensor("spade." + k_, tuple(p.shape)) * (3.0 if p.dim() == 4 else 1.0))
            S["sp." + k_] = rounded(p.detach().clone(), prec).double().requires_grad_(True) if p.dim() == 4 else p.detach().clone().double().requires_grad_(True)
    mod = mod.to(dev())
    xr = rounded(x, prec).double().requires_grad_(True)
    xin = O.upsample2x(xr) if up else xr
    out_ref = torch.relu(O.spade(S, "sp", xin, rounded(seg, prec).double()))
    g = torch.randn(out_ref.shape)
    gr = rounded(g, prec).double()
    keys = [k_ for k_ in S]
    grads = torch.autograd.grad(out_ref, [xr] + [S[k_] for k_ in keys], gr)

    xg = ops.to_nhwc(x.to(dev()), prec).requires_grad_(True)
    out = mod(xg, seg.to(dev()), up=up)
    # bf16: gamma/beta/actv intermediates are rounded to bf16 as well -> looser
    tol = TOL[pname] * (3 if pname == "bf16" else 1)
    assert maxrel(ops.to_nchw(out, C), out_ref) < tol
    out.backward(nhwc_ref(g, C).to(dev()).to(prec.dtype))
    assert maxrel(ops.to_nchw(xg.grad, C), grads[0]) < tol * 2, "dx"
    named = dict(mod.named_parameters())
    for k_, gref in zip(keys, grads[1:]):
        if pname == "bf16":
            # gamma/beta/actv are themselves rounded to bf16 in the product path, so a few ReLU masks flip relative
            # to the reference; each flip is a full-magnitude outlier in one element -> judge by relative L2
            got = named[k_[3:]].grad.double().cpu()
            assert ((got - gref).norm() / gref.norm()).item() < 5e-2, k_
        else:
            assert maxrel(named[k_[3:]].grad, gref) < tol * 2, k_


@pytest.mark.parametrize("pname", ["f32", "bf16"])
def test_compose_and_nan_guard(ops, pname):
    prec = ops.get_precision(pname)
    torch.manual_seed(5)
    N, H, W = 2, 9, 7
    raw = torch.randn(N, 4, H, W) * 1.5
    x = torch.rand(N, 3, H, W) * 2 - 1
    rr = rounded(raw, prec).double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    p = torch.sigmoid(rr[:, 3:4])
    out_ref = xr * (1 - p) + torch.tanh(rr[:, :3]) * p
    go, gp = torch.randn(N, 3, H, W), torch.randn(N, 1, H, W)
    grads = torch.autograd.grad([out_ref, p], [rr, xr], [go.double(), gp.double()])
    rawg = ops.to_nhwc(raw.to(dev()), prec).requires_grad_(True)
    xg = x.to(dev()).requires_grad_(True)
    out, prob = ops.compose(rawg, xg)
    assert maxrel(out, out_ref) < 1e-5 and maxrel(prob, p) < 1e-5
    torch.autograd.backward([out, prob], [go.to(dev()), gp.to(dev())])
    assert maxrel(ops.to_nchw(rawg.grad, 4), grads[0]) < TOL[pname]
    assert maxrel(xg.grad, grads[1]) < 1e-5
    # NaN guard: untouched without NaN; nan->0, inf->max with one
    t = ops.to_nhwc(torch.randn(1, 8, 4, 4).to(dev()), prec)
    before = t.clone()
    ops.nan_guard_(t)
    assert torch.equal(t, before)
    t[0, 1, 2, 3] = float("nan")
    t[0, 0, 0, 0] = float("inf")
    ops.nan_guard_(t)
    assert t[0, 1, 2, 3].item() == 0 and torch.isfinite(t).all() and t[0, 0, 0, 0].item() > 1e38


def test_losses(ops):
    torch.manual_seed(9)
    x = torch.randn(4, 1, 5, 5) * 3
    t = torch.rand(4, 1, 5, 5).round()
    for target in (t, 1.0, 0.0):
        xr = x.double().requires_grad_(True)
        tt = target.double() if isinstance(target, torch.Tensor) else torch.full_like(xr, target)
        ref = O.bce_logits(xr, tt)
        (gref,) = torch.autograd.grad(ref * 1.7, xr)
        xg = x.to(dev()).requires_grad_(True)
        out = ops.bce_logits(xg, target.to(dev()) if isinstance(target, torch.Tensor) else target)
        (out * 1.7).backward()
        assert abs(out.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
        assert maxrel(xg.grad, gref) < 1e-5
    a, b = torch.randn(3, 3, 8, 8), torch.randn(3, 3, 8, 8)
    for bb in (b, None):
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if bb is not None else None
        ref = O.l1(ar, br if bb is not None else torch.zeros_like(ar))
        gref = torch.autograd.grad(ref, [ar] + ([br] if bb is not None else []))
        ag = a.to(dev()).requires_grad_(True)
        bgpu = b.to(dev()).requires_grad_(True) if bb is not None else None
        out = ops.l1(ag, bgpu)
        out.backward()
        assert abs(out.item() - ref.item()) < 1e-6
        assert maxrel(ag.grad, gref[0]) < 1e-6
        if bb is not None:
            assert maxrel(bgpu.grad, gref[1]) < 1e-6


def test_fused_adam_matches_oracle():
    from de_i2i_gan_amd.optim import FusedAdam
    torch.manual_seed(2)
    shapes = [(7,), (3, 5, 2, 2), (129,), (64, 3, 4, 4), (1,)]
    ps = [torch.randn(s) for s in shapes]
    cfg = O.Cfg()
    S = {str(i): p.clone() for i, p in enumerate(ps)}
    st = O.AdamState()
    gp = [p.clone().to(dev()).requires_grad_(True) for p in ps]
    opt = FusedAdam(gp, lr=cfg.lr, betas=cfg.betas, eps=cfg.eps)
    for step in range(3):
        gs = [torch.randn(s) * (10 ** (step - 1)) for s in shapes]
        grads = {str(i): g for i, g in enumerate(gs)}
        if step == 1:
            grads["2"] = None                       # a parameter without grad is skipped, its step count lags
        O.adam_update(S, grads, st, cfg)
        for i, p in enumerate(gp):
            p.grad = None if grads[str(i)] is None else gs[i].to(dev())
        opt.step()
    for i, p in enumerate(gp):
        assert maxrel(p, S[str(i)]) < 2e-6, i
        assert getattr(p, "_dei2i_epoch", 0) >= 2


@pytest.mark.parametrize("shape", [(16, 8, 3, 3), (6, 32, 4, 4), (1, 24, 3, 3), (40, 3, 7, 7), (256, 64, 3, 3)])
@pytest.mark.parametrize("iterate", [True, False])
def test_spectral_weight_op(ops, shape, iterate):
    """ops.spectral_weight (csrc/spectral.hip) against the oracle's restatement of torch.nn.utils.spectral_norm: the
    iterated buffers, w / sigma, and the weight gradient -- with the parameter used TWICE in one backward pass, so the
    second node adds into the first one's gradient tensor inside the kernel (K % 4 != 0 cases take the scalar kernel)."""
    torch.manual_seed(3)
    w = torch.randn(shape) * 0.2
    u = F.normalize(torch.randn(shape[0]), dim=0)
    v = F.normalize(torch.randn(w[0].numel()), dim=0)
    g1, g2 = torch.randn(shape), torch.randn(shape)
    # reference in float64 through the oracle (two successive forwards, like two calls of the module in one step)
    S = {"c.weight_orig": w.double().requires_grad_(True), "c.weight_u": u.double().clone(), "c.weight_v": v.double().clone()}
    r1 = O.weight_of(S, "c.weight", iterate)
    r2 = O.weight_of(S, "c.weight", iterate)
    ((r1 * g1.double()).sum() + (r2 * g2.double()).sum()).backward()
    wg = torch.nn.Parameter(w.to(dev()))
    ug, vg = u.to(dev()), v.to(dev())
    o1 = ops.spectral_weight(wg, ug, vg, iterate)
    o2 = ops.spectral_weight(wg, ug, vg, iterate)
    ((o1 * g1.to(dev())).sum() + (o2 * g2.to(dev())).sum()).backward()
    assert maxrel(o1, r1) < 2e-5 and maxrel(o2, r2) < 2e-5
    assert maxrel(ug, S["c.weight_u"]) < 2e-5 and maxrel(vg, S["c.weight_v"]) < 2e-5
    assert maxrel(wg.grad, S["c.weight_orig"].grad) < 5e-5


@pytest.mark.parametrize("pname", ["f32", "bf16"])
def test_noise_inject_op(ops, pname):
    """ops.noise_inject: x + w * noise (one value per pixel) and its gradients against float64 torch; the weight is used
    twice in the backward pass (in-kernel accumulation of the second node)."""
    from de_i2i_gan_amd.ops import BF16, F32
    prec = BF16 if pname == "bf16" else F32
    torch.manual_seed(5)
    n, h, w_, c = 3, 9, 7, 16
    x1, x2 = torch.randn(n, h, w_, c), torch.randn(n, h, w_, c)
    nz1, nz2 = torch.randn(n, 1, h, w_), torch.randn(n, 1, h, w_)
    gy1, gy2 = torch.randn(n, h, w_, c), torch.randn(n, h, w_, c)
    wt = torch.tensor(0.37).reshape(1, 1, 1, 1)
    xr1, xr2 = rounded(x1, prec).double().requires_grad_(True), rounded(x2, prec).double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    y1 = xr1 + wr.reshape(()) * nz1.double().reshape(n, h, w_, 1)
    y2 = xr2 + wr.reshape(()) * nz2.double().reshape(n, h, w_, 1)
    gr1, gr2 = rounded(gy1, prec).double(), rounded(gy2, prec).double()
    ((y1 * gr1).sum() + (y2 * gr2).sum()).backward()
    wg = torch.nn.Parameter(wt.to(dev()))
    xg1 = x1.to(dev(), prec.dtype).requires_grad_(True)
    xg2 = x2.to(dev(), prec.dtype).requires_grad_(True)
    o1 = ops.noise_inject(xg1, wg, nz1.to(dev()))
    o2 = ops.noise_inject(xg2, wg, nz2.to(dev()))
    torch.autograd.backward([o1, o2], [gy1.to(dev(), prec.dtype), gy2.to(dev(), prec.dtype)])
    assert maxrel(o1, y1) < TOL[pname] and maxrel(o2, y2) < TOL[pname]
    assert maxrel(xg1.grad, gr1) < 1e-6 and maxrel(xg2.grad, gr2) < 1e-6
    assert wg.grad.shape == wt.shape and maxrel(wg.grad, wr.grad) < 2e-5


def test_halo_conv_tile_variants_bit_identical(ops):
    """The halo conv's tile options (output-channel width 64 | 128, weight ring 4 / 6 / 8 deep) change the LDS schedule,
    not the arithmetic: forward and dgrad must come out bit-identical to the shipped tile, launch after launch.  (A
    3-deep ring failed exactly this -- a race between the two wave groups -- and was removed.)"""
    from de_i2i_gan_amd import _lib
    lib = _lib.load()
    lib.dei2i_set_option(b"halo16", 0)          # the sweep options belong to the 8 x 32 tile kernel: keep the 16 x 32 one out
    try:
        for cin, cout, hw, n, up in ((128, 64, 128, 8, False), (256, 128, 64, 16, False), (128, 128, 32, 16, True)):
            torch.manual_seed(11)
            geom = ops.ConvGeom(cin, cout, 3, 1, 1, True, up)
            x = torch.randn(n, hw, hw, cin, device=dev()).bfloat16()
            w = torch.randn(cout, cin, 3, 3, device=dev()) * 0.05
            ho = hw * 2 if up else hw
            gy = torch.randn(n, ho, ho, cout, device=dev()).bfloat16()

            def run():
                xd = x.detach().requires_grad_(True)
                y = ops.conv2d(xd, w, None, ops.PackedWeights(), geom, "none")
                (dx,) = torch.autograd.grad(y, xd, gy)
                return y.detach().view(torch.int16).clone(), dx.detach().view(torch.int16).clone()

            lib.dei2i_set_option(b"halo_bn", 0)
            lib.dei2i_set_option(b"halo_stages", 0)
            y0, d0 = run()
            for bn, stg in ((0, 0), (64, 4), (64, 6), (64, 8)):
                lib.dei2i_set_option(b"halo_bn", bn)
                lib.dei2i_set_option(b"halo_stages", stg)
                for _ in range(3):
                    y, d = run()
                    assert torch.equal(y, y0) and torch.equal(d, d0), (cin, cout, hw, bn, stg)
    finally:
        lib.dei2i_set_option(b"halo_bn", 0)
        lib.dei2i_set_option(b"halo_stages", 0)
        lib.dei2i_set_option(b"halo16", 3)      # (the shipped default: pipelined loop)


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("class_mode", [False, True])
def test_spade_relu_skip_branch(ops, pname, class_mode):
    """SPADE(..., skip=True) hands x through for the res block's identity branch; the gradient arriving there is added
    inside the backward-apply kernel: loss(z) + loss(xs) must give the gradients of the two-node formulation."""
    from de_i2i_gan_amd.networks.architecture import SPADE
    prec = ops.get_precision(pname)
    torch.manual_seed(13)
    N, C, H, W, label_nc = 2, 16, 8, 8, 6
    mod = SPADE(label_nc, C, hidden_nc=16, kernel_size=(3, 3), padding="same")
    with torch.no_grad():
        for k_, p in mod.named_parameters():
            p.copy_(O.formula_tensor("spade." + k_, tuple(p.shape)) * (3.0 if p.dim() == 4 else 1.0))
    mod = mod.to(dev())
    seg = torch.zeros(N, label_nc, 1, 1) if class_mode else torch.rand(N, label_nc, 4, 4)
    if class_mode:
        seg[0, 2], seg[1, 4] = 1, 1
    seg = seg.to(dev())
    x = ops.to_nhwc((torch.randn(N, C, H, W) * 1.2).to(dev()), prec)
    gz = torch.randn(N, H, W, C, device=dev()).to(prec.dtype)
    gs = torch.randn(N, H, W, C, device=dev()).to(prec.dtype)

    def grads(skip):
        for p in mod.parameters():
            p.grad = None
        mod._gb_cache.clear()                # the class table is memoised WITH its graph per loss evaluation
        xg = x.detach().clone().requires_grad_(True)
        if skip:
            z, xs = mod(xg, seg, skip=True)
            assert xs.data_ptr() == xg.data_ptr()
        else:
            z, xs = mod(xg, seg), xg
        torch.autograd.backward([z, xs * 1], [gz, gs]) if not skip else torch.autograd.backward([z, xs], [gz, gs])
        return z.detach(), xg.grad.detach(), [p.grad.detach().clone() for p in mod.parameters()]

    z0, dx0, pg0 = grads(False)
    z1, dx1, pg1 = grads(True)
    assert torch.equal(z0, z1)
    # one rounding of (dx + skip) in the kernel against two in the two-node formulation
    assert maxrel(dx1, dx0) < (1e-6 if pname == "f32" else 1e-2)
    for a, b in zip(pg1, pg0):
        assert torch.equal(a, b)


@pytest.mark.parametrize("pname", ["f32", "bf16"])
def test_weight_used_twice_activation_gradient_only_and_with_foreign_contribution(ops, pname):
    """A conv weight applied twice in one graph (the G step applies every generator layer four times): the second
    backward node adds into the first node's gradient tensor (ops._grad_target).  (a) torch.autograd.grad(inputs=[x]) --
    the engine never runs the weight's AccumulateGrad and drops the first node's tensor, so a later node must not write
    into it; (b) a full backward where an op outside this library contributes to the same parameter too.  Both against
    the oracle's conv in float64 on the rounded operands."""
    prec = ops.get_precision(pname)
    torch.manual_seed(21)
    N, C, H, W = 2, 16, 12, 16
    x = torch.randn(N, C, H, W)
    w = torch.randn(C, C, 3, 3) * math.sqrt(2.0 / (C * 9))
    gy = torch.randn(N, C, H, W)
    xr, wr = rounded(x, prec).double().requires_grad_(True), rounded(w, prec).double().requires_grad_(True)
    h_ref = O.conv2d(xr, wr, stride=1, pad=1, mode="reflect")
    y_ref = O.conv2d(h_ref, wr, stride=1, pad=1, mode="reflect")
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xr, wr], rounded(gy, prec).double())
    geom = ops.ConvGeom(C, C, 3, 1, 1, True, False)
    wg = torch.nn.Parameter(w.to(dev()))
    cache = ops.PackedWeights()
    gyh = nhwc_ref(gy, C).to(dev()).to(prec.dtype)
    tol = TOL[pname] * 2            # the intermediate activation is rounded once more than in the reference

    def graph():
        xg = x.to(dev()).requires_grad_(True)
        y = ops.conv2d(ops.conv2d(ops.to_nhwc(xg, prec), wg, None, cache, geom), wg, None, cache, geom)
        return xg, y

    xg, y = graph()
    (gx,) = torch.autograd.grad(y, [xg], gyh)                       # (a): nothing may be written for the weight
    canary = torch.zeros(1 << 20, device=dev())                     # would-be victim of a write into freed memory
    torch.cuda.synchronize()
    assert maxrel(gx, gx_ref) < tol and float(canary.abs().max()) == 0.0 and wg.grad is None
    xg, y = graph()                                                 # (b): a foreign contribution to the same parameter
    torch.autograd.backward([y, (wg * 3.0).sum()], [gyh, torch.ones((), device=dev())])
    assert maxrel(wg.grad, gw_ref + 3.0) < tol and maxrel(xg.grad, gx_ref) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("pname", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(256, 256, 3, 1), (128, 64, 4, 2), (136, 72, 3, 1), (2048, 1024, 4, 2), (64, 250, 3, 1),
                                   (48, 33, 4, 2)])
def test_pack_weight_both_matches_the_separate_packers(ops, pname, shape):
    """dei2i_pack_weight_both (one launch; LDS-tiled for the large weights) writes exactly the bytes of dei2i_pack_weight_fwd +
    dei2i_pack_weight_dgrad (the per-element kernels whose index functions geom.h documents)."""
    from de_i2i_gan_amd import _lib as L
    cout, cin, k, s = shape
    prec = ops.get_precision(pname)
    lib = L.load()
    geom = ops.ConvGeom(cin, cout, k, s, 1, True, False)
    d = ops._desc(prec, geom, 1, 8, 8, prec.pad(cin), prec.pad(cout))
    torch.manual_seed(cout + cin)
    w = torch.randn(cout, cin, k, k, device=dev())
    nf, nd = lib.dei2i_packed_fwd_elems(byref(d)), lib.dei2i_packed_dgrad_elems(byref(d))
    f1 = torch.full((nf,), 7.0, dtype=prec.dtype, device=dev())
    g1 = torch.full((nd,), 7.0, dtype=prec.dtype, device=dev())
    f2, g2 = torch.full_like(f1, 5.0), torch.full_like(g1, 5.0)
    L.check(lib.dei2i_pack_weight_both(byref(d), ops._p(w), ops._p(f1), ops._p(g1), ops._stream()), "both")
    L.check(lib.dei2i_pack_weight_fwd(byref(d), ops._p(w), ops._p(f2), ops._stream()), "fwd")
    L.check(lib.dei2i_pack_weight_dgrad(byref(d), ops._p(w), ops._p(g2), ops._stream()), "dgrad")
    torch.cuda.synchronize()
    assert torch.equal(f1, f2)
    assert torch.equal(g1, g2)


@pytest.mark.gpu
@pytest.mark.parametrize("pname,N,C,H,act", [("f32", 2, 8, 8, "leaky_relu"), ("f32", 3, 20, 16, "none"), ("bf16", 4, 64, 32, "leaky_relu"),
                                             ("bf16", 2, 256, 8, "none"), ("bf16", 2, 16, 4, "relu")])
def test_instance_norm_act_and_avgpool_match_torch(pname, N, C, H, act):
    """The conv StyleExtractor's blocks (extractor.py:50-80 with architecture.py:79-176): act(InstanceNorm2d(x)) (affine=False, eps
    1e-5, biased variance) and AvgPool2d(2, 2), forward and backward, against torch's own ops in float64 on the rounded operands."""
    import torch.nn.functional as F
    from de_i2i_gan_amd import ops
    prec = ops.get_precision(pname)
    torch.manual_seed(3)
    cs = prec.pad(C)
    x = torch.zeros(N, H, H, cs)
    x[..., :C] = torch.randn(N, H, H, C) * 1.7 + 0.3
    x = x.to(prec.dtype)
    g = torch.zeros(N, H // 2, H // 2, cs)
    g[..., :C] = torch.randn(N, H // 2, H // 2, C)
    g = g.to(prec.dtype)
    xg = x.to(dev()).requires_grad_(True)
    y = ops.avgpool2(ops.instance_norm_act(xg, act))
    y.backward(g.to(dev()))
    xr = x.double().permute(0, 3, 1, 2)[:, :C].requires_grad_(True)
    z = F.instance_norm(xr, eps=1e-5)
    z = F.leaky_relu(z, 0.2) if act == "leaky_relu" else (torch.relu(z) if act == "relu" else z)
    yr = F.avg_pool2d(z, 2, 2)
    yr.backward(g.double().permute(0, 3, 1, 2)[:, :C])
    tol = 2e-5 if pname == "f32" else 1.5e-2
    got_y = y.detach().double().cpu().permute(0, 3, 1, 2)[:, :C]
    got_dx = xg.grad.double().cpu().permute(0, 3, 1, 2)[:, :C]
    assert float((got_y - yr.detach()).abs().max() / yr.detach().abs().max()) < tol
    # InstanceNorm's backward subtracts means: the bf16 rounding of the output is amplified by the cancellation -> relative L2 there
    if pname == "f32":
        assert float((got_dx - xr.grad).abs().max() / xr.grad.abs().max()) < 1e-4
    else:
        assert float((got_dx - xr.grad).norm() / xr.grad.norm()) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sgd", "rmsprop"])
def test_fused_sgd_and_rmsprop_match_torch(kind):
    """trainers/base_trainer.py:71-74: ``optim.SGD(params, lr=...)`` / ``optim.RMSprop(params, lr=...)`` (torch's defaults); three
    steps over tensors of odd sizes, one of them without a gradient (skipped: no state), against torch's own optimizer in float64."""
    from de_i2i_gan_amd.optim import FusedRMSprop, FusedSGD
    torch.manual_seed(9)
    shapes = [(7,), (33, 5), (64, 64, 3, 3), (1,)]
    ps = [torch.randn(s) for s in shapes]
    mine = [p.clone().to(dev()).requires_grad_(True) for p in ps]
    ref = [p.clone().double().requires_grad_(True) for p in ps]
    o_mine = (FusedSGD if kind == "sgd" else FusedRMSprop)(mine, lr=1e-2)
    o_ref = (torch.optim.SGD if kind == "sgd" else torch.optim.RMSprop)(ref, lr=1e-2)
    for step in range(3):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 3:
                a.grad = b.grad = None                       # a parameter the loss does not reach
                continue
            g = torch.randn(shapes[i]) * (1.0 + step)
            a.grad, b.grad = g.to(dev()), g.double()
        o_mine.step()
        o_ref.step()
    for a, b in zip(mine, ref):
        assert float((a.detach().double().cpu() - b.detach()).abs().max()) < 2e-6 * max(1.0, float(b.detach().abs().max()))
    assert len(o_mine.state.get(mine[3], {})) == 0
