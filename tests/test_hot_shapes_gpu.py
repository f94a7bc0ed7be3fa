"""Parity of the bf16 hot kernels AT THE BENCHMARKED CONFIGURATION (BASELINE.json configs[1]: 256x256, batch 16, default
widths), which the tiny goldens and the f32-mode full-size tests never reach ("what's weak" #1):

  * op level: forward, dgrad and wgrad of the exact shapes the 256x256 / batch-16 step launches, against the oracle's
    conv (oracle/defectgan_oracle.py::conv2d, architecture.py:51-56,95-100,228-233; discriminator.py:60-77) evaluated
    in fp32 on the bf16-ROUNDED operands -- so the only differences are the fp32 accumulation order and the bf16
    rounding of the stored output (2^-9 relative per element);
  * which kernel ran: the library's host-side launch counters (dei2i_launch_counts) must show the tuned kernel family
    for every phase -- a silent hipErrorNotSupported fall-through to the generic GEMM fails the test;
  * step level: one D + G loss/backward at 256x256, batch 16, default widths, bf16 mode against this build's exact-f32
    mode with the same seed (the f32 mode is what the reference-generated goldens pin, tests/test_model_gpu.py).

Tolerances (written here, measured values in the comments of each assert): outputs are judged by relative L2 and by the
max error relative to the tensor's max; bf16 rounding of a stored output gives ~1.1e-3 relative L2 (2^-9 / sqrt(3) on
the mantissa step), fp32 gradients of weights only differ by summation order."""
import json
import math
import os

import pytest
import torch

from helpers import make_opt
from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from de_i2i_gan_amd import ops as _ops
    return _ops


def _counts(reset=True, splitk=False):
    """launches per MFMA kernel family since the last call (``splitk``: keep the split-K finalize launches too)"""
    from de_i2i_gan_amd import _lib
    return {k: v for k, v in _lib.launch_counts(reset=reset).items() if v and (splitk or k != "splitk_finalize")}


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def maxrel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# (name, cin, cout, k, stride, pad, up, H, W, N, act, forward kernel, dgrad kernels, wgrad kernel)  -- all reflect-padded, no
# bias, as the reference builds them (generator.py:67-73,107-126,139-152,178-191,224-241; discriminator.py:60-77).
# N = 16: one train-mode generator pass; N = 64: the discriminator sees the 4 image batches of a D step as one batch
# (DESIGN.md "Batched where the function allows it").  The kernel columns are the library's CURRENT dispatch, pinned:
# a change of dispatch must be made here on purpose.  "gather_v1" next to a tuned kernel in a dgrad is the thin reflect
# ring of a stride-1 conv (conv_api.hip: decomposed reflect dgrad).  Shapes still on the generic kernels (listed in
# DESIGN.md as open): D's first-conv dgrad (64 -> 3 channels, stride 2), the wgrads of the 64-channel-input stride-2 convs
# and of D's first and last conv, D's last conv forward (M = 1024 rows: split-K generic GEMM).
R, V2, V1 = {"halo_conv": 1, "gather_v1": 1}, {"gather_v2": 1}, {"gather_v1": 1}
# "halo16_s2": the 16 x 32 tile kernel's stride-2 form (parity planes in LDS) -- taken for the 64-channel inputs, where it beats the
# gather GEMM (conv_halo16.hip: halo16_s2_shape_ok)
# V2R: the stride-2 reflect dgrad of a LARGE frame, decomposed (conv_api.hip dei2i_conv2d_dgrad_input): the four parity classes of the
# zero-boundary dgrad in one launch straight into dx + two small launches for the frame's ring rows / columns (then the border fold)
V2R = {"gather_v2": 1, "gather_v1": 2}
R16 = {"halo16_conv": 1}                            # 16 x 32 tile kernel: interior, reflect ring and the four frame corners in ONE launch
HOT = [
    ("res 256->256 3x3 @64^2 N=16", 256, 256, 3, 1, 1, False, 64, 64, 16, "none", "halo16_conv", R16, "wgrad_halo"),
    ("res 256->256 3x3 @64^2 N=8", 256, 256, 3, 1, 1, False, 64, 64, 8, "none", "halo_conv", R, "wgrad_halo"),
    ("dec0 256->128 up @128^2 N=16", 256, 128, 3, 1, 1, True, 64, 64, 16, "none", "halo16_conv", V2, "wgrad_halo"),
    ("dec1 128->64 up @256^2 N=16", 128, 64, 3, 1, 1, True, 128, 128, 16, "none", "halo16_conv", V2, "wgrad_halo"),
    ("enc0 64->128 4x4 s2 @256^2 N=16", 64, 128, 4, 2, 1, False, 256, 256, 16, "none", "halo16_s2", V2R, "wgrad_v1"),
    ("enc1 128->256 4x4 s2 @128^2 N=16", 128, 256, 4, 2, 1, False, 128, 128, 16, "none", "gather_v2", V2, "wgrad_v2"),
    ("enc1 128->256 4x4 s2 @128^2 N=32 (the paired passes' batch)", 128, 256, 4, 2, 1, False, 128, 128, 32, "none", "gather_v2", V2R, "wgrad_v2"),
    ("stem 3->64 7x7 @256^2 N=16", 3, 64, 7, 1, 3, False, 256, 256, 16, "none", "thin_cin", {"thin_cout": 1, "gather_v1": 1}, "wgrad_thin"),
    ("heads 64->4 3x3 @256^2 N=16", 64, 4, 3, 1, 1, False, 256, 256, 16, "none", "thin_cout", {"thin_cin": 1, "gather_v1": 1}, "wgrad_halo"),
    ("D0 3->64 4x4 s2 @256^2 N=64 +LReLU", 3, 64, 4, 2, 1, False, 256, 256, 64, "leaky_relu", "thin_cin", V1, "wgrad_v1"),
    ("D1 64->128 4x4 s2 @128^2 N=64 +LReLU", 64, 128, 4, 2, 1, False, 128, 128, 64, "leaky_relu", "halo16_s2", V2R, "wgrad_v1"),
    ("D2 128->256 4x4 s2 @64^2 N=64 +LReLU", 128, 256, 4, 2, 1, False, 64, 64, 64, "leaky_relu", "gather_v2", V2, "wgrad_v2"),
    ("D3 256->512 4x4 s2 @32^2 N=64 +LReLU", 256, 512, 4, 2, 1, False, 32, 32, 64, "leaky_relu", "gather_v2", V2, "wgrad_v2"),   # 16x16 outputs: below the 16x32 tile
    ("D4 512->1024 4x4 s2 @16^2 N=64 +LReLU", 512, 1024, 4, 2, 1, False, 16, 16, 64, "leaky_relu", "gather_v2", V2, "wgrad_v2"),
    ("D5 1024->2048 4x4 s2 @8^2 N=64 +LReLU", 1024, 2048, 4, 2, 1, False, 8, 8, 64, "leaky_relu", "gather_v1", V2, "wgrad_v1"),
]
_seen = {}


@pytest.mark.parametrize("case", HOT, ids=[c[0] for c in HOT])
def test_hot_shape_conv_fwd_dgrad_wgrad_vs_oracle(ops, case):
    name, cin, cout, k, s, pad, up, H, W, N, act, fam_fwd, fam_dgrad, fam_wgrad = case
    prec = ops.BF16
    torch.manual_seed(101 + [c[0] for c in HOT].index(name))
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    x = torch.randn(N, cin, H, W).bfloat16().float()
    w = (torch.randn(cout, cin, k, k) * math.sqrt(2.0 / (cin * k * k))).bfloat16().float()
    # ---- oracle, fp32 on the rounded operands ----
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = O.conv2d(O.upsample2x(xr) if up else xr, wr, stride=s, pad=pad, mode="reflect")
    if act == "leaky_relu":
        y_ref = O.leaky_relu(y_ref)
    gy = torch.randn(y_ref.shape).bfloat16().float()
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xr, wr], gy)
    y_ref = y_ref.detach()
    # ---- product path ----
    xg = ops.to_nhwc(x.to(DEV), prec).requires_grad_(True)          # NHWC bf16 leaf (3 channels are padded to 8)
    wg = w.to(DEV).requires_grad_(True)
    geom = ops.ConvGeom(cin, cout, k, s, pad, True, up)
    _counts()
    y = ops.conv2d(xg, wg, None, ops.PackedWeights(), geom, act)
    torch.cuda.synchronize()
    c_fwd = _counts()
    got_y = ops.to_nchw(y, cout)
    gyh = torch.zeros(N, y.shape[1], y.shape[2], prec.pad(cout), dtype=torch.bfloat16, device=DEV)
    gyh[..., :cout] = gy.to(DEV).permute(0, 2, 3, 1).to(torch.bfloat16)
    _counts()
    (dx,) = torch.autograd.grad(y, [xg], gyh, retain_graph=True)
    torch.cuda.synchronize()
    c_dgrad = _counts()
    (dw,) = torch.autograd.grad(y, [wg], gyh)
    torch.cuda.synchronize()
    c_wgrad = _counts()
    _seen[name] = {"fwd": c_fwd, "dgrad": c_dgrad, "wgrad": c_wgrad}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/hot_shape_families.json", "w") as f:
        json.dump(_seen, f, indent=1)

    # ---- numbers ----
    # forward: bf16-rounded output of an fp32-accumulated sum (measured ~1.2e-3 relative L2, the rounding alone)
    assert rel_l2(got_y, y_ref) < 3e-3 and maxrel(got_y, y_ref) < 1.5e-2, ("fwd", rel_l2(got_y, y_ref), maxrel(got_y, y_ref))
    got_dx = dx[..., :cin].permute(0, 3, 1, 2)
    if act == "none":
        assert rel_l2(got_dx, gx_ref) < 4e-3 and maxrel(got_dx, gx_ref) < 1.5e-2, ("dgrad", rel_l2(got_dx, gx_ref), maxrel(got_dx, gx_ref))
        assert rel_l2(dw, gw_ref) < 1e-3 and maxrel(dw, gw_ref) < 2e-3, ("wgrad", rel_l2(dw, gw_ref), maxrel(dw, gw_ref))
    else:
        # fused LeakyReLU: the mask is taken from the bf16-ROUNDED output, the oracle's from the fp32 pre-activation -- they
        # differ where |pre| < 2^-9 relative, a handful of full-magnitude outliers -> relative L2 only
        assert rel_l2(got_dx, gx_ref) < 1e-2, ("dgrad", rel_l2(got_dx, gx_ref))
        assert rel_l2(dw, gw_ref) < 1e-2, ("wgrad", rel_l2(dw, gw_ref))
    if prec.pad(cin) > cin:
        assert float(dx[..., cin:].abs().max()) == 0.0                  # padded input channels get no gradient

    # ---- which kernel ran ----
    assert c_fwd == {fam_fwd: 1}, ("forward was served by", c_fwd)
    assert c_dgrad == fam_dgrad, ("dgrad was served by", c_dgrad)
    assert c_wgrad == {fam_wgrad: 1}, ("wgrad was served by", c_wgrad)


# The small-M GEMMs of the step, all on the generic gather GEMM with split-K slabs + splitk_finalize (conv_gemm.hip: 128 x {32, 64,
# 128} tiles; fewer tiles than CUs -> the k range is split):  D's two heads on the 64-image batch (discriminator.py:78-98: src_clf
# 3x3 reflect 2048 -> 1, cls_clf 4x4 valid 2048 -> 6, both on the 4x4 map) and SPADE's table convs on the 5x5 border-class image
# (normalization.py:17-22 collapsed, DESIGN.md "SPADE collapse": shared 6 -> 128 and gamma|beta 128 -> 2C, 3x3 zero pad, bias; one
# 5x5 image per label set).
# (name, cin, cout, k, pad, reflect, H, W, N, bias, act, split-K expected)
SMALL_M = [
    ("D src_clf 2048->1 3x3 reflect @4^2 N=64", 2048, 1, 3, 1, True, 4, 4, 64, False, "none", True),
    ("D cls_clf 2048->6 4x4 valid @4^2 N=64", 2048, 6, 4, 0, False, 4, 4, 64, False, "none", True),
    ("SPADE shared 6->128 3x3 zero @5^2 N=32 +bias +ReLU", 6, 128, 3, 1, False, 5, 5, 32, True, "relu", False),
    ("SPADE gamma|beta 128->512 3x3 zero @5^2 N=32 +bias", 128, 512, 3, 1, False, 5, 5, 32, True, "none", True),
    ("SPADE gamma|beta 128->128 3x3 zero @5^2 N=32 +bias", 128, 128, 3, 1, False, 5, 5, 32, True, "none", True),
]


@pytest.mark.parametrize("case", SMALL_M, ids=[c[0] for c in SMALL_M])
def test_small_m_shapes_on_the_split_k_gemm_vs_oracle(ops, case):
    name, cin, cout, k, pad, reflect, H, W, N, has_bias, act, want_split = case
    prec = ops.BF16
    torch.manual_seed(301 + [c[0] for c in SMALL_M].index(name))
    x = torch.randn(N, cin, H, W).bfloat16().float()
    w = (torch.randn(cout, cin, k, k) * math.sqrt(2.0 / (cin * k * k))).bfloat16().float()
    b = torch.randn(cout) * 0.1 if has_bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = O.conv2d(xr, wr, stride=1, pad=pad, mode="reflect" if reflect else "zeros")
    if b is not None:
        y_ref = y_ref + b.view(1, -1, 1, 1)
    if act == "relu":
        y_ref = torch.relu(y_ref)
    gy = torch.randn(y_ref.shape).bfloat16().float()
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xr, wr], gy)
    xg = ops.to_nhwc(x.to(DEV), prec).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg_ = b.to(DEV) if b is not None else None
    geom = ops.ConvGeom(cin, cout, k, 1, pad, reflect, False)
    _counts()
    y = ops.conv2d(xg, wg, bg_, ops.PackedWeights(), geom, act)
    torch.cuda.synchronize()
    c_fwd = _counts(splitk=True)
    gyh = torch.zeros(N, y.shape[1], y.shape[2], prec.pad(cout), dtype=torch.bfloat16, device=DEV)
    gyh[..., :cout] = gy.to(DEV).permute(0, 2, 3, 1).to(torch.bfloat16)
    dx, dw = torch.autograd.grad(y, [xg, wg], gyh)
    torch.cuda.synchronize()
    c_bwd = _counts(splitk=True)
    got_y = ops.to_nchw(y, cout)
    assert got_y.shape == y_ref.shape
    assert rel_l2(got_y, y_ref.detach()) < 3e-3, ("fwd", rel_l2(got_y, y_ref.detach()))
    tol = 4e-3 if act == "none" else 1e-2           # fused ReLU: mask from the bf16-rounded output (see the hot-shape test)
    assert rel_l2(dx[..., :cin].permute(0, 3, 1, 2), gx_ref) < tol, ("dgrad", rel_l2(dx[..., :cin].permute(0, 3, 1, 2), gx_ref))
    assert rel_l2(dw, gw_ref) < (1e-3 if act == "none" else 1e-2), ("wgrad", rel_l2(dw, gw_ref))
    assert c_fwd.get("gather_v1") == 1 and (c_fwd.get("splitk_finalize", 0) == 1) == want_split, c_fwd
    assert set(c_bwd) <= {"gather_v1", "splitk_finalize", "wgrad_v1"}, c_bwd
    _seen[name] = {"fwd": c_fwd, "bwd": c_bwd}


COS_MIN, L2_MAX = {"D": 0.99, "G": 0.95}, {"D": 0.12, "G": 0.30}
C256 = dict(image_size=256, batch=16, num_layers=5, ngf=64, ndf=64, hidden_nc=128)


def _losses_and_grads(pname):
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    bg, labels, df = O.synthetic_batch(16, 256)
    torch.manual_seed(123)                       # the reference's init (N(0, 0.02)), same weights in both modes
    tr = DefectGanTrainer(make_opt(C256, DEV, pname))
    G, D = tr.model.netG, tr.model.netD
    with torch.no_grad():
        G.eval()
        D.eval()
        out, prob = G(bg.to(DEV), labels.to(DEV))
        src, cls = D(out)
    fwd = [t.float().cpu() for t in (out, prob, src, cls)]
    _counts()
    g1, c1 = tr.model("discriminator", bg, labels, df)
    (g1 + 2 * c1).backward()
    dgr = {k: p.grad.detach().double().cpu() for k, p in D.named_parameters()}
    for p in D.parameters():
        p.grad = None
    ls = tr.model("generator", bg, labels, df)
    (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
    ggr = {k: p.grad.detach().double().cpu() for k, p in G.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    fams = _counts()
    losses = [float(g1), float(c1)] + [float(v) for v in ls]
    del tr
    torch.cuda.empty_cache()
    return losses, fwd, dgr, ggr, fams


def test_bf16_step_tracks_f32_mode_at_256_batch_16():
    """The benchmarked configuration itself: 256x256, batch 16, ngf = ndf = 64, reference init, one D loss + backward and
    one G loss + backward in bf16 mode against the exact-f32 mode of this build (same seed, same inputs).

    Bounds and what was measured (gpurun_out/bf16_vs_f32_256x16.json, round 2):
      * 7 losses: 1e-3 relative (north_star's figure; measured 3e-5 .. 8e-4);
      * G(x), p, D_src(G(x)), D_cls(G(x)): 2e-2 relative L2 (measured 1.0e-2, 3.3e-3, 1.9e-2, 1.2e-2);
      * D gradients: every parameter cosine >= 0.99, relative L2 <= 0.12 (measured 0.9952 / 0.098 at the first conv -- the
        end of the backward chain -- rising to 1.0000 / 0.003 at the heads); full gradient cosine >= 0.999 (0.9994);
      * G gradients: full-gradient cosine >= 0.999 (measured 0.9997); per parameter cosine >= 0.95, relative L2 <= 0.30.
        Measured profile by depth from the output: heads 0.002 .. 0.017 relative L2, dec_blk.1 0.006, dec_blk.0 0.14 .. 0.17,
        the res blocks and the encoder 0.18 .. 0.26 (cosine 0.967 .. 0.984).  The step from 0.6 % to 15 % happens at the
        first InstanceNorm backward the gradient meets: the G loss has a constant-gradient term (sd_con = mean |p - 0|,
        defectgan_model.py:231-236), so the gradient entering SPADE's InstanceNorm is dominated by a per-channel
        constant that the norm's backward removes (dx = rstd * (g - mean(g) - xhat * mean(g * xhat))); bf16 rounds every
        stored gradient element relative to its FULL value (2^-9), which after the mean is subtracted is 2^-9 times the
        mean-to-fluctuation ratio of what is left.  That is a property of bf16 gradient storage (the op-level tests above
        pin every kernel to its own rounding); the exact-f32 mode is the one held to 1e-3 against the reference;
      * parameters whose gradient is STRUCTURALLY zero are judged by absolute size: the bias of a ResBlock's second
        BatchNorm shifts the block output by a per-channel constant, which reflect-padded convs carry to the next
        BatchNorm / InstanceNorm unchanged and that norm removes -- f32 mode leaves 1e-8-scale noise there, bf16 mode more;
        both must stay below 1e-3 of the net's full gradient norm."""
    f_loss, f_fwd, f_d, f_g, f_fam = _losses_and_grads("f32")
    b_loss, b_fwd, b_d, b_g, b_fam = _losses_and_grads("bf16")
    # f32 mode runs the generic exact-f32 GEMMs only; bf16 mode must have run the tuned families
    assert set(f_fam) <= {"gather_v1", "wgrad_v1"}, f_fam
    for fam in ("halo16_conv", "gather_v2", "thin_cin", "thin_cout", "wgrad_halo", "wgrad_v2", "wgrad_thin"):
        assert b_fam.get(fam, 0) > 0, (fam, b_fam)
    for a, b in zip(b_loss, f_loss):
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (b_loss, f_loss)
    for a, b, nm in zip(b_fwd, f_fwd, ("G(x)", "p", "D_src(G(x))", "D_cls(G(x))")):
        assert rel_l2(a, b) < 2e-2, (nm, rel_l2(a, b))
    report, bad = {}, []
    for net, fb, ff in (("D", b_d, f_d), ("G", b_g, f_g)):
        assert fb.keys() == ff.keys() and len(ff) > 5
        va = torch.cat([fb[k].flatten() for k in ff])
        vb = torch.cat([ff[k].flatten() for k in ff])
        full_cos = float(torch.dot(va, vb) / (va.norm() * vb.norm()))
        per = {}
        for k in ff:
            a, b = fb[k].flatten(), ff[k].flatten()
            cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
            l2 = float((a - b).norm() / b.norm().clamp_min(1e-300))
            per[k] = (round(cos, 5), round(l2, 5))
            negligible = float(b.norm()) <= 1e-3 * float(vb.norm()) and float(a.norm()) <= 1e-3 * float(vb.norm())
            if not negligible and not (cos >= COS_MIN[net] and l2 <= L2_MAX[net]):
                bad.append((net, k, cos, l2))
        report[net] = {"full_cos": full_cos, "worst_param_cos": min(v[0] for v in per.values()),
                       "worst_param_rel_l2": max(v[1] for v in per.values()), "per_param": per}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_vs_f32_256x16.json", "w") as f:
        json.dump({"losses_bf16": b_loss, "losses_f32": f_loss, "grad": report, "families_bf16": b_fam,
                   "fwd_rel_l2": [rel_l2(a, b) for a, b in zip(b_fwd, f_fwd)]}, f, indent=1)
    assert not bad, bad
    assert report["D"]["full_cos"] >= 0.999 and report["G"]["full_cos"] >= 0.999, (report["D"]["full_cos"], report["G"]["full_cos"])


def _place(arena, want_bit31, tensors):
    """Views into ``arena`` holding copies of ``tensors``, back to back from an address whose LOW 32 bits have bit 31 == want_bit31
    (and stay on that side of the 2 GiB line for all of them)."""
    base = arena.data_ptr()
    need = sum((t.numel() * t.element_size() + 4095) // 4096 * 4096 for t in tensors)
    assert need < (1 << 31)
    off = (-base) % 4096
    while True:
        lo, hi = (base + off) & 0xFFFFFFFF, (base + off + need - 1) & 0xFFFFFFFF
        if (lo >> 31) == want_bit31 and (hi >> 31) == want_bit31 and hi > lo:
            break
        off += ((1 << 31) - (lo & 0x7FFFFFFF))             # to the next 2 GiB line of the low word
        assert off + need <= arena.numel()
    out = []
    for t in tensors:
        nb = t.numel() * t.element_size()
        v = arena[off:off + nb].view(t.dtype).view(t.shape)
        v.copy_(t)
        assert ((v.data_ptr() & 0xFFFFFFFF) >> 31) == want_bit31
        out.append(v)
        off += (nb + 4095) // 4096 * 4096
    return out


def test_operand_addresses_on_both_sides_of_the_2gib_line():
    """Regression test for a GPU memory-access fault on record (round 2, development build of wgrad_halo's lean LDS-DMA issue
    path): ``glds16_asm_s`` (csrc/common.h) hands ``global_load_lds_dwordx4`` a wave-uniform 64-bit base in an SGPR pair, built
    from two ``readfirstlane``s of the pointer's halves -- the builtin returns *int*, and widening the low half as a signed value
    put 0xffffffff into the high half whenever the operand's low address word had bit 31 set (fault address 0xffff_b1400000).
    Whether a run hit it depended on where the allocator put the operand.  Here the operands of the kernels that use the helper
    (wgrad_halo: x and dy; thin_cout: x) and of their siblings (thin_cin) are placed, inside one 5 GiB arena, once at an address
    whose low word has bit 31 SET and once CLEAR: same bits out, and the oracle's numbers."""
    from de_i2i_gan_amd import ops
    prec = ops.BF16
    arena = torch.empty(5 << 30, dtype=torch.uint8, device=DEV)
    cases = [("res 256->256 3x3 @64^2 N=8", 256, 256, 3, 1, 64, 64, 8, {"wgrad_halo"}),
             ("heads 64->4 3x3 @256^2 N=4", 64, 4, 3, 1, 256, 256, 4, {"thin_cout", "thin_cin", "wgrad_halo"}),
             ("stem 3->64 7x7 @256^2 N=4", 3, 64, 7, 3, 256, 256, 4, {"thin_cin", "thin_cout", "wgrad_thin"})]
    for name, cin, cout, k, pad, H, W, N, fams in cases:
        torch.manual_seed(77)
        x = torch.randn(N, cin, H, W).bfloat16().float()
        w = (torch.randn(cout, cin, k, k) * math.sqrt(2.0 / (cin * k * k))).bfloat16().float()
        gy = torch.randn(N, cout, H, W).bfloat16().float()
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y_ref = O.conv2d(xr, wr, stride=1, pad=pad, mode="reflect")
        gx_ref, gw_ref = torch.autograd.grad(y_ref, [xr, wr], gy)
        x_h = ops.to_nhwc(x.to(DEV), prec)
        g_h = torch.zeros(N, H, W, prec.pad(cout), dtype=torch.bfloat16, device=DEV)
        g_h[..., :cout] = gy.to(DEV).permute(0, 2, 3, 1).to(torch.bfloat16)
        geom = ops.ConvGeom(cin, cout, k, 1, pad, True, False)
        res = {}
        for bit in (1, 0):
            xa, ga = _place(arena, bit, [x_h, g_h])
            xa = xa.detach().requires_grad_(True)
            wg = w.to(DEV).requires_grad_(True)
            _counts()
            y = ops.conv2d(xa, wg, None, ops.PackedWeights(), geom, "none")
            dx, dw = torch.autograd.grad(y, [xa, wg], ga)
            torch.cuda.synchronize()
            ran = _counts()
            assert fams <= set(ran), (name, bit, ran)
            res[bit] = (y.detach().clone(), dx.clone(), dw.clone())
        for a, b in zip(res[1], res[0]):
            assert torch.equal(a, b), name
        y, dx, dw = res[1]
        assert rel_l2(ops.to_nchw(y, cout), y_ref.detach()) < 3e-3, name
        assert rel_l2(dx[..., :cin].permute(0, 3, 1, 2), gx_ref) < 4e-3, name
        assert rel_l2(dw, gw_ref) < 1e-3, name


@pytest.mark.parametrize("cin,cout,H,W,N,reflect,act", [(128, 256, 128, 128, 16, True, "none"), (64, 128, 64, 128, 64, False, "leaky_relu"),
                                                        (256, 136, 64, 64, 64, True, "none"), (96, 128, 64, 128, 64, True, "leaky_relu")])
def test_stride2_halo_form_on_other_channel_counts_vs_gather_gemm_and_oracle(ops, cin, cout, H, W, N, reflect, act):
    """The stride-2 form of the 16 x 32 tile kernel (conv_halo16.hip S2) is dispatched for 64-channel inputs only (where it is the
    faster kernel); option halo16_s2 = 2 sends every shape it can take to it: 128 / 256 / 96 input channels (several channel blocks per
    parity plane, two output-channel tiles, a ragged last channel tile), zero padding, non-square images -- against the oracle's conv and
    bit-for-bit statistics against the stored output."""
    from de_i2i_gan_amd import _lib
    lib = _lib.load()
    torch.manual_seed(41)
    x = torch.randn(N, cin, H, W).bfloat16().float()
    w = (torch.randn(cout, cin, 4, 4) * math.sqrt(2.0 / (cin * 16))).bfloat16().float()
    y_ref = O.conv2d(x, w, stride=2, pad=1, mode="reflect" if reflect else "zeros")
    if act == "leaky_relu":
        y_ref = O.leaky_relu(y_ref)
    xg = ops.to_nhwc(x.to(DEV), ops.BF16)
    geom = ops.ConvGeom(cin, cout, 4, 2, 1, reflect, False)
    _lib.check(lib.dei2i_set_option(b"halo16_s2", 2), "set_option")
    try:
        _counts()
        y = ops.conv2d(xg, w.to(DEV), None, ops.PackedWeights(), geom, act, stats=True)
        torch.cuda.synchronize()
        fam = _counts()
    finally:
        _lib.check(lib.dei2i_set_option(b"halo16_s2", 1), "set_option")
    assert fam == {"halo16_s2": 1}, fam
    got = ops.to_nchw(y, cout)
    assert rel_l2(got, y_ref) < 3e-3 and maxrel(got, y_ref) < 1.5e-2, (rel_l2(got, y_ref), maxrel(got, y_ref))
    if ops.BF16.pad(cout) > cout:
        assert float(y[..., cout:].abs().max()) == 0.0
    partial, chunks = y._dei2i_stats                       # the epilogue's statistics records of the stored (rounded) output
    yf = y.float()
    assert maxrel(partial.double().sum(dim=1)[:, 0], yf.sum(dim=(1, 2)).double()) < 1e-4
    assert maxrel(partial.double().sum(dim=1)[:, 1], (yf * yf).sum(dim=(1, 2)).double()) < 1e-5
