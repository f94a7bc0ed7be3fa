"""GPU checks of the fp8 (OCP e4m3) forward mode -- BASELINE.json configs[4], `compute_dtype="fp8"`.

What the mode is: the FORWARD GEMM of the stride-1 3x3 convolutions the halo kernel takes runs on e4m3 operands
(v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation); tensors in HBM, the backward pass and every other kernel are those of
the bf16 mode.  The reference has no fp8 path, so the checker is the bf16 mode of this build (itself pinned to the
reference by test_model_gpu.py) -- "parity unpinned by the reference", tolerance vs bf16 stated here:

  * quantiser: bit-exact against torch's float8_e4m3fn conversion (round-to-nearest-even, saturating at +-448);
  * one convolution: e4m3 has 3 mantissa bits (relative step 2^-3, rms rounding error 3.6 % per operand element); a
    K = 9*Cin term dot product of independently rounded operands averages that down: measured 3-4 % of the output rms
    against the bf16 kernel on the same bf16 inputs; bound 6 %;
  * the train step (256x256, default widths, reference init): step-1 losses within 3 % of the bf16 mode, full-gradient
    cosine > 0.97, two steps finite."""
from ctypes import byref

import numpy as np
import pytest
import torch

from helpers import make_opt
from oracle import defectgan_oracle as O          # synthetic_batch only

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from de_i2i_gan_amd import ops as o
    return o


def test_quantizer_matches_torch_e4m3fn(ops):
    from de_i2i_gan_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(4096, generator=g) * 4.0, torch.tensor([0.0, -0.0, 447.0, 448.0, 460.0, 1e4, -1e4, 2.0 ** -9,
                                                                        2.0 ** -10, 0.0019, 1.0625, 1.1875, 30.0]),
                   torch.zeros(3)]).to(torch.bfloat16)
    assert x.numel() % 8 == 0
    xd = x.to(DEV)
    out = torch.empty(x.numel(), dtype=torch.uint8, device=DEV)
    for scale in (1.0, 16.0):
        L.check(lib.dei2i_quantize_fp8(x.numel(), ops._p(xd), scale, ops._p(out), ops._stream()), "quantize")
        ref = (x.float() * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        got = out.cpu()
        # -0.0 and +0.0 may differ in the sign bit only
        same = (got == ref) | (((got | ref) & 0x7F) == 0)
        assert bool(same.all()), (got[~same][:8], ref[~same][:8], x[~same][:8])


@pytest.mark.parametrize("cin,cout,h,n,reflect,up,act", [
    (256, 256, 64, 8, True, False, "none"),          # res-block conv
    (256, 128, 64, 4, True, True, "none"),           # decoder conv behind the fused x2 upsample
    (128, 64, 128, 4, True, False, "leaky_relu"),    # BN = 64 variant, fused activation
    (128, 192, 32, 16, False, False, "relu"),        # zero padding, Cout not a multiple of 128
])
def test_conv_forward_fp8_tracks_bf16(ops, cin, cout, h, n, reflect, up, act):
    torch.manual_seed(5)
    geom = ops.ConvGeom(cin, cout, 3, 1, 1, reflect, up)
    x = torch.randn(n, h, h, cin, device=DEV).relu_().to(torch.bfloat16)          # post-activation-like input
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.02
    b = torch.randn(cout, device=DEV) * 0.1
    with torch.no_grad():
        ref = ops.conv2d(x, w, b, ops.PackedWeights(), geom, act)
        with ops.fp8_forward(True):
            got = ops.conv2d(x, w, b, ops.PackedWeights(), geom, act)
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    err = (got.float() - ref.float()).pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt()
    assert 1e-4 < float(err) < 0.06, float(err)      # > 0: the fp8 kernel really ran; < 6 %: see module docstring


def test_fused_e4m3_copies_of_the_normalisation_kernels_equal_the_quantiser(ops):
    """In the fp8 scope the BatchNorm-act and SPADE-act kernels write the e4m3 operand copy in the same pass; it must be
    byte-identical to quantising their bf16 output (so a conv gives the same result whichever way its operand came)."""
    from de_i2i_gan_amd import _lib as L
    lib = L.load()
    torch.manual_seed(9)
    n, h, c = 2, 32, 128
    y = torch.randn(n, h, h, c, device=DEV).to(torch.bfloat16)
    w, b = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    gb = (torch.randn(n, 5, 5, 2 * c, device=DEV) * 0.3).to(torch.bfloat16)
    with torch.no_grad(), ops.fp8_forward(True):
        outs = [ops.batchnorm_act(y, w, b, rm, rv, True, "leaky_relu"), ops.spade_relu(y, gb, True, 1)]
    with torch.no_grad():
        plain = ops.batchnorm_act(y, w, b, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), True, "leaky_relu")
    assert not hasattr(plain, "_dei2i_fp8") and torch.equal(plain, outs[0])
    for out in outs:
        q = out._dei2i_fp8
        ref = torch.empty(out.numel(), dtype=torch.uint8, device=DEV)
        L.check(lib.dei2i_quantize_fp8(out.numel(), ops._p(out), ops.FP8_ACT_SCALE, ops._p(ref), ops._stream()), "quantize")
        assert q.dtype == torch.uint8 and q.numel() == out.numel() and torch.equal(q, ref)


def test_unsupported_shapes_stay_bf16_and_fp8_call_refuses_them(ops):
    from de_i2i_gan_amd import _lib as L
    lib = L.load()
    geom = ops.ConvGeom(64, 64, 3, 1, 1, True, False)          # 64 input channels: not a 128-byte e4m3 slice
    d = ops._desc(ops.BF16, geom, 4, 64, 64, 64, 64)
    assert lib.dei2i_conv2d_fp8_supported(byref(d)) == 0
    x = torch.randn(4, 64, 64, 64, device=DEV).to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, device=DEV) * 0.02
    with torch.no_grad():
        ref = ops.conv2d(x, w, None, ops.PackedWeights(), geom, "none")
        with ops.fp8_forward(True):
            got = ops.conv2d(x, w, None, ops.PackedWeights(), geom, "none")
    assert torch.equal(ref, got)                                 # the bf16 kernel, bit for bit
    y = torch.empty_like(ref)
    dq = torch.ones(1, device=DEV)
    rc = lib.dei2i_conv2d_fwd_fp8(byref(d), ops._p(x), ops._p(w), None, ops._p(dq), 0, ops._p(y), ops._stream())
    assert rc != 0                                               # refused: there is no fallback inside the fp8 entry point


def test_train_step_fp8_tracks_bf16_at_256():
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=256, batch=2, num_layers=5, ngf=64, ndf=64, hidden_nc=128)
    bg, labels, df = O.synthetic_batch(2, 256)
    res = {}
    for pname in ("bf16", "fp8"):
        torch.manual_seed(123)
        tr = DefectGanTrainer(make_opt(c, DEV, pname))
        G, D = tr.model.netG, tr.model.netD
        assert G.fp8 == (pname == "fp8")
        g1, c1 = tr.model("discriminator", bg, labels, df)
        (g1 + 2 * c1).backward()
        dgr = torch.cat([p.grad.double().flatten() for p in D.parameters()])
        ls = tr.model("generator", bg, labels, df)
        (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
        ggr = torch.cat([p.grad.double().flatten() for p in G.parameters() if p.grad is not None])
        res[pname] = (np.array([float(g1), float(c1)] + [float(v) for v in ls]), dgr, ggr)
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        tr.step(bg, labels, df)
        tr.step(bg, labels, df)
        L = tr.losses
        last = [L["gan"]["D"][-1], L["clf"]["D"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1], L["aux"]["rec"][-1]]
        assert np.isfinite(last).all()
    a, b = res["bf16"], res["fp8"]
    assert np.max(np.abs(b[0] - a[0]) / np.abs(a[0])) < 0.03, (a[0], b[0])
    assert not np.array_equal(a[0], b[0])                        # the fp8 path really ran
    for i in (1, 2):
        cos = float(torch.dot(a[i], b[i]) / (a[i].norm() * b[i].norm()))
        assert cos > 0.97, (i, cos)
