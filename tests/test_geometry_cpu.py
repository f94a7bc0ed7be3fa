"""CPU check of the product's conv GEOMETRY (de-i2i-gan_amd/csrc/geom.h): the same descriptor builders and index
maps the HIP kernels use are executed with plain loops (tests/hostcheck/hostcheck.cpp, built with g++) and compared
with torch's conv forward / autograd on every conv shape family of the reference's G and D."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
SO = HERE / "hostcheck" / "libhostcheck.so"


@pytest.fixture(scope="module")
def hc():
    src = HERE / "hostcheck" / "hostcheck.cpp"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(SO), str(src)])
    return ctypes.CDLL(str(SO))


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def nhwc(t, cs):
    n, c, h, w = t.shape
    out = np.zeros((n, h, w, cs), np.float32)
    out[..., :c] = t.permute(0, 2, 3, 1).numpy()
    return out


# (Cin, Cout, k, stride, pad, mode, up, H)   mode: 0 zero, 1 reflect
CASES = [
    (3, 8, 7, 1, 3, 1, 0, 12),     # stem 7x7 reflect (generator.py:67-73)
    (8, 12, 4, 2, 1, 1, 0, 12),    # encoder / D 4x4 s2 reflect (generator.py:107-116, discriminator.py:60-76)
    (8, 8, 3, 1, 1, 1, 0, 8),      # res / decoder 3x3 reflect
    (8, 4, 3, 1, 1, 1, 1, 6),      # NormConvBlock: upsample fused into the conv (architecture.py:203,241-245)
    (6, 8, 3, 1, 1, 0, 0, 7),      # SPADE 3x3 zero pad (normalization.py:17-22)
    (8, 6, 4, 1, 0, 0, 0, 4),      # cls_clf full-extent valid conv (discriminator.py:78-83)
    (8, 1, 3, 1, 1, 1, 0, 4),      # src_clf on a 4x4 map
    (4, 8, 4, 2, 1, 1, 0, 4),      # deepest D layer: 4x4 -> 2x2
]


@pytest.mark.parametrize("case", CASES)
def test_conv_geometry(hc, case):
    cin, cout, k, s, pad, mode, up, H = case
    torch.manual_seed(sum(case))
    N, W = 2, H + 2 if (s == 1 and k != H) else H
    if s == 2:
        W = H
    cins, couts = -(-cin // 4) * 4, -(-cout // 4) * 4
    x = torch.randn(N, cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    xl = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x
    if mode == 1 and pad:
        y = F.conv2d(F.pad(xl, (pad,) * 4, mode="reflect"), w, stride=s)
    else:
        y = F.conv2d(xl, w, stride=s, padding=pad)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, [x, w], gy)

    p = np.array([N, H, W, cin, cout, cins, couts, k, k, s, pad, mode, up], dtype=np.int32)
    dims = np.zeros(4, np.int32)
    hc.hc_out_shape(ptr(p), ptr(dims))
    assert (dims[0], dims[1]) == (y.shape[2], y.shape[3])

    wp = np.zeros(cout * k * k * cins, np.float32)
    hc.hc_pack_fwd(ptr(p), ptr(np.ascontiguousarray(w.detach().float().numpy())), ptr(wp))
    xn = nhwc(x.detach().float(), cins)
    yn = np.zeros((N, dims[0], dims[1], couts), np.float32)
    hc.hc_conv_fwd(ptr(p), ptr(xn), ptr(wp), ptr(yn))
    np.testing.assert_allclose(yn[..., :cout], y.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)

    # dgrad: extended-frame output, then fold
    wd = np.zeros(cin * 4 * k * k * couts, np.float32)
    hc.hc_pack_dgrad.restype = ctypes.c_longlong
    n_wd = hc.hc_pack_dgrad(ptr(p), ptr(np.ascontiguousarray(w.detach().float().numpy())), ptr(wd))
    assert n_wd == cin * k * k * couts
    gyn = nhwc(gy.float(), couts)
    ext = np.full((N, dims[2], dims[3], cins), np.nan, np.float32)      # every element must be written
    hc.hc_conv_dgrad(ptr(p), ptr(gyn), ptr(wd), ptr(ext))
    assert not np.isnan(ext[..., :cin]).any()
    dx = np.zeros((N, H, W, cins), np.float32)
    ext0 = np.nan_to_num(ext)
    hc.hc_fold(N, H, W, cins, pad, int(mode == 1 and pad > 0), up, ptr(ext0), ptr(dx))
    np.testing.assert_allclose(dx[..., :cin], gx.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)

    # decomposed grad_input of stride-1 reflect convs: interior GEMM straight into dx + reflect-ring GEMM + border fold
    if mode == 1 and pad > 0 and s == 1 and not up and H >= 2 * pad + 2 and W >= 2 * pad + 2:
        ext2 = np.full((N, dims[2], dims[3], cins), np.nan, np.float32)
        dx2 = np.full((N, H, W, cins), np.nan, np.float32)                 # every element must be written
        hc.hc_conv_dgrad_decomposed.restype = ctypes.c_longlong
        written = hc.hc_conv_dgrad_decomposed(ptr(p), ptr(gyn), ptr(wd), ptr(ext2), ptr(dx2))
        ring = np.ones((dims[2], dims[3]), bool)
        ring[pad:pad + H, pad:pad + W] = False
        assert written == N * int(ring.sum()) * cins
        assert np.isnan(ext2[:, ~ring]).all() and not np.isnan(ext2[:, ring]).any()      # the ring and only the ring
        np.testing.assert_allclose(dx2[..., :cin], gx.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)

    # ... and of the 4x4 stride-2 pad-1 reflect convs, per parity class
    if mode == 1 and pad == 1 and s == 2 and k == 4 and not up and H >= 8 and H % 2 == 0 and W % 2 == 0:
        ext2 = np.full((N, dims[2], dims[3], cins), np.nan, np.float32)
        dx2 = np.full((N, H, W, cins), np.nan, np.float32)
        hc.hc_conv_dgrad_decomposed_s2.restype = ctypes.c_longlong
        written = hc.hc_conv_dgrad_decomposed_s2(ptr(p), ptr(gyn), ptr(wd), ptr(ext2), ptr(dx2))
        ring = np.ones((dims[2], dims[3]), bool)
        ring[1:1 + H, 1:1 + W] = False
        assert written == N * int(ring.sum()) * cins
        assert np.isnan(ext2[:, ~ring]).all() and not np.isnan(ext2[:, ring]).any()
        np.testing.assert_allclose(dx2[..., :cin], gx.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)

    # wgrad
    dwp = np.zeros(cout * k * k * cins, np.float32)
    hc.hc_conv_wgrad(ptr(p), ptr(xn), ptr(gyn), ptr(dwp))
    dw = dwp.reshape(cout, k * k, cins)[:, :, :cin].transpose(0, 2, 1).reshape(cout, cin, k, k)
    np.testing.assert_allclose(dw, gw.numpy(), rtol=1e-4, atol=1e-4)


def test_fastdiv(hc):
    assert hc.hc_fastdiv_selftest() == 0
