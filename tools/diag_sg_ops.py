"""Component check of the ops the stargan-v2 generator adds (in_affine_act, upsample2, scale, leaky_relu, conv with fused upsample and
bias) against torch autograd, f32 mode."""
import sys
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from de_i2i_gan_amd import ops

DEV = "cuda:0"
torch.manual_seed(0)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


N, C, H = 2, 64, 16
x = torch.randn(N, C, H, H, device=DEV)
gam, bet = torch.randn(N, C, device=DEV) * 0.3, torch.randn(N, C, device=DEV) * 0.3
gy = torch.randn(N, C, H, H, device=DEV)
# ---- in_affine_act
xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
ref = F.leaky_relu((1 + gr.view(N, C, 1, 1)) * F.instance_norm(xr, eps=1e-5) + br.view(N, C, 1, 1), 0.2)
rg = torch.autograd.grad(ref, [xr, gr, br], gy)
xh, gh, bh = nhwc(x).requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
out = ops.in_affine_act(xh, gh, bh, "leaky_relu")
og = torch.autograd.grad(out, [xh, gh, bh], nhwc(gy))
print("in_affine_act fwd", rel(out.permute(0, 3, 1, 2), ref), "dx", rel(og[0].permute(0, 3, 1, 2), rg[0]), "dgamma", rel(og[1], rg[1]), "dbeta", rel(og[2], rg[2]))
# ---- upsample2 / scale / leaky
xr = x.clone().requires_grad_(True)
ref = F.interpolate(xr, scale_factor=2, mode="nearest") * 0.7
gy2 = torch.randn_like(ref)
rg = torch.autograd.grad(ref, [xr], gy2)
xh = nhwc(x).requires_grad_(True)
out = ops.scale(ops.upsample2(xh), 0.7)
og = torch.autograd.grad(out, [xh], nhwc(gy2))
print("upsample2*scale fwd", rel(out.permute(0, 3, 1, 2), ref), "dx", rel(og[0].permute(0, 3, 1, 2), rg[0]))
# ---- conv 3x3 zero pad + bias with fused upsample; conv1x1 with upsample
for k, pad in ((3, 1), (1, 0)):
    w = torch.randn(32, C, k, k, device=DEV) * 0.05
    b = torch.randn(32, device=DEV) * 0.1 if k == 3 else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, b, padding=pad)
    gy3 = torch.randn_like(ref)
    rg = torch.autograd.grad(ref, [xr, wr], gy3)
    xh, wh = nhwc(x).requires_grad_(True), w.clone().requires_grad_(True)
    out = ops.conv2d(xh, wh, b, ops.PackedWeights(), ops.ConvGeom(C, 32, k, 1, pad, False, True), "none")
    og = torch.autograd.grad(out, [xh, wh], nhwc(gy3))
    print(f"conv{k}x{k} up fwd", rel(out[..., :32].permute(0, 3, 1, 2), ref), "dx", rel(og[0].permute(0, 3, 1, 2), rg[0]), "dw", rel(og[1], rg[1]))
