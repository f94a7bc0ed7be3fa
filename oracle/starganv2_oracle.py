"""CPU restatement of the reference's stargan-v2 G/D train step -- TEST INFRASTRUCTURE ONLY.

Only ``tests/`` (and the fixture generator ``tests/golden/gen_starganv2_golden.py``) may import this module; the product never does.
Functional style like ``defectgan_oracle``: networks are state dicts with the reference modules' ``state_dict()`` keys, every function
cites the reference lines it restates (``/root/reference/stargan-v2/core/model.py`` = M, ``core/solver.py`` = S).  Pinned by
``tests/golden/gen_starganv2_golden.py``, which imports the reference's own ``core.model`` / ``core.solver`` (the absent ``munch``,
``torchvision``, ``cv2``, ``skimage``, ``ffmpeg`` stubbed: none of them does arithmetic on this path) and asserts oracle == reference on
the forward passes, the four loss graphs (incl. the R1 penalty's double backward), the gradients, the Adam(0, 0.99, weight decay) updates
and the EMA; the fixture ``tests/golden/sg0_*`` holds the reference's numbers.

Scope: ``--norm_type adain`` (the reference's default), ``--w_hpf 0`` (every documented AFHQ command; w_hpf > 0 needs the FAN landmark
network and its downloaded weights), no DiffAugment."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


@dataclass
class Cfg:
    img_size: int = 64
    style_dim: int = 16
    latent_dim: int = 8
    num_domains: int = 2
    max_conv_dim: int = 64
    lambda_reg: float = 1.0
    lambda_sty: float = 1.0
    lambda_ds: float = 1.0
    lambda_cyc: float = 1.0
    lr: float = 1e-4
    f_lr: float = 1e-6
    beta1: float = 0.0
    beta2: float = 0.99
    weight_decay: float = 1e-4
    ema_beta: float = 0.999


# --------------------------------------------------------------------------- #
# state shapes (keys and shapes of the reference modules' state_dict()) -- M:26-46, 69-101, 321-363, 442-505, 508-524
# --------------------------------------------------------------------------- #
def _resblk_shapes(p: str, din: int, dout: int, normalize: bool) -> Dict[str, tuple]:
    s = {p + "conv1.weight": (din, din, 3, 3), p + "conv1.bias": (din,), p + "conv2.weight": (dout, din, 3, 3), p + "conv2.bias": (dout,)}
    if normalize:                                          # nn.InstanceNorm2d(affine=True): weight, bias (no running stats)
        s.update({p + "norm1.weight": (din,), p + "norm1.bias": (din,), p + "norm2.weight": (din,), p + "norm2.bias": (din,)})
    if din != dout:
        s[p + "conv1x1.weight"] = (dout, din, 1, 1)
    return s


def _adain_resblk_shapes(p: str, din: int, dout: int, style_dim: int) -> Dict[str, tuple]:
    s = {p + "conv1.weight": (dout, din, 3, 3), p + "conv1.bias": (dout,), p + "conv2.weight": (dout, dout, 3, 3), p + "conv2.bias": (dout,),
         p + "norm1.fc.weight": (2 * din, style_dim), p + "norm1.fc.bias": (2 * din,),
         p + "norm2.fc.weight": (2 * dout, style_dim), p + "norm2.fc.bias": (2 * dout,)}
    if din != dout:
        s[p + "conv1x1.weight"] = (dout, din, 1, 1)
    return s


def generator_plan(cfg: Cfg) -> Tuple[int, List[Tuple[int, int, bool]], List[Tuple[int, int, bool]]]:
    """-> (dim_in, encode [(din, dout, downsample)], decode [(din, dout, upsample)]) -- M:321-363 with w_hpf = 0"""
    dim_in = 2 ** 14 // cfg.img_size
    d0 = dim_in
    enc, dec = [], []
    for _ in range(int(math.log2(cfg.img_size)) - 4):
        dout = min(dim_in * 2, cfg.max_conv_dim)
        enc.append((dim_in, dout, True))
        dec.insert(0, (dout, dim_in, True))
        dim_in = dout
    for _ in range(2):
        enc.append((dim_in, dim_in, False))
        dec.insert(0, (dim_in, dim_in, False))
    return d0, enc, dec


def generator_state_shapes(cfg: Cfg) -> Dict[str, tuple]:
    d0, enc, dec = generator_plan(cfg)
    s = {"from_rgb.weight": (d0, 3, 3, 3), "from_rgb.bias": (d0,)}
    tail = {"to_rgb.0.weight": (d0,), "to_rgb.0.bias": (d0,), "to_rgb.2.weight": (3, d0, 1, 1), "to_rgb.2.bias": (3,)}
    blocks = {}
    for i, (a, b, _) in enumerate(enc):
        blocks.update(_resblk_shapes(f"encode.{i}.", a, b, True))
    for i, (a, b, _) in enumerate(dec):
        blocks.update(_adain_resblk_shapes(f"decode.{i}.", a, b, cfg.style_dim))
    # nn.Module registration order: from_rgb, encode, decode, to_rgb (M:326-334)
    s.update(blocks)
    s.update(tail)
    return s


def mapping_state_shapes(cfg: Cfg) -> Dict[str, tuple]:
    s = {"shared.0.weight": (512, cfg.latent_dim), "shared.0.bias": (512,)}
    for i in (2, 4, 6):
        s.update({f"shared.{i}.weight": (512, 512), f"shared.{i}.bias": (512,)})
    for d in range(cfg.num_domains):
        for i in (0, 2, 4):
            s.update({f"unshared.{d}.{i}.weight": (512, 512), f"unshared.{d}.{i}.bias": (512,)})
        s.update({f"unshared.{d}.6.weight": (cfg.style_dim, 512), f"unshared.{d}.6.bias": (cfg.style_dim,)})
    return s


def _trunk_plan(cfg: Cfg) -> Tuple[int, List[Tuple[int, int]]]:
    dim_in = 2 ** 14 // cfg.img_size
    d0, blocks = dim_in, []
    for _ in range(int(math.log2(cfg.img_size)) - 2):
        dout = min(dim_in * 2, cfg.max_conv_dim)
        blocks.append((dim_in, dout))
        dim_in = dout
    return d0, blocks


def style_encoder_state_shapes(cfg: Cfg) -> Dict[str, tuple]:
    d0, blocks = _trunk_plan(cfg)
    s = {"shared.0.weight": (d0, 3, 3, 3), "shared.0.bias": (d0,)}
    for i, (a, b) in enumerate(blocks):
        s.update(_resblk_shapes(f"shared.{i + 1}.", a, b, False))
    n, dl = len(blocks), blocks[-1][1]
    s.update({f"shared.{n + 2}.weight": (dl, dl, 4, 4), f"shared.{n + 2}.bias": (dl,)})
    for d in range(cfg.num_domains):
        s.update({f"unshared.{d}.weight": (cfg.style_dim, dl), f"unshared.{d}.bias": (cfg.style_dim,)})
    return s


def discriminator_state_shapes(cfg: Cfg) -> Dict[str, tuple]:
    d0, blocks = _trunk_plan(cfg)
    s = {"main.0.weight": (d0, 3, 3, 3), "main.0.bias": (d0,)}
    for i, (a, b) in enumerate(blocks):
        s.update(_resblk_shapes(f"main.{i + 1}.", a, b, False))
    n, dl = len(blocks), blocks[-1][1]
    s.update({f"main.{n + 2}.weight": (dl, dl, 4, 4), f"main.{n + 2}.bias": (dl,),
              f"main.{n + 4}.weight": (cfg.num_domains, dl, 1, 1), f"main.{n + 4}.bias": (cfg.num_domains,)})
    return s


def formula_tensor(key: str, shape: tuple) -> Tensor:
    """RNG-free fill (the defectGAN oracle's scheme, its own phase per key): He-like gain for weights, small biases, IN weights ~ 1."""
    n = 1
    for d in shape:
        n *= d
    h = sum((i + 1) * ord(c) for i, c in enumerate(key)) % 9973
    idx = torch.arange(n, dtype=torch.float64)
    base = torch.sin(idx * 0.7391 + h * 0.011) * 0.6 + torch.cos(idx * 0.2113 + h * 0.07) * 0.4
    if len(shape) >= 2:
        fan_in = n // shape[0]
        v = base * math.sqrt(2.0 / fan_in)
    elif key.endswith("weight"):                           # InstanceNorm2d(affine=True) weight
        v = 1.0 + 0.1 * base
    else:
        v = 0.05 * base
    return v.reshape(shape).float()


def make_state(shapes: Dict[str, tuple], prefix: str = "") -> State:
    return {k: formula_tensor(prefix + k, s) for k, s in shapes.items()}


def synthetic_inputs(cfg: Cfg, batch: int):
    """x_real, x_ref, x_ref2 ~ formula images in [-1, 1]; y_org / y_trg domain labels; z_trg, z_trg2 latent codes (S:262-266)"""
    def img(tag):
        return torch.tanh(formula_tensor(tag, (batch, 3, cfg.img_size, cfg.img_size)) * 40.0)
    y_org = torch.arange(batch) % cfg.num_domains
    y_trg = (torch.arange(batch) + 1) % cfg.num_domains
    z = formula_tensor("z_trg", (batch, cfg.latent_dim)) * 3.0
    z2 = formula_tensor("z_trg2", (batch, cfg.latent_dim)) * 3.0
    return img("x_real"), y_org, y_trg, img("x_ref"), img("x_ref2"), z, z2


# --------------------------------------------------------------------------- #
# blocks
# --------------------------------------------------------------------------- #
def lrelu(x: Tensor) -> Tensor:
    return F.leaky_relu(x, 0.2)


def instance_norm_affine(S: State, p: str, x: Tensor) -> Tensor:
    """nn.InstanceNorm2d(C, affine=True): eps 1e-5, biased variance per (n, c), no running stats"""
    return F.instance_norm(x, weight=S[p + "weight"], bias=S[p + "bias"], eps=1e-5)


def resblk(S: State, p: str, x: Tensor, normalize: bool, downsample: bool) -> Tensor:
    """ResBlk -- M:26-67: (shortcut + residual) / sqrt(2); shortcut = [conv1x1] -> [avg_pool2d 2]; residual = [IN] -> LReLU ->
    conv1 -> [avg_pool2d 2] -> [IN] -> LReLU -> conv2 (3x3, zero padding 1, bias)"""
    sc = x
    if p + "conv1x1.weight" in S:
        sc = F.conv2d(sc, S[p + "conv1x1.weight"])
    if downsample:
        sc = F.avg_pool2d(sc, 2)
    h = x
    if normalize:
        h = instance_norm_affine(S, p + "norm1.", h)
    h = F.conv2d(lrelu(h), S[p + "conv1.weight"], S[p + "conv1.bias"], padding=1)
    if downsample:
        h = F.avg_pool2d(h, 2)
    if normalize:
        h = instance_norm_affine(S, p + "norm2.", h)
    h = F.conv2d(lrelu(h), S[p + "conv2.weight"], S[p + "conv2.bias"], padding=1)
    return (sc + h) / math.sqrt(2)


def adain(S: State, p: str, x: Tensor, s: Tensor) -> Tensor:
    """AdaIN -- M:69-80: (1 + gamma) * IN(x) + beta, (gamma | beta) = fc(s)"""
    h = F.linear(s, S[p + "fc.weight"], S[p + "fc.bias"])
    gamma, beta = torch.chunk(h.view(h.size(0), h.size(1), 1, 1), 2, dim=1)
    return (1 + gamma) * F.instance_norm(x, eps=1e-5) + beta


def adain_resblk(S: State, p: str, x: Tensor, s: Tensor, upsample: bool) -> Tensor:
    """AdainResBlk -- M:83-123 with w_hpf = 0: (residual + shortcut) / sqrt(2)"""
    sc = x
    if upsample:
        sc = F.interpolate(sc, scale_factor=2, mode="nearest")
    if p + "conv1x1.weight" in S:
        sc = F.conv2d(sc, S[p + "conv1x1.weight"])
    h = lrelu(adain(S, p + "norm1.", x, s))
    if upsample:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    h = F.conv2d(h, S[p + "conv1.weight"], S[p + "conv1.bias"], padding=1)
    h = F.conv2d(lrelu(adain(S, p + "norm2.", h, s)), S[p + "conv2.weight"], S[p + "conv2.bias"], padding=1)
    return (h + sc) / math.sqrt(2)


def generator(S: State, x: Tensor, s: Tensor, cfg: Cfg) -> Tensor:
    """Generator.forward -- M:365-382 (masks None): from_rgb -> encode -> decode(s) -> to_rgb (IN affine, LReLU, conv 1x1)"""
    _, enc, dec = generator_plan(cfg)
    h = F.conv2d(x, S["from_rgb.weight"], S["from_rgb.bias"], padding=1)
    for i, (_, _, down) in enumerate(enc):
        h = resblk(S, f"encode.{i}.", h, True, down)
    for i, (_, _, up) in enumerate(dec):
        h = adain_resblk(S, f"decode.{i}.", h, s, up)
    h = lrelu(instance_norm_affine(S, "to_rgb.0.", h))
    return F.conv2d(h, S["to_rgb.2.weight"], S["to_rgb.2.bias"])


def mapping_network(S: State, z: Tensor, y: Tensor, cfg: Cfg) -> Tensor:
    """MappingNetwork.forward -- M:442-471"""
    h = z
    for i in (0, 2, 4, 6):
        h = F.relu(F.linear(h, S[f"shared.{i}.weight"], S[f"shared.{i}.bias"]))
    outs = []
    for d in range(cfg.num_domains):
        o = h
        for i in (0, 2, 4):
            o = F.relu(F.linear(o, S[f"unshared.{d}.{i}.weight"], S[f"unshared.{d}.{i}.bias"]))
        outs.append(F.linear(o, S[f"unshared.{d}.6.weight"], S[f"unshared.{d}.6.bias"]))
    out = torch.stack(outs, dim=1)
    return out[torch.arange(y.size(0)), y]


def _trunk(S: State, p: str, x: Tensor, cfg: Cfg) -> Tensor:
    """the shared trunk of StyleEncoder / Discriminator -- M:474-492, 508-524: conv3x3 -> ResBlk(downsample) x (log2(size) - 2) ->
    LReLU -> conv4x4 valid -> LReLU"""
    _, blocks = _trunk_plan(cfg)
    h = F.conv2d(x, S[p + "0.weight"], S[p + "0.bias"], padding=1)
    for i in range(len(blocks)):
        h = resblk(S, f"{p}{i + 1}.", h, False, True)
    n = len(blocks)
    h = F.conv2d(lrelu(h), S[f"{p}{n + 2}.weight"], S[f"{p}{n + 2}.bias"])
    return lrelu(h)


def style_encoder(S: State, x: Tensor, y: Tensor, cfg: Cfg) -> Tensor:
    """StyleEncoder.forward -- M:494-505"""
    h = _trunk(S, "shared.", x, cfg)
    h = h.view(h.size(0), -1)
    out = torch.stack([F.linear(h, S[f"unshared.{d}.weight"], S[f"unshared.{d}.bias"]) for d in range(cfg.num_domains)], dim=1)
    return out[torch.arange(y.size(0)), y]


def discriminator(S: State, x: Tensor, y: Tensor, cfg: Cfg) -> Tensor:
    """Discriminator.forward -- M:526-532"""
    _, blocks = _trunk_plan(cfg)
    n = len(blocks)
    h = _trunk(S, "main.", x, cfg)
    h = F.conv2d(h, S[f"main.{n + 4}.weight"], S[f"main.{n + 4}.bias"])
    h = h.view(h.size(0), -1)
    return h[torch.arange(y.size(0)), y]


# --------------------------------------------------------------------------- #
# losses -- S:467-546, 566-583
# --------------------------------------------------------------------------- #
def adv_loss(logits: Tensor, target: int) -> Tensor:
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, fill_value=float(target)))


def r1_reg(d_out: Tensor, x_in: Tensor) -> Tensor:
    """S:573-583: 0.5 * mean_n sum (d sum(D(x)) / dx)^2, differentiable (create_graph)"""
    g = torch.autograd.grad(d_out.sum(), x_in, create_graph=True, retain_graph=True, only_inputs=True)[0]
    return 0.5 * g.pow(2).view(x_in.size(0), -1).sum(1).mean(0)


def style_code(N: Dict[str, State], y: Tensor, cfg: Cfg, x_ref: Optional[Tensor] = None, z: Optional[Tensor] = None) -> Tensor:
    """core/utils.py:485-490 (adain): the mapping network on a latent code, else the style encoder on a reference image"""
    return mapping_network(N["mapping_network"], z, y, cfg) if z is not None else style_encoder(N["style_encoder"], x_ref, y, cfg)


def compute_d_loss(N, x_real, y_org, y_trg, cfg: Cfg, z_trg=None, x_ref=None):
    """S:467-491 -> (loss, {real, fake, reg})"""
    assert (z_trg is None) != (x_ref is None)
    x_real = x_real.detach().requires_grad_(True)
    out = discriminator(N["discriminator"], x_real, y_org, cfg)
    loss_real = adv_loss(out, 1)
    loss_reg = r1_reg(out, x_real)
    with torch.no_grad():
        s_trg = style_code(N, y_trg, cfg, x_ref, z_trg)
        x_fake = generator(N["generator"], x_real, s_trg, cfg)
    loss_fake = adv_loss(discriminator(N["discriminator"], x_fake, y_trg, cfg), 0)
    loss = loss_real + loss_fake + cfg.lambda_reg * loss_reg
    return loss, {"real": float(loss_real), "fake": float(loss_fake), "reg": float(loss_reg)}


def compute_g_loss(N, x_real, y_org, y_trg, cfg: Cfg, z_trgs=None, x_refs=None):
    """S:494-546 (w_hpf = 0) -> (loss, {adv, sty, ds, cyc})"""
    assert (z_trgs is None) != (x_refs is None)
    z_trg, z_trg2 = z_trgs if z_trgs is not None else (None, None)
    x_ref, x_ref2 = x_refs if x_refs is not None else (None, None)
    s_trg = style_code(N, y_trg, cfg, x_ref, z_trg)
    x_fake = generator(N["generator"], x_real, s_trg, cfg)
    loss_adv = adv_loss(discriminator(N["discriminator"], x_fake, y_trg, cfg), 1)
    s_pred = style_encoder(N["style_encoder"], x_fake, y_trg, cfg)
    loss_sty = torch.mean(torch.abs(s_pred - s_trg))
    s_trg2 = style_code(N, y_trg, cfg, x_ref2, z_trg2)
    x_fake2 = generator(N["generator"], x_real, s_trg2, cfg).detach()
    loss_ds = torch.mean(torch.abs(x_fake - x_fake2))
    s_org = style_encoder(N["style_encoder"], x_real, y_org, cfg)
    x_rec = generator(N["generator"], x_fake, s_org, cfg)
    loss_cyc = torch.mean(torch.abs(x_rec - x_real))
    loss = loss_adv + cfg.lambda_sty * loss_sty - cfg.lambda_ds * loss_ds + cfg.lambda_cyc * loss_cyc
    return loss, {"adv": float(loss_adv), "sty": float(loss_sty), "ds": float(loss_ds), "cyc": float(loss_cyc)}


# --------------------------------------------------------------------------- #
# optimizer and EMA -- S:48-56 (torch.optim.Adam(lr, betas=[beta1, beta2], weight_decay): L2 decay added to the gradient), S:549-551
# --------------------------------------------------------------------------- #
@dataclass
class AdamState:
    step: Dict[str, int] = field(default_factory=dict)
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)


def adam_update(S: State, grads: Dict[str, Optional[Tensor]], st: AdamState, lr: float, cfg: Cfg) -> None:
    """torch.optim.Adam single-tensor path with weight_decay (coupled L2); a parameter with grad None is skipped"""
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            p = S[k]
            g = g + cfg.weight_decay * p
            if k not in st.step:
                st.step[k], st.m[k], st.v[k] = 0, torch.zeros_like(p), torch.zeros_like(p)
            st.step[k] += 1
            t = st.step[k]
            st.m[k].lerp_(g, 1 - cfg.beta1)
            st.v[k].mul_(cfg.beta2).addcmul_(g, g, value=1 - cfg.beta2)
            denom = (st.v[k].sqrt() / math.sqrt(1 - cfg.beta2 ** t)).add_(1e-8)
            p.addcdiv_(st.m[k], denom, value=-lr / (1 - cfg.beta1 ** t))


def moving_average(S: State, S_ema: State, beta: float) -> None:
    """S:549-551: param_test = lerp(param, param_test, beta)"""
    with torch.no_grad():
        for k in S:
            S_ema[k] = torch.lerp(S[k], S_ema[k], beta)


def grads_of(loss: Tensor, S: State) -> Dict[str, Optional[Tensor]]:
    keys = [k for k, v in S.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [S[k] for k in keys], allow_unused=True, retain_graph=True)
    return dict(zip(keys, gs))


def require_grad(S: State, on: bool = True) -> None:
    for v in S.values():
        v.requires_grad_(on)


def train_iteration(N: Dict[str, State], N_ema: Dict[str, State], opt: Dict[str, AdamState], inputs, cfg: Cfg):
    """One iteration of Solver.train with norm_type adain -- S:262-296: D update on the latent branch, D update on the reference
    branch, G (+ mapping network + style encoder) update on the latent branch, G update on the reference branch, EMA of the three
    generator-side networks.  Returns the four loss dicts (and, for the fixture, the gradients of each update)."""
    x_real, y_org, y_trg, x_ref, x_ref2, z_trg, z_trg2 = inputs
    lrs = {"generator": cfg.lr, "style_encoder": cfg.lr, "discriminator": cfg.lr, "mapping_network": cfg.f_lr}
    out, grads = {}, {}
    for net in N.values():
        require_grad(net, True)
    for tag, kw in (("d_latent", dict(z_trg=z_trg)), ("d_ref", dict(x_ref=x_ref))):
        loss, out[tag] = compute_d_loss(N, x_real, y_org, y_trg, cfg, **kw)
        grads[tag] = grads_of(loss, N["discriminator"])
        adam_update(N["discriminator"], grads[tag], opt["discriminator"], lrs["discriminator"], cfg)
    loss, out["g_latent"] = compute_g_loss(N, x_real, y_org, y_trg, cfg, z_trgs=(z_trg, z_trg2))
    grads["g_latent"] = {n: grads_of(loss, N[n]) for n in ("generator", "mapping_network", "style_encoder")}
    for n in ("generator", "mapping_network", "style_encoder"):
        adam_update(N[n], grads["g_latent"][n], opt[n], lrs[n], cfg)
    loss, out["g_ref"] = compute_g_loss(N, x_real, y_org, y_trg, cfg, x_refs=(x_ref, x_ref2))
    grads["g_ref"] = {"generator": grads_of(loss, N["generator"])}
    adam_update(N["generator"], grads["g_ref"]["generator"], opt["generator"], lrs["generator"], cfg)
    for n in ("generator", "mapping_network", "style_encoder"):
        moving_average(N[n], N_ema[n], cfg.ema_beta)
    return out, grads
