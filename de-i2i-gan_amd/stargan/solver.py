"""The reference's stargan-v2 loss functions and train iteration (stargan-v2/core/solver.py) on the product's networks: same function
names, argument meaning and return values (loss tensor + a namespace of floats); ``Solver.train_iteration`` is one pass of the body of
``Solver.train`` (solver.py:262-296) for ``norm_type adain``."""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from ..optim import FusedAdam


def adv_loss(logits, target):
    """solver.py:566-570"""
    assert target in [1, 0]
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, fill_value=float(target)))


def r1_reg(d_out, x_in):
    """solver.py:573-583: zero-centred gradient penalty on real images -- a DOUBLE backward through the discriminator (ops._ConvDgradFn &c.)"""
    batch_size = x_in.size(0)
    grad_dout = torch.autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True, only_inputs=True)[0]
    grad_dout2 = grad_dout.pow(2)
    assert grad_dout2.size() == x_in.size()
    return 0.5 * grad_dout2.view(batch_size, -1).sum(1).mean(0)


def get_style_code(nets, norm_type, num_embeds, y_trg, x_ref, z_trg):
    """core/utils.py:485-490"""
    if norm_type != "adain":
        raise NotImplementedError("stargan: norm_type 'adain' is built")
    return nets.mapping_network(z_trg, y_trg) if z_trg is not None else nets.style_encoder(x_ref, y_trg)


def compute_d_loss(nets, args, x_real, y_org, y_trg, z_trg=None, x_ref=None, masks=None):
    """solver.py:467-491"""
    assert (z_trg is None) != (x_ref is None)
    x_real.requires_grad_()
    out = nets.discriminator(x_real, y_org)
    loss_real = adv_loss(out, 1)
    loss_reg = r1_reg(out, x_real)
    with torch.no_grad():
        s_trg = get_style_code(nets, args.norm_type, 1, y_trg, x_ref, z_trg)
        x_fake = nets.generator(x_real, s_trg, labels=y_trg, masks=masks)
    out = nets.discriminator(x_fake, y_trg)
    loss_fake = adv_loss(out, 0)
    loss = loss_real + loss_fake + args.lambda_reg * loss_reg
    return loss, SimpleNamespace(real=loss_real.item(), fake=loss_fake.item(), reg=loss_reg.item())


def compute_g_loss(nets, args, x_real, y_org, y_trg, z_trgs=None, x_refs=None, masks=None):
    """solver.py:494-546 (w_hpf = 0: no heat-map masks)"""
    assert (z_trgs is None) != (x_refs is None)
    z_trg, z_trg2 = z_trgs if z_trgs is not None else (None, None)
    x_ref, x_ref2 = x_refs if x_refs is not None else (None, None)
    s_trg = get_style_code(nets, args.norm_type, args.num_embeds, y_trg, x_ref, z_trg)
    x_fake = nets.generator(x_real, s_trg, labels=y_trg, masks=masks)
    out = nets.discriminator(x_fake, y_trg)
    loss_adv = adv_loss(out, 1)
    s_pred = get_style_code(nets, args.norm_type, args.num_embeds, y_trg, x_fake, z_trg=None)
    loss_sty = torch.mean(torch.abs(s_pred - s_trg))
    s_trg2 = get_style_code(nets, args.norm_type, args.num_embeds, y_trg, x_ref2, z_trg=z_trg2)
    x_fake2 = nets.generator(x_real, s_trg2, labels=y_trg, masks=masks).detach()
    loss_ds = torch.mean(torch.abs(x_fake - x_fake2))
    s_org = get_style_code(nets, args.norm_type, args.num_embeds, y_org, x_real, z_trg=None)
    x_rec = nets.generator(x_fake, s_org, labels=y_org, masks=None)
    loss_cyc = torch.mean(torch.abs(x_rec - x_real))
    loss = loss_adv + args.lambda_sty * loss_sty - args.lambda_ds * loss_ds + args.lambda_cyc * loss_cyc
    return loss, SimpleNamespace(adv=loss_adv.item(), sty=loss_sty.item(), ds=loss_ds.item(), cyc=loss_cyc.item())


def moving_average(model, model_test, beta=0.999):
    """solver.py:549-551"""
    with torch.no_grad():
        for param, param_test in zip(model.parameters(), model_test.parameters()):
            param_test.data = torch.lerp(param.data, param_test.data, beta)


class Solver:
    """solver.py:33-56 (networks, EMA copies, one Adam per network: lr / f_lr for the mapping network, betas, coupled weight decay)
    + one iteration of ``train`` (solver.py:262-296, norm_type adain).  The optimizers are the product's fused multi-tensor Adam."""

    def __init__(self, args, nets, nets_ema, device="cuda:0"):
        self.args, self.nets, self.nets_ema, self.device = args, nets, nets_ema, torch.device(device)
        for ns in (nets, nets_ema):
            for m in vars(ns).values():
                m.to(self.device)
        self.optims = SimpleNamespace()
        for name, net in vars(nets).items():
            lr = args.f_lr if name == "mapping_network" else args.lr
            setattr(self.optims, name, FusedAdam(net.parameters(), lr=lr, betas=(args.beta1, args.beta2), weight_decay=args.weight_decay,
                                                 decoupled=False))

    def _reset_grad(self):
        for opt in vars(self.optims).values():
            opt.zero_grad()

    def train_iteration(self, x_real, y_org, y_trg, x_ref, x_ref2, z_trg, z_trg2):
        args, nets, optims = self.args, self.nets, self.optims
        out = {}
        d_loss, out["d_latent"] = compute_d_loss(nets, args, x_real, y_org, y_trg, z_trg=z_trg)
        self._reset_grad()
        d_loss.backward()
        optims.discriminator.step()
        d_loss, out["d_ref"] = compute_d_loss(nets, args, x_real, y_org, y_trg, x_ref=x_ref)
        self._reset_grad()
        d_loss.backward()
        optims.discriminator.step()
        g_loss, out["g_latent"] = compute_g_loss(nets, args, x_real, y_org, y_trg, z_trgs=[z_trg, z_trg2])
        self._reset_grad()
        g_loss.backward()
        optims.generator.step()
        optims.mapping_network.step()
        optims.style_encoder.step()
        g_loss, out["g_ref"] = compute_g_loss(nets, args, x_real, y_org, y_trg, x_refs=[x_ref, x_ref2])
        self._reset_grad()
        g_loss.backward()
        optims.generator.step()
        for name in ("generator", "mapping_network", "style_encoder"):
            moving_average(getattr(nets, name), getattr(self.nets_ema, name), beta=0.999)
        return out
