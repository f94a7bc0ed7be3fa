#!/usr/bin/env python3
"""Golden fixture of the stargan-v2 G/D train iteration, captured from the REFERENCE on CPU.

Runs only in the build container (needs /root/reference); never on the GPU box, never imported by product code.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_starganv2_golden.py

Imports the reference's own ``core.model`` (Generator, MappingNetwork, StyleEncoder, Discriminator) and ``core.solver``
(compute_d_loss with its R1 penalty, compute_g_loss, moving_average).  Modules that are absent here and do no arithmetic on this path are
stubbed in ``sys.modules`` BEFORE the import (the list is recorded in the fixture): ``munch`` (an attribute dict: stubbed with a ten-line
attribute dict), ``torchvision`` / ``cv2`` / ``skimage`` / ``ffmpeg`` (image I/O, augmentation transforms, video, the FAN landmark
network's pre/post-processing) and the reference's own data-loader / metrics modules (they import torchvision datasets and downloaded
metric networks).  Networks are built at a small size (img_size 64, max_conv_dim 64, style_dim 16, latent_dim 8, w_hpf 0, norm_type
adain), formula-filled, and one iteration of ``Solver.train`` (solver.py:262-296) is driven by hand -- D update on the latent branch, D on
the reference branch, G + mapping network + style encoder on the latent branch, G on the reference branch, EMA -- with optimizers built
as ``Solver.__init__`` builds them (solver.py:48-56).  The oracle (oracle/starganv2_oracle.py) is ASSERTED equal to the reference at
every stage; only data is stored: sg0_img64_b2.{npz,json}."""
import json
import os
import sys
from pathlib import Path
from types import ModuleType, SimpleNamespace
from unittest.mock import MagicMock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/stargan-v2")


class _Munch(dict):
    """munch.Munch as the reference uses it: a dict with attribute access (a return container, solver.py:489-491,543-546)"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


from transformers import ViTForImageClassification  # noqa: E402,F401  (core.model names it at import; resolved BEFORE the stubs below so
#                                                         that transformers' own "is torchvision installed" probe sees the truth)

_munch = ModuleType("munch")
_munch.Munch = _Munch
sys.modules["munch"] = _munch
STUBS = ["torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets", "torchvision.models", "cv2", "skimage",
         "skimage.filters", "ffmpeg", "core.data_loader", "metrics", "metrics.eval", "metrics.fid", "metrics.lpips"]
for _m in STUBS:
    sys.modules.setdefault(_m, MagicMock())

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import starganv2_oracle as O  # noqa: E402
from core import model as RM  # noqa: E402  (the reference)
from core import solver as RS  # noqa: E402  (the reference)

torch.set_num_threads(8)
NAME = "sg0_img64_b2"
BATCH = 2


def close(a, b, what, rtol=2e-4, atol=2e-6):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"oracle != reference for {what}: err={err:.3e} ref={ref:.3e}"
    return err


def fill(net, prefix):
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(O.formula_tensor(prefix + k, tuple(v.shape)))


def main():
    # lambda_reg: on this formula fill D's input gradient is small (R1 ~ 2e-6); 2e5 makes the penalty -- whose gradient is a DOUBLE
    # backward through every conv / pool / LeakyReLU of D -- carry a large share of the D updates' gradients, so the fixture pins it
    cfg = O.Cfg(lambda_reg=2e5)
    args = SimpleNamespace(norm_type="adain", num_embeds=1, lambda_reg=cfg.lambda_reg, lambda_sty=cfg.lambda_sty, lambda_ds=cfg.lambda_ds,
                           lambda_cyc=cfg.lambda_cyc, w_hpf=0, DiffAugment="")
    nets = _Munch(generator=RM.Generator(cfg.img_size, cfg.style_dim, max_conv_dim=cfg.max_conv_dim, w_hpf=0),
                  mapping_network=RM.MappingNetwork(cfg.latent_dim, cfg.style_dim, cfg.num_domains),
                  style_encoder=RM.StyleEncoder(cfg.img_size, cfg.style_dim, cfg.num_domains, cfg.max_conv_dim),
                  discriminator=RM.Discriminator(cfg.img_size, cfg.num_domains, cfg.max_conv_dim))
    shapes = {"generator": O.generator_state_shapes(cfg), "mapping_network": O.mapping_state_shapes(cfg),
              "style_encoder": O.style_encoder_state_shapes(cfg), "discriminator": O.discriminator_state_shapes(cfg)}
    for n, net in nets.items():
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == shapes[n] and list(got) == list(shapes[n]), f"{n}: state_dict manifest mismatch (keys, shapes or ORDER)"
        fill(net, n + ".")
    import copy
    nets_ema = _Munch({n: copy.deepcopy(nets[n]) for n in ("generator", "mapping_network", "style_encoder")})
    N = {n: {k: v.detach().clone() for k, v in net.state_dict().items()} for n, net in nets.items()}
    N_ema = {n: {k: v.clone() for k, v in N[n].items()} for n in nets_ema}
    inputs = O.synthetic_inputs(cfg, BATCH)
    x_real, y_org, y_trg, x_ref, x_ref2, z_trg, z_trg2 = inputs
    meta = {"name": NAME, "batch": BATCH, "config": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "stubbed_modules": ["munch (attribute dict)"] + STUBS,
            "keys": {n: list(shapes[n]) for n in shapes}}
    arrays, errs = {}, {}

    # ---- forward passes ----
    with torch.no_grad():
        s_map = nets.mapping_network(z_trg, y_trg)
        s_enc = nets.style_encoder(x_ref, y_trg)
        x_fake = nets.generator(x_real, s_map)
        d_out = nets.discriminator(x_real, y_org)
        o_map = O.mapping_network(N["mapping_network"], z_trg, y_trg, cfg)
        o_enc = O.style_encoder(N["style_encoder"], x_ref, y_trg, cfg)
        o_fake = O.generator(N["generator"], x_real, o_map, cfg)
        o_d = O.discriminator(N["discriminator"], x_real, y_org, cfg)
    errs.update(mapping=close(o_map, s_map, "mapping network"), style_encoder=close(o_enc, s_enc, "style encoder"),
                generator=close(o_fake, x_fake, "generator"), discriminator=close(o_d, d_out, "discriminator"))
    arrays.update(s_map=s_map.numpy(), s_enc=s_enc.numpy(), x_fake=x_fake.numpy(), d_out=d_out.numpy())

    # ---- one training iteration through the reference's loss functions and optimizers (solver.py:48-56, 262-296) ----
    lrs = {"generator": cfg.lr, "style_encoder": cfg.lr, "discriminator": cfg.lr, "mapping_network": cfg.f_lr}
    optims = {n: torch.optim.Adam(params=nets[n].parameters(), lr=lrs[n], betas=[cfg.beta1, cfg.beta2], weight_decay=cfg.weight_decay)
              for n in nets}

    def reset_grad():
        for o in optims.values():
            o.zero_grad()

    ref_losses, ref_grads = {}, {}

    def gnorms(net):
        return np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in net.named_parameters()])

    for tag, kw in (("d_latent", dict(z_trg=z_trg)), ("d_ref", dict(x_ref=x_ref))):
        loss, ls = RS.compute_d_loss(nets, args, x_real.clone(), y_org, y_trg, masks=None, **kw)
        reset_grad()
        loss.backward()
        ref_losses[tag] = [ls.real, ls.fake, ls.reg]
        ref_grads[tag] = {"discriminator": gnorms(nets.discriminator)}
        if tag == "d_latent":
            arrays["d_latent_grad::main.0.weight"] = nets.discriminator.main[0].weight.grad.numpy().copy()     # carries the R1 term's double backward
        optims["discriminator"].step()
    loss, ls = RS.compute_g_loss(nets, args, x_real, y_org, y_trg, z_trgs=[z_trg, z_trg2], masks=None)
    reset_grad()
    loss.backward()
    ref_losses["g_latent"] = [ls.adv, ls.sty, ls.ds, ls.cyc]
    ref_grads["g_latent"] = {n: gnorms(nets[n]) for n in ("generator", "mapping_network", "style_encoder")}
    for n in ("generator", "mapping_network", "style_encoder"):
        optims[n].step()
    loss, ls = RS.compute_g_loss(nets, args, x_real, y_org, y_trg, x_refs=[x_ref, x_ref2], masks=None)
    reset_grad()
    loss.backward()
    ref_losses["g_ref"] = [ls.adv, ls.sty, ls.ds, ls.cyc]
    ref_grads["g_ref"] = {"generator": gnorms(nets.generator)}
    optims["generator"].step()
    for n in ("generator", "mapping_network", "style_encoder"):
        RS.moving_average(nets[n], nets_ema[n], beta=cfg.ema_beta)

    # ---- the oracle, same iteration ----
    opt = {n: O.AdamState() for n in N}
    o_losses, o_grads = O.train_iteration(N, N_ema, opt, inputs, cfg)
    order = {"d_latent": ("real", "fake", "reg"), "d_ref": ("real", "fake", "reg"), "g_latent": ("adv", "sty", "ds", "cyc"),
             "g_ref": ("adv", "sty", "ds", "cyc")}
    for tag, keys in order.items():
        errs["loss_" + tag] = close([o_losses[tag][k] for k in keys], ref_losses[tag], "losses " + tag,
                                    rtol=1e-5 if tag == "d_latent" else 2e-3)        # later stages sit behind Adam updates of lr-sized steps
        arrays["losses_" + tag] = np.array(ref_losses[tag], np.float64)
    close(o_grads["d_latent"]["main.0.weight"], arrays["d_latent_grad::main.0.weight"], "dL/d(main.0.weight) incl. the R1 double backward", rtol=2e-4)
    for tag in order:
        for n, ref in ref_grads[tag].items():
            og = o_grads[tag][n] if tag.startswith("g_") else o_grads[tag]
            on = np.array([float(og[k].double().norm()) if og.get(k) is not None else -1.0 for k in shapes[n]])
            assert ((on < 0) == (ref < 0)).all(), (tag, n)
            m = ref > 1e-6 * ref.max()
            errs[f"gradnorm_{tag}_{n}"] = float(np.max(np.abs(on[m] - ref[m]) / ref[m]))
            assert errs[f"gradnorm_{tag}_{n}"] < (1e-3 if tag == "d_latent" else 5e-2), (tag, n, errs[f"gradnorm_{tag}_{n}"])
            arrays[f"gradnorm_{tag}_{n}"] = ref
    for n, net in nets.items():
        sd = net.state_dict()
        ref = np.array([float(sd[k].double().norm()) for k in shapes[n]])
        mine = np.array([float(N[n][k].detach().double().norm()) for k in shapes[n]])
        errs["post_" + n] = close(mine, ref, "post-step norms " + n, rtol=1e-4)
        arrays["post_norm_" + n] = ref
        # sign-like first Adam steps (beta1 = 0: the update is lr * g / |g| per element): elementwise within 2.2 lr
        dmax = max(float((sd[k] - N[n][k].detach()).abs().max()) for k in shapes[n])
        assert dmax <= 2.2 * 2 * lrs[n], (n, dmax)
    for n, net in nets_ema.items():
        sd = net.state_dict()
        ref = np.array([float(sd[k].double().norm()) for k in shapes[n]])
        mine = np.array([float(N_ema[n][k].double().norm()) for k in shapes[n]])
        errs["ema_" + n] = close(mine, ref, "EMA norms " + n, rtol=1e-5)
        arrays["ema_norm_" + n] = ref
    meta["oracle_vs_reference_max_abs_err"] = errs
    meta["torch_version"] = torch.__version__
    out_dir = Path(__file__).resolve().parent
    np.savez_compressed(out_dir / f"{NAME}.npz", **arrays)
    with open(out_dir / f"{NAME}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(NAME, "ok; oracle-vs-reference errs:", {k: f"{v:.2e}" for k, v in errs.items()})
    for tag in order:
        print("  ", tag, ref_losses[tag])


if __name__ == "__main__":
    main()
