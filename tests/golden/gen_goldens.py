#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference); never on the GPU box and never
imported by product code.  Usage (from the repo root):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_goldens.py

What it does, per tiny config:
  1. imports the reference's DefectGanTrainer (viz/metric-only deps that are absent here and not
     on the hot path are stubbed with MagicMock -- SURVEY.md section 8c),
  2. overwrites every parameter/buffer with the RNG-free formula fill of
     oracle.defectgan_oracle.formula_tensor,
  3. records G(x), D(G(x)), two consecutive D+G steps (losses, grad norms, post-step parameter
     checksums, BN running stats), a spatial-label inference case and a 2-way micro-batch
     (DDP-equivalent) step,
  4. runs the oracle restatement on the same inputs and ASSERTS it agrees with the reference
     (this is what pins the oracle), then writes <config>.npz + <config>.json.
Only data (inputs are formula/seed generated, outputs are arrays) is stored -- no reference source.
"""
import copy
import json
import os
import sys
import tempfile
from pathlib import Path
from types import SimpleNamespace
from unittest.mock import MagicMock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/defectGAN")
for _m in ["torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.models", "cv2",
           "torchmetrics", "torchmetrics.image", "torchmetrics.image.lpip", "torch.utils.tensorboard",
           "tensorboard"]:
    sys.modules.setdefault(_m, MagicMock())

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import defectgan_oracle as O  # noqa: E402
from trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402  (the reference)

torch.set_num_threads(8)

CONFIGS = {
    # tol_step2: relative tolerance on the step-2 losses.  They sit behind one Adam update whose first step is
    # sign-like (every element moves +-lr whatever |g|), which amplifies the fp32 gradient noise floor documented
    # in run_config(); measured oracle-vs-reference (both fp32, same maths): 2e-4 on t0, 4e-2 on t1.
    "t0_img32_b2": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16, tol_step2=1e-3),
    "t1_img64_b4": dict(image_size=64, batch=4, num_layers=4, ngf=16, ndf=16, hidden_nc=32, tol_step2=8e-2),
    # the deeper encoder/decoder of the 512x512 configuration (num_scales=3: three stride-2 stages, three SPADE
    # up-blocks), shrunk to 64x64 so the reference finishes in seconds.  tol_gradnorm: the 8x8 bottleneck (128 pixels
    # per channel) makes every fp32 ReLU/L1 branch flip weigh more: the reference's fp32 G grad norms sit up to 16 %
    # from the oracle's fp32 ones, while an fp64 run of both agrees to 1e-9 on every key (checked when this config
    # was added: reference modules .double(), same losses to 1e-15).
    # the README recipes' options: every ConvBlock / decoder conv under torch.nn.utils.spectral_norm (one power iteration
    # per training-mode forward) and NoiseInjection after the decoder convs.  The noise draw is replaced by a
    # deterministic, shape-keyed provider (shape_noise below) in reference, oracle and product alike.
    "t3_img32_b2_sn_noise": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16, use_spectral=True,
                                 add_noise=True, tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
    # DiffAugment on everything the discriminator sees (host RNG: the seed below reproduces the draws)
    "t4_img32_b2_diffaug": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16,
                                diff_aug="color,translation,cutout", tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2),
    # --cycle_gan: the generator returns (foreground, spatial_prob) uncomposed and the G loss has no cyc / con terms
    # (generator.py:272-273, defectgan_model.py:222-227)
    "t7_img32_b2_cycle": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16, cycle_gan=True, tol_step2=1e-3),
    "t2_img64_s3_b2": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, num_scales=3,
                           tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
}


def make_opt(c):
    return SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=c.get("cycle_gan", False), label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"],
        input_nc=3, use_spectral=c.get("use_spectral", False), num_scales=c.get("num_scales", 2), style_norm_block_type="spade", hidden_nc=c["hidden_nc"],
        style_distill=False, embed_nc=768, add_noise=c.get("add_noise", False), num_layers=c["num_layers"],
        image_size=c["image_size"], batch_size=c["batch"], device=torch.device("cpu"), is_train=True,
        clf_loss_type="bce", continue_training=False, load_model_name=None, init_type="normal",
        init_variance=0.02, phase="train", ckpt_dir=Path(tempfile.mkdtemp()), name="golden",
        iters_per_epoch=10, num_epochs=-1, num_iters=100, lr=[2e-4], optimizer="adam", scheduler="step",
        lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug=c.get("diff_aug", ""), sean_alpha=None,
        use_running_stats=False, save_latest_freq=10 ** 9)


def fill(net):
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(O.formula_tensor(k, tuple(v.shape)))


def clone_state(net):
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def checks(sd):
    keys = sorted(sd.keys())
    s = np.array([float(sd[k].double().sum()) for k in keys])
    n = np.array([float(sd[k].double().norm()) for k in keys])
    return keys, s, n


def grad_norms(net):
    keys, vals, heads = [], [], []
    for k, p in net.named_parameters():
        keys.append(k)
        if p.grad is None:
            vals.append(-1.0)
            heads.append(np.zeros(4, np.float32))
        else:
            vals.append(float(p.grad.double().norm()))
            h4 = np.zeros(4, np.float32)
            g4 = p.grad.flatten()[:4].float().numpy()
            h4[:g4.size] = g4                      # (noise weights have a single element)
            heads.append(h4)
    return keys, np.array(vals), np.stack(heads)


def assert_close(a, b, what, rtol=2e-4, atol=2e-6):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"oracle != reference for {what}: err={err:.3e} ref={ref:.3e}"
    return err


def rel_dev(a, b, what, tol, floor=1e-4):
    """max relative deviation over entries whose reference magnitude is above `floor` (legit-zero grads skipped)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert ((a < 0) == (b < 0)).all(), f"{what}: grad-is-None pattern differs"
    m = b > floor
    dev = float(np.max(np.abs(a[m] - b[m]) / b[m]))
    if dev > tol:
        print(what, "per-key rel dev:", np.round(np.abs(a[m] - b[m]) / b[m], 3).tolist(), "ref norms:", b[m].tolist())
    assert dev <= tol, f"oracle != reference for {what}: max rel dev {dev:.3e} > {tol}"
    return dev


def run_config(name, c):
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"],
                hidden_nc=c["hidden_nc"], num_scales=c.get("num_scales", 2), use_spectral=c.get("use_spectral", False),
                add_noise=c.get("add_noise", False), diff_aug=c.get("diff_aug", ""), cycle_gan=c.get("cycle_gan", False))
    opt = make_opt(c)
    tr = DefectGanTrainer(opt)
    if c.get("add_noise"):
        # NoiseInjection draws image.new_empty(N,1,H,W).normal_(): from here on (the weights are initialised) every
        # normal_() fill -- there is no other one on the step -- comes from the deterministic provider
        torch.Tensor.normal_ = lambda self, mean=0.0, std=1.0, generator=None: self.copy_(O.shape_noise(tuple(self.shape)))
        O.NOISE_SOURCE = O.shape_noise
    G, D = tr.model.netG, tr.model.netD
    fill(G)
    fill(D)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    arrays, meta = {}, {"config": c, "name": name}

    # manifest check: oracle's key/shape manifest == reference's state_dict
    for net, shapes in ((G, O.generator_state_shapes(cfg)), (D, O.discriminator_state_shapes(cfg))):
        ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert ref_shapes == shapes, "state_dict manifest mismatch"
    meta["G_keys"] = list(G.state_dict().keys())
    meta["D_keys"] = list(D.state_dict().keys())

    SG = {k: v.clone() for k, v in G.state_dict().items()}
    SD = {k: v.clone() for k, v in D.state_dict().items()}

    # ---- forward-only goldens (eval + train mode G, D on G(x)) ----
    seg = labels.reshape(c["batch"], 6, 1, 1)
    with torch.no_grad():
        G.eval()
        D.eval()                      # (spectral norm: no power iteration here, the steps below start from the filled u, v)
        out_e, prob_e = G(bg, seg)
        src_e, cls_e = D(out_e)
        # spatial (2x2) label map: inference path (test_defectgan.py:221-228)
        seg22 = torch.zeros(c["batch"], 6, 2, 2)
        for i in range(c["batch"]):
            seg22[i, 1 + i % 5, 0, 0] = 1
            seg22[i, 1 + (i + 1) % 5, 0, 1] = 1
            seg22[i, 0, 1, 0] = 1
            seg22[i, 1 + (i + 2) % 5, 1, 1] = 1
        out_s, prob_s = G(bg, seg22)
        saved = copy.deepcopy(G.state_dict())
        G.train()
        out_t, prob_t = G(bg, seg)
        G.load_state_dict(saved)      # undo the running-stat update
    o_out_e, o_prob_e = O.generator_forward(SG, bg, seg, cfg, training=False)
    o_src_e, o_cls_e = O.discriminator_forward(SD, o_out_e, cfg)
    o_out_s, o_prob_s = O.generator_forward(SG, bg, seg22, cfg, training=False)
    SGt = {k: v.clone() for k, v in SG.items()}
    o_out_t, o_prob_t = O.generator_forward(SGt, bg, seg, cfg, training=True)
    errs = {}
    errs.update({
        "G_eval": assert_close(o_out_e, out_e, "G eval out"), "P_eval": assert_close(o_prob_e, prob_e, "G eval prob"),
        "D_src": assert_close(o_src_e, src_e, "D src"), "D_cls": assert_close(o_cls_e, cls_e, "D cls"),
        "G_spatial": assert_close(o_out_s, out_s, "G spatial-label out"),
        "G_train": assert_close(o_out_t, out_t, "G train out"),
    })
    arrays.update(G_out_eval=out_e.numpy(), G_prob_eval=prob_e.numpy(), D_src=src_e.numpy(), D_cls=cls_e.numpy(),
                  G_out_spatial=out_s.numpy(), G_prob_spatial=prob_s.numpy(), seg22=seg22.numpy(),
                  G_out_train=out_t.numpy(), G_prob_train=prob_t.numpy())

    # ---- two consecutive D+G steps through the reference trainer ----
    stG, stD = O.AdamState(), O.AdamState()
    ref_losses, ora_losses = [], []
    meta["step_seed"] = 4100          # iteration `it` runs under torch.manual_seed(step_seed + it) (DiffAugment's host draws)
    for it in range(2):
        torch.manual_seed(meta["step_seed"] + it)
        tr._train_discriminator_once(bg, labels, df)
        if it == 0:
            k, v, h = grad_norms(D)
            meta["D_grad_keys"] = k
            arrays["D_grad_norms_step1"], arrays["D_grad_heads_step1"] = v, h
        tr._train_generator_once(bg, labels, df)
        if it == 0:
            k, v, h = grad_norms(G)
            meta["G_grad_keys"] = k
            arrays["G_grad_norms_step1"], arrays["G_grad_heads_step1"] = v, h
        L = tr.losses
        ref_losses.append([L["gan"]["D"][-1], L["clf"]["D"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1],
                           L["aux"]["rec"][-1], L["aux"]["cyc"][-1], L["aux"]["con"][-1]])
        torch.manual_seed(meta["step_seed"] + it)              # the oracle consumes the RNG in the same order (D step, G step)
        ol, gD, gG = O.step(SG, SD, stG, stD, bg, labels, df, cfg)
        ora_losses.append([ol[k] for k in ("d_gan", "d_clf", "g_gan", "g_clf", "g_rec", "g_cyc", "g_con")])
        if it == 0:
            # G-step gradients are ill-conditioned in fp32: a handful of ReLU-mask / L1-sign flips (pre-activations
            # within rounding of 0) each inject a full-magnitude error that propagates through the chained G(G(x))
            # graph.  Measured here: the reference's own fp32 grads sit 1e-3 (t0) .. 5e-2 (t1) in relative L2 from
            # an fp64 run of the reference, and so do the oracle's; in fp64 oracle == reference to 1e-12.  So the
            # per-key grad-norm pin is loose by necessity (documented in DESIGN.md); tight backward checks live in
            # the op-level tests.
            on = np.array([float(gG[k].double().norm()) if gG[k] is not None else -1.0 for k in meta["G_grad_keys"]])
            errs["G_grad_norm_rel"] = rel_dev(on, arrays["G_grad_norms_step1"], "G grad norms", c.get("tol_gradnorm", 8e-2))
            on = np.array([float(gD[k].double().norm()) if gD[k] is not None else -1.0 for k in meta["D_grad_keys"]])
            errs["D_grad_norm_rel"] = rel_dev(on, arrays["D_grad_norms_step1"], "D grad norms", 2e-3)
    # step 1 is a pure forward/backward comparison; step 2 sits behind one Adam update whose first step is
    # sign-like (update = +-lr whatever |g|), so fp32 rounding of near-zero grads (the reference's own fp32
    # grads are ~1e-3 from an fp64 run of itself, dominated by L1 sign flips) moves the losses by ~1e-4.
    errs["losses_step1"] = assert_close(ora_losses[0], ref_losses[0], "losses step 1", rtol=1e-5)
    errs["losses_step2"] = assert_close(ora_losses[1], ref_losses[1], "losses step 2", rtol=c["tol_step2"])
    arrays["losses"] = np.array(ref_losses, dtype=np.float64)      # rows: step, cols: d_gan d_clf g_gan g_clf rec cyc con
    for tag, net, S in (("G", G, SG), ("D", D, SD)):
        keys, s, n = checks(net.state_dict())
        okeys, os_, on_ = checks({k: v.detach() for k, v in S.items()})
        assert keys == okeys
        errs[f"{tag}_post_norm"] = assert_close(on_, n, f"{tag} post-step norms", rtol=c.get("tol_post", 1e-2))
        meta[f"{tag}_check_keys"] = keys
        arrays[f"{tag}_post_sum"], arrays[f"{tag}_post_norm"] = s, n
    for k, v in G.state_dict().items():
        if "running_" in k:
            arrays["bn::" + k] = v.numpy().copy()
            assert_close(SG[k], v, "running stat " + k, rtol=c.get("tol_running", 5e-2))
    # a few complete post-step tensors (small ones) for element-wise checks
    def stored(sd, k):                                   # spectral convs keep their parameter under key + "_orig"
        return sd[k] if k in sd else sd[k + "_orig"]

    for k in ("stem.conv_block.0.weight", "dec_blk.1.conv.weight", "foreground_head.de_conv_block.0.weight"):
        arrays["Gp::" + k] = stored(G.state_dict(), k).numpy().copy()
    for k in ("enc_blk.0.conv_block.0.weight", "src_clf.conv_block.0.weight"):
        arrays["Dp::" + k] = stored(D.state_dict(), k).numpy().copy()

    # ---- DDP-equivalent golden: world=2 micro-batches, grad accumulation, local BN ----
    world = 2
    tr2 = DefectGanTrainer(make_opt(c))
    G2, D2 = tr2.model.netG, tr2.model.netD
    fill(G2)
    fill(D2)
    per = c["batch"] // world
    ddp_losses = []
    tr2.optimizers["D"].zero_grad()
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        gan, clf = tr2.model("discriminator", bg[sl], labels[sl], df[sl])
        ((gan + clf * 2) / world).backward()
        ddp_losses.append([gan.item(), clf.item()])
    tr2.optimizers["D"].step()
    tr2.optimizers["G"].zero_grad()
    buf0 = {k: v.clone() for k, v in G2.state_dict().items() if "running_" in k or "num_batches" in k}
    buf_rank0 = None
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        with torch.no_grad():
            for k, v in G2.state_dict().items():
                if k in buf0:
                    v.copy_(buf0[k])
        gl = tr2.model("generator", bg[sl], labels[sl], df[sl])
        g_loss = gl[0] + gl[1] * 5 + gl[2] * 5 + gl[3] * 5 + gl[4] * 1
        (g_loss / world).backward()
        ddp_losses[r] += [x.item() for x in gl]
        if r == 0:
            buf_rank0 = {k: v.clone() for k, v in G2.state_dict().items() if k in buf0}
    tr2.optimizers["G"].step()
    with torch.no_grad():
        for k, v in G2.state_dict().items():
            if k in buf0:
                v.copy_(buf_rank0[k])            # broadcast_buffers semantics: rank 0's buffers win
    arrays["ddp2_losses"] = np.array(ddp_losses, dtype=np.float64)   # rows: rank; cols: d_gan d_clf g_gan g_clf rec cyc con
    for tag, net in (("G", G2), ("D", D2)):
        keys, s, n = checks(net.state_dict())
        arrays[f"ddp2_{tag}_post_sum"], arrays[f"ddp2_{tag}_post_norm"] = s, n

    meta["oracle_vs_reference_max_abs_err"] = errs
    meta["torch_version"] = torch.__version__
    out_dir = Path(__file__).resolve().parent
    np.savez_compressed(out_dir / f"{name}.npz", **arrays)
    with open(out_dir / f"{name}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(name, "ok; oracle-vs-reference errs:", {k: f"{v:.2e}" for k, v in errs.items()})
    print("  losses step1:", ref_losses[0])
    print("  losses step2:", ref_losses[1])


if __name__ == "__main__":
    for n, c in CONFIGS.items():
        if len(sys.argv) == 1 or n in sys.argv[1:]:
            run_config(n, c)
