#!/usr/bin/env python3
"""Golden fixture of the SEAN decoder variant (--style_norm_block_type sean, --sean_alpha 1: style embeddings from an
embeddings file mixed with the label latent), made by running the REFERENCE's DefectGanTrainer on CPU (build container only).

Like gen_goldens.py: every parameter / buffer of G and D is overwritten with the RNG-free formula fill; the embeddings file
is synthetic (oracle.synthetic_embeddings: formula-filled embeddings for the labels with one or two set bits, empty lists --
the reference then feeds zeros -- for the rest; no download); python's ``random`` (the reference draws the embeddings with
random.choices) is seeded identically before every reference / oracle call; the oracle restatement is ASSERTED equal to the
reference, and only data is stored: t6_img64_b2_sean.{npz,json}.

t8_img64_b2_sean_distill adds --style_distill and --use_running_stats (normalization.py:104-190): the distillation terms the
SEAN layers back-propagate inside their own forward, the per-label lists of mixed codes tracked over the G loss's four passes,
``update_stats`` after the two steps (buffers compared key by key) and a forward with ``inference_running_stats`` (codes from a
noise vector and those buffers).  batch == num_embeds there: the reference's KL of the (N, num_embeds, hidden) encoder
features against the (N, hidden) target only broadcasts when the two agree."""
import json
import os
import random
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
           "torchmetrics", "torchmetrics.image", "torchmetrics.image.lpip", "torch.utils.tensorboard", "tensorboard"]:
    sys.modules.setdefault(_m, MagicMock())

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import defectgan_oracle as O  # noqa: E402
from trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402  (the reference)

torch.set_num_threads(8)
CONFIGS = {
    "t6_img64_b2_sean": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, style_norm="sean", embed_nc=24,
                             num_embeds=3, tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
    "t8_img64_b2_sean_distill": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, style_norm="sean", embed_nc=24,
                                     num_embeds=2, style_distill=True, use_running_stats=True,
                                     tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
}
SEED = 4242


def close(a, b, what, rtol=2e-4, atol=2e-6):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"oracle != reference for {what}: err={err:.3e} ref={ref:.3e}"
    return err


def fill(net):
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(O.formula_tensor(k, tuple(v.shape)))


def norms(sd):
    keys = sorted(sd.keys())
    return keys, np.array([float(sd[k].double().norm()) for k in keys])


def gnorms(net):
    keys = [k for k, _ in net.named_parameters()]
    return keys, np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in net.named_parameters()])


def main(NAME):
    c = CONFIGS[NAME]
    distill, running = bool(c.get("style_distill")), bool(c.get("use_running_stats"))
    O.SEAN_CTX.reset()
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"],
                style_norm="sean", embed_nc=c["embed_nc"], num_embeds=c["num_embeds"], style_distill=distill, use_running_stats=running)
    embeddings = O.synthetic_embeddings(cfg)
    embed_path = Path(tempfile.mkdtemp()) / "embeds.pth"
    torch.save(embeddings, embed_path)
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"], input_nc=3,
        use_spectral=False, num_scales=2, style_norm_block_type="sean", hidden_nc=c["hidden_nc"], style_distill=distill,
        embed_nc=c["embed_nc"], num_embeds=c["num_embeds"], embed_path=embed_path,
        add_noise=False, num_layers=c["num_layers"], image_size=c["image_size"], batch_size=c["batch"], device=torch.device("cpu"),
        is_train=True, clf_loss_type="bce", continue_training=False, load_model_name=None, init_type="normal", init_variance=0.02,
        phase="train", ckpt_dir=Path(tempfile.mkdtemp()), name="golden", iters_per_epoch=10, num_epochs=-1, num_iters=100, lr=[2e-4],
        optimizer="adam", scheduler="step", lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug="", sean_alpha=1.0,
        use_running_stats=running, save_latest_freq=10 ** 9, latent_dim=16)
    tr = DefectGanTrainer(opt)
    G, D = tr.model.netG, tr.model.netD
    for net in (G, D):
        fill(net)
    if True:
        for net, shapes in ((G, O.generator_state_shapes(cfg)), (D, O.discriminator_state_shapes(cfg))):
            got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            assert got == shapes and list(got) == list(shapes), "state_dict manifest mismatch (keys, shapes or ORDER)"
        meta = {"config": c, "name": NAME, "seed": SEED, "G_keys": list(G.state_dict().keys()), "D_keys": list(D.state_dict().keys())}
        arrays, errs = {}, {}
        SG, SD = ({k: v.clone() for k, v in n.state_dict().items()} for n in (G, D))
        SE = (embeddings, random)                          # the oracle's sean "extractor": the embeddings and the RNG they are drawn with
        bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
        # ---- forward: inference mode of the model (defectgan_model.py:304-306: embeddings of the labels -> netG(data, labels, feat)) ----
        with torch.no_grad():
            random.seed(SEED)
            out_e, prob_e = tr.model("inference", bg, labels)          # _generate_fake: embeddings drawn for `labels`
            src_e, cls_e = D(out_e)
            random.seed(SEED)
            o_feat = O.get_style_embeds(embeddings, labels, cfg, random)
            o_out, o_prob = O.generator_forward(SG, bg, labels.reshape(c["batch"], 6, 1, 1), cfg, training=False, style_feat=o_feat)
            o_src, o_cls = O.discriminator_forward(SD, o_out, cfg)
        errs.update(G_eval=close(o_out, out_e, "G eval"), P_eval=close(o_prob, prob_e, "prob"),
                    D_src=close(o_src, src_e, "D src"), D_cls=close(o_cls, cls_e, "D cls"))
        arrays.update(style_embeds=o_feat.numpy(), G_out_eval=out_e.numpy(), G_prob_eval=prob_e.numpy(), D_src=src_e.numpy(), D_cls=cls_e.numpy())
        # ---- two D+G steps through the reference trainer ----
        stG, stD = O.AdamState(), O.AdamState()
        ref_losses, ora_losses = [], []
        for it in range(2):
            random.seed(SEED + 10 * it + 1)
            tr._train_discriminator_once(bg, labels, df)
            if it == 0:
                meta["D_grad_keys"], arrays["D_grad_norms_step1"] = gnorms(D)
            random.seed(SEED + 10 * it + 2)
            tr._train_generator_once(bg, labels, df)
            if it == 0:
                meta["G_grad_keys"], arrays["G_grad_norms_step1"] = gnorms(G)
            L = tr.losses
            ref_losses.append([L["gan"]["D"][-1], L["clf"]["D"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1], L["aux"]["rec"][-1],
                               L["aux"]["cyc"][-1], L["aux"]["con"][-1]] +
                              ([L["distill"]["latent"][-1], L["distill"]["embed"][-1]] if distill else []))
            random.seed(SEED + 10 * it + 1)
            d_gan, d_clf, gD = O.train_discriminator_once(SG, SD, stD, bg, labels, df, cfg, SE=SE)
            O.adam_update(SD, gD, stD, cfg)
            random.seed(SEED + 10 * it + 2)
            gl, gG = O.train_generator_once(SG, SD, stG, bg, labels, df, cfg, SE=SE)
            O.adam_update(SG, gG, stG, cfg)
            ora_losses.append([float(d_gan), float(d_clf)] + [float(v) for v in gl])
            if it == 0:
                for tag, gr in (("G", gG), ("D", gD)):
                    on = np.array([float(gr[k].double().norm()) if gr[k] is not None else -1.0 for k in meta[f"{tag}_grad_keys"]])
                    ref = arrays[f"{tag}_grad_norms_step1"]
                    assert ((on < 0) == (ref < 0)).all(), tag
                    m = ref > 1e-4
                    errs[f"{tag}_grad_norm_rel"] = float(np.max(np.abs(on[m] - ref[m]) / ref[m]))
                    assert errs[f"{tag}_grad_norm_rel"] <= (c["tol_gradnorm"] if tag != "D" else 2e-3), (tag, errs)
        if running:
            # ---- per epoch: the tracked codes -> the mean_* / std_* buffers (generator.py:283-284 -> normalization.py:111-125) ----
            n_tracked = {k_: len(v) for m_ in G.modules() if hasattr(m_, "embeds") for k_, v in m_.embeds.items() if v}
            o_tracked = {k_: len(v) for lists in O.SEAN_CTX.embeds.values() for k_, v in lists.items() if v}
            assert n_tracked == o_tracked and n_tracked, (n_tracked, o_tracked)
            meta["tracked_codes_per_label"] = {O.label_to_str(k_): v for k_, v in n_tracked.items()}
            G.update_stats()
            O.sean_update_stats(SG, O.SEAN_CTX.embeds)
            touched = [k_ for k_, v in G.state_dict().items() if (".mean_" in k_ or ".std_" in k_) and float(v.abs().sum()) > 0]
            assert touched
            for k_ in touched:
                close(SG[k_], G.state_dict()[k_], "running-stat buffer " + k_, rtol=c["tol_running"])
            meta["running_stat_keys"] = touched
            arrays["running_stat_norms"] = np.array([float(G.state_dict()[k_].double().norm()) for k_ in touched])
            # ---- inference from the buffers: one noise vector per sample instead of embeddings (normalization.py:162-168) ----
            noise = O.formula_tensor("sean.inference_noise", (c["batch"], c["hidden_nc"]))
            G.eval()
            G.inference_running_stats = True
            O.SEAN_CTX.inference_running_stats = True
            with torch.no_grad():
                out_r, prob_r = G(bg, labels.reshape(c["batch"], 6, 1, 1), noise)
                # (on the REFERENCE's post-step state: after two steps of this formula-filled toy net the oracle's own state has
                #  drifted by the tolerances above, and this check is about the running-stats code path)
                SGr = {k_: v.clone() for k_, v in G.state_dict().items()}
                o_out_r, o_prob_r = O.generator_forward(SGr, bg, labels.reshape(c["batch"], 6, 1, 1), cfg, training=False, style_feat=noise)
            G.inference_running_stats = False
            O.SEAN_CTX.inference_running_stats = False
            G.train()
            errs["G_running_stats_inference"] = close(o_out_r, out_r, "G with inference_running_stats")
            arrays.update(inference_noise=noise.numpy(), G_out_running=out_r.numpy(), G_prob_running=prob_r.numpy())
            for k_, v in SGr.items():                    # the state that forward ran on (the product test loads it)
                arrays["post::" + k_] = v.numpy().copy()
        errs["losses_step1"] = close(ora_losses[0], ref_losses[0], "losses step 1", rtol=1e-5)
        errs["losses_step2"] = close(ora_losses[1], ref_losses[1], "losses step 2", rtol=c["tol_step2"])
        arrays["losses"] = np.array(ref_losses, dtype=np.float64)
        for tag, net, S in (("G", G, SG), ("D", D, SD)):
            keys, n = norms(net.state_dict())
            okeys, on = norms({k: v.detach() for k, v in S.items()})
            assert keys == okeys
            errs[f"{tag}_post_norm"] = close(on, n, f"{tag} post-step norms", rtol=c["tol_post"])
            meta[f"{tag}_check_keys"], arrays[f"{tag}_post_norm"] = keys, n
        for k, v in G.state_dict().items():
            if "running_" in k:
                arrays["bn::" + k] = v.numpy().copy()
    meta["oracle_vs_reference_max_abs_err"] = errs
    meta["torch_version"] = torch.__version__
    out_dir = Path(__file__).resolve().parent
    np.savez_compressed(out_dir / f"{NAME}.npz", **arrays)
    with open(out_dir / f"{NAME}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(NAME, "ok; oracle-vs-reference errs:", {k: f"{v:.2e}" for k, v in errs.items()})
    print("  losses step1:", ref_losses[0])
    print("  losses step2:", ref_losses[1])


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CONFIGS)):
        main(name)
