#!/usr/bin/env python3
"""Golden fixture of the AdaIN decoder variant (--style_norm_block_type adain, sean_alpha 0: the StyleExtractor is the
five-layer MLP on [labels | noise]), made by running the REFERENCE's DefectGanTrainer on CPU (build container only).

Like gen_goldens.py: every parameter / buffer of G, D and the StyleExtractor E is overwritten with the RNG-free formula fill,
the extractor's torch.randn draw is replaced by the shape-keyed provider (reference, oracle and product alike), the oracle
restatement is ASSERTED equal to the reference, and only data is stored: t5_img64_b2_adain.{npz,json}."""
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
           "torchmetrics", "torchmetrics.image", "torchmetrics.image.lpip", "torch.utils.tensorboard", "tensorboard"]:
    sys.modules.setdefault(_m, MagicMock())

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import defectgan_oracle as O  # noqa: E402
from trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402  (the reference)

torch.set_num_threads(8)
CONFIGS = {
    "t5_img64_b2_adain": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, style_norm="adain", latent_dim=16,
                              tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
    # --sean_alpha 1: the StyleExtractor is the conv encoder on the image (extractor.py:50-80) instead of the MLP on [labels | noise]
    "t9_img64_b2_adain_conv": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, style_norm="adain", latent_dim=16,
                                   sean_alpha=1, tol_step2=8e-2, tol_gradnorm=0.3, tol_post=5e-2, tol_running=0.2),
}


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
    alpha = c.get("sean_alpha", 0)
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"],
                style_norm="adain", latent_dim=c["latent_dim"], sean_alpha=alpha)
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"], input_nc=3,
        use_spectral=False, num_scales=2, style_norm_block_type="adain", hidden_nc=c["hidden_nc"], style_distill=False, embed_nc=768,
        add_noise=False, num_layers=c["num_layers"], image_size=c["image_size"], batch_size=c["batch"], device=torch.device("cpu"),
        is_train=True, clf_loss_type="bce", continue_training=False, load_model_name=None, init_type="normal", init_variance=0.02,
        phase="train", ckpt_dir=Path(tempfile.mkdtemp()), name="golden", iters_per_epoch=10, num_epochs=-1, num_iters=100, lr=[2e-4],
        optimizer="adam", scheduler="step", lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug="", sean_alpha=alpha,
        use_running_stats=False, save_latest_freq=10 ** 9, latent_dim=c["latent_dim"])
    tr = DefectGanTrainer(opt)
    G, D, E = tr.model.netG, tr.model.netD, tr.model.netE
    for net in (G, D, E):
        fill(net)
    # the extractor draws torch.randn(N, latent_dim - label_nc) per call (extractor.py:89): from here on the deterministic provider
    real_randn = torch.randn
    torch.randn = lambda *shape, **kw: O.shape_noise(tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape))
    O.NOISE_SOURCE = O.shape_noise
    try:
        for net, shapes in ((G, O.generator_state_shapes(cfg)), (D, O.discriminator_state_shapes(cfg)), (E, O.extractor_state_shapes(cfg))):
            got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            assert got == shapes and (alpha == 0 or list(got) == list(shapes)), "state_dict manifest mismatch"
        meta = {"config": c, "name": NAME, "G_keys": list(G.state_dict().keys()), "D_keys": list(D.state_dict().keys()),
                "E_keys": list(E.state_dict().keys())}
        arrays, errs = {}, {}
        SG, SD, SE = ({k: v.clone() for k, v in n.state_dict().items()} for n in (G, D, E))
        bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
        # ---- forward: inference mode of the model (defectgan_model.py:310-312: netE(data, labels) -> netG(data, labels, feat)) ----
        with torch.no_grad():
            G.eval(); D.eval()
            feat = E(bg, labels)
            out_e, prob_e = G(bg, labels, feat)
            src_e, cls_e = D(out_e)
            o_feat = O.style_extractor(SE, bg, labels, cfg)
            o_out, o_prob = O.generator_forward(SG, bg, labels.reshape(c["batch"], 6, 1, 1), cfg, training=False, style_feat=o_feat)
            o_src, o_cls = O.discriminator_forward(SD, o_out, cfg)
        errs.update(E_feat=close(o_feat, feat, "E feat"), G_eval=close(o_out, out_e, "G eval"), P_eval=close(o_prob, prob_e, "prob"),
                    D_src=close(o_src, src_e, "D src"), D_cls=close(o_cls, cls_e, "D cls"))
        arrays.update(E_feat=feat.numpy(), G_out_eval=out_e.numpy(), G_prob_eval=prob_e.numpy(), D_src=src_e.numpy(), D_cls=cls_e.numpy())
        # ---- two D+G steps through the reference trainer (optimizers G, D and E) ----
        stG, stD, stE = O.AdamState(), O.AdamState(), O.AdamState()
        ref_losses, ora_losses = [], []
        for it in range(2):
            tr._train_discriminator_once(bg, labels, df)
            if it == 0:
                meta["D_grad_keys"], arrays["D_grad_norms_step1"] = gnorms(D)
            tr._train_generator_once(bg, labels, df)
            if it == 0:
                meta["G_grad_keys"], arrays["G_grad_norms_step1"] = gnorms(G)
                meta["E_grad_keys"], arrays["E_grad_norms_step1"] = gnorms(E)
            L = tr.losses
            ref_losses.append([L["gan"]["D"][-1], L["clf"]["D"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1], L["aux"]["rec"][-1],
                               L["aux"]["cyc"][-1], L["aux"]["con"][-1]])
            d_gan, d_clf, gD = O.train_discriminator_once(SG, SD, stD, bg, labels, df, cfg, SE=SE)
            O.adam_update(SD, gD, stD, cfg)
            gl, gG, gE = O.train_generator_once(SG, SD, stG, bg, labels, df, cfg, SE=SE)
            O.adam_update(SG, gG, stG, cfg)
            O.adam_update(SE, gE, stE, cfg)
            ora_losses.append([float(d_gan), float(d_clf)] + [float(v) for v in gl])
            if it == 0:
                for tag, gr in (("G", gG), ("D", gD), ("E", gE)):
                    on = np.array([float(gr[k].double().norm()) if gr[k] is not None else -1.0 for k in meta[f"{tag}_grad_keys"]])
                    ref = arrays[f"{tag}_grad_norms_step1"]
                    assert ((on < 0) == (ref < 0)).all(), tag
                    m = ref > 1e-4
                    errs[f"{tag}_grad_norm_rel"] = float(np.max(np.abs(on[m] - ref[m]) / ref[m]))
                    assert errs[f"{tag}_grad_norm_rel"] <= (c["tol_gradnorm"] if tag != "D" else 2e-3), (tag, errs)
        errs["losses_step1"] = close(ora_losses[0], ref_losses[0], "losses step 1", rtol=1e-5)
        errs["losses_step2"] = close(ora_losses[1], ref_losses[1], "losses step 2", rtol=c["tol_step2"])
        arrays["losses"] = np.array(ref_losses, dtype=np.float64)
        for tag, net, S in (("G", G, SG), ("D", D, SD), ("E", E, SE)):
            keys, n = norms(net.state_dict())
            okeys, on = norms({k: v.detach() for k, v in S.items()})
            assert keys == okeys
            errs[f"{tag}_post_norm"] = close(on, n, f"{tag} post-step norms", rtol=c["tol_post"])
            meta[f"{tag}_check_keys"], arrays[f"{tag}_post_norm"] = keys, n
        for k, v in G.state_dict().items():
            if "running_" in k:
                arrays["bn::" + k] = v.numpy().copy()
    finally:
        torch.randn = real_randn
        O.NOISE_SOURCE = None
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
