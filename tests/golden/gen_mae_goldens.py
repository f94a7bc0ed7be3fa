#!/usr/bin/env python3
"""Golden fixture of the MAE-GAN pre-training step, captured from the REFERENCE on CPU.

Runs only in the build container (needs /root/reference); never on the GPU box, never imported by product code.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_mae_goldens.py

Per configuration: import the reference's MAETrainer (viz / metric modules that are absent here and off the step are
stubbed), formula-fill G, D and the mask token, seed the torch RNG, run two iterations (D update + G update each; every
update draws its own shifted patch mask from the global RNG), record the losses, the masks' checksums, post-step parameter
norms and the mask token -- and ASSERT that oracle/mae_oracle.py reproduces them (that is what pins the oracle)."""
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
for _m in ["torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.models", "cv2", "torchmetrics",
           "torchmetrics.image", "torchmetrics.image.lpip", "torch.utils.tensorboard", "tensorboard", "metrics.fid_score"]:
    sys.modules.setdefault(_m, MagicMock())

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import defectgan_oracle as O  # noqa: E402
from oracle import mae_oracle as M  # noqa: E402
from trainers.mae_trainer import MAETrainer  # noqa: E402  (the reference)

torch.set_num_threads(8)
SEED = 20240917

CONFIGS = {
    "m0_img32_b2_position": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16, patch_size=8,
                                 mask_ratio=0.75, mask_token_type="position"),
    "m1_img64_b2_vector": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, patch_size=8,
                               mask_ratio=0.5, mask_token_type="vector"),
    # --split_training: G is trained by the L1 loss alone, D by the classifier loss on the real images alone
    # (defectgan_model.py:119-120,157-158); the D update draws no mask
    "m2_img32_b2_split": dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16, patch_size=8,
                              mask_ratio=0.75, mask_token_type="position", split_training=True),
    # the MAE stage with SEAN blocks + --style_distill (mae_trainer.py:121-131, defectgan_model.py:106-128,361-372): style
    # embeddings from a synthetic embeddings file (oracle.synthetic_embeddings, drawn with python's ``random``), the two
    # distillation terms logged per G update, their gradients taken inside the SEAN layers.  batch == num_embeds, as in
    # gen_sean_golden.py's t8 (the reference's KL broadcast needs it)
    "m3_img64_b2_sean_distill": dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, patch_size=8,
                                     mask_ratio=0.5, mask_token_type="position", style_norm="sean", embed_nc=24, num_embeds=2,
                                     style_distill=True),
}


def make_opt(c, embed_path=None):
    sean = c.get("style_norm") == "sean"
    extra = dict(embed_path=embed_path, num_embeds=c["num_embeds"], latent_dim=16) if sean else {}
    return SimpleNamespace(
        **extra,
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"], input_nc=3,
        use_spectral=False, num_scales=2, style_norm_block_type=c.get("style_norm", "spade"), hidden_nc=c["hidden_nc"],
        style_distill=bool(c.get("style_distill")), embed_nc=c.get("embed_nc", 768), add_noise=False, num_layers=c["num_layers"], image_size=c["image_size"], batch_size=c["batch"],
        device=torch.device("cpu"), is_train=True, clf_loss_type="bce", continue_training=False, load_model_name=None,
        init_type="normal", init_variance=0.02, phase="train", ckpt_dir=Path(tempfile.mkdtemp()), log_dir=Path(tempfile.mkdtemp()),
        name="golden", iters_per_epoch=10, num_epochs=8, num_iters=100, lr=[1.5e-4], optimizer="adamw", scheduler="cos",
        lr_decay=0.05, loss_weight=[10, 3, 1], num_critics=1, diff_aug="", sean_alpha=1.0 if sean else None, use_running_stats=False,
        save_latest_freq=10 ** 9, save_img_freq=10 ** 9, save_ckpt_freq=10 ** 9, split_training=c.get("split_training", False),
        mask_token_type=c["mask_token_type"], mask_ratio=c["mask_ratio"], patch_size=c["patch_size"])


def fill(net):
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(O.formula_tensor(k, tuple(v.shape)))


def close(a, b, what, rtol, atol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * np.abs(b).max(), f"oracle != reference for {what}: {err:.3e} vs {np.abs(b).max():.3e}"
    return float(err)


def run_config(name, c):
    sean = c.get("style_norm") == "sean"
    O.SEAN_CTX.reset()
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"],
                **(dict(style_norm="sean", embed_nc=c["embed_nc"], num_embeds=c["num_embeds"], style_distill=bool(c.get("style_distill")))
                   if sean else {}))
    embed_path, SE = None, None
    if sean:
        embeddings = O.synthetic_embeddings(cfg)
        embed_path = Path(tempfile.mkdtemp()) / "embeds.pth"
        torch.save(embeddings, embed_path)
        SE = (embeddings, random)
    tr = MAETrainer(make_opt(c, embed_path), ["fusion"])
    G, D, MT = tr.model.netG, tr.model.netD, tr.model.mask_token
    fill(G)
    fill(D)
    shape = M.mask_token_shape(c["mask_token_type"], 3, c["image_size"])
    with torch.no_grad():
        MT.mask_token.copy_(O.formula_tensor("mask_token", shape) * 0.25)
    imgs, labels, _ = O.synthetic_batch(c["batch"], c["image_size"])
    SG = {k: v.clone() for k, v in G.state_dict().items()}
    SD = {k: v.clone() for k, v in D.state_dict().items()}
    token = {"mask_token": MT.mask_token.detach().clone()}
    stG, stD = O.AdamState(), O.AdamState()
    arrays, meta, errs = {}, {"config": c, "name": name, "seed": SEED}, {}
    ref_losses, masks_sum = [], []
    torch.manual_seed(SEED)
    distill = sean and bool(c.get("style_distill"))
    for it in range(2):
        random.seed(SEED + 10 * it + 1)                  # (sean: the embeddings are drawn with python's ``random``)
        tr._train_discriminator_once(imgs, labels)
        random.seed(SEED + 10 * it + 2)
        tr._train_generator_once(imgs, labels)
        L = tr.losses
        ref_losses.append([L["gan"]["D"][-1], L["clf"]["D"][-1], L["rec"]["train"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1]] +
                          ([L["distill"]["latent"][-1], L["distill"]["embed"][-1]] if distill else []))
    # the oracle consumes the RNG in the same order: one shifted mask per update
    # the reference steps its cosine scheduler once at construction (first_epoch = 1, base_trainer.py:121-123), so the
    # optimizers run at lr(1), not opt.lr
    lr_eff = tr.optimizers["D"].param_groups[0]["lr"]
    assert abs(tr.optimizers["G"].param_groups[0]["lr"] - lr_eff) < 1e-12
    meta["lr_effective"] = lr_eff
    torch.manual_seed(SEED)
    ora_losses = []
    for it in range(2):
        split = c.get("split_training", False)
        md = None if split else M.generate_shifted_mask(tuple(imgs.shape), c["patch_size"], c["mask_ratio"])
        mg = M.generate_shifted_mask(tuple(imgs.shape), c["patch_size"], c["mask_ratio"])
        masks_sum.append([0.0 if split else float(md.sum()), float(mg.sum())])
        random.seed(SEED + 10 * it + 1)
        ol, _, _ = M.step(SG, SD, token, stG, stD, imgs, labels, md, mg, cfg, lr=lr_eff, kind=c["mask_token_type"],
                          mask_ratio=c["mask_ratio"], split_training=split, SE=SE,
                          before_g=lambda it=it: random.seed(SEED + 10 * it + 2))
        ora_losses.append([ol[k] for k in ("d_gan", "d_clf", "g_rec", "g_gan", "g_clf")] +
                          ([ol["distill_latent"], ol["distill_embed"]] if distill else []))
    errs["losses_step1"] = close(ora_losses[0], ref_losses[0], "losses step 1", 1e-5)
    errs["losses_step2"] = close(ora_losses[1], ref_losses[1], "losses step 2", 5e-2)      # behind sign-like first AdamW steps
    arrays["losses"] = np.array(ref_losses, np.float64)        # rows: step; cols: d_gan d_clf g_rec g_gan g_clf [distill latent, embed]
    arrays["mask_sums"] = np.array(masks_sum, np.float64)
    for tag, net, S in (("G", G, SG), ("D", D, SD)):
        keys = sorted(net.state_dict().keys())
        n = np.array([float(net.state_dict()[k].double().norm()) for k in keys])
        on = np.array([float(S[k].detach().double().norm()) for k in keys])
        errs[f"{tag}_post_norm"] = close(on, n, f"{tag} post-step norms", 1e-2)
        meta[f"{tag}_check_keys"] = keys
        arrays[f"{tag}_post_norm"] = n
    arrays["mask_token_post"] = MT.mask_token.detach().numpy().copy()
    # (two sign-like first AdamW steps move every element by ~lr each whatever its gradient's size: an element whose fp32
    #  gradient sits within rounding of 0 may differ by 2 lr per step between two correct evaluations)
    errs["mask_token"] = close(token["mask_token"].detach().numpy(), arrays["mask_token_post"], "mask token", 2e-2, atol=2.2 * lr_eff)
    meta["oracle_vs_reference_max_abs_err"] = errs
    meta["torch_version"] = torch.__version__
    out_dir = Path(__file__).resolve().parent
    np.savez_compressed(out_dir / f"{name}.npz", **arrays)
    with open(out_dir / f"{name}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(name, "ok; oracle-vs-reference errs:", {k: f"{v:.2e}" for k, v in errs.items()})
    print("  losses:", ref_losses)


if __name__ == "__main__":
    for n, c in CONFIGS.items():
        if len(sys.argv) == 1 or n in sys.argv[1:]:
            run_config(n, c)
