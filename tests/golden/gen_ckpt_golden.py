#!/usr/bin/env python3
"""Checkpoint fixture written BY THE REFERENCE (build container only; needs /root/reference):

  * the reference's DefectGanTrainer is built with its own random init (seed 11: init_weights N(0, 0.02), spectral /
    noise options off), runs ONE D+G step (so BatchNorm buffers, counters and Adam-updated weights are non-trivial),
  * the reference's own `model.save('latest')` (models/networks/__init__.py:4-12 -> torch.save(net.state_dict())) writes
    latest_net_G.pth / latest_net_D.pth, and the trainer's iteration record `iter.txt` is written the way
    defectgan_trainer.py:111-113 does (np.savetxt((epoch, iters))),
  * G(x) (eval mode) and D(G(x)) of the saved weights are recorded.

Stored under tests/golden/ckpt_ref/: the two .pth files (plain tensors: loaded with weights_only=True), iter.txt and
ckpt_ref.npz (inputs are the seeded synthetic batch; outputs are arrays).  No reference source is stored."""
import os
import shutil
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

C = dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16)
OUT = Path(__file__).resolve().parent / "ckpt_ref"


def main():
    tmp = Path(tempfile.mkdtemp())
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=C["ngf"], ndf=C["ndf"], input_nc=3,
        use_spectral=False, num_scales=2, style_norm_block_type="spade", hidden_nc=C["hidden_nc"], style_distill=False,
        embed_nc=768, add_noise=False, num_layers=C["num_layers"], image_size=C["image_size"], batch_size=C["batch"],
        device=torch.device("cpu"), is_train=True, clf_loss_type="bce", continue_training=False, load_model_name=None,
        init_type="normal", init_variance=0.02, phase="train", ckpt_dir=tmp, name="ref_run", iters_per_epoch=10,
        num_epochs=-1, num_iters=100, lr=[2e-4], optimizer="adam", scheduler="step", lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1],
        num_critics=1, diff_aug="", sean_alpha=None, use_running_stats=False, save_latest_freq=10 ** 9)
    torch.manual_seed(11)
    tr = DefectGanTrainer(opt)
    bg, labels, df = O.synthetic_batch(C["batch"], C["image_size"])
    tr.iters += 1
    tr._train_discriminator_once(bg, labels, df)
    tr._train_generator_once(bg, labels, df)
    tr.model.save("latest")                                                   # the reference's own writer
    np.savetxt(tr.iter_record_path, (1, tr.iters), fmt="%i", delimiter=",")    # defectgan_trainer.py:113
    G, D = tr.model.netG, tr.model.netD
    G.eval(); D.eval()
    with torch.no_grad():
        out, prob = G(bg, labels.reshape(C["batch"], 6, 1, 1))
        src, cls = D(out)
    OUT.mkdir(exist_ok=True)
    for f in ("latest_net_G.pth", "latest_net_D.pth", "iter.txt"):
        shutil.copy(tmp / "ref_run" / f, OUT / f)
    np.savez_compressed(OUT / "ckpt_ref.npz", out=out.numpy(), prob=prob.numpy(), src=src.numpy(), cls=cls.numpy(),
                        g_keys=np.array(sorted(G.state_dict().keys())), d_keys=np.array(sorted(D.state_dict().keys())))
    print("wrote", sorted(p.name for p in OUT.iterdir()), {p.name: p.stat().st_size for p in OUT.iterdir()})


if __name__ == "__main__":
    main()
