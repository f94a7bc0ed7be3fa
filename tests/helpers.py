"""Shared test helpers: opt namespace with the reference's field names, formula weight fill, golden loading."""
import json
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from oracle import defectgan_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    meta = json.loads((GOLD / f"{name}.json").read_text())
    arr = np.load(GOLD / f"{name}.npz")
    c = meta["config"]
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"])
    return meta, arr, c, cfg


def make_opt(c, device, compute_dtype="f32", **over):
    """The attribute set the reference's DefectGanTrainer reads, plus compute_dtype."""
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"], input_nc=3,
        use_spectral=False, num_scales=c.get("num_scales", 2), style_norm_block_type="spade", hidden_nc=c["hidden_nc"],
        style_distill=False, embed_nc=768, add_noise=False, num_layers=c["num_layers"], image_size=c["image_size"],
        batch_size=c["batch"], device=torch.device(device), is_train=True, clf_loss_type="bce", continue_training=False,
        load_model_name=None, init_type="normal", init_variance=0.02, phase="train", ckpt_dir=Path(tempfile.mkdtemp()),
        name="t", iters_per_epoch=10, num_epochs=-1, num_iters=100, lr=[2e-4], optimizer="adam", scheduler="step",
        lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug="", sean_alpha=None, use_running_stats=False,
        save_latest_freq=10 ** 9, compute_dtype=compute_dtype)
    for k, v in over.items():
        setattr(opt, k, v)
    return opt


def formula_fill(net):
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(O.formula_tensor(k, tuple(v.shape)).to(v.device))
