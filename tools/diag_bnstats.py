"""Diagnostic: per-layer running-stat deviation from the reference golden after the two golden train steps."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from helpers import formula_fill, load_golden, make_opt
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
name = sys.argv[1] if len(sys.argv) > 1 else "t1_img64_b4"
for pname in ("f32", "bf16", "bf16"):
    meta, arr, c, cfg = load_golden(name)
    tr = DefectGanTrainer(make_opt(c, "cuda:0", pname))
    G, D = tr.model.netG, tr.model.netD
    formula_fill(G); formula_fill(D)
    from oracle import defectgan_oracle as O
    bg, lab, df = O.synthetic_batch(c["batch"], c["image_size"])
    for it in range(2):
        tr.step(bg, lab, df)
    tr.flush_losses()
    sdg = G.state_dict()
    print(pname)
    for k in meta["G_keys"]:
        if "running_" in k:
            mine, ref = sdg[k].float().cpu().numpy(), arr["bn::" + k]
            print("  %-55s maxrel %.3f  mine[:3] %s ref[:3] %s" % (k, float(np.max(np.abs(mine - ref) / (np.abs(ref) + 1e-6))) if "var" in k else float(np.max(np.abs(mine - ref))), np.round(mine[:3], 3), np.round(ref[:3], 3)))
