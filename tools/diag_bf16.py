"""Diagnostic: bf16 vs f32 HIP paths with the reference's real init (N(0,0.02)): losses and gradient cosine."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from helpers import make_opt
from oracle import defectgan_oracle as O
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
DEV = "cuda:0"
def run(c):
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    res = {}
    for pname in ("f32", "bf16"):
        torch.manual_seed(123)
        tr = DefectGanTrainer(make_opt(c, DEV, pname))
        G, D = tr.model.netG, tr.model.netD
        tr.optimizers["D"].zero_grad()
        g1, c1 = tr.model("discriminator", bg, labels, df)
        (g1 + 2 * c1).backward()
        dgr = {k: p.grad.clone() for k, p in D.named_parameters()}
        tr.optimizers["G"].zero_grad()
        ls = tr.model("generator", bg, labels, df)
        (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
        ggr = {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None}
        res[pname] = (float(g1), float(c1), [float(x) for x in ls], dgr, ggr)
    a, b = res["f32"], res["bf16"]
    print(c, "\n losses f32 ", a[0], a[1], a[2], "\n losses bf16", b[0], b[1], b[2])
    for tag, i in (("D", 3), ("G", 4)):
        cs = []
        for k in a[i]:
            x, y = a[i][k].double().flatten(), b[i][k].double().flatten()
            if x.norm() < 1e-9: continue
            cs.append((float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-30)), float((x - y).norm() / x.norm()), k))
        cs.sort()
        print(f" {tag} grads: min cos {cs[0][0]:.4f} ({cs[0][2]}), median cos {np.median([c_[0] for c_ in cs]):.4f}, median relL2 {np.median([c_[1] for c_ in cs]):.3e}")
        allx = torch.cat([a[i][k].double().flatten() for k in a[i]]); ally = torch.cat([b[i][k].double().flatten() for k in a[i]])
        print(f"   whole-gradient cos {float(torch.dot(allx, ally) / (allx.norm() * ally.norm())):.5f}")
run(dict(image_size=64, batch=4, num_layers=4, ngf=16, ndf=16, hidden_nc=32))
run(dict(image_size=128, batch=4, num_layers=5, ngf=32, ndf=32, hidden_nc=64))
