"""Paired generator passes vs the four-pass form, beside the noise floor of the four-pass form itself (the same graph with the
inputs moved by one fp32 ulp): per-parameter gradient differences.  GPU only.  python3 tools/diag_pair.py [f32|bf16]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from helpers import make_opt  # noqa: E402
from oracle import defectgan_oracle as O  # noqa: E402  (synthetic inputs only)
from de_i2i_gan_amd import ops  # noqa: E402
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402

pname = sys.argv[1] if len(sys.argv) > 1 else "f32"
c = dict(image_size=128, batch=4, num_layers=4, ngf=32, ndf=32, hidden_nc=64)
bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
ops.forked_chains = False


def run(paired, scale=1.0):
    ops.paired_passes = paired
    torch.manual_seed(11)
    tr = DefectGanTrainer(make_opt(c, "cuda:0", pname))
    G = tr.model.netG
    ls = tr.model("generator", bg * scale, labels, df * scale)
    (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
    torch.cuda.synchronize()
    return [float(v) for v in ls], {k: p.grad.double().clone() for k, p in G.named_parameters() if p.grad is not None}


lp, gp = run(True)
l4, g4 = run(False)
ln, gn = run(False, 1.0 + 2.0 ** -22)
gmax = max(float(v.norm()) for v in g4.values())
print("losses paired", lp, "\nlosses four  ", l4, "\nlosses four' ", ln)
rows = []
for k in g4:
    own = float(g4[k].norm())
    rows.append((float((gp[k] - g4[k]).norm()) / (own + 1e-30), float((gn[k] - g4[k]).norm()) / (own + 1e-30), own / gmax, k))
rows.sort(reverse=True)
print("%-60s %12s %12s %10s" % ("parameter", "paired-four", "four'-four", "own/max"))
for r in rows[:25]:
    print("%-60s %12.3e %12.3e %10.2e" % (r[3], r[0], r[1], r[2]))
tot_p = sum(float((gp[k] - g4[k]).norm()) ** 2 for k in g4) ** 0.5 / sum(float(g4[k].norm()) ** 2 for k in g4) ** 0.5
tot_n = sum(float((gn[k] - g4[k]).norm()) ** 2 for k in g4) ** 0.5 / sum(float(g4[k].norm()) ** 2 for k in g4) ** 0.5
print("whole gradient: paired-four %.3e   four'-four %.3e" % (tot_p, tot_n))
