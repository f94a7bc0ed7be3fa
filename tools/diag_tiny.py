"""Diagnostic: per-parameter G-step gradient error, f32 HIP path vs fp64 oracle (tiny16 / t0 configs)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from helpers import formula_fill, make_opt
from oracle import defectgan_oracle as O
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "tiny16"
c = dict(tiny16=dict(image_size=16, batch=1, num_layers=1, ngf=8, ndf=8, hidden_nc=8),
         t0=dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16))[name]
cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"])
bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
SG = {k: (v.double() if v.is_floating_point() else v) for k, v in O.make_state(O.generator_state_shapes(cfg)).items()}
SD = {k: v.double() for k, v in O.make_state(O.discriminator_state_shapes(cfg)).items()}
g_losses, gG = O.train_generator_once({k: v.clone() for k, v in SG.items()}, SD, None, bg.double(), labels.double(), df.double(), cfg)
tr = DefectGanTrainer(make_opt(c, DEV, "f32"))
G, D = tr.model.netG, tr.model.netD
formula_fill(G); formula_fill(D)
if "dfirst" in sys.argv:
    gan, clf = tr.model("discriminator", bg, labels, df)
    (gan + 2 * clf).backward()
if "gnograd" in sys.argv:          # what the D step does to G: two no_grad train-mode forwards
    nm_l, df_l = tr.model._get_labels(labels.to(DEV))
    with torch.no_grad():
        G(bg.to(DEV), df_l); G(df.to(DEV), nm_l)
if "dfwdbwd" in sys.argv:          # what the D step does to D: forward + backward on some images
    src, cls = D(df.to(DEV))
    (src.mean() + cls.mean()).backward()
    D.zero_grad(set_to_none=True)
if "dfwd" in sys.argv:
    with torch.no_grad():
        D(df.to(DEV))
ls = tr.model("generator", bg, labels, df)
(ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
print("losses", [float(a) - float(b) for a, b in zip(ls, g_losses)])
for k, p in G.named_parameters():
    if gG[k] is None or p.grad is None:
        continue
    ref = gG[k]
    print("%-50s %.3e  |ref| %.3e" % (k, ((p.grad.double().cpu() - ref).norm() / (ref.norm() + 1e-30)).item(), float(ref.norm())))
