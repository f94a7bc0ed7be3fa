"""Diagnostic: NaN-poison every torch.empty the ops allocate; report the first autograd Function whose backward (or
forward) emits a NaN -> finds reads of memory no kernel wrote."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from helpers import formula_fill, make_opt
from oracle import defectgan_oracle as O
from de_i2i_gan_amd import ops
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "tiny16"
prec = "bf16" if "bf16" in sys.argv else "f32"
c = dict(tiny16=dict(image_size=16, batch=1, num_layers=1, ngf=8, ndf=8, hidden_nc=8),
         t0=dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16),
         t1=dict(image_size=64, batch=4, num_layers=4, ngf=16, ndf=16, hidden_nc=32))[name]
_empty, _empty_like = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = empty, empty_like
ops._workspaces.clear()
seen = set()
def wrap(cls):
    fwd, bwd = cls.forward, cls.backward
    def f(ctx, *a):
        out = fwd(ctx, *a)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point() and torch.isnan(o).any() and (cls.__name__, "f") not in seen:
                seen.add((cls.__name__, "f"))
                print("NaN in forward of", cls.__name__, "out", i, tuple(o.shape), "nan frac", float(torch.isnan(o).float().mean()),
                      [tuple(x.shape) for x in a if torch.is_tensor(x)], [x for x in a if isinstance(x, ops.ConvGeom)])
        return out
    def b(ctx, *g):
        out = bwd(ctx, *g)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point() and torch.isnan(o).any():
                key = (cls.__name__, "b", i, tuple(o.shape))
                if key not in seen:
                    seen.add(key)
                    print("NaN in backward of", cls.__name__, "grad", i, tuple(o.shape), "nan frac", float(torch.isnan(o).float().mean()),
                          "grad_out nan:", [bool(torch.isnan(x).any()) for x in g if torch.is_tensor(x)], getattr(ctx, "geom", None))
        return out
    cls.forward, cls.backward = staticmethod(f), staticmethod(b)
for n_ in dir(ops):
    o = getattr(ops, n_)
    if isinstance(o, type) and issubclass(o, torch.autograd.Function) and o is not torch.autograd.Function:
        wrap(o)
bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
tr = DefectGanTrainer(make_opt(c, DEV, prec))
G, D = tr.model.netG, tr.model.netD
formula_fill(G); formula_fill(D)
if "dfirst" in sys.argv:
    gan, clf = tr.model("discriminator", bg, labels, df)
    (gan + 2 * clf).backward()
    print("D losses", float(gan), float(clf), "nan D grads:", [k for k, p in D.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()])
ls = tr.model("generator", bg, labels, df)
(ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
print("G losses", [float(a) for a in ls])
print("nan G grads:", [k for k, p in G.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()][:10])
