"""Diagnostic: run the G loss graph twice on one trainer (same weights, grads zeroed in between) and report the first
autograd Function, in backward order, whose gradient outputs differ between the first and the second run."""
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
trace = []
def wrap(cls):
    fwd, bwd = cls.forward, cls.backward
    def f(ctx, *a):
        out = fwd(ctx, *a)
        outs = out if isinstance(out, tuple) else (out,)
        trace.append(("F", cls.__name__, [o.detach().double().cpu().clone() for o in outs if torch.is_tensor(o)],
                      str([x for x in a if isinstance(x, ops.ConvGeom)])))
        return out
    def b(ctx, *g):
        out = bwd(ctx, *g)
        outs = out if isinstance(out, tuple) else (out,)
        trace.append(("B", cls.__name__, [o.detach().double().cpu().clone() for o in outs if torch.is_tensor(o)],
                      str(getattr(ctx, "geom", ""))))
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
G.eval() if "eval" in sys.argv else None
runs = []
for it in range(2):
    trace.clear()
    G.zero_grad(set_to_none=True); D.zero_grad(set_to_none=True)
    ls = tr.model("generator", bg, labels, df)
    (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
    torch.cuda.synchronize()
    runs.append(list(trace))
a, b = runs
print("trace lengths", len(a), len(b))
shown = 0
for i, (x, y) in enumerate(zip(a, b)):
    assert x[0] == y[0] and x[1] == y[1], (i, x[:2], y[:2])
    for j, (u, v) in enumerate(zip(x[2], y[2])):
        d = ((u - v).norm() / (v.norm() + 1e-30)).item()
        if d > float(sys.argv[-1]):
            print("#%d %s %s out%d shape %s rel diff %.3e  |v| %.3e  %s" % (i, x[0], x[1], j, tuple(u.shape), d, float(v.norm()), x[3]))
            shown += 1
    if shown >= 40:
        break
