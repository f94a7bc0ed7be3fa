"""Diagnostic (not a test): N train steps at the bench configuration, then a SHA-256 over every parameter and the last
losses.  Two runs of this script must print the same digest (bit-reproducible step: no atomics, no races) -- a longer
version of tests/test_full_size_gpu.py's two-step check.   usage: diag_soak.py [steps] [--use-spectral] [--add-noise] [--dtype fp8]"""
import argparse
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int, nargs="?", default=60)
ap.add_argument("--use-spectral", action="store_true")
ap.add_argument("--add-noise", action="store_true")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--stage", default="defectgan")
a = ap.parse_args()
args = argparse.Namespace(image_size=256, batch=16, dtype=a.dtype, use_spectral=a.use_spectral, add_noise=a.add_noise, stage=a.stage)
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402

opt = bench.make_opt(args, "cuda:0")
torch.manual_seed(123)
tr = DefectGanTrainer(opt)
bg, lab, df = bench.synthetic_batch(16, 256, seed=7)
bg, lab, df = bg.cuda(), lab.cuda(), df.cuda()
for i in range(a.steps):
    tr.step(bg, lab, df)
tr.flush_losses()
torch.cuda.synchronize()
h = hashlib.sha256()
bad = 0
for net in (tr.model.netG, tr.model.netD):
    for k, v in sorted(net.state_dict().items()):
        t = v.detach().float().cpu()
        bad += int((~torch.isfinite(t)).sum())
        h.update(k.encode())
        h.update(t.numpy().tobytes())
last = {k: v[-1] for kind in tr.losses.values() for k, v in kind.items() if v}
print("steps", a.steps, "non-finite parameter values", bad, "digest", h.hexdigest()[:32], "last losses", {k: round(v, 6) for k, v in last.items()})
