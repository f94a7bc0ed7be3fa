"""Diagnostic (not a test): where a MAE step's wall time goes."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from types import SimpleNamespace
args = SimpleNamespace(image_size=256, batch=16, dtype="bf16", stage="mae")
opt = bench.make_opt(args, "cuda:0")
from de_i2i_gan_amd.trainers.mae_trainer import MAETrainer
torch.manual_seed(1)
tr = MAETrainer(opt)
bg, lab, df = bench.synthetic_batch(16, 256, 7)
bg, lab = bg.cuda(), lab.cuda()
for _ in range(3):
    tr.step(bg, lab)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(bg, lab)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
