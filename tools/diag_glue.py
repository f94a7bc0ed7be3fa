"""Which Python lines launch the step's torch glue kernels (fills, copies, cats, adds): torch.profiler with stacks over 2 steps of the
bench workload.  GPU box.  python3 tools/diag_glue.py"""
import sys
from collections import Counter
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

sys.argv = ["bench.py"]
args = bench.parse()
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402

device = "cuda:0"
torch.cuda.set_device(0)
torch.manual_seed(123)
tr = DefectGanTrainer(bench.make_opt(args, device))
bg, lab, df = bench.synthetic_batch(args.batch, args.image_size, seed=7)
bg, lab, df = bg.to(device), lab.to(device), df.to(device)
for _ in range(3):
    tr.step(bg, lab, df)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        tr.step(bg, lab, df)
torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::cat", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::clone",
        "aten::contiguous", "aten::_foreach_add_", "aten::stack", "aten::mean", "aten::sum", "aten::to", "aten::_to_copy")
by = Counter()
for ev in prof.events():
    if ev.name in WANT:
        frames = [f for f in (ev.stack or []) if "de-i2i-gan_amd" in f or "de_i2i_gan_amd" in f]
        where = frames[0].split("/")[-1] if frames else ((ev.stack or ["?"])[0].split("/")[-1])
        by[(ev.name, where, str(ev.input_shapes)[:60])] += 1
for (name, where, shapes), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print("%6.1f/step  %-18s %-60s %s" % (n / STEPS, name, where, shapes))
