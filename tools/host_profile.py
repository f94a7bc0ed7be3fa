"""Where the HOST's time per train step goes (cProfile over a few steps of the bench workload; GPU box).
python3 tools/host_profile.py [steps]"""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
argv_keep = list(sys.argv); sys.argv = ["bench.py"]
args = bench.parse()
from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer  # noqa: E402

device = "cuda:0"
torch.cuda.set_device(0)
MAE = "mae" in argv_keep[2:]
if MAE:
    args.stage = "mae"
opt = bench.make_opt(args, device)
torch.manual_seed(123)
bg, lab, df = bench.synthetic_batch(args.batch, args.image_size, seed=7)
bg, lab, df = bg.to(device), lab.to(device), df.to(device)
if MAE:
    from de_i2i_gan_amd.trainers.mae_trainer import MAETrainer  # noqa: E402
    _tr = MAETrainer(opt)

    class tr:                                   # same call shape as the defectGAN trainer below
        model = _tr.model

        @staticmethod
        def step(a, b, c):
            return _tr.step(a, b)
else:
    tr = DefectGanTrainer(opt)
if "ddp" in argv_keep[2:]:                       # the gradient reducer on a one-rank RCCL group (bench.py --force-collectives)
    import os
    import socket
    import torch.distributed as dist
    from de_i2i_gan_amd.parallel import attach_ddp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    sk.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
    attach_ddp(tr, measure="measure" in argv_keep[2:], force_collectives=True)
for _ in range(3):
    tr.step(bg, lab, df)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(bg, lab, df)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("unprofiled: host %.2f ms/step, wall %.2f ms/step" % (1e3 * host / steps, 1e3 * wall / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.step(bg, lab, df)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(s.getvalue())
