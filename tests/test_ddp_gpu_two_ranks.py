"""GPU, world_size 2: the PRODUCT's data-parallel step end to end -- DefectGanTrainer + HIP kernels + attach_ddp (hook-driven
gradient all-reduce on a side stream, 1/world folded into the fused Adam, rank-0 BatchNorm buffers) -- on two processes.

The test box has one GPU, so both ranks use cuda:0 and the process group is gloo (RCCL refuses two ranks on one device;
gloo moves the CUDA gradients through the host): the transport differs from production (RCCL over xGMI, exercised
single-rank in test_model_gpu.py), everything else is the production code path.  Expected result = the reference driven
micro-batch by micro-batch with gradient accumulation (goldens `ddp2_*`)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import formula_fill, load_golden, make_opt
from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from de_i2i_gan_amd.parallel import attach_ddp
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    meta, arr, c, cfg = load_golden(name)
    tr = DefectGanTrainer(make_opt(dict(c, batch=c["batch"] // world), "cuda:0", "f32"))
    formula_fill(tr.model.netG)
    formula_fill(tr.model.netD)
    red = attach_ddp(tr, bucket_bytes=1 << 14, direct_bytes=1 << 12)          # tiny thresholds: buckets AND direct tensors
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    per = c["batch"] // world
    sl = slice(rank * per, (rank + 1) * per)
    tr.step(bg[sl], labels[sl], df[sl])
    torch.cuda.synchronize()
    L = tr.losses
    losses = [L["gan"]["D"][0], L["clf"]["D"][0], L["gan"]["G"][0], L["clf"]["G"][0], L["aux"]["rec"][0], L["aux"]["cyc"][0],
              L["aux"]["con"][0]]
    torch.save({"losses": losses, "G": {k: v.detach().cpu() for k, v in tr.model.netG.state_dict().items()},
                "D": {k: v.detach().cpu() for k, v in tr.model.netD.state_dict().items()}, "stats": red.stats},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_product_step_equals_reference_microbatch_accumulation(tmp_path):
    name, world = "t0_img32_b2", 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    meta, arr, c, cfg = load_golden(name)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    for i in range(world):                                   # per-rank losses = the reference's per-micro-batch losses
        assert np.allclose(r[i]["losses"], arr["ddp2_losses"][i], rtol=2e-4, atol=1e-6), (i, r[i]["losses"], arr["ddp2_losses"][i])
    for net in ("G", "D"):                                   # both ranks end with identical parameters and buffers
        for k in r[0][net]:
            assert torch.equal(r[0][net][k], r[1][net][k]), (net, k)
    for net, keys in (("G", meta["G_check_keys"]), ("D", meta["D_check_keys"])):
        mine = np.array([float(r[0][net][k].double().norm()) for k in keys])
        ref = arr[f"ddp2_{net}_post_norm"]
        assert np.max(np.abs(mine - ref) / np.maximum(ref, 1e-6)) < 2e-3, net
    assert r[0]["stats"]["collectives"] > 2 and r[0]["stats"]["bytes"] > 0
