"""CPU, world_size 2 over gloo: the product's GradReducer (de_i2i_gan_amd.parallel) sums gradients across ranks with
hook-driven bucketed all-reduce, the 1/world average is applied at the optimizer, BatchNorm buffers follow rank 0.
The two-rank result must equal the reference driven micro-batch by micro-batch with gradient accumulation
(goldens `ddp2_*`).  The model maths on CPU is the oracle -- the reducer is the unit under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_golden
from oracle import defectgan_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Holder(torch.nn.Module):
    """nn.Module view of an oracle state dict so GradReducer can attach hooks / broadcast buffers."""

    def __init__(self, state):
        super().__init__()
        self.keys = list(state.keys())
        for i, k in enumerate(self.keys):
            t = state[k]
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.register_buffer(f"b{i}", t.clone())
            else:
                self.register_parameter(f"p{i}", torch.nn.Parameter(t.clone()))

    def state(self):
        named = dict(self.named_parameters())
        named.update(dict(self.named_buffers()))
        return {k: named[("b" if k.endswith(("running_mean", "running_var", "num_batches_tracked")) else "p") + str(i)]
                for i, k in enumerate(self.keys)}


def _worker(rank, world, port, name, out_dir, overlap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from de_i2i_gan_amd.parallel import GradReducer
    meta, arr, c, cfg = load_golden(name)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    per = c["batch"] // world
    sl = slice(rank * per, (rank + 1) * per)
    G, D = Holder(O.make_state(O.generator_state_shapes(cfg))), Holder(O.make_state(O.discriminator_state_shapes(cfg)))
    red = GradReducer(bucket_bytes=1 << 14, direct_bytes=1 << 12, overlap=overlap)     # tiny thresholds: exercise both paths
    red.attach(G)
    red.attach(D)
    if rank != 0:
        # a replica that was seeded / resumed differently: attach_ddp's broadcast must bring it back to rank 0's state
        with torch.no_grad():
            for t in list(G.parameters()) + list(D.parameters()) + [b for b in G.buffers() if b.is_floating_point()]:
                t.add_(0.37)
            for b in G.buffers():
                if not b.is_floating_point():
                    b.add_(5)
    red.broadcast_parameters(G)
    red.broadcast_parameters(D)
    SG, SD = G.state(), D.state()
    stG, stD = O.AdamState(), O.AdamState()
    w = cfg.loss_weight
    # ---- D step on this rank's shard ----
    gan, clf = O.discriminator_losses(SG, SD, bg[sl], labels[sl], df[sl], cfg)
    (gan + clf * w[0]).backward()
    red.reduce(D)
    O.adam_update(SD, {k: (SD[k].grad / world if SD[k].grad is not None else None) for k in O.param_keys(SD)}, stD, cfg)
    losses = [float(gan), float(clf)]
    # ---- G step ----
    for p in D.parameters():
        p.grad = None
        p.requires_grad_(False)
    g = O.generator_losses(SG, SD, bg[sl], labels[sl], df[sl], cfg)
    (g[0] + g[1] * w[1] + g[2] * w[2] + g[3] * w[3] + g[4] * w[4]).backward()
    red.reduce(G)
    O.adam_update(SG, {k: (SG[k].grad / world if SG[k].grad is not None else None) for k in O.param_keys(SG)}, stG, cfg)
    red.broadcast_buffers(G)
    losses += [float(x) for x in g]
    torch.save({"losses": losses, "G": {k: v.detach().clone() for k, v in SG.items()},
                "D": {k: v.detach().clone() for k, v in SD.items()}, "stats": red.stats}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_step_equals_reference_microbatch_accumulation(tmp_path, overlap):
    name, world = "t0_img32_b2", 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), overlap), nprocs=world, join=True)
    meta, arr, c, cfg = load_golden(name)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    # per-rank losses are the reference's per-micro-batch losses
    for i in range(world):
        assert np.allclose(r[i]["losses"], arr["ddp2_losses"][i], rtol=2e-4, atol=1e-6), (i, r[i]["losses"], arr["ddp2_losses"][i])
    # both ranks hold identical parameters and buffers after the step
    for net in ("G", "D"):
        for k in r[0][net]:
            assert torch.equal(r[0][net][k], r[1][net][k]), (net, k)
    # ... equal to the reference with grad accumulation over the two micro-batches (rank 0's BatchNorm buffers)
    for net, keys in (("G", meta["G_check_keys"]), ("D", meta["D_check_keys"])):
        mine = np.array([float(r[0][net][k].double().norm()) for k in keys])
        ref = arr[f"ddp2_{net}_post_norm"]
        assert np.max(np.abs(mine - ref) / np.maximum(ref, 1e-6)) < 2e-3, net
    assert r[0]["stats"]["collectives"] > 2 and r[0]["stats"]["bytes"] > 0


def _bf16_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from de_i2i_gan_amd.parallel import GradReducer
    torch.manual_seed(100 + rank)                          # different gradients on every rank
    shapes = [(64, 32, 3, 3), (32,), (128, 64, 4, 4), (7,), (256, 128)]          # direct (>= 4 KB here) and bucketed tensors
    res = {}
    for comm in ("fp32", "bf16"):
        net = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(s)) for s in shapes])
        red = GradReducer(bucket_bytes=1 << 12, direct_bytes=1 << 12, comm_dtype=comm)
        red.attach(net)
        g = torch.Generator().manual_seed(7 + rank)
        loss = sum((p * torch.randn(p.shape, generator=g) * (10.0 ** (i - 2))).sum() for i, p in enumerate(net))    # gradient scales 1e-2 .. 1e2
        loss.backward()
        red.reduce(net)
        res[comm] = [p.grad.clone() for p in net]
        assert all(p.grad.dtype == torch.float32 for p in net)
        res[comm + "_bytes"] = red.stats["bytes"]
    torch.save(res, os.path.join(out_dir, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_bf16_gradient_exchange_error_against_the_fp32_exchange(tmp_path):
    """GradReducer(comm_dtype="bf16"): every addend is rounded to bf16 and the sum is bf16 (one more rounding per partial sum):
    relative to the fp32 exchange of the same gradients each tensor is within 2^-8 in relative L2 (measured ~3e-3 at world 2: two
    roundings of 2^-9 / sqrt(3) rms each), elementwise within 2^-7 of the tensor's largest sum; half the bytes travel; both ranks
    end with identical gradients; the gradients stay fp32 (the optimizer's 1/world is applied to them as before)."""
    world = 2
    mp.spawn(_bf16_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"b{i}.pt", weights_only=True) for i in range(world)]
    assert r[0]["bf16_bytes"] * 2 == r[0]["fp32_bytes"]
    for a, b in zip(r[0]["bf16"], r[1]["bf16"]):
        assert torch.equal(a, b)
    for exact, narrow in zip(r[0]["fp32"], r[0]["bf16"]):
        rel = float((narrow - exact).norm() / exact.norm())
        assert rel < 2.0 ** -8, rel
        assert float((narrow - exact).abs().max()) <= 2.0 ** -7 * float(exact.abs().max())
