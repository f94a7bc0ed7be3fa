"""bench.py --gpus N invoked directly (no torch.distributed.run around it) must start the N rank processes itself
(it exited).  CPU: the launcher's rendezvous plumbing over gloo.  GPU: the real 2-rank RCCL step when the
box has two GPUs (skipped on the 1-GPU test box; the driver's 8-GPU node runs bench.py itself)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def test_direct_invocation_spawns_ranks_that_rendezvous(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    tag = tmp_path / "self"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "3", "--spawn-selftest", str(tag)], env=env, timeout=300)
    assert r.returncode == 0
    for rank in range(3):
        got = Path(f"{tag}.{rank}").read_text().split()
        assert got == [str(rank), str(rank), "3", "127.0.0.1", "6"], got       # 1 + 2 + 3: the all-reduce saw every rank


def test_a_failing_rank_fails_the_launcher(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""            # no GPU visible: every rank exits with bench.py's "needs a GPU" message
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"needs a GPU" in r.stderr


@pytest.mark.gpu
def test_two_rank_rccl_step_through_the_launcher():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL across ranks); the 1-GPU box covers the 2-rank step over gloo in test_ddp_gpu_two_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--image-size", "128",
                        "--batch", "4", "--no-cpu-baseline"], env=env, timeout=900, stdout=subprocess.PIPE)
    assert r.returncode == 0
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert line["ddp"]["collectives_per_step"] > 2 and line["ddp"]["comm_ms_per_pass"] > 0
    assert all(v == v for v in line["losses_last_step"].values())               # finite (no NaN)


@pytest.mark.gpu
def test_the_one_gpu_bench_line_keeps_the_drivers_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` -> ONE JSON line on stdout with the keys the driver reads (metric / unit of
    BASELINE.json, whole-job value, the roofline object of the dominant kernel); the CPU baseline leg is skipped here (it is 10-30 s
    of oracle time; its own switch)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], env=env,
                       timeout=900, stdout=subprocess.PIPE)
    assert r.returncode == 0
    out = [ln for ln in r.stdout.decode().strip().splitlines() if ln.startswith("{")]
    assert len(out) == 1
    line = json.loads(out[0])
    base = json.loads((REPO / "BASELINE.json").read_text())
    assert base["metric"].startswith(line["metric"].replace("256x256", "256×256")) and line["unit"] == "pairs/s"     # BASELINE.json's headline metric
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "bf16" and line["data"] == "synthetic"
    assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.2 < rf["frac"] < 1.0
    # the dominant instance kernel-alone (extra serial steps), the other instance, and the forward instance sampled in the timed region
    assert rf["launches_per_step"] > 0 and rf["avg_launch_ms"] > 0 and "kernel-alone" in rf["timing"]
    assert rf["other_instance"]["launches_per_step"] > 0 and rf["ms_per_step"] >= rf["other_instance"]["ms_per_step"]
    assert rf["forward_instance_in_timed_region"]["event_bracketed_launches"] > 0
    for k in ("defer_loss_sync", "discriminator_passes", "eval_generator_passes_in_d_step", "d_weight_gradients_in_g_step"):
        assert k in line["config"]
    assert all(v == v for v in line["losses_last_step"].values())
