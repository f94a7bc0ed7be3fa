#!/usr/bin/env python3
"""bench.py -- paired 256x256 images/sec of the defectGAN G+D train step on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = DefectGanTrainer.step(): one discriminator update + one generator update (num_critics = 1) on one
synthetic batch that is already resident in HBM.  value = pairs/s over ALL ranks (weak scaling: 16 pairs per GPU).
One JSON line is printed by rank 0.  Extra objects:
  roofline     -- the step's dominant kernel: whichever instance of the halo-resident 3x3 conv (its FOLD launches = the reflect
                  dgrads with the ring and the norm-backward reductions, or its forward launches) holds more device time,
                  algorithmic FLOPs / HIP-event time of its launches, KERNEL-ALONE (extra steps after the timed region with the
                  weight gradients on the main stream, so no bracket spans two concurrent kernels), against the dense bf16 MFMA
                  peak; sub-objects: the other instance, the forward instance sampled INSIDE the timed region, the family
  cpu_baseline -- the CPU oracle restatement of the same step timed on this box's host cores (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 2500.0}      # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
# (fp8 mode: priced at the bf16 peak -- its e4m3 GEMMs use the non-scaled v_mfma_f32_16x16x32_fp8_fp8, which runs at the
#  bf16 rate, and most of the step still runs bf16 kernels)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16, help="pairs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: e4m3 forward GEMMs of the stride-1 3x3 convs, bf16 everywhere else (BASELINE.json configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--stage", default="defectgan", choices=["defectgan", "mae"],
                    help="mae: the MAE-GAN pre-training step (SURVEY.md section 8f rank 1) instead of the headline defectGAN step")
    ap.add_argument("--use-spectral", action="store_true", help="the README recipes' --use_spectral (spectral-normalised convs)")
    ap.add_argument("--add-noise", action="store_true", help="the README recipes' --add_noise (NoiseInjection after decoder convs)")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=INT",
                    help="library option for A/B runs on one box, e.g. halo_conv=0 (dei2i_set_option)")
    ap.add_argument("--no-fuse-norm", action="store_true",
                    help="A/B: separate statistics passes instead of conv-epilogue statistics (ops.fuse_norm = False)")
    ap.add_argument("--no-fuse-ring", action="store_true",
                    help="A/B: SPADE -> upsample -> conv writes the upsampled normalised tensor (ops.fuse_ring = False)")
    ap.add_argument("--no-wgrad-stream", action="store_true",
                    help="A/B: leaf-weight gradients on the main stream instead of the side stream (ops.wgrad_side_stream = False)")
    ap.add_argument("--no-fuse-bwd", action="store_true",
                    help="A/B: SPADE / BatchNorm backward reductions by a streaming pass instead of the dgrad epilogue (ops.fuse_bwd = False)")
    ap.add_argument("--no-fold-eval-bn", action="store_true",
                    help="A/B: eval-mode BatchNorm as its own apply pass instead of folded into the conv weights (ops.fold_eval_bn = False)")
    ap.add_argument("--no-paired-passes", action="store_true",
                    help="A/B: the G loss's four generator passes as four passes (ops.paired_passes = False) instead of two over 2 x batch")
    ap.add_argument("--no-forked-chains", action="store_true",
                    help="A/B: the G loss's four generator passes on one stream instead of two chains on two streams (ops.forked_chains = False)")
    ap.add_argument("--fuse-pro", action="store_true",
                    help="A/B: BatchNorm / SPADE apply on the consumer conv's operand path (ops.fuse_pro = True; measured slower)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="--gpus 1 only: attach the data-parallel reducer on a ONE-rank RCCL group -- hooks, bucket cat / copy-back, "
                         "the all-reduce launches on the side stream and the wgrad-stream join all run -- and report what it costs "
                         "(ddp: collectives and MB per step, side-stream occupancy, exposed ms, step time with / without the reducer)")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="gradient all-reduce message type (parallel.GradReducer comm_dtype; bf16 halves the bytes on xGMI)")
    ap.add_argument("--ddp-no-overlap", action="store_true", help="A/B: issue every gradient collective after backward (GradReducer overlap=False)")
    ap.add_argument("--ddp-no-measure", action="store_true", help="A/B: no event brackets around the collectives (GradReducer measure=False)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # child process of the default run
    ap.add_argument("--spawn-selftest", default=None, help=argparse.SUPPRESS)             # tests: rendezvous of the spawned ranks over gloo, no GPU
    return ap.parse_args()


def make_opt(args, device):
    """The reference's option attribute names with its defaults (options/defectgan_options.py:22-48,93-109)."""
    import tempfile
    from pathlib import Path
    from types import SimpleNamespace
    import torch
    s = args.image_size
    mae = getattr(args, "stage", "defectgan") == "mae"     # options/defectgan_options.py:150-189 (MAE stage defaults)
    extra = dict(optimizer="adamw", scheduler="cos", lr=[1.5e-4], lr_decay=0.05, loss_weight=[10, 3, 1], num_epochs=200,
                 split_training=False, mask_token_type="position", mask_ratio=0.75, patch_size=8) if mae else {}
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=False, label_nc=6, skip_conn=False, ngf=64, ndf=64, input_nc=3,
        use_spectral=bool(getattr(args, "use_spectral", False)), num_scales=3 if s >= 512 else 2,
        style_norm_block_type="spade", hidden_nc=128, style_distill=False, embed_nc=768,
        add_noise=bool(getattr(args, "add_noise", False)), num_layers=5 if s >= 128 else 4, image_size=s,
        batch_size=args.batch, device=torch.device(device), is_train=True, clf_loss_type="bce", continue_training=False,
        load_model_name=None, init_type="normal", init_variance=0.02, phase="train", ckpt_dir=Path(tempfile.mkdtemp()),
        name="bench", iters_per_epoch=1000, num_epochs=-1, num_iters=10 ** 6, lr=[2e-4], optimizer="adam", scheduler="step",
        lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug="", sean_alpha=None, use_running_stats=False,
        save_latest_freq=10 ** 9, compute_dtype=args.dtype, defer_loss_sync=True)
    for k, v in extra.items():
        setattr(opt, k, v)
    return opt


def synthetic_batch(n, size, seed, label_nc=6):
    """bg, df ~ U(-1,1) fp32, multi-hot labels with labels[i, 1 + i % 5] = 1."""
    import torch
    g = torch.Generator().manual_seed(seed)
    bg = torch.rand(n, 3, size, size, generator=g) * 2 - 1
    df = torch.rand(n, 3, size, size, generator=g) * 2 - 1
    labels = torch.zeros(n, label_nc)
    for i in range(n):
        labels[i, 1 + i % (label_nc - 1)] = 1
    return bg, labels, df


def cpu_baseline():
    """Oracle (CPU restatement, fp32, torch CPU ops) timed on this host -- a reported baseline, bounded to well under a
    minute: C1 = 64x64 batch 4 (BASELINE.json configs[0]; 1 warm-up + 2 timed steps), then up to 1 + 4 256x256 batch-1
    steps (about 10-20 s of CPU work) if the C1 timing predicts they fit the budget.  Checker code used here only as the reported baseline."""
    import torch
    from oracle import defectgan_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, os.cpu_count() or 1, 16))      # cgroup CPU share of a 1-GPU box is 16
    torch.set_num_threads(cores)

    def run(size, n, layers, warm, timed):
        cfg = O.Cfg(image_size=size, num_layers=layers)
        SG, SD = O.make_state(O.generator_state_shapes(cfg)), O.make_state(O.discriminator_state_shapes(cfg))
        stG, stD = O.AdamState(), O.AdamState()
        bg, lab, df = O.synthetic_batch(n, size)
        for _ in range(warm):
            O.step(SG, SD, stG, stD, bg, lab, df, cfg)
        t0 = time.perf_counter()
        for _ in range(timed):
            O.step(SG, SD, stG, stD, bg, lab, df, cfg)
        return n * timed / (time.perf_counter() - t0)

    print("[bench] cpu baseline: oracle at 64x64 batch 4 on %d threads ..." % cores, file=sys.stderr, flush=True)
    c1 = run(64, 4, 4, 1, 2)
    # conv FLOPs per pair: 64x64 (num_layers=4) 158.3 GF, 256x256 2549.4 GF
    est_256 = 2.5 * (2549.4 / 158.3) / c1      # measured: large-image steps run ~2.5x slower per FLOP (cache misses)
    out = {"value": c1, "unit": "pairs/s (64x64 pairs)", "cores": cores, "kind": "port",
           "sample": "oracle D+G step, 64x64 batch 4 num_layers=4 (BASELINE.json configs[0]), 1 warm-up + 2 timed steps"}
    if est_256 < 60.0:
        # ~10-20 s of CPU work at the bench's own image size: a warm-up step when steps are short, then 1..4 timed steps
        warm = 1 if est_256 < 8.0 else 0
        timed = max(1, min(4, int(16.0 / est_256)))
        print("[bench] cpu baseline: %d + %d 256x256 batch-1 steps (estimated %.0f s each) ..." % (warm, timed, est_256),
              file=sys.stderr, flush=True)
        v = run(256, 1, 5, warm, timed)
        out.update({"value": v, "unit": "pairs/s", "c1_64px_pairs_per_s": c1,
                    "sample": "oracle D+G step at 256x256 batch 1, %d warm-up + %d timed steps (value); 64x64 batch 4: %.3f pairs/s"
                              % (warm, timed, c1)})
    else:
        out["extrapolated_256px_pairs_per_s"] = 1.0 / est_256
        out["sample"] += "; a 256x256 step was skipped (estimated %.0f s > 60 s budget), FLOP-scaled estimate given" % est_256
    return out


def cpu_baseline_bounded(limit_s=240):
    """Run cpu_baseline() in a CPU-only child process under a hard wall-clock limit, so a slow or oversubscribed host
    can never stall the bench line (the child never touches the GPU)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], stdout=subprocess.PIPE,
                           timeout=limit_s, check=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="", WORLD_SIZE="1"))
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "pairs/s", "cores": None, "kind": "port",
                "sample": "oracle run exceeded the %d s wall-clock limit on this host and was stopped" % limit_s}


def spawn_ranks(n):
    """`python bench.py --gpus N` invoked directly (no launcher set WORLD_SIZE): start the N rank processes -- one per
    GPU -- as CHILDREN of this process, which itself never touches the GPU, hand each the torch.distributed.run
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), and return the worst exit code.
    Rank 0 prints the one JSON line on the shared stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:               # a rank died: its peers would wait in a collective forever
                    q.terminate()
    return rc


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    if args.spawn_selftest:                  # CPU test of the launcher: the ranks find each other and agree on a sum
        dist.init_process_group("gloo")
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        with open(f"{args.spawn_selftest}.{os.environ['RANK']}", "w") as f:
            f.write("%s %s %s %s %g" % (os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], t.item()))
        dist.destroy_process_group()
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if args.force_collectives and world != 1:
        raise SystemExit("--force-collectives prices the reducer on ONE GPU (--gpus 1); with --gpus N > 1 the reducer is attached anyway")
    if world > 1 or args.force_collectives:
        if world == 1:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            sk.close()
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))

    from de_i2i_gan_amd import _lib
    from de_i2i_gan_amd.parallel import attach_ddp
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer

    from de_i2i_gan_amd import ops as _ops
    _ops.fuse_norm = not args.no_fuse_norm
    _ops.fuse_pro = bool(args.fuse_pro)
    _ops.fuse_ring = not args.no_fuse_ring
    _ops.wgrad_side_stream = not args.no_wgrad_stream
    _ops.fuse_bwd = not args.no_fuse_bwd
    _ops.fold_eval_bn = not args.no_fold_eval_bn
    _ops.forked_chains = not args.no_forked_chains
    _ops.paired_passes = not args.no_paired_passes
    for kv in args.set_option:
        name, val = kv.split("=")
        _lib.check(_lib.load().dei2i_set_option(name.encode(), int(val)), "set_option " + kv)
    opt = make_opt(args, device)
    torch.manual_seed(123)                           # reference default (utils/util.py:21); same weights on every rank
    if args.stage == "mae":
        from de_i2i_gan_amd.trainers.mae_trainer import MAETrainer
        tr = MAETrainer(opt)
        step = lambda: tr.step(bg, lab)                      # noqa: E731
    else:
        tr = DefectGanTrainer(opt)
        step = lambda: tr.step(bg, lab, df)                  # noqa: E731
    bg, lab, df = synthetic_batch(args.batch, args.image_size, seed=7 + rank)
    bg, lab, df = bg.to(device), lab.to(device), df.to(device)
    plain_ms = None
    if args.force_collectives:
        # the same steps WITHOUT the reducer first, same process, same box: what the hooks, buckets and stream joins cost is the
        # difference (the one-rank all-reduces themselves move no data over xGMI)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        plain_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    red = attach_ddp(tr, measure=not args.ddp_no_measure, force_collectives=args.force_collectives, comm_dtype=args.comm_dtype,
                     overlap=not args.ddp_no_overlap) \
        if (world > 1 or args.force_collectives) else None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        print("[bench] model built; %d warm-up + %d timed steps ..." % (args.warmup, args.steps), file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    sync()
    if red is not None:
        red.overlap_report()                          # drop the warm-up steps' events
    # Inside the timed region only the DOMINANT kernel (the halo-resident 3x3 conv's forward launches: 65 per step, the
    # step's largest single kernel) is bracketed with HIP events, and only every 4th of its launches (4 is coprime to 65, so over
    # the timed steps every launch of the step's sequence is sampled equally often) -- two event records per launch break the
    # back-to-back dispatch of the stream: bracketing all ~600 conv launches of a step cost 2.4 ms of a 44.2 ms step, all 126
    # halo launches 0.9 ms (measured round 2, same box).  The same kernel's FOLD launches (the reflect dgrads, 56 per step) are
    # a family of their own: they run during the backward passes, BESIDE the weight-gradient kernels of ops' side stream, so
    # an event bracket around one of them spans two kernels sharing the GPU and says nothing about either.
    # The other conv families only count launches and FLOPs there (no events) and are timed in a few EXTRA steps afterwards.
    HALO_SAMPLE = 4
    fams = (("gather_gemm", _lib.PROF_GATHER_GEMM), ("wgrad", _lib.PROF_WGRAD), ("halo", _lib.PROF_HALO_CONV),
            ("halo_fold", _lib.PROF_HALO_FOLD))

    def collect():
        """-> {family: (launches, FLOPs of all launches, event-bracketed launches, their ms, their FLOPs)}"""
        out = {}
        for name, fid in fams:
            tn, tfl = ctypes.c_int64(), ctypes.c_double()
            _lib.check(lib.dei2i_prof_collect_timed(fid, ctypes.byref(tn), ctypes.byref(tfl)), "prof_collect_timed")
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            _lib.check(lib.dei2i_prof_collect(fid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "prof_collect")
            out[name] = (n.value, fl.value, tn.value, ms.value, tfl.value)
            lib.dei2i_prof_enable(fid, 0)
        return out

    if not args.no_roofline:
        for name, fid in fams:
            lib.dei2i_prof_enable(fid, HALO_SAMPLE if name == "halo" else 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = time.perf_counter() - t0         # the host's share: every launch of the timed steps is enqueued, nothing awaited
    sync()
    elapsed = time.perf_counter() - t0
    ddp = red.overlap_report() if red is not None else None
    red_stats = dict(red.stats) if red is not None else None      # before the extra steps below add to them
    fam, fam_extra, extra_steps = {}, {}, 0
    if not args.no_roofline:
        fam = collect()
        # kernel-alone figures: a few EXTRA steps after the timed region with events on every conv launch and the weight
        # gradients on the MAIN stream -- with the side stream a bracket around a backward conv spans that kernel and a wgrad
        # kernel sharing the GPU and prices neither
        extra_steps = min(3, args.steps)
        _ops.wgrad_side_stream = False
        _ops.forked_chains = False                      # (one stream: nothing runs beside the bracketed kernel)
        for name, fid in fams:
            lib.dei2i_prof_enable(fid, 1)
        for _ in range(extra_steps):
            step()
        sync()
        fam_extra = collect()
        _ops.wgrad_side_stream = not args.no_wgrad_stream
        _ops.forked_chains = not args.no_forked_chains
    if hasattr(tr, "flush_losses"):
        tr.flush_losses()
    if rank == 0:
        print("[bench] timed region done: %.2f ms/step" % (1e3 * elapsed / args.steps), file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    pairs = args.batch * world * args.steps
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        "metric": (f"paired {args.image_size}x{args.image_size} images/sec (G+D train step)" if args.stage == "defectgan" else
                   f"{args.image_size}x{args.image_size} images/sec (MAE-GAN pre-training D+G step)"), "value": pairs / elapsed, "unit": "pairs/s" if args.stage == "defectgan" else "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype if args.dtype != "fp8" else "fp8-e4m3 forward GEMMs of the 3x3 convs + bf16", "data": "synthetic",
        "config": {"workload": (f"defectGAN D+G train step, {args.image_size}x{args.image_size} paired RGB, "
                                f"batch {args.batch}/GPU, ngf=ndf=64 num_res=6 num_layers={opt.num_layers} SPADE, Adam(0.5,0.999)"
                                if args.stage == "defectgan" else
                                f"MAE-GAN pre-training D+G step (mask_ratio 0.75, patch 8, position mask token), "
                                f"{args.image_size}x{args.image_size} RGB, batch {args.batch}/GPU, ngf=ndf=64 num_res=6 "
                                f"num_layers={opt.num_layers} SPADE, AdamW(0.9,0.95) through GradScaler"),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "spade_path": "collapsed (5x5 border-class gamma/beta for 1x1 label maps)",
                   "use_spectral": bool(args.use_spectral), "add_noise": bool(args.add_noise),
                   "conv_epilogue_statistics": not args.no_fuse_norm, "operand_path_norm": bool(args.fuse_pro),
                   "spade_upsample_at_source_resolution": not args.no_fuse_ring,
                   "weight_gradients_on_side_stream": not args.no_wgrad_stream,
                   "norm_backward_reductions_in_dgrad_epilogue": not args.no_fuse_bwd,
                   "eval_batchnorm_folded_into_conv_weights": not args.no_fold_eval_bn,
                   "generator_passes_paired_over_2x_batch": not args.no_paired_passes and not args.use_spectral and not args.add_noise,
                   "generator_chains_on_two_streams": (args.no_paired_passes and not args.no_forked_chains and not args.use_spectral
                                                       and not args.add_noise),
                   # what the step does differently from a literal transcription of the reference's step (same function):
                   "defer_loss_sync": bool(opt.defer_loss_sync),      # losses stay on the device; no .item() per update
                   "discriminator_passes": "one batched D pass per step phase (4 image batches in the D step, 2 in the G step: "
                                           "D has no batch statistics)" if not args.use_spectral else "one D pass per image batch (spectral norm iterates u, v per call)",
                   "eval_generator_passes_in_d_step": "one batched eval-mode G pass (2 x batch; running BatchNorm statistics)",
                   "d_weight_gradients_in_g_step": "skipped (the reference computes and discards them)"},
        "losses_last_step": {k: round(v[-1], 5) for kind in tr.losses.values() for k, v in kind.items() if v},
    }
    if fam:
        hn, hfl, htn, hms, htfl = fam["halo"]            # the dominant kernel: the halo-resident stride-1 3x3 conv, forward launches
        fn_, ffl = fam["halo_fold"][:2]                  # its FOLD launches (reflect dgrads; timed region: counts only)
        on, ofl = fam["gather_gemm"][:2]                 # the other conv forward / dgrad kernels (timed region: counts only)
        on, ofl = on + fn_, ofl + ffl
        wn, wfl = fam["wgrad"][:2]
        xhn, _, _, xhms, xhfl = fam_extra["halo"]        # all families with events on every launch, extra steps after the timed region
        xfn, _, _, xfms, xffl = fam_extra["halo_fold"]
        xon, _, _, xoms, xofl = fam_extra["gather_gemm"]
        xon, xoms, xofl = xon + xfn, xoms + xfms, xofl + xffl
        xwn, _, _, xwms, xwfl = fam_extra["wgrad"]
        peak = PEAK_TFLOPS[args.dtype]
        ach = htfl / (hms * 1e-3) / 1e12 if hms > 0 else 0.0
        traffic, traffic_src, mfma_pmc, traffic_fold, mfma_pmc_fold = None, None, None, None, None
        if args.dtype == "bf16" and args.image_size == 256 and args.batch == 16:
            for tag in ("r03", "r02", "r01_g"):           # newest committed rocprofv3 --pmc summaries (profiles/)
                pmc = os.path.join(REPO, "profiles", tag + "_pmc_traffic.json")
                if traffic is None and os.path.exists(pmc):
                    with open(pmc) as f:
                        t = json.load(f)
                    hk = next((k for k in ("halo16_conv_fwd", "halo16_conv", "halo_conv") if k in t), None)
                    if hk in t:
                        traffic = t[hk]["hbm_bytes_per_launch"]
                        traffic_fold = t.get("halo16_conv_fold", {}).get("hbm_bytes_per_launch")
                        traffic_src = ("profiles/%s_pmc_traffic.json (profiles/collect.sh + summarize.py): (2*FETCH_SIZE + WRITE_SIZE)"
                                       "*1024 bytes per launch of the kernel instance, separate --pmc passes" % tag)
                pmc2 = os.path.join(REPO, "profiles", tag + "_pmc_mfma.json")
                if mfma_pmc is None and os.path.exists(pmc2):
                    with open(pmc2) as f:
                        t2 = json.load(f)
                    mfma_pmc = t2.get("halo16_conv_fwd", t2.get("halo16_conv", t2.get("halo_conv", {}))).get("mfma_busy_frac")
                    mfma_pmc_fold = t2.get("halo16_conv_fold", {}).get("mfma_busy_frac")
        xn, xms, xfl = xhn + xon, xhms + xoms, xhfl + xofl

        def inst(n, ms, fl, what):
            a = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"kernel": what, "achieved": a, "frac": a / peak, "launches_per_step": n / max(extra_steps, 1),
                    "avg_launch_ms": ms / max(n, 1), "flops_per_launch": fl / max(n, 1), "ms_per_step": ms / max(extra_steps, 1)}
        K_FOLD = ("halo16_conv_kernel<128, 8, 0, true, 0, false, true> (halo-resident stride-1 3x3 conv, 16x32-pixel tiles: the FOLD "
                  "launches = input gradients of the reflect-padded convs, with the reflect ring and the SPADE / BatchNorm backward "
                  "reductions in the epilogue)")
        K_FWD = ("halo16_conv_kernel<128, 8, 0, false, 0, true, false> (the same kernel's forward / zero-boundary launches, "
                 "software-pipelined loop)")
        fold_i, fwd_i = inst(xfn, xfms, xffl, K_FOLD), inst(xhn, xhms, xhfl, K_FWD)
        dom, other = (fold_i, fwd_i) if xfms >= xhms else (fwd_i, fold_i)
        fam_ach = (xhfl + xffl) / ((xhms + xfms) * 1e-3) / 1e12 if (xhms + xfms) > 0 else 0.0
        line["roofline"] = {"bound": "mfma", "kernel": dom["kernel"] + " -- the instance with the most device time per step",
                            "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dom["frac"],
                            "traffic": traffic if dom is fwd_i else traffic_fold, "traffic_source": traffic_src,
                            "mfma_busy_frac_pmc": mfma_pmc if dom is fwd_i else mfma_pmc_fold,   # same kernel, counters of a separate profiled run (profiles/)
                            "launches_per_step": dom["launches_per_step"], "avg_launch_ms": dom["avg_launch_ms"],
                            "flops_per_launch": dom["flops_per_launch"], "ms_per_step": dom["ms_per_step"],
                            "timing": "kernel-alone: HIP events on the launch stream around EVERY launch in %d extra steps after the timed "
                                      "region, run on ONE stream (weight gradients on the main stream, the generator chains not forked: with "
                                      "side streams a bracket around a conv spans two kernels sharing the GPU); "
                                      "profiles/r03_serial_kernel_stats.csv is the rocprofv3 table of the same serial configuration" % extra_steps,
                            "other_instance": other,
                            "halo16_family_flop_weighted": {"achieved": fam_ach, "frac": fam_ach / peak,
                                                            "launches_per_step": (xhn + xfn) / max(extra_steps, 1)},
                            "forward_instance_in_timed_region": {
                                "achieved": ach, "frac": ach / peak, "launches_per_step": hn / args.steps,
                                "avg_launch_ms": hms / max(htn, 1), "flops_per_launch": htfl / max(htn, 1),
                                "event_bracketed_launches": htn, "traffic": traffic, "mfma_busy_frac_pmc": mfma_pmc,
                                "timing": "HIP events around every %dth forward launch of halo16_conv_kernel inside the timed region, "
                                          "on the launch stream (%d of %d launches)" % (HALO_SAMPLE, htn, hn)},
                            "all_conv_fwd_dgrad_kernels": {"achieved": xfl / (xms * 1e-3) / 1e12 if xms > 0 else 0.0,
                                                           "launches_per_step": xn / max(extra_steps, 1),
                                                           "avg_launch_ms": xms / max(xn, 1),
                                                           "timing": "%d extra serial steps after the timed region" % extra_steps}}
        conv_flops_step = (hfl + ofl + wfl) / args.steps
        line["mfma"] = {"executed_conv_tflop_per_step": conv_flops_step / 1e12,
                        "step_mfma_util": conv_flops_step / (ms_per_step * 1e-3) / (peak * 1e12),
                        "conv_launches_per_step": (hn + on + wn) / args.steps,
                        "wgrad_tflops": xwfl / (xwms * 1e-3) / 1e12 if xwms > 0 else 0.0,
                        "conv_kernel_time_over_step_time": (xms + xwms) / max(extra_steps, 1) / ms_per_step,
                        "timing": "FLOPs counted in the timed region; wgrad / all-conv times kernel-alone from %d extra serial steps "
                                  "after it (weight gradients on the main stream)" % extra_steps}
    if ddp is not None:
        # rank 0's view: time the gradient all-reduces occupied the side stream per backward pass, and the part that ran
        # after backward's last kernel (the optimizer waits for it) -- stream events, measured inside the timed region
        ddp.update({"collectives_per_step": red_stats["collectives"] / (args.steps + args.warmup),
                    "allreduce_mb_per_step": red_stats["bytes"] / (args.steps + args.warmup) / 1e6,
                    "host_ms_per_step_in_reduce": red_stats.get("reduce_host_ms", 0.0) / (args.steps + args.warmup),
                    "host_ms_per_step_issuing_collectives": red_stats.get("launch_host_ms", 0.0) / (args.steps + args.warmup),
                    "comm_dtype": args.comm_dtype})
        if plain_ms is not None:
            ddp.update({"ranks": 1, "step_ms_without_reducer": plain_ms, "step_ms_with_reducer": ms_per_step,
                        "reducer_cost_ms_per_step": ms_per_step - plain_ms,
                        "note": "one-rank RCCL group (--force-collectives): the reducer's host and stream work at full size, no xGMI traffic"})
        line["ddp"] = ddp
    if world == 1 and not args.no_cpu_baseline and not args.force_collectives:
        line["cpu_baseline"] = cpu_baseline_bounded()
    print(json.dumps(line))
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
