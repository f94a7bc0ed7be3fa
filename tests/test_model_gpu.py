"""GPU parity of the product's Generator / Discriminator / Trainer (HIP kernels through the C ABI) against the golden
fixtures captured from the reference (tests/golden/gen_goldens.py) and against the oracle restatement.

Tolerances
  f32 mode (exact-f32 MFMA)  -- forward <= 1e-3 of the tensor max (north_star's bound; measured ~5e-6), losses 1e-4,
                                gradients vs the oracle in fp64, same branches (KinkTape) <= 1e-3 relative L2 per tensor (measured ~1e-4).
  bf16 mode (bf16 MFMA, fp32 accumulate) -- bf16 rounds every stored activation to 2^-9.  On the formula-filled tiny
      nets of the goldens (He-gain weights, 8..16 channels) that noise is amplified by the normalisation layers:
      forward RMS error <= 0.12 (measured 4e-2..9e-2), step-1 losses 2e-2, step-2 losses 0.2.  With the reference's real init (N(0,0.02)) bf16 tracks the f32 path to
      1e-4 on the losses and > 0.995 cosine on the full gradient -- checked in test_bf16_tracks_f32_with_reference_init.
  Quantities behind an Adam update use the noise floors recorded with the goldens (Adam's first steps are sign-like);
  in addition the t0 golden's G step sits behind a D update that leaves a LeakyReLU pre-activation of the 2x2 final D
  feature map within fp32 rounding of 0 for `fake_defects`, so the golden G grad norms are only matched to 15%
  (diagnosed with tools/diag_gstep.py: with identical D inputs the HIP gradient equals the fp64 oracle to 3e-6)."""
import functools
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import record_kinks, unbatch_discriminator, unpair_generator, formula_fill, load_golden, make_opt
from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def maxrel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def rmsrel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12)).item()


def check_fwd(got, ref, pname, bf16_rms=0.12):
    if pname == "f32":
        assert maxrel(got, ref) < 1e-3
    else:
        assert rmsrel(got, ref) < bf16_rms and maxrel(got, ref) < 4.2 * bf16_rms


def build(c, pname, **over):
    from de_i2i_gan_amd import ops
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    ops.noise_source = O.shape_noise if c.get("add_noise") else None        # the goldens' stand-in for the N(0,1) draw
    tr = DefectGanTrainer(make_opt(c, DEV, pname, **over))
    formula_fill(tr.model.netG)
    formula_fill(tr.model.netD)
    return tr


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["t0_img32_b2", "t1_img64_b4", "t2_img64_s3_b2", "t3_img32_b2_sn_noise"])
def test_forward_matches_reference_goldens(name, pname):
    meta, arr, c, cfg = load_golden(name)
    tr = build(c, pname)
    G, D = tr.model.netG, tr.model.netD
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    bg_d, lab_d = bg.to(DEV), labels.to(DEV)
    # bf16 on the formula-filled nets: the three-scale t2 net normalises over an 8x8 bottleneck (64 pixels per
    # instance-norm channel), which amplifies the 2^-9 activation rounding further than t0/t1 (measured 0.18 RMS)
    # (t3: spectral normalisation keeps the bf16 rms error small -- 0.08 -- but single output pixels deviate more)
    check = functools.partial(check_fwd, bf16_rms=0.3 if (c.get("num_scales", 2) == 3 or c.get("use_spectral")) else 0.12)
    _forward_checks(G, D, tr, arr, c, bg, bg_d, lab_d, pname, check)


def _forward_checks(G, D, tr, arr, c, bg, bg_d, lab_d, pname, check_fwd):
    with torch.no_grad():
        G.eval()
        D.eval()                          # as in the golden script (spectral norm: no power iteration in eval mode)
        out, prob = G(bg_d, lab_d.reshape(c["batch"], 6, 1, 1))
        assert out.shape == (c["batch"], 3, c["image_size"], c["image_size"]) and out.dtype == torch.float32
        check_fwd(out, arr["G_out_eval"], pname)
        check_fwd(prob, arr["G_prob_eval"], pname)
        src, cls = D(torch.from_numpy(arr["G_out_eval"]).to(DEV))
        assert src.shape == arr["D_src"].shape and cls.shape == arr["D_cls"].shape
        check_fwd(src, arr["D_src"], pname)
        check_fwd(cls, arr["D_cls"], pname)
        out_s, prob_s = tr.model("inference", bg, torch.from_numpy(arr["seg22"]))       # spatial (N,6,2,2) labels
        check_fwd(out_s, arr["G_out_spatial"], pname)
        check_fwd(prob_s, arr["G_prob_spatial"], pname)
        saved = {k: v.clone() for k, v in G.state_dict().items()}
        G.train()
        out_t, prob_t = G(bg_d, lab_d)                      # (N,6) labels are accepted too
        check_fwd(out_t, arr["G_out_train"], pname)
        check_fwd(prob_t, arr["G_prob_train"], pname)
        assert int(G.state_dict()["stem.conv_block.1.num_batches_tracked"]) == 1
        G.load_state_dict(saved)


@pytest.mark.parametrize("name,c,gtol", [
    ("tiny16", dict(image_size=16, batch=1, num_layers=1, ngf=8, ndf=8, hidden_nc=8), 1e-3),
    ("t0", dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16), 1e-3),
    # three scales: the backward chain is two blocks longer and crosses batch / instance norms over 8x8 maps, whose
    # backward amplifies fp32 rounding.  Branches are replayed, so what is left is linear rounding: the fp32 CPU
    # restatement itself, replayed the same way, sits up to 1.1e-2 from its fp64 run on this config (3e-4 on t0);
    # the HIP f32 path measures 3.4e-3
    ("t2_scales3", dict(image_size=64, batch=2, num_layers=4, ngf=8, ndf=8, hidden_nc=16, num_scales=3), 1.5e-2),
])
def test_step_gradients_match_oracle_fp64(name, c, gtol):
    """Every live parameter gradient of the D loss graph and of the G loss graph (4 chained G passes + 2 D passes),
    f32 HIP path vs the oracle evaluated in fp64 on the same weights and the same piecewise-linear branches: the
    branch the HIP path took at every ReLU / LeakyReLU / |a-b| is recorded and replayed by the oracle (KinkTape), and
    the elements where that differs from the oracle's own branch must all sit within fp32 noise of the kink."""
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"],
                num_scales=c.get("num_scales", 2))
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    SG = {k: (v.double() if v.is_floating_point() else v) for k, v in O.make_state(O.generator_state_shapes(cfg)).items()}
    SD = {k: v.double() for k, v in O.make_state(O.discriminator_state_shapes(cfg)).items()}
    tr = build(c, "f32")
    G, D = tr.model.netG, tr.model.netD
    with record_kinks() as d_tape:
        gan, clf = tr.model("discriminator", bg, labels, df)
    (gan + 2 * clf).backward()
    with record_kinks() as g_tape:
        ls = tr.model("generator", bg, labels, df)
    (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()

    def oracle(fn, decisions, *a):
        O.KINK_TAPE = tape = O.KinkTape(decisions)
        try:
            out = fn(*a)
        finally:
            O.KINK_TAPE = None
        assert tape.exhausted() and tape.sites == len(decisions)
        assert tape.worst < 2e-3, (tape.flips, tape.worst)      # replayed branches differ only within noise of a kink
        return out

    d_tape = unbatch_discriminator(d_tape, 4, c["batch"])        # product: one D pass over 4 batches; oracle: 4 passes
    if tr.model._pairs_generator_passes(bg.to("cuda:0"), df.to("cuda:0"), None):
        g_tape = unpair_generator(g_tape, c["batch"])            # product: two G passes over 2 x batch; oracle: four passes
    g_tape = unbatch_discriminator(g_tape, 2, c["batch"])
    d_gan, d_clf, gD = oracle(O.train_discriminator_once, d_tape, SG, SD, None, bg.double(), labels.double(), df.double(), cfg)
    g_losses, gG = oracle(O.train_generator_once, g_tape, {k: v.clone() for k, v in SG.items()},
                          {k: v.detach() for k, v in SD.items()}, None, bg.double(), labels.double(), df.double(), cfg)
    assert abs(float(gan) - float(d_gan)) < 1e-5 and abs(float(clf) - float(d_clf)) < 1e-5
    for k, p in D.named_parameters():
        assert ((p.grad.double().cpu() - gD[k]).norm() / gD[k].norm()).item() < 1e-4, k
    for a, b in zip(ls, g_losses):
        assert abs(float(a) - float(b)) < 1e-5
    scale = max(float(v.norm()) for v in gG.values() if v is not None)
    for k, p in G.named_parameters():
        if gG[k] is None:
            assert p.grad is None, k                      # never-executed norm_s / conv_s
            continue
        ref = gG[k]
        if float(ref.norm()) < 1e-7 * scale:              # gradients that are zero by construction (bias before IN)
            assert float(p.grad.double().norm()) < 1e-4 * scale, k
            continue
        assert ((p.grad.double().cpu() - ref).norm() / ref.norm()).item() < gtol, k


@pytest.mark.parametrize("pname", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["t0_img32_b2", "t1_img64_b4", "t2_img64_s3_b2", "t3_img32_b2_sn_noise", "t4_img32_b2_diffaug",
                                  "t7_img32_b2_cycle"])
def test_two_train_steps_match_reference_goldens(name, pname):
    meta, arr, c, cfg = load_golden(name)
    tr = build(c, pname)
    G, D = tr.model.netG, tr.model.netD
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    ltol = {"f32": 1e-4, "bf16": 6e-2 if c.get("num_scales", 2) == 3 else 2e-2}[pname]     # t2 bf16: measured 4e-2
    for it in range(2):
        torch.manual_seed(meta.get("step_seed", 0) + it)        # DiffAugment draws from the host RNG like the reference
        tr._train_discriminator_once(bg, labels, df)            # CPU tensors in, like the reference's loaders
        if it == 0:
            dn = np.array([float(p.grad.double().norm()) for _, p in D.named_parameters()])
            assert [k for k, _ in D.named_parameters()] == meta["D_grad_keys"]
            # bf16: D sees G's fake images, i.e. the bf16 forward noise of the formula-filled G (0.12 RMS, 0.3 on t2)
            dtol = 2e-3 if pname == "f32" else (0.3 if (c.get("num_scales", 2) == 3 or c.get("use_spectral")) else 0.15)
            assert np.max(np.abs(dn - arr["D_grad_norms_step1"]) / arr["D_grad_norms_step1"]) < dtol
        tr._train_generator_once(bg, labels, df)
        if it == 0:
            ref = arr["G_grad_norms_step1"]
            gn = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in G.named_parameters()])
            assert [k for k, _ in G.named_parameters()] == meta["G_grad_keys"]
            assert ((gn < 0) == (ref < 0)).all(), "grad-is-None pattern (never-executed norm_s / conv_s)"
            if pname == "f32":
                m = ref > 1e-4
                assert np.max(np.abs(gn[m] - ref[m]) / ref[m]) < max(0.15, c.get("tol_gradnorm", 0))   # see module docstring
            else:
                # coarse guard (see the step-2 comment below); the tight bf16 gradient bounds are the reference-init tests
                assert np.isfinite(gn).all()
                m = ref > 1e-4 * ref.max()
                worst = float(np.max(np.abs(np.log(gn[m] / ref[m]))))          # measured: up to log 3.1 (t7), log 5 on the three-scale t2
                assert worst < math.log(8.0), worst
        L = tr.losses
        got = [L["gan"]["D"][-1], L["clf"]["D"][-1], L["gan"]["G"][-1], L["clf"]["G"][-1], L["aux"]["rec"][-1],
               L["aux"]["cyc"][-1], L["aux"]["con"][-1]]
        # step 2 sits behind two sign-like Adam updates.  f32 keeps the noise floor recorded with the golden.  bf16 on these
        # formula-filled nets (He-gain weights, 8..16 channels) is a COARSE guard only: the 2^-9 rounding of every stored
        # activation grows ~4x per res-block layer there (tools/diag_first_use.py), so against the fp32 golden step-2 losses sit
        # 0.05..0.22 off (the runs themselves are bit-reproducible: tests/test_full_size_gpu.py).  The tight bf16 bound over two
        # steps -- every BatchNorm buffer, post-Adam parameters, Adam moments -- is test_bf16_two_steps_track_f32_* below, at the
        # reference's init and default widths.
        tol = ltol if it == 0 else max(c["tol_step2"], 2e-2 if pname == "f32" else 0.4)
        assert maxrel(np.array(got), arr["losses"][it]) < tol, (it, got, arr["losses"][it].tolist())
    # post-step state: Adam moved the parameters, BatchNorm running stats tracked 8 train-mode forwards
    keys, n = meta["D_check_keys"], arr["D_post_norm"]
    sd = D.state_dict()
    mine = np.array([float(sd[k].double().norm()) for k in keys])
    assert maxrel(mine, n) < 1e-3
    sdg = G.state_dict()
    assert int(sdg["stem.conv_block.1.num_batches_tracked"]) == 8
    for k in meta["G_keys"]:
        if "running_" in k:
            if pname == "f32":
                assert maxrel(sdg[k], arr["bn::" + k]) < c.get("tol_running", 5e-2), k
            elif k.startswith(("stem.", "enc_blk.0.", "enc_blk.1.")):
                # bf16, coarse guard: on the formula-filled nets the 2^-9 activation rounding grows ~4x per res-block layer
                # (tools/diag_first_use.py t1: 1e-4 -> 3e-2 over five layers), which swamps the deep res-block statistics; the
                # first three BatchNorms are comparable with the fp32 golden (measured <= 0.09).  EVERY BatchNorm buffer is held
                # tightly against the f32 mode in test_bf16_two_steps_track_f32_* (reference init, default widths).
                assert maxrel(sdg[k], arr["bn::" + k]) < 0.2, k
    for k in ("enc_blk.0.conv_block.0.weight", "src_clf.conv_block.0.weight"):
        mine = sd[k] if k in sd else sd[k + "_orig"]          # spectral convs keep their parameter under key + "_orig"
        d = (mine.cpu() - torch.from_numpy(arr["Dp::" + k])).abs()
        assert d.max().item() <= 5 * cfg.lr          # sign-like early Adam steps (|update| <= ~1.1 lr each): see test_oracle_goldens.py


def test_bf16_tracks_f32_with_reference_init():
    """The reference's own init (N(0, 0.02), base_network.py:27-56): bf16 path vs f32 path, same seed."""
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=64, batch=4, num_layers=4, ngf=16, ndf=16, hidden_nc=32)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    res = {}
    for pname in ("f32", "bf16"):
        torch.manual_seed(123)
        tr = DefectGanTrainer(make_opt(c, DEV, pname))
        G, D = tr.model.netG, tr.model.netD
        g1, c1 = tr.model("discriminator", bg, labels, df)
        (g1 + 2 * c1).backward()
        dgr = torch.cat([p.grad.double().flatten() for p in D.parameters()])
        ls = tr.model("generator", bg, labels, df)
        (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
        ggr = torch.cat([p.grad.double().flatten() for p in G.parameters() if p.grad is not None])
        res[pname] = ([float(g1), float(c1)] + [float(x) for x in ls], dgr, ggr)
    a, b = res["f32"], res["bf16"]
    assert maxrel(np.array(b[0]), np.array(a[0])) < 1e-3
    for i in (1, 2):
        cos = float(torch.dot(a[i], b[i]) / (a[i].norm() * b[i].norm()))
        assert cos > 0.995, (i, cos)


def test_bf16_two_steps_track_f32_at_the_reference_init_and_default_widths():
    """bf16 mode against the exact-f32 mode (the one the reference-generated goldens pin to 1e-4) over TWO full trainer steps
    (D update, G update, D update, G update) at the reference's init (N(0, 0.02), base_network.py:27-56) and DEFAULT widths
    (ngf = ndf = 64, hidden_nc = 128), 64 x 64, batch 4 -- step 2 runs on Adam-updated parameters and updated BatchNorm running
    statistics.  Runs are bit-reproducible, so these are properties of bf16 arithmetic, not noise bands.  Bounds (measured values
    are written to gpurun_out/bf16_vs_f32_two_steps.json):
      * the 7 losses of step 1: 1e-3 relative (measured 2.3e-4); of step 2: 1e-2 (measured 4.7e-3 -- behind two sign-like Adam
        updates: |update| ~ lr per element whatever the gradient's size, so a bf16 sign flip of a near-zero gradient element moves
        a weight by 2 lr = 2 % of the init's sigma);
      * EVERY BatchNorm running_mean / running_var buffer after the two steps (8 train-mode forwards): 1.5e-2 of the buffer's max
        (measured: means <= 5.5e-3, variances <= 2.5e-3);
      * every parameter after two Adam updates: |p_bf16 - p_f32| <= 4.4 lr (two updates, each bounded by ~1.1 lr in either mode;
        measured 4.1), relative L2 of the whole parameter vector 1e-2 (measured 4.6e-3), norms of the weight tensors within 1e-3;
      * Adam's first moments (the gradients' running mean): cosine >= 0.99 per network."""
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=64, batch=4, num_layers=4, ngf=64, ndf=64, hidden_nc=128)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    res = {}
    for pname in ("f32", "bf16"):
        torch.manual_seed(123)
        tr = DefectGanTrainer(make_opt(c, DEV, pname))
        G, D = tr.model.netG, tr.model.netD
        for _ in range(2):
            tr.step(bg, labels, df)
        if hasattr(tr, "flush_losses"):
            tr.flush_losses()
        L = tr.losses
        losses = np.array([[L["gan"]["D"][i], L["clf"]["D"][i], L["gan"]["G"][i], L["clf"]["G"][i], L["aux"]["rec"][i],
                            L["aux"]["cyc"][i], L["aux"]["con"][i]] for i in range(2)])
        sd = {"G." + k: v.detach().double().cpu() for k, v in G.state_dict().items()}
        sd.update({"D." + k: v.detach().double().cpu() for k, v in D.state_dict().items()})
        moments = {}
        for net in ("G", "D"):
            st = tr.optimizers[net].state
            moments[net] = torch.cat([st[p]["exp_avg"].detach().double().cpu().flatten() for p in tr.optimizers[net].param_groups[0]["params"]
                                      if p in st and "exp_avg" in st[p]])
        res[pname] = (losses, sd, moments)
        lr = tr.optimizers["G"].param_groups[0]["lr"]
    (lf, sf, mf), (lb, sb, mb) = res["f32"], res["bf16"]
    report = {"loss_rel_step1": float(np.max(np.abs(lb[0] - lf[0]) / np.maximum(np.abs(lf[0]), 1e-3))),
              "loss_rel_step2": float(np.max(np.abs(lb[1] - lf[1]) / np.maximum(np.abs(lf[1]), 1e-3)))}
    bn = {k: maxrel(sb[k], sf[k]) for k in sf if "running_" in k}
    assert len(bn) == 2 * 9, len(bn)                      # stem, two encoder blocks, six res-block convs
    assert int(sb["G.stem.conv_block.1.num_batches_tracked"]) == int(sf["G.stem.conv_block.1.num_batches_tracked"]) == 8
    params = [k for k in sf if "running_" not in k and "num_batches" not in k]
    dmax = max(float((sb[k] - sf[k]).abs().max()) for k in params)
    va, vb = torch.cat([sb[k].flatten() for k in params]), torch.cat([sf[k].flatten() for k in params])
    report.update({"bn_buffer_maxrel_worst": max(bn.values()), "bn_buffers": bn, "param_max_abs_diff_over_lr": dmax / lr,
                   "param_rel_l2": float((va - vb).norm() / vb.norm()),
                   # (norms of the tensors that START away from zero: conv / linear weights N(0, 0.02), BatchNorm weights N(1, 0.02);
                   #  the zero-initialised biases are 2 Adam steps long and are covered by the absolute bound)
                   "param_norm_rel": max(abs(float(sb[k].norm() / sf[k].norm()) - 1) for k in params
                                         if float(sf[k].norm()) > 1e-2 * math.sqrt(sf[k].numel())),
                   "exp_avg_cos": {n: float(torch.dot(mb[n], mf[n]) / (mb[n].norm() * mf[n].norm())) for n in ("G", "D")}})
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "bf16_vs_f32_two_steps.json").write_text(json.dumps(report, indent=1))
    assert report["loss_rel_step1"] < 1e-3 and report["loss_rel_step2"] < 1e-2, report              # measured 2.3e-4 / 4.7e-3
    assert report["bn_buffer_maxrel_worst"] < 1.5e-2, bn                                            # measured 5.5e-3 (running_var: 2.5e-3)
    assert report["param_max_abs_diff_over_lr"] <= 4.4 and report["param_rel_l2"] < 1e-2 and report["param_norm_rel"] < 1e-3, report   # 4.1 / 4.6e-3
    assert min(report["exp_avg_cos"].values()) > 0.99, report["exp_avg_cos"]


def test_bf16_default_widths_against_the_oracle_with_reference_init():
    """The benchmarked mode at the reference's DEFAULT widths (ngf = ndf = 64, hidden_nc = 128) and the reference's own init
    (N(0, 0.02)), checked against the ORACLE -- the fp32 CPU restatement the goldens pin to the reference -- not against the
    build's own f32 mode: one D loss/backward and one G loss/backward at 64 x 64, batch 4 (what the oracle finishes in
    seconds).  Bounds: the 7 losses 5e-4 relative (measured 4e-5), cosine of the full D and G gradients >= 0.995 (measured
    0.9986 / 0.9990), their norms within 1 % (measured 4e-4 / 1e-3); the measured values go to gpurun_out/."""
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    c = dict(image_size=64, batch=4, num_layers=4, ngf=64, ndf=64, hidden_nc=128)
    cfg = O.Cfg(image_size=64, ngf=64, ndf=64, num_layers=4, hidden_nc=128)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    torch.manual_seed(321)
    tr = DefectGanTrainer(make_opt(c, DEV, "bf16"))
    G, D = tr.model.netG, tr.model.netD
    SG = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    SD = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    assert list(SG) == list(O.generator_state_shapes(cfg)) and list(SD) == list(O.discriminator_state_shapes(cfg))
    g1, c1 = tr.model("discriminator", bg, labels, df)
    (g1 + 2 * c1).backward()
    d_mine = {k: p.grad.double().cpu() for k, p in D.named_parameters()}
    ls = tr.model("generator", bg, labels, df)
    (ls[0] + 5 * ls[1] + 5 * ls[2] + 5 * ls[3] + ls[4]).backward()
    g_mine = {k: p.grad.double().cpu() for k, p in G.named_parameters() if p.grad is not None}
    torch.set_num_threads(16)
    # the oracle's D step leaves G untouched (eval mode); its G loss then runs on the same states as the product's did
    o_gan, o_clf, gD = O.train_discriminator_once(SG, SD, O.AdamState(), bg, labels, df, cfg)
    for k in O.param_keys(SD):
        SD[k].requires_grad_(False)
    o_ls, gG = O.train_generator_once(SG, SD, O.AdamState(), bg, labels, df, cfg)
    mine = np.array([float(g1), float(c1)] + [float(v) for v in ls])
    ref = np.array([float(o_gan), float(o_clf)] + [float(v) for v in o_ls])
    assert maxrel(mine, ref) < 5e-4, (mine.tolist(), ref.tolist())
    report = {"loss_maxrel": maxrel(mine, ref)}
    for tag, got, want in (("D", d_mine, gD), ("G", g_mine, gG)):
        keys = [k for k in got if want.get(k) is not None]
        assert len(keys) == len(got)
        a = torch.cat([got[k].flatten() for k in keys])
        b = torch.cat([want[k].double().flatten() for k in keys])
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        assert cos > 0.995, (tag, cos)
        assert abs(float(a.norm() / b.norm()) - 1) < 1e-2, (tag, float(a.norm()), float(b.norm()))
        report[tag + "_grad_cos"], report[tag + "_grad_norm_ratio"] = cos, float(a.norm() / b.norm())
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "bf16_vs_oracle_default_widths.json").write_text(json.dumps(report, indent=1))


def test_step_wrapper_and_deferred_losses():
    meta, arr, c, cfg = load_golden("t0_img32_b2")
    tr = build(c, "f32", defer_loss_sync=True)
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    tr.step(bg, labels, df)
    assert len(tr.losses["gan"]["D"]) == 0
    tr.flush_losses()
    got = [tr.losses["gan"]["D"][0], tr.losses["clf"]["D"][0], tr.losses["gan"]["G"][0], tr.losses["clf"]["G"][0],
           tr.losses["aux"]["rec"][0], tr.losses["aux"]["cyc"][0], tr.losses["aux"]["con"][0]]
    assert maxrel(np.array(got), arr["losses"][0]) < 1e-4


def test_ddp_reducer_on_gpu_single_rank_rccl():
    """One-rank RCCL group on the GPU: the hook-driven side-stream all-reduce path (forced on) must leave the step's
    numbers unchanged.  Multi-rank correctness is covered on CPU/gloo (tests/test_ddp_gloo.py)."""
    import os
    import torch.distributed as dist
    from de_i2i_gan_amd.parallel import attach_ddp
    meta, arr, c, cfg = load_golden("t0_img32_b2")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        tr = build(c, "f32")
        red = attach_ddp(tr, force_collectives=True, bucket_bytes=1 << 14, direct_bytes=1 << 12)
        bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
        tr.step(bg, labels, df)
        got = [tr.losses["gan"]["D"][0], tr.losses["clf"]["D"][0], tr.losses["gan"]["G"][0], tr.losses["clf"]["G"][0],
               tr.losses["aux"]["rec"][0], tr.losses["aux"]["cyc"][0], tr.losses["aux"]["con"][0]]
        assert maxrel(np.array(got), arr["losses"][0]) < 1e-4
        tr.step(bg, labels, df)
        torch.cuda.synchronize()
        assert red.stats["collectives"] > 10
        sd = tr.model.netD.state_dict()
        mine = np.array([float(sd[k].double().norm()) for k in meta["D_check_keys"]])
        assert maxrel(mine, arr["D_post_norm"]) < 1e-3
    finally:
        dist.destroy_process_group()

@pytest.mark.parametrize("c", [
    dict(image_size=80, batch=3, num_layers=3, ngf=16, ndf=16, hidden_nc=32),      # 80 = 5*16: no 8x32 tiling anywhere
    dict(image_size=48, batch=1, num_layers=2, ngf=24, ndf=8, hidden_nc=8),        # batch 1, channel counts off the 16s
    dict(image_size=32, batch=2, num_layers=3, ngf=12, ndf=12, hidden_nc=8),       # BatchNorm widths below the padded channel stride (12 -> 16 in bf16)
])
def test_ragged_sizes_forward_matches_oracle(c):
    """Generator / Discriminator modules on shapes the tile-aligned kernels refuse (image sides that are not multiples of
    the 8x32 tile -- the Model class itself asserts a power of two like the reference, defectgan_model.py:23, the network
    classes do not -- odd batches, channel counts off the multiples of 16): the generic kernels must give the oracle's
    numbers (f32 1e-3), and the bf16 kernels run the same shapes."""
    from de_i2i_gan_amd.networks.generator import DefectGanGenerator
    from de_i2i_gan_amd.networks.discriminator import DefectGanDiscriminator
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"])
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    SG, SD = O.make_state(O.generator_state_shapes(cfg)), O.make_state(O.discriminator_state_shapes(cfg))
    seg = labels.reshape(c["batch"], 6, 1, 1)
    with torch.no_grad():
        o_out, o_prob = O.generator_forward(SG, bg, seg, cfg, training=False)
        o_src, o_cls = O.discriminator_forward(SD, o_out, cfg)
    for pname in ("f32", "bf16"):
        opt = make_opt(c, DEV, pname)
        G, D = DefectGanGenerator(opt).to(DEV), DefectGanDiscriminator(opt).to(DEV)
        formula_fill(G)
        formula_fill(D)
        with torch.no_grad():
            G.eval()
            out, prob = G(bg.to(DEV), labels.to(DEV))
            src, cls = D(o_out.to(DEV))
        for got, ref in ((out, o_out), (prob, o_prob), (src, o_src), (cls, o_cls)):
            assert got.shape == ref.shape
            check_fwd(got, ref, pname, 0.3)          # bf16 on the formula-filled nets: measured up to 0.22 rms at batch 1


@pytest.mark.parametrize("ngf", [24, 6])        # 6: every BatchNorm width (6, 12, 24) has a padded channel stride in f32 mode too (6 -> 8)
def test_odd_batch_and_channel_counts_step_matches_oracle(ngf):
    """One D step and one G step on a power-of-two image with batch 3 and channel counts that are not multiples of 16
    (f32 mode): step-1 losses 1e-4, D gradients 1e-3 (D's loss graph has no near-tie branches on this seed)."""
    c = dict(image_size=32, batch=3, num_layers=3, ngf=ngf, ndf=8, hidden_nc=8)
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"])
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    tr = build(c, "f32")
    D = tr.model.netD
    SG, SD = O.make_state(O.generator_state_shapes(cfg)), O.make_state(O.discriminator_state_shapes(cfg))
    gan, clf = tr.model("discriminator", bg, labels, df)
    (gan + 2 * clf).backward()
    o_gan, o_clf, gD = O.train_discriminator_once(SG, SD, None, bg, labels, df, cfg)
    assert abs(float(gan) - float(o_gan)) < 1e-4 and abs(float(clf) - float(o_clf)) < 1e-4
    for k, p in D.named_parameters():
        assert ((p.grad.cpu().double() - gD[k].double()).norm() / gD[k].double().norm()).item() < 1e-3, k
    ls = tr.model("generator", bg, labels, df)
    o_ls, _ = O.train_generator_once({k: v.clone() for k, v in SG.items()}, {k: v.detach() for k, v in SD.items()}, None, bg,
                                     labels, df, cfg)
    for a, b in zip(ls, o_ls):
        assert abs(float(a) - float(b)) < 1e-4 * max(1.0, abs(float(b)))
