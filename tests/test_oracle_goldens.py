"""CPU: the oracle restatement against the golden fixtures captured from the reference itself
(tests/golden/gen_goldens.py).  Tolerances are the measured fp32 noise floors recorded by the generator."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import defectgan_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = ["t0_img32_b2", "t1_img64_b4", "t2_img64_s3_b2", "t3_img32_b2_sn_noise", "t4_img32_b2_diffaug", "t7_img32_b2_cycle"]
# t2: num_scales=3; t3: spectral norm + noise; t4: DiffAugment; t7: --cycle_gan


def load(name):
    meta = json.loads((GOLD / f"{name}.json").read_text())
    arr = np.load(GOLD / f"{name}.npz")        # allow_pickle defaults to False
    c = meta["config"]
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"],
                hidden_nc=c["hidden_nc"], num_scales=c.get("num_scales", 2), use_spectral=c.get("use_spectral", False),
                add_noise=c.get("add_noise", False), diff_aug=c.get("diff_aug", ""), cycle_gan=c.get("cycle_gan", False))
    O.NOISE_SOURCE = O.shape_noise if c.get("add_noise") else None      # the goldens' deterministic stand-in for N(0,1)
    return meta, arr, c, cfg


def close(a, b, rtol, atol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= atol + rtol * np.abs(b).max()


@pytest.mark.parametrize("name", NAMES)
def test_manifest_matches_reference_state_dict(name):
    meta, arr, c, cfg = load(name)
    assert list(O.generator_state_shapes(cfg).keys()) == meta["G_keys"]
    assert list(O.discriminator_state_shapes(cfg).keys()) == meta["D_keys"]
    if not c.get("use_spectral"):
        assert len(meta["G_keys"]) == 133 + 13 * (c.get("num_scales", 2) - 2)   # one more ConvBlock (6) + NormConvBlock (7) per scale
    else:
        assert len(meta["G_keys"]) == 181 and "stem.conv_block.0.weight_u" in meta["G_keys"]   # + u, v per spectral conv, + noise weights


@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference(name):
    meta, arr, c, cfg = load(name)
    torch.set_num_threads(8)
    SG = O.make_state(O.generator_state_shapes(cfg))
    SD = O.make_state(O.discriminator_state_shapes(cfg))
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    seg = labels.reshape(c["batch"], 6, 1, 1)
    with torch.no_grad():
        out, prob = O.generator_forward(SG, bg, seg, cfg, training=False)
        src, cls = O.discriminator_forward(SD, out, cfg)
        out_s, prob_s = O.generator_forward(SG, bg, torch.from_numpy(arr["seg22"]), cfg, training=False)
        out_t, prob_t = O.generator_forward({k: v.clone() for k, v in SG.items()}, bg, seg, cfg, training=True)
    assert close(out, arr["G_out_eval"], 1e-4)
    assert close(prob, arr["G_prob_eval"], 1e-4)
    assert close(src, arr["D_src"], 1e-4)
    assert close(cls, arr["D_cls"], 1e-4)
    assert close(out_s, arr["G_out_spatial"], 1e-4)
    assert close(prob_s, arr["G_prob_spatial"], 1e-4)
    assert close(out_t, arr["G_out_train"], 2e-4)
    assert close(prob_t, arr["G_prob_train"], 2e-4)


def test_two_steps_match_reference_t0():
    meta, arr, c, cfg = load("t0_img32_b2")
    torch.set_num_threads(8)
    SG = O.make_state(O.generator_state_shapes(cfg))
    SD = O.make_state(O.discriminator_state_shapes(cfg))
    bg, labels, df = O.synthetic_batch(c["batch"], c["image_size"])
    stG, stD = O.AdamState(), O.AdamState()
    order = ("d_gan", "d_clf", "g_gan", "g_clf", "g_rec", "g_cyc", "g_con")
    for it in range(2):
        losses, gD, gG = O.step(SG, SD, stG, stD, bg, labels, df, cfg)
        got = [losses[k] for k in order]
        assert close(got, arr["losses"][it], 1e-5 if it == 0 else c["tol_step2"]), (it, got, arr["losses"][it])
        if it == 0:
            ref = arr["D_grad_norms_step1"]
            mine = np.array([float(gD[k].double().norm()) for k in meta["D_grad_keys"]])
            assert close(mine, ref, 2e-4)
            refg = arr["G_grad_norms_step1"]
            mineg = np.array([float(gG[k].double().norm()) if gG[k] is not None else -1.0 for k in meta["G_grad_keys"]])
            assert ((mineg < 0) == (refg < 0)).all()          # never-executed norm_s/conv_s: grad None
            m = refg > 1e-4
            assert np.max(np.abs(mineg[m] - refg[m]) / refg[m]) < 1e-2
    # Adam moved the small tensors the same way.  Early Adam steps are sign-like (each element moves ~lr per
    # step whatever |g|), so an element whose ~0 gradient flips sign under fp32 rounding ends up to 2*lr away
    # per step: bound = 2 steps * 2 * lr; the bulk must agree far tighter than that.
    for k in ("enc_blk.0.conv_block.0.weight", "src_clf.conv_block.0.weight"):
        d = np.abs(SD[k].detach().numpy() - arr["Dp::" + k])
        assert d.max() <= 4 * cfg.lr + 1e-6
        assert np.median(d) < 2e-5
    for k in meta["G_keys"]:
        if "running_" in k:
            assert close(SG[k], arr["bn::" + k], 5e-2)


def test_kink_tape_replay_makes_fp32_and_fp64_gradients_agree():
    """The branch tape (oracle.KinkTape): record the fp32 oracle run's branch at every ReLU / LeakyReLU / |a-b| on the
    gradient path in the HIP recorder's format, replay it in fp64 -> the G-step gradients agree to ~1e-4, where the
    free-running fp64 gradients differ by percents whenever an element sits within rounding noise of a kink."""
    import torch
    cfg = O.Cfg(image_size=16, ngf=8, ndf=8, num_layers=1, hidden_nc=8)
    bg, labels, df = O.synthetic_batch(1, 16)
    SG32, SD32 = O.make_state(O.generator_state_shapes(cfg)), O.make_state(O.discriminator_state_shapes(cfg))
    tape = []
    saved = O.relu, O.leaky_relu, O.l1

    def relu(x):
        y = torch.relu(x)
        if x.requires_grad:                                         # NHWC, channels zero-padded like the HIP layout
            tape.append(("relu", torch.nn.functional.pad(y.detach().permute(0, 2, 3, 1), (0, 3)).double()))
        return y

    def leaky(x):
        y = torch.where(x >= 0, x, 0.2 * x)
        if x.requires_grad:
            tape.append(("leaky", y.detach().permute(0, 2, 3, 1).double()))
        return y

    def l1(a, b, kink=True):
        d = a - b
        if kink and d.requires_grad:
            tape.append(("l1", torch.sign(d.detach()).double()))
        return d.abs().mean()

    O.relu, O.leaky_relu, O.l1 = relu, leaky, l1
    try:
        _, g32 = O.train_generator_once({k: v.clone() for k, v in SG32.items()}, SD32, None, bg, labels, df, cfg)
    finally:
        O.relu, O.leaky_relu, O.l1 = saved
    SG = {k: (v.double() if v.is_floating_point() else v) for k, v in SG32.items()}
    SD = {k: v.double() for k, v in SD32.items()}
    O.KINK_TAPE = t = O.KinkTape(tape)
    try:
        _, g64 = O.train_generator_once({k: v.clone() for k, v in SG.items()}, SD, None, bg.double(), labels.double(),
                                        df.double(), cfg)
    finally:
        O.KINK_TAPE = None
    assert t.exhausted() and t.sites == len(tape) == 64
    assert t.worst < 1e-4                                           # any replayed flip sits on the kink
    worst = max(((g32[k].double() - g64[k]).norm() / g64[k].norm()).item() for k in g64
                if g64[k] is not None and g64[k].norm() > 1e-6)
    assert worst < 2e-3, worst
