"""CPU: the stargan-v2 oracle (oracle/starganv2_oracle.py) against the fixture captured from the reference's own core.model /
core.solver (tests/golden/gen_starganv2_golden.py): forward passes of the four networks, the four loss graphs of one training
iteration (incl. the R1 penalty: a double backward through the discriminator), gradient norms, post-Adam parameter norms, EMA."""
import json
from pathlib import Path

import numpy as np
import torch

from oracle import starganv2_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
NAME = "sg0_img64_b2"


def load():
    meta = json.loads((GOLD / f"{NAME}.json").read_text())
    arr = np.load(GOLD / f"{NAME}.npz")
    cfg = O.Cfg(**meta["config"])
    return meta, arr, cfg


def states(cfg):
    shapes = {"generator": O.generator_state_shapes(cfg), "mapping_network": O.mapping_state_shapes(cfg),
              "style_encoder": O.style_encoder_state_shapes(cfg), "discriminator": O.discriminator_state_shapes(cfg)}
    return shapes, {n: O.make_state(s, n + ".") for n, s in shapes.items()}


def test_state_manifest_is_the_references():
    meta, arr, cfg = load()
    shapes, _ = states(cfg)
    for n in shapes:
        assert list(shapes[n]) == meta["keys"][n], n


def test_one_training_iteration_matches_the_reference_fixture():
    meta, arr, cfg = load()
    torch.set_num_threads(8)
    shapes, N = states(cfg)
    N_ema = {n: {k: v.clone() for k, v in N[n].items()} for n in ("generator", "mapping_network", "style_encoder")}
    inputs = O.synthetic_inputs(cfg, meta["batch"])
    x_real, y_org, y_trg, x_ref, x_ref2, z_trg, z_trg2 = inputs
    with torch.no_grad():
        s_map = O.mapping_network(N["mapping_network"], z_trg, y_trg, cfg)
        assert np.abs(s_map.numpy() - arr["s_map"]).max() < 1e-5
        assert np.abs(O.style_encoder(N["style_encoder"], x_ref, y_trg, cfg).numpy() - arr["s_enc"]).max() < 1e-5
        assert np.abs(O.generator(N["generator"], x_real, s_map, cfg).numpy() - arr["x_fake"]).max() < 1e-5
        assert np.abs(O.discriminator(N["discriminator"], x_real, y_org, cfg).numpy() - arr["d_out"]).max() < 1e-6
    opt = {n: O.AdamState() for n in N}
    losses, grads = O.train_iteration(N, N_ema, opt, inputs, cfg)
    order = {"d_latent": ("real", "fake", "reg"), "d_ref": ("real", "fake", "reg"), "g_latent": ("adv", "sty", "ds", "cyc"),
             "g_ref": ("adv", "sty", "ds", "cyc")}
    for tag, keys in order.items():
        got, ref = np.array([losses[tag][k] for k in keys]), arr["losses_" + tag]
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)) < (1e-5 if tag == "d_latent" else 5e-3), (tag, got, ref)
    g0 = grads["d_latent"]["main.0.weight"].numpy()                 # carries the R1 term (lambda_reg 2e5 in the fixture)
    assert np.abs(g0 - arr["d_latent_grad::main.0.weight"]).max() < 1e-4 * np.abs(arr["d_latent_grad::main.0.weight"]).max()
    for n in N:
        mine = np.array([float(N[n][k].detach().double().norm()) for k in shapes[n]])
        assert np.max(np.abs(mine - arr["post_norm_" + n]) / np.maximum(arr["post_norm_" + n], 1e-9)) < 1e-4, n
    for n in N_ema:
        mine = np.array([float(N_ema[n][k].double().norm()) for k in shapes[n]])
        assert np.max(np.abs(mine - arr["ema_norm_" + n]) / np.maximum(arr["ema_norm_" + n], 1e-9)) < 1e-5, n
