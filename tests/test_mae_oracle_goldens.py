"""CPU: the MAE-stage oracle (oracle/mae_oracle.py) against the fixtures captured from the reference's MAETrainer
(tests/golden/gen_mae_goldens.py) -- masks from the same seeded RNG, two D+G iterations."""
import json
import random
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import defectgan_oracle as O
from oracle import mae_oracle as M

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = ["m0_img32_b2_position", "m1_img64_b2_vector", "m2_img32_b2_split",         # m2: --split_training
         "m3_img64_b2_sean_distill"]                                                # m3: SEAN blocks + --style_distill


def load(name):
    meta = json.loads((GOLD / f"{name}.json").read_text())
    arr = np.load(GOLD / f"{name}.npz")
    c = meta["config"]
    sean = dict(style_norm="sean", embed_nc=c["embed_nc"], num_embeds=c["num_embeds"], style_distill=bool(c.get("style_distill"))) \
        if c.get("style_norm") == "sean" else {}
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"], **sean)
    return meta, arr, c, cfg


@pytest.mark.parametrize("name", NAMES)
def test_mae_two_iterations_match_reference(name):
    meta, arr, c, cfg = load(name)
    torch.set_num_threads(8)
    SG, SD = O.make_state(O.generator_state_shapes(cfg)), O.make_state(O.discriminator_state_shapes(cfg))
    token = {"mask_token": O.formula_tensor("mask_token", M.mask_token_shape(c["mask_token_type"], 3, c["image_size"])) * 0.25}
    imgs, labels, _ = O.synthetic_batch(c["batch"], c["image_size"])
    stG, stD = O.AdamState(), O.AdamState()
    sean = cfg.style_norm == "sean"
    O.SEAN_CTX.reset()
    SE = (O.synthetic_embeddings(cfg), random) if sean else None       # the embeddings file of the fixture, drawn with ``random``
    torch.manual_seed(meta["seed"])
    got = []
    for it in range(2):
        split = c.get("split_training", False)          # (the D update of --split_training draws no mask)
        md = None if split else M.generate_shifted_mask(tuple(imgs.shape), c["patch_size"], c["mask_ratio"])
        mg = M.generate_shifted_mask(tuple(imgs.shape), c["patch_size"], c["mask_ratio"])
        assert [0.0 if split else float(md.sum()), float(mg.sum())] == arr["mask_sums"][it].tolist()          # the reference's masks
        random.seed(meta["seed"] + 10 * it + 1)
        ol, _, _ = M.step(SG, SD, token, stG, stD, imgs, labels, md, mg, cfg, lr=meta["lr_effective"],
                          kind=c["mask_token_type"], mask_ratio=c["mask_ratio"], split_training=split, SE=SE,
                          before_g=lambda it=it: random.seed(meta["seed"] + 10 * it + 2))
        got.append([ol[k] for k in ("d_gan", "d_clf", "g_rec", "g_gan", "g_clf")] +
                   ([ol["distill_latent"], ol["distill_embed"]] if sean and cfg.style_distill else []))
    ref = arr["losses"]
    scale = np.maximum(np.abs(ref), 1.0)                # (the embed distillation term is O(500): relative there)
    assert (np.abs(np.array(got[0]) - ref[0]) / scale[0]).max() < 1e-5
    assert (np.abs(np.array(got[1]) - ref[1]) / scale[1]).max() < 5e-3           # behind sign-like first AdamW steps
    assert np.abs(token["mask_token"].detach().numpy() - arr["mask_token_post"]).max() < 2e-3


def test_mask_token_kinds():
    imgs = torch.rand(2, 3, 16, 16) * 2 - 1
    torch.manual_seed(0)
    masks = M.generate_shifted_mask((2, 3, 16, 16), 4, 0.5)
    assert masks.shape == (2, 1, 16, 16) and set(masks.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(M.apply_mask_token(None, imgs, masks, "zero", 0.5), imgs * masks)
    tok = torch.full((1, 3, 1, 1), 0.25)
    out = M.apply_mask_token(tok, imgs, masks, "vector", 0.5)
    assert torch.equal(out[masks.expand_as(out) == 1], imgs[masks.expand_as(imgs) == 1])
    assert bool((out[masks.expand_as(out) == 0] == 0.25).all())
    mean = M.apply_mask_token(None, imgs, masks, "mean", 0.5)
    expect = (imgs * masks).mean(dim=(2, 3), keepdim=True) / 0.5
    assert torch.allclose(mean[masks.expand_as(mean) == 0], expect.expand_as(mean)[masks.expand_as(mean) == 0])
