"""Checkpoint / resume / option I/O (reference: models/networks/__init__.py:4-23,
trainers/base_trainer.py:24-25,38-44, trainers/defectgan_trainer.py:75-120, options/base_options.py:116-149).

  * a checkpoint WRITTEN BY THE REFERENCE (tests/golden/ckpt_ref/, made by tests/golden/gen_ckpt_golden.py with the
    reference's own trainer, init and save_network) loads into the product's networks with weights_only=True and
    reproduces the reference's G(x) / D(G(x)) of those weights; iter.txt resumes the counters;
  * the product's own train loop writes latest_net_*.pth / iter.txt / numbered checkpoints like the reference and a new
    trainer with continue_training picks them up; the files load back into the key set the reference uses;
  * opt.pkl / opt.txt round trip."""
import shutil
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import make_opt
from oracle import defectgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REF = Path(__file__).resolve().parent / "golden" / "ckpt_ref"
C = dict(image_size=32, batch=2, num_layers=3, ngf=8, ndf=8, hidden_nc=16)


def maxrel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def test_reference_written_checkpoint_loads_and_reproduces_its_outputs(tmp_path):
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    run = tmp_path / "ref_run"
    run.mkdir()
    for f in ("latest_net_G.pth", "latest_net_D.pth", "iter.txt"):
        shutil.copy(REF / f, run / f)
    arr = np.load(REF / "ckpt_ref.npz")
    opt = make_opt(C, DEV, "f32", ckpt_dir=tmp_path, name="ref_run", load_model_name="ref_run", continue_training=True)
    tr = DefectGanTrainer(opt)                       # continue_training: model.load('latest') + iter.txt
    assert (tr.first_epoch, tr.iters) == (1, 1)
    G, D = tr.model.netG, tr.model.netD
    assert sorted(G.state_dict().keys()) == list(arr["g_keys"]) and sorted(D.state_dict().keys()) == list(arr["d_keys"])
    bg, labels, df = O.synthetic_batch(C["batch"], C["image_size"])
    with torch.no_grad():
        G.eval()
        D.eval()
        out, prob = G(bg.to(DEV), labels.to(DEV))
        src, cls = D(out)
    for got, key in ((out, "out"), (prob, "prob"), (src, "src"), (cls, "cls")):
        assert maxrel(got.cpu(), arr[key]) < 1e-3, key        # exact-f32 mode: north_star's bound (measured ~1e-5)
    assert int(G.stem.conv_block[1].num_batches_tracked) == 4   # buffers came along (four train-mode passes of one G step)


def _loaders(n_batches, batch, size, seed=3):
    g = torch.Generator().manual_seed(seed)

    def one():
        x = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
        lab = torch.zeros(batch, 6)
        lab[torch.arange(batch), 1 + torch.arange(batch) % 5] = 1
        return x, lab, None

    def background():
        while True:
            yield one()

    return {"defects": [one() for _ in range(n_batches)], "background": background()}


def test_train_loop_writes_and_resumes_checkpoints_like_the_reference(tmp_path):
    from de_i2i_gan_amd.trainers.defectgan_trainer import DefectGanTrainer
    from de_i2i_gan_amd.utils.options_io import load_options, option_file_path, save_options, update_options_from_file
    over = dict(ckpt_dir=tmp_path, name="run", iters_per_epoch=3, num_epochs=8, num_iters=24, save_latest_freq=2, save_ckpt_freq=1)
    opt = make_opt(C, DEV, "bf16", **over)
    save_options(opt, defaults={"ngf": 64})
    assert option_file_path(opt).exists() and "[default: 64]" in option_file_path(opt).with_suffix(".txt").read_text()
    torch.manual_seed(5)
    tr = DefectGanTrainer(opt)
    tr.opt.num_epochs = 2                              # two epochs of three iterations
    tr.train(_loaders(3, C["batch"], C["image_size"]))
    run = tmp_path / "run"
    assert tr.iters == 6
    assert sorted(p.name for p in run.glob("*.pth")) == ["1_net_D.pth", "1_net_G.pth", "2_net_D.pth", "2_net_G.pth",
                                                           "latest_net_D.pth", "latest_net_G.pth"]
    assert np.loadtxt(run / "iter.txt", delimiter=",", dtype=int).tolist() == [2, 6]
    stored = torch.load(run / "latest_net_G.pth", map_location="cpu", weights_only=True)
    assert sorted(stored.keys()) == list(np.load(REF / "ckpt_ref.npz")["g_keys"])       # the reference's key set, NCHW fp32
    assert all(v.dtype in (torch.float32, torch.int64) for v in stored.values())
    # resume: a fresh process would rebuild opt, read opt.pkl back and continue
    opt2 = make_opt(C, DEV, "bf16", **dict(over, continue_training=True, load_model_name="run", ngf=999))
    update_options_from_file(opt2)
    assert opt2.ngf == 8 and load_options(opt2).save_latest_freq == 2
    tr2 = DefectGanTrainer(opt2)
    assert (tr2.first_epoch, tr2.iters) == (2, 6)
    for k, v in tr2.model.netG.state_dict().items():
        assert torch.equal(v.cpu(), stored[k]), k
    # schedulers were fast-forwarded first_epoch times (base_trainer.py:124-126)
    assert tr2.schedulers["G"].last_epoch == 2
