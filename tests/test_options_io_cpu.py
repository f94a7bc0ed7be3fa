"""Option files (options/base_options.py:116-149) -- CPU only: the round trip through opt.pkl, the precedence of values given
when continuing a run (the reference installs the stored values as parser DEFAULTS, so the command line still wins), and the
loader refusing a pickle that names anything but plain option values."""
import pickle
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

from de_i2i_gan_amd.utils.options_io import load_options, option_file_path, save_options, update_options_from_file


def _opt(tmp_path, **over):
    d = dict(ckpt_dir=Path(tmp_path), name="run", load_model_name=None, continue_training=False, device=torch.device("cpu"),
             ngf=64, lr=[2e-4], num_epochs=20, diff_aug="", loss_weight=[2, 5, 5, 5, 1], sean_alpha=None)
    d.update(over)
    return SimpleNamespace(**d)


def test_round_trip_and_precedence(tmp_path):
    save_options(_opt(tmp_path, ngf=8, num_epochs=12, lr=[1e-4]))
    assert option_file_path(_opt(tmp_path)).exists()
    again = _opt(tmp_path, continue_training=True, load_model_name="run", num_epochs=40)
    stored = load_options(again)
    assert (stored.ngf, stored.num_epochs, stored.lr, stored.device) == (8, 12, [1e-4], torch.device("cpu"))
    # nothing marked as given: the stored run's values replace the defaults (ngf, lr, num_epochs) ...
    a = update_options_from_file(_opt(tmp_path, continue_training=True, load_model_name="run", num_epochs=40))
    assert (a.ngf, a.num_epochs, a.lr, a.load_model_name, a.continue_training) == (8, 12, [1e-4], "run", True)
    # ... while an option given on the continuing run's command line wins over the file (base_options.py:135-141)
    b = update_options_from_file(again, explicit=("num_epochs",))
    assert (b.ngf, b.num_epochs, b.lr) == (8, 40, [1e-4])


def test_loader_refuses_anything_but_option_values(tmp_path):
    opt = _opt(tmp_path, continue_training=True)
    path = option_file_path(opt, make_dir=True)

    class Evil:
        def __reduce__(self):
            return (print, ("executed from an option file",))

    with path.open("wb") as f:
        pickle.dump(SimpleNamespace(name="run", hook=Evil()), f)
    with pytest.raises(pickle.UnpicklingError):
        load_options(opt)
