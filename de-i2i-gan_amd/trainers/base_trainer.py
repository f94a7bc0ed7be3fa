"""BaseTrainer (trainers/base_trainer.py:12-131): model creation, one optimizer + one LR scheduler per network."""
import math
from collections import defaultdict

import numpy as np
import torch
from torch import optim

from ..models import create_model
from ..optim import FusedAdam, FusedRMSprop, FusedSGD


# opt.optimizer -> keyword arguments of the fused optimizer (the reference builds torch.optim.Adam(betas=(0.5, 0.999)) /
# torch.optim.AdamW(betas=(0.9, 0.95)) with AdamW's default weight decay: trainers/base_trainer.py:75-80)
_OPTIMIZER_ARGS = {
    "adam": {"betas": (0.5, 0.999)},
    "adamw": {"betas": (0.9, 0.95), "weight_decay": 1e-2},
}
_STEP_LR_STAGES = 4          # 'step' schedule: lr_decay is reached after four equal stages (base_trainer.py:94-98)


class BaseTrainer:
    """Same attributes and resume rules as the reference's BaseTrainer (trainers/base_trainer.py:12-131): ``model``,
    ``losses`` / ``dis_outputs`` (dict of lists), ``iters`` / ``first_epoch`` (restored from ``iter.txt`` when continuing),
    ``optimizers`` / ``schedulers`` keyed by network name; the schedulers are stepped ``first_epoch`` times at construction
    like the reference does."""

    def __init__(self, opt):
        self.opt = opt
        self.model = create_model(opt)
        self._restore_or_init_weights(opt)
        self.losses, self.dis_outputs = defaultdict(list), defaultdict(list)
        if opt.phase == "val":
            self.metrics = {}
        self._resume_progress(opt)
        self._init_lr(opt)
        self._create_optimizer(opt)
        self._create_scheduler(opt)
        # defer_loss_sync: keep the per-step losses on the device and convert them in one batch in flush_losses()
        # instead of the reference's .item() per loss (one host sync each, defectgan_trainer.py:164-168,179-180)
        self.defer_loss_sync = bool(getattr(opt, "defer_loss_sync", False))
        self._pending = []

    def _restore_or_init_weights(self, opt):
        if opt.continue_training:
            self.model.load("latest")
        elif opt.load_model_name is not None:
            self.model.load(opt.which_epoch)
        else:
            self.model.init_weights()

    def _resume_progress(self, opt):
        """Epoch / iteration counters and the derived run length (num_epochs == -1 means 'from num_iters')."""
        if not hasattr(opt, "iters_per_epoch"):
            raise AssertionError("opt must have attribute {iters_per_epoch}, it can be calculated by length of loader")
        self.iter_record_path = opt.ckpt_dir / opt.name / "iter.txt"
        self.first_epoch, self.iters = 1, 0
        if opt.continue_training:
            self.first_epoch, self.iters = np.loadtxt(self.iter_record_path, delimiter=",", dtype=int)
        if opt.num_epochs == -1:
            opt.num_epochs = math.ceil(opt.num_iters / (opt.iters_per_epoch + 1e-12))
        opt.num_iters = opt.num_epochs * opt.iters_per_epoch
        if not self.first_epoch < opt.num_epochs:
            raise AssertionError(f"first_epoch {self.first_epoch} should not larger than num_epochs {opt.num_epochs}")
        if not self.iters < opt.num_iters:
            raise AssertionError(f"iters {self.iters} should not larger than num_iters {opt.num_iters}")
        opt.first_epoch = self.first_epoch

    def _init_lr(self, opt):
        self.lr = opt.lr[0]

    def _lr_of(self, network_name):
        return self.lr[network_name] if isinstance(self.lr, dict) else self.lr

    def _create_optimizer(self, opt):
        if not isinstance(self.lr, (int, float, dict)):
            raise AssertionError("type of lr should be scalar or dict")
        if opt.optimizer in ("sgd", "rmsprop"):          # torch's defaults, lr only (base_trainer.py:71-74)
            cls = FusedSGD if opt.optimizer == "sgd" else FusedRMSprop
            self.optimizers = {name: cls(net.parameters(), lr=self._lr_of(name)) for name, net in self.model.networks.items()}
            return
        if opt.optimizer not in _OPTIMIZER_ARGS:
            raise NameError(f"optimizer named {opt.optimizer} not defined")
        self.optimizers = {name: FusedAdam(net.parameters(), lr=self._lr_of(name), **_OPTIMIZER_ARGS[opt.optimizer])
                           for name, net in self.model.networks.items()}

    def _scheduler_for(self, opt, name, optimizer):
        kind = opt.scheduler
        if kind == "step":
            return optim.lr_scheduler.StepLR(optimizer, step_size=opt.num_epochs // _STEP_LR_STAGES,
                                             gamma=opt.lr_decay ** (1 / _STEP_LR_STAGES))
        if kind == "exp":
            return optim.lr_scheduler.ExponentialLR(optimizer, gamma=opt.lr_decay ** (1 / opt.num_epochs))
        if kind == "cos":                          # anneals to lr * lr_decay over the run (base_trainer.py:103-111)
            return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.num_epochs, eta_min=self._lr_of(name) * opt.lr_decay)
        raise NameError(f"scheduler named {kind} not defined")

    def _create_scheduler(self, opt):
        self.schedulers = {name: self._scheduler_for(opt, name, o) for name, o in self.optimizers.items()}
        for _ in range(self.first_epoch):          # the reference fast-forwards to first_epoch (>= 1) at construction
            for scheduler in self.schedulers.values():
                scheduler.step()

    # ---- loss bookkeeping (self.losses[kind][name] lists, like the reference) ----------------------------------
    def _record(self, keys, tensors):
        stacked = torch.stack([t.detach() for t in tensors])
        if self.defer_loss_sync:
            self._pending.append((keys, stacked))
        else:
            for (kind, name), v in zip(keys, stacked.tolist()):        # ONE device->host read for the group
                self.losses[kind][name].append(v)

    def flush_losses(self):
        if self._pending:
            flat = torch.cat([s for _, s in self._pending]).tolist()
            i = 0
            for keys, s in self._pending:
                for kind, name in keys:
                    self.losses[kind][name].append(flat[i])
                    i += 1
            self._pending = []

    def _update_per_epoch(self, epoch=None):
        for scheduler in self.schedulers.values():
            scheduler.step()
