"""BaseTrainer (trainers/base_trainer.py:12-131): model creation, one optimizer + one LR scheduler per network."""
import math
from collections import defaultdict

import numpy as np
import torch
from torch import optim

from ..models import create_model
from ..optim import FusedAdam


class BaseTrainer:
    def __init__(self, opt):
        self.opt = opt
        self.model = create_model(opt)
        if opt.continue_training:
            self.model.load("latest")
        elif opt.load_model_name is not None:
            self.model.load(opt.which_epoch)
        else:
            self.model.init_weights()

        self.losses = defaultdict(list)
        self.dis_outputs = defaultdict(list)
        if opt.phase == "val":
            self.metrics = dict()

        self.iter_record_path = opt.ckpt_dir / opt.name / "iter.txt"
        self.first_epoch = 1
        self.iters = 0
        assert hasattr(self.opt, "iters_per_epoch"), "opt must have attribute {iters_per_epoch}, " \
                                                     "it can be calculated by length of loader"
        if opt.continue_training:
            self.first_epoch, self.iters = np.loadtxt(self.iter_record_path, delimiter=",", dtype=int)
        if self.opt.num_epochs == -1:
            self.opt.num_epochs = math.ceil(self.opt.num_iters / (self.opt.iters_per_epoch + 1e-12))
        self.opt.num_iters = self.opt.num_epochs * self.opt.iters_per_epoch
        assert self.first_epoch < self.opt.num_epochs, f"first_epoch {self.first_epoch} should not larger than " \
                                                       f"num_epochs {self.opt.num_epochs}"
        assert self.iters < self.opt.num_iters, f"iters {self.iters} should not larger than num_iters {self.opt.num_iters}"
        self.opt.first_epoch = self.first_epoch

        self._init_lr(opt)
        self._create_optimizer(opt)
        self._create_scheduler(opt)
        # defer_loss_sync: keep the per-step losses on the device and convert them in one batch in flush_losses()
        # instead of the reference's .item() per loss (one host sync each, defectgan_trainer.py:164-168,179-180)
        self.defer_loss_sync = bool(getattr(opt, "defer_loss_sync", False))
        self._pending = []

    def _init_lr(self, opt):
        self.lr = opt.lr[0]

    def _create_optimizer(self, opt):
        assert isinstance(self.lr, (int, float, dict)), "type of lr should be scalar or dict"
        optim_args = dict()
        if opt.optimizer == "adam":
            optim_cls = FusedAdam                       # torch.optim.Adam(betas=(0.5, 0.999)) in the reference (:75-77)
            optim_args["betas"] = (0.5, 0.999)
        elif opt.optimizer == "adamw":
            optim_cls = FusedAdam                       # torch.optim.AdamW(betas=(0.9, 0.95)) in the reference (:78-80)
            optim_args["betas"] = (0.9, 0.95)
            optim_args["weight_decay"] = 1e-2           # torch.optim.AdamW's default, which the reference relies on
        elif opt.optimizer in ("sgd", "rmsprop"):
            raise NotImplementedError(f"optimizer [{opt.optimizer}]: only 'adam' / 'adamw' have a fused kernel")
        else:
            raise NameError(f"optimizer named {opt.optimizer} not defined")
        self.optimizers = {}
        for network_name, network in self.model.networks.items():
            optim_args["lr"] = self.lr[network_name] if isinstance(self.lr, dict) else self.lr
            self.optimizers[network_name] = optim_cls(network.parameters(), **optim_args)

    def _create_scheduler(self, opt):
        sched_args = dict()
        ext_args = defaultdict(dict)
        if opt.scheduler == "step":
            sched_cls = optim.lr_scheduler.StepLR
            step_cnt = 4
            sched_args["step_size"] = opt.num_epochs // step_cnt
            sched_args["gamma"] = opt.lr_decay ** (1 / step_cnt)
        elif opt.scheduler == "exp":
            sched_cls = optim.lr_scheduler.ExponentialLR
            sched_args["gamma"] = opt.lr_decay ** (1 / opt.num_epochs)
        elif opt.scheduler == "cos":
            sched_cls = optim.lr_scheduler.CosineAnnealingLR
            sched_args["T_max"] = opt.num_epochs
        else:
            raise NameError(f"scheduler named {opt.scheduler} not defined")
        if opt.scheduler == "cos":
            for network_name in self.model.networks:
                base = self.lr[network_name] if isinstance(self.lr, dict) else self.lr
                ext_args["eta_min"][network_name] = base * opt.lr_decay
        self.schedulers = dict()
        for model_name, optimizer in self.optimizers.items():
            for key, value in ext_args.items():
                sched_args[key] = value[model_name]
            self.schedulers[model_name] = sched_cls(optimizer, **sched_args)
        for _ in range(self.first_epoch):
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
        for model_name in self.schedulers.keys():
            self.schedulers[model_name].step()
