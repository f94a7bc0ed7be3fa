"""DefectGanTrainer (trainers/defectgan_trainer.py:19-188): the two ``_train_*_once`` methods with the reference's
signatures and side effects, ``step()`` = one D update then (every ``num_critics`` iterations) one G update, and the
epoch loop ``train()`` / ``_train_epoch()`` with the reference's checkpoint / resume behaviour (``latest`` weights and
``iter.txt`` every ``save_latest_freq`` iterations, numbered weights every ``save_ckpt_freq`` epochs, one scheduler step per
epoch).

TensorBoard logging, progress bars, image grids and FID/IS/LPIPS validation are host-side tooling outside the hot path
 and are not part of this package."""
from collections import defaultdict

import numpy as np
import torch

from .base_trainer import BaseTrainer


class DefectGanTrainer(BaseTrainer):
    def __init__(self, opt):
        super().__init__(opt)
        assert len(opt.loss_weight) == 5, f"length of loss weights must be 5, not {len(opt.loss_weight)}"
        self.loss_weights = {"clf_d": opt.loss_weight[0], "clf_g": opt.loss_weight[1], "rec": opt.loss_weight[2],
                             "sd_cyc": opt.loss_weight[3], "sd_con": opt.loss_weight[4]}
        self.loss_types = ["gan", "clf", "aux"]
        self.distill = opt.style_norm_block_type == "sean" and getattr(opt, "style_distill", False)      # defectgan_trainer.py:29-30
        if self.distill:
            self.loss_types.append("distill")
        self._init_losses()
        if opt.phase == "val":
            raise NotImplementedError("phase='val' builds FID/LPIPS metric networks (downloaded weights): out of scope")
        self.reducer = None          # set by parallel.attach_ddp(): gradient all-reduce across ranks

    def _init_lr(self, opt):
        assert len(opt.lr) in (1, 2), f"length of lr must be 1 or 2, not {len(opt.lr)}"
        self.lr = {"D": opt.lr[0], "G": opt.lr[1]} if len(opt.lr) == 2 else opt.lr[0]

    def _init_losses(self):
        self.losses = {loss_type: defaultdict(list) for loss_type in self.loss_types}

    # ---- the step ---------------------------------------------------------------------------------------------
    def _train_generator_once(self, bg_data, df_labels, df_data):
        """defectgan_trainer.py:138-168"""
        self.optimizers["G"].zero_grad()
        with_e = "E" in self.optimizers                  # adain: the StyleExtractor is trained by the G loss (:140-141,161-163)
        if with_e:
            self.optimizers["E"].zero_grad()
        losses = self.model("generator", bg_data, df_labels, df_data)
        gan_loss, clf_loss, rec_loss, sd_cyc_loss, sd_con_loss = losses[:5]
        g_loss = gan_loss + clf_loss * self.loss_weights["clf_g"] + rec_loss * self.loss_weights["rec"] + \
            sd_cyc_loss * self.loss_weights["sd_cyc"] + sd_con_loss * self.loss_weights["sd_con"]
        g_loss.backward()
        if self.reducer is not None:
            self.reducer.reduce(self.model.netG)
            if with_e:
                self.reducer.reduce(self.model.netE)
        self.optimizers["G"].step()
        if with_e:
            self.optimizers["E"].step()
        if self.reducer is not None:
            self.reducer.broadcast_buffers(self.model.netG)      # BatchNorm running stats follow rank 0
        self._record([("gan", "G"), ("clf", "G"), ("aux", "rec"), ("aux", "cyc"), ("aux", "con")],
                     [gan_loss, clf_loss, rec_loss, sd_cyc_loss, sd_con_loss])
        if self.distill:                                  # (:142-146) logged; the gradients were taken inside the SEAN layers
            self._record([("distill", "latent"), ("distill", "embed")], list(losses[5:7]))

    def _train_discriminator_once(self, bg_data, df_labels, df_data):
        """defectgan_trainer.py:170-180"""
        self.optimizers["D"].zero_grad()
        gan_loss, clf_loss = self.model("discriminator", bg_data, df_labels, df_data)
        d_loss = gan_loss + clf_loss * self.loss_weights["clf_d"]
        d_loss.backward()
        if self.reducer is not None:
            self.reducer.reduce(self.model.netD)
        self.optimizers["D"].step()
        self._record([("gan", "D"), ("clf", "D")], [gan_loss, clf_loss])

    def step(self, bg_data, df_labels, df_data):
        """One iteration of the reference's hot loop (defectgan_trainer.py:96-109)."""
        self.iters += 1
        self._train_discriminator_once(bg_data, df_labels, df_data)
        if self.iters % self.opt.num_critics == 0:
            self._train_generator_once(bg_data, df_labels, df_data)

    # ---- the epoch loop with the reference's checkpoint / resume rules (defectgan_trainer.py:75-120) -------------------
    def train(self, train_loaders, val_loaders=None):
        """Epochs ``first_epoch .. num_epochs`` (1-based, like the reference): ``train_loaders['defects']`` is iterated once
        per epoch, ``train_loaders['background']`` is an (infinite) iterator drawn with ``next`` -- both yield
        ``(images, labels, _)``.  ``val_loaders`` is accepted for signature compatibility (metrics are out of scope)."""
        for epoch in range(self.first_epoch, self.opt.num_epochs + 1):
            self._init_losses()
            self._train_epoch(train_loaders, epoch)
            if epoch % getattr(self.opt, "save_ckpt_freq", 10 ** 9) == 0:
                self.model.save(epoch)
            self._update_per_epoch(epoch)

    def _train_epoch(self, data_loaders, epoch):
        for df_data, df_labels, _ in data_loaders["defects"]:
            bg_data, bg_labels, _ = next(data_loaders["background"])
            bg_data = bg_data[:df_data.size(0)]            # truncate to the defect batch (defectgan_trainer.py:102-105)
            self.step(bg_data, df_labels, df_data)
            if self.iters % self.opt.save_latest_freq == 0:
                self.save_latest(epoch)

    def save_latest(self, epoch):
        """``latest_net_{G,D}.pth`` + ``iter.txt`` = (epoch, iters), what ``--continue_training`` restores
        (defectgan_trainer.py:111-113, base_trainer.py:38-44).  Optimizer state is not saved -- the reference does not."""
        self.model.save("latest")
        self.iter_record_path.parent.mkdir(parents=True, exist_ok=True)
        np.savetxt(self.iter_record_path, (epoch, self.iters), fmt="%i", delimiter=",")

    def _update_per_epoch(self, epoch=None):
        super()._update_per_epoch(epoch)
        self.model.update_per_epoch(epoch)
