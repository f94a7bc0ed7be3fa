"""MAETrainer (trainers/mae_trainer.py:13-158): the MAE-GAN pre-training stage -- the documented first stage of the
reference recipe (defectGAN/README.md:6-10).  Per iteration one discriminator update and (every ``num_critics``
iterations) one generator update on the same image batch; each update draws its own shifted patch mask
(``utils/masks.py``), the generator repairs ``MaskToken(imgs, mask)``.

Same kernels as the defectGAN stage (one generator pass per update instead of four).

GradScaler.  The reference wraps both updates in ``torch.cuda.amp.GradScaler`` (base_trainer.py:61, mae_trainer.py:139-158)
although it never enables autocast (defectgan_model.py:54 is commented out): with fp32 gradients, scaling the loss by 2^16
and un-scaling before the update is the identity unless a gradient overflows fp32, and on the CPU -- the path the goldens
come from -- torch disables the scaler altogether.  ``opt.grad_scaler = True`` routes the updates through
``torch.amp.GradScaler`` exactly like the reference's GPU path (one host sync per update for its overflow check); the
default applies the update directly, which is the reference's CPU semantics and keeps the step free of host syncs.
TensorBoard logging, image grids and FID validation are host-side tooling outside the step and are not part of this
package."""
from collections import defaultdict

import torch

from .base_trainer import BaseTrainer


class MAETrainer(BaseTrainer):
    def __init__(self, opt, data_types=("fusion",)):
        super().__init__(opt)
        assert len(opt.loss_weight) == 3, f"length of loss weights must be 3, not {len(opt.loss_weight)}"
        self.loss_weights = {"rec": opt.loss_weight[0], "clf_D": opt.loss_weight[1], "clf_G": opt.loss_weight[2]}
        self.loss_types = ["rec", "gan", "clf"]
        self.distill = opt.style_norm_block_type == "sean" and getattr(opt, "style_distill", False)      # mae_trainer.py:20-21
        if self.distill:
            self.loss_types.append("distill")
        self._init_losses()
        self.data_types = data_types
        if opt.phase == "val":
            raise NotImplementedError("phase='val' builds FID metric networks (downloaded weights): out of scope")
        # the mask token is trained by the generator's optimizer (mae_trainer.py:28)
        self.optimizers["G"].add_param_group({"params": list(self.model.mask_token.parameters())})
        device_type = torch.device(opt.device).type
        self.scaler = torch.amp.GradScaler(device_type, enabled=device_type == "cuda" and bool(getattr(opt, "grad_scaler", False)))
        self.reducer = None          # set by parallel.attach_ddp(): gradient all-reduce across ranks

    def _init_lr(self, opt):
        assert len(opt.lr) in (1, 2), f"length of lr must be 1 or 2, not {len(opt.lr)}"
        self.lr = {"D": opt.lr[0], "G": opt.lr[1]} if len(opt.lr) == 2 else opt.lr[0]

    def _init_losses(self):
        self.losses = {loss_type: defaultdict(list) for loss_type in self.loss_types}

    def _scaled_update(self, loss, name):
        net = self.model.networks[name]
        self.scaler.scale(loss).backward()
        if self.reducer is not None:
            self.reducer.reduce(net)
            if name == "G":
                self.reducer.reduce(self.model.mask_token)
        self.scaler.step(self.optimizers[name])
        self.scaler.update()

    def _train_generator_once(self, data, labels):
        """mae_trainer.py:124-147"""
        self.optimizers["G"].zero_grad()
        losses = self.model("mae_generator", data, labels)
        rec_loss, gan_loss, clf_loss = losses[:3]
        g_loss = gan_loss + rec_loss * self.loss_weights["rec"] + clf_loss * self.loss_weights["clf_G"]
        self._scaled_update(g_loss, "G")
        if self.reducer is not None:
            self.reducer.broadcast_buffers(self.model.netG)
        self._record([("rec", "train"), ("gan", "G"), ("clf", "G")], [rec_loss, gan_loss, clf_loss])
        if self.distill and len(losses) == 5:             # (:125-131) logged only
            self._record([("distill", "latent"), ("distill", "embed")], list(losses[3:5]))

    def _train_discriminator_once(self, data, labels):
        """mae_trainer.py:149-158"""
        self.optimizers["D"].zero_grad()
        gan_loss, clf_loss = self.model("mae_discriminator", data, labels)
        d_loss = gan_loss + clf_loss * self.loss_weights["clf_D"]
        self._scaled_update(d_loss, "D")
        self._record([("gan", "D"), ("clf", "D")], [gan_loss, clf_loss])

    def step(self, data, labels):
        """One iteration of the reference's loop (mae_trainer.py:92-99)."""
        self.iters += 1
        self._train_discriminator_once(data, labels)
        if self.iters % self.opt.num_critics == 0:
            self._train_generator_once(data, labels)

    def _update_per_epoch(self, epoch=None):
        super()._update_per_epoch(epoch)
        self.model.update_per_epoch(epoch)
