"""DefectGanModel (models/defectgan_model.py:18-104,173-314): the loss graphs of the D step and the G step."""
import os

import torch

from ..networks.discriminator import DefectGanDiscriminator
from ..networks.generator import DefectGanGenerator
from .base_model import BaseModel


class DefectGanModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        image_size = opt.image_size
        assert image_size & (image_size - 1) == 0, "Image size must be a power of 2"
        if opt.style_norm_block_type != "spade":
            raise NotImplementedError("only style_norm_block_type='spade' (the reference default) is implemented")
        if getattr(opt, "diff_aug", ""):
            raise NotImplementedError("DiffAugment policies are not implemented yet (SURVEY.md section 8f rank 2)")
        self.netG = DefectGanGenerator(opt).to(opt.device, non_blocking=True)
        self.netD = DefectGanDiscriminator(opt).to(opt.device, non_blocking=True)
        if self.opt.is_train or hasattr(opt, "clf_loss_type"):
            assert opt.clf_loss_type is not None, "clf_loss_type should be initialized in dataset"
            self.clf_loss_type = opt.clf_loss_type

    def __call__(self, mode, data, labels, df_data=None, img_only=False, mask=None):
        data, labels = data.to(self.opt.device, non_blocking=True), labels.to(self.opt.device, non_blocking=True)
        if df_data is not None:
            df_data = df_data.to(self.opt.device, non_blocking=True)
        if mode == "generator":
            self.netD.eval()
            self.netG.train()
            return self._compute_generator_loss(data, labels, df_data)
        if mode == "discriminator":
            self.netD.train()
            self.netG.eval()
            return self._compute_discriminator_loss(data, labels, df_data)
        if mode == "inference":
            self.netD.eval()
            self.netG.eval()
            return self._generate_fake(data, labels)
        if mode == "inference_classifier":
            self.netD.eval()
            self.netG.eval()
            return self._compute_clf_loss(data, labels)
        raise ValueError(f"|mode {mode}| is invalid")

    @staticmethod
    def _mean(terms):
        return torch.stack(terms).mean()

    def _netD_batched(self, *images):
        """netD(x) for several image batches in ONE pass over their concatenation.  The discriminator has no batch
        statistics (PatchGAN convs + LeakyReLU only, discriminator.py:60-90), so this is the same function as the
        reference's one call per batch -- with 2-4x fewer, 2-4x larger kernels on its small-M deep layers and one weight
        gradient per layer instead of one per call."""
        if os.environ.get("DEI2I_SPLIT_D"):               # A/B switch: the reference's call-per-batch structure
            return [self.netD(t) for t in images]
        sizes = [t.shape[0] for t in images]
        src, cls = self.netD(torch.cat(images, 0))
        return list(zip(src.split(sizes), cls.split(sizes)))

    def _compute_generator_loss(self, bg_data, df_labels, df_data):
        """defectgan_model.py:173-249"""
        nm_labels, df_labels = self._get_labels(df_labels)
        self.netG.clear_spade_cache()
        if not os.environ.get("DEI2I_SPLIT_D"):
            self.netG.prime_spade((df_labels, nm_labels))        # both label sets' SPADE tables in one pass
        fake_defects, df_prob = self.netG(bg_data, df_labels)
        recover_normals, rec_df_prob = self.netG(fake_defects, nm_labels)
        fake_normals, nm_prob = self.netG(df_data, nm_labels)
        recover_defects, rec_nm_prob = self.netG(fake_normals, df_labels)

        # The reference lets autograd compute (and then discards) the discriminator's weight gradients here; the
        # G optimizer never reads them and optimizers['D'].zero_grad() drops them, so they are skipped.
        d_params = [p for p in self.netD.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            (fake_defects_src, fake_defects_cls), (fake_normals_src, fake_normals_cls) = \
                self._netD_batched(fake_defects, fake_normals)
        finally:
            for p in d_params:
                p.requires_grad_(True)

        gan_loss = [self._cal_loss(fake_defects_src, 1.0, "bce"), self._cal_loss(fake_normals_src, 1.0, "bce")]
        clf_loss = [self._cal_loss(fake_defects_cls, df_labels.view_as(fake_defects_cls), self.clf_loss_type),
                    self._cal_loss(fake_normals_cls, nm_labels.view_as(fake_normals_cls), self.clf_loss_type)]
        rec_loss = [self._cal_loss(recover_defects, df_data, "l1"), self._cal_loss(recover_normals, bg_data, "l1")]
        sd_cyc_loss = [self._cal_loss(df_prob, rec_df_prob, "l1"), self._cal_loss(nm_prob, rec_nm_prob, "l1")]
        sd_con_loss = [self._cal_loss(df_prob, None, "l1"), self._cal_loss(nm_prob, None, "l1"),
                       self._cal_loss(rec_df_prob, None, "l1"), self._cal_loss(rec_nm_prob, None, "l1")]
        return (self._mean(gan_loss), self._mean(clf_loss), self._mean(rec_loss), self._mean(sd_cyc_loss),
                self._mean(sd_con_loss))

    def _compute_discriminator_loss(self, bg_data, df_labels, df_data):
        """defectgan_model.py:251-292"""
        nm_labels, df_labels = self._get_labels(df_labels)
        self.netG.clear_spade_cache()
        with torch.no_grad():
            if self.netG.training or os.environ.get("DEI2I_SPLIT_D"):
                fake_defects, _ = self.netG(bg_data, df_labels)
                fake_normals, _ = self.netG(df_data, nm_labels)
            else:
                # netG is in eval mode here (defectgan_model.py:87-90): BatchNorm uses running statistics and SPADE's
                # InstanceNorm is per sample, so one pass over both batches is the same function as two passes
                fakes, _ = self.netG(torch.cat([bg_data, df_data], 0), torch.cat([df_labels, nm_labels], 0))
                fake_defects, fake_normals = fakes.split([bg_data.shape[0], df_data.shape[0]])
        (fake_defects_src, _), (fake_normals_src, _), (real_defects_src, real_defects_cls), \
            (real_normals_src, real_normals_cls) = self._netD_batched(fake_defects.detach(), fake_normals.detach(), df_data, bg_data)
        gan_loss = [self._cal_loss(fake_defects_src, 0.0, "bce"), self._cal_loss(fake_normals_src, 0.0, "bce"),
                    self._cal_loss(real_defects_src, 1.0, "bce"), self._cal_loss(real_normals_src, 1.0, "bce")]
        clf_loss = [self._cal_loss(real_defects_cls, df_labels.view_as(real_defects_cls), self.clf_loss_type),
                    self._cal_loss(real_normals_cls, nm_labels.view_as(real_normals_cls), self.clf_loss_type)]
        return self._mean(gan_loss), self._mean(clf_loss)

    def _compute_clf_loss(self, imgs, labels):
        _, df_logits = self.netD(imgs)
        return df_logits, self._cal_loss(df_logits, labels, self.clf_loss_type)

    @torch.no_grad()
    def _generate_fake(self, data, labels):
        """defectgan_model.py:302-314 (spade branch): labels (N,C) or a spatial (N,C,h,w) map"""
        self.netG.clear_spade_cache()
        return self.netG(data, self._expand_seg(labels))

    @staticmethod
    def _expand_seg(labels):
        if labels.dim() == 2:
            return labels.reshape(labels.size(0), labels.size(1), 1, 1)
        if labels.dim() == 4:
            return labels
        raise ValueError(f"|labels dim {labels.dim()}| is invalid")

    def _get_labels(self, df_labels):
        """defectgan_model.py:413-428: nm_labels = one-hot class 0; both expanded to (N,C,1,1) for SPADE"""
        df_labels = df_labels.float()
        nm_labels = torch.zeros_like(df_labels)
        nm_labels[:, 0] = 1
        return self._expand_seg(nm_labels), self._expand_seg(df_labels)
