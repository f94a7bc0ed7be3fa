"""DefectGanModel (models/defectgan_model.py:18-171,173-314,361-383): the loss graphs of the D step and the G step of the
defectGAN stage, and of the MAE-GAN pre-training stage (modes ``mae_*``)."""
import os
import random

import torch

from .. import ops
from ..networks.architecture import MaskToken
from ..networks.discriminator import DefectGanDiscriminator
from ..networks.generator import DefectGanGenerator
from ..utils.diffaug import diff_augment
from ..utils.masks import draw_shifted_mask, expand_shifted_mask
from .base_model import BaseModel


class DefectGanModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        image_size = opt.image_size
        assert image_size & (image_size - 1) == 0, "Image size must be a power of 2"
        if opt.style_norm_block_type not in ("spade", "adain", "sean"):
            raise ValueError(f"|style_norm_block_type {opt.style_norm_block_type}| is invalid")
        self.netG = DefectGanGenerator(opt).to(opt.device, non_blocking=True)
        self.netD = DefectGanDiscriminator(opt).to(opt.device, non_blocking=True)
        if opt.style_norm_block_type == "sean":          # style embeddings (defectgan_model.py:34-45)
            if opt.sean_alpha is not None:
                self.netG.set_sean_alpha(opt.sean_alpha)
            if opt.sean_alpha != 0:
                assert opt.embed_path is not None, ("embed_path should be initialized if style_norm_block_type is sean and "
                                                    "sean_alpha is not 0")
                # {label tuple: [embedding (embed_nc,), ...]} -- the user's own file; loaded without executing anything from it
                self.embeddings = torch.load(opt.embed_path, weights_only=True)
                for label, embeds in self.embeddings.items():
                    self.embeddings[label] = [e.to(opt.device, non_blocking=True) for e in embeds]
        if opt.style_norm_block_type == "adain":         # style embedding network, trained by the G loss (defectgan_model.py:46-47)
            from ..networks.extractor import StyleExtractor
            self.netE = StyleExtractor(opt).to(opt.device, non_blocking=True)
        if self.opt.is_train or hasattr(opt, "clf_loss_type"):
            assert opt.clf_loss_type is not None, "clf_loss_type should be initialized in dataset"
            self.clf_loss_type = opt.clf_loss_type
        if hasattr(opt, "mask_token_type"):              # learnable mask token of the MAE stage (defectgan_model.py:31-32)
            self.mask_token = MaskToken(opt).to(opt.device, non_blocking=True)

    def __call__(self, mode, data, labels, df_data=None, img_only=False, mask=None):
        data, labels = data.to(self.opt.device, non_blocking=True), labels.to(self.opt.device, non_blocking=True)
        if df_data is not None:
            df_data = df_data.to(self.opt.device, non_blocking=True)
        if mode.startswith("mae"):
            if mode == "mae_generator":
                self.netD.eval()
                self.netG.train()
                return self._compute_mae_generator_loss(data, labels)
            if mode == "mae_discriminator":
                self.netD.train()
                self.netG.eval()
                return self._compute_mae_discriminator_loss(data, labels)
            if mode == "mae_inference":
                self.netD.eval()
                self.netG.eval()
                with torch.no_grad():
                    if getattr(self.opt, "split_training", False):       # defectgan_model.py:69-72
                        rec_loss, gan_loss, _ = self._compute_mae_generator_loss(data, labels)
                        _, clf_loss = self._compute_mae_discriminator_loss(data, labels)
                        return rec_loss, gan_loss, clf_loss
                    return self._compute_mae_generator_loss(data, labels)
            raise ValueError(f"|mode {mode}| is invalid")
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
        # --use_spectral: every training-mode call runs one power iteration on D's (u, v), so the reference's four /
        # two calls are NOT one call on the concatenation -- keep its call-per-batch structure then (eval mode is fine)
        per_call_state = getattr(self.opt, "use_spectral", False) and self.netD.training
        if os.environ.get("DEI2I_SPLIT_D") or per_call_state:     # (env: A/B switch)
            return [self.netD(t) for t in images]
        sizes = [t.shape[0] for t in images]
        whole = images[0]._base if len(images) > 1 else None
        if (whole is not None and all(t._base is whole and t.is_contiguous() for t in images) and whole.is_contiguous()
                and whole.shape[0] == sum(sizes) and whole.shape[1:] == images[0].shape[1:]
                and all(t.data_ptr() == whole[sum(sizes[:i]):].data_ptr() for i, t in enumerate(images))):
            batch = whole                                   # the batches ARE the consecutive slices of one tensor (paired generator passes)
        else:
            batch = torch.cat(images, 0)
        src, cls = self.netD(batch)
        return list(zip(src.split(sizes), cls.split(sizes)))

    # ---- MAE-GAN pre-training stage (defectgan_model.py:106-171, 361-383) ----------------------------------------
    def _repair_mask(self, imgs, labels, mask=None):
        """Draw a shifted patch mask (host RNG, like the reference), fill the masked pixels with the mask token and
        let the generator repair the image."""
        if mask is not None:
            masks = self._upload_mask(mask)
        else:       # the reference's RNG draws on the host (utils/util.py:61-71); the expansion to pixels on the device
            row0, col0, keep = draw_shifted_mask(imgs.size(), self.opt.patch_size, self.opt.mask_ratio)
            masks = expand_shifted_mask(self._upload_mask(keep), row0, col0, self.opt.patch_size, imgs.size(2), imgs.size(3))
        self.netG.clear_spade_cache()
        if self.opt.style_norm_block_type == "sean":             # defectgan_model.py:370-372
            predicted, _ = self.netG(self.mask_token(imgs, masks), labels, self._get_style_embeds(labels))
        elif self.opt.style_norm_block_type == "adain":          # :377-379
            predicted, _ = self.netG(self.mask_token(imgs, masks), labels, self.netE(imgs, labels))
        else:
            seg = self._expand_seg(labels)
            if seg.shape[2:] == (1, 1):
                self.netG.prime_spade((seg,))                # every SPADE module's class table in the batched launches (label_path.hip)
            predicted, _ = self.netG(self.mask_token(imgs, masks), seg)
        return predicted, masks

    def _upload_mask(self, masks):
        """Host mask -> device without stalling the host: a pageable host->device copy waits for the GPU work queued
        before it, so the mask goes through one of a few persistent pinned staging buffers (each guarded by an event)."""
        dev = torch.device(self.opt.device)
        if masks.device.type != "cpu" or dev.type != "cuda":
            return masks.to(dev, non_blocking=True)
        ring = self.__dict__.setdefault("_mask_ring", [])
        slot = self.__dict__.get("_mask_slot", 0)
        self.__dict__["_mask_slot"] = (slot + 1) % 4
        if len(ring) <= slot or ring[slot][0].shape != masks.shape:
            entry = (torch.empty(masks.shape, dtype=masks.dtype, pin_memory=True), torch.cuda.Event())
            if len(ring) <= slot:
                ring.append(entry)
            else:
                ring[slot] = entry
        buf, done = ring[slot]
        done.synchronize()                            # the copy that last used this buffer (four uploads ago) has finished
        buf.copy_(masks)
        out = buf.to(dev, non_blocking=True)
        done.record(torch.cuda.current_stream(dev))
        return out

    def _sean_distill(self):
        return self.opt.style_norm_block_type == "sean" and getattr(self.opt, "style_distill", False)

    def _compute_mae_generator_loss(self, imgs, labels):
        """(rec, gan, clf): l1(G(masked), x), bce(D(G(masked)), 1), bce(cls(G(masked)), labels); with SEAN's --style_distill
        (defectgan_model.py:106-128) the two distillation terms are appended (logged only: their gradients were taken inside
        the SEAN layers' forward) -- but not in the --split_training return, like the reference"""
        if self._sean_distill():
            self.netG.enable_sean_distill_loss(True)
        predicted, _ = self._repair_mask(imgs, labels)
        distill = None
        if self._sean_distill():
            distill = self.netG.get_sean_distill_loss()
            self.netG.enable_sean_distill_loss(False)
        rec_loss = self._cal_loss(predicted, imgs, "l1")
        if getattr(self.opt, "split_training", False):      # --split_training (defectgan_model.py:119-120): G sees only the L1 loss
            zero = torch.zeros([], device=rec_loss.device)
            return rec_loss, zero, zero
        d_params = [p for p in self.netD.parameters() if p.requires_grad]     # D's weight gradients are never read here
        for p in d_params:
            p.requires_grad_(False)
        try:
            fake_src, fake_cls = self.netD(predicted)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        out = (rec_loss, self._cal_loss(fake_src, 1.0, "bce"), self._cal_loss(fake_cls, labels.view_as(fake_cls), self.clf_loss_type))
        return out + (distill["latent"], distill["embed"]) if distill is not None else out

    def _compute_mae_discriminator_loss(self, imgs, labels):
        """(gan, clf): mean(bce(D(G(masked)), 0), bce(D(x), 1)), bce(cls(x), labels); G runs in eval mode, no grad"""
        if getattr(self.opt, "split_training", False):      # :157-158: D only learns the classifier, on the real images
            _, real_cls = self.netD(imgs)
            clf_loss = self._cal_loss(real_cls, labels.view_as(real_cls), self.clf_loss_type)
            return torch.zeros([], device=clf_loss.device), clf_loss
        with torch.no_grad():
            predicted, _ = self._repair_mask(imgs, labels)
        (real_src, real_cls), (fake_src, _) = self._netD_batched(imgs, predicted.detach())
        clf_loss = self._cal_loss(real_cls, labels.view_as(real_cls), self.clf_loss_type)
        gan_loss = self._mean([self._cal_loss(fake_src, 0.0, "bce"), self._cal_loss(real_src, 1.0, "bce")])
        return gan_loss, clf_loss

    def _forks_generator_chains(self, bg_data, nm_feat, flag=None):
        """The plain SPADE generator only: spectral norm iterates (u, v) in place per forward and NoiseInjection draws from one RNG
        -- both are ordered by the reference's pass order --, the style variants bring a second trained network into the passes."""
        o = self.opt
        return ((ops.forked_chains if flag is None else flag) and bg_data.is_cuda and nm_feat is None and o.style_norm_block_type == "spade"
                and not getattr(o, "use_spectral", False) and not getattr(o, "add_noise", False) and not getattr(o, "cycle_gan", False)
                and not os.environ.get("DEI2I_SPLIT_D"))

    def _pairs_generator_passes(self, bg_data, df_data, nm_feat):
        """One pass over [bg | df] and one over [fake_defects | fake_normals] instead of four (ops.bn_batch_groups): the same
        generators as _forks_generator_chains takes, in training mode (BatchNorm per group of the batch), equal batch sizes."""
        return (ops.paired_passes and self._forks_generator_chains(bg_data, nm_feat, flag=True) and self.netG.training
                and bg_data.shape == df_data.shape)

    def _paired_label_sets(self, labels_in, nm_labels, df_labels):
        """-> ([df | nm], [nm | df], same): the label tensors of the paired passes.  ``same``: they are the very tensors the last
        call made for this incoming label tensor (the D step and the G step of one iteration get the same one) -- the SPADE class
        tables are memoized per label TENSOR and parameter state, so the tables the D step computed serve the G step's passes (the
        generator's parameters do not change in between).  The incoming tensor is held, so no other tensor can take its identity."""
        st = getattr(self, "_label_sets", None)
        if st is not None and st[0] is labels_in and st[1] == labels_in._version:
            return st[2], st[3], True
        head, tail = torch.cat([df_labels, nm_labels], 0), torch.cat([nm_labels, df_labels], 0)
        self._label_sets = (labels_in, labels_in._version, head, tail)
        return head, tail, False

    def _compute_generator_loss(self, bg_data, df_labels, df_data):
        """defectgan_model.py:173-249"""
        labels_in = df_labels
        nm_labels, df_labels = self._get_labels(df_labels)
        nm_feat, df_feat = self._style_feats(bg_data, nm_labels, df_labels, df_data)
        paired = self._pairs_generator_passes(bg_data, df_data, nm_feat)
        if paired:
            head_labels, tail_labels, same = self._paired_label_sets(labels_in, nm_labels, df_labels)
            if not same:
                self.netG.clear_spade_cache()
            self.netG.prime_spade((head_labels, tail_labels))     # (a no-op for the modules whose tables the D step left)
        else:
            self.netG.clear_spade_cache()
            if not os.environ.get("DEI2I_SPLIT_D") and nm_feat is None:
                self.netG.prime_spade((df_labels, nm_labels))    # both label sets' SPADE tables in one pass
        sean = self.opt.style_norm_block_type == "sean"       # defectgan_model.py:177-182,192-197: what the four passes also feed
        if self._sean_distill():
            self.netG.enable_sean_distill_loss(True)
        if sean and getattr(self.opt, "use_running_stats", False):
            self.netG.track_running_stats = True
        if paired:
            # the chain heads bg -> fake_defects | df -> fake_normals as one pass over 2 x batch, the chain tails -> recover_normals |
            # -> recover_defects as another: everything in this generator acts per sample except training-mode BatchNorm, which
            # takes its statistics per half of the batch (ops.bn_batch_groups) and replays the four running-statistics updates in
            # the reference's pass order (ops.bn_running_deferred: heads are passes 0 and 2, tails 1 and 3)
            n = bg_data.shape[0]
            with ops.bn_running_deferred() as running, ops.bn_batch_groups(2):
                running.pass_index = (0, 2)
                reals = torch.cat([bg_data, df_data], 0)
                fakes, probs = self.netG(reals, head_labels, None)
                running.pass_index = (1, 3)
                recovers, rec_probs = self.netG(fakes, tail_labels, None)
                running.apply()
            fake_defects, fake_normals = fakes[:n], fakes[n:]
            df_prob, nm_prob = probs[:n], probs[n:]
            recover_normals, recover_defects = recovers[:n], recovers[n:]
            rec_df_prob, rec_nm_prob = rec_probs[:n], rec_probs[n:]
        elif self._forks_generator_chains(bg_data, nm_feat):
            # bg -> fake_defects -> recover_normals and df -> fake_normals -> recover_defects share nothing but the parameters: two
            # streams (ops.forked_chains).  BatchNorm's four running-statistics updates are replayed in the reference's pass order
            # after the join (ops.bn_running_deferred); everything downstream (D, the losses) runs on the joining stream.
            main = torch.cuda.current_stream(bg_data.device)
            chain_a, chain_b = ops.chain_streams(bg_data.device)
            with ops.bn_running_deferred() as running:
                chain_a.wait_stream(main)
                chain_b.wait_stream(main)
                with torch.cuda.stream(chain_a):
                    running.pass_index = 0
                    fake_defects, df_prob = self.netG(bg_data, df_labels, df_feat)
                    running.pass_index = 1
                    recover_normals, rec_df_prob = self.netG(fake_defects, nm_labels, nm_feat)
                with torch.cuda.stream(chain_b):
                    running.pass_index = 2
                    fake_normals, nm_prob = self.netG(df_data, nm_labels, nm_feat)
                    running.pass_index = 3
                    recover_defects, rec_nm_prob = self.netG(fake_normals, df_labels, df_feat)
                main.wait_stream(chain_a)
                main.wait_stream(chain_b)
                running.apply()
            for t in (fake_defects, df_prob, recover_normals, rec_df_prob, fake_normals, nm_prob, recover_defects, rec_nm_prob):
                t.record_stream(main)                       # (allocated on a chain's stream, consumed on the joining one)
        else:
            fake_defects, df_prob = self.netG(bg_data, df_labels, df_feat)
            recover_normals, rec_df_prob = self.netG(fake_defects, nm_labels, nm_feat)
            fake_normals, nm_prob = self.netG(df_data, nm_labels, nm_feat)
            recover_defects, rec_nm_prob = self.netG(fake_normals, df_labels, df_feat)
        distill = None
        if self._sean_distill():
            distill = self.netG.get_sean_distill_loss()
            self.netG.enable_sean_distill_loss(False)
        if sean and getattr(self.opt, "use_running_stats", False):
            self.netG.track_running_stats = False

        # The reference lets autograd compute (and then discards) the discriminator's weight gradients here; the
        # G optimizer never reads them and optimizers['D'].zero_grad() drops them, so they are skipped.
        d_params = [p for p in self.netD.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        policy = getattr(self.opt, "diff_aug", "")            # DiffAugment on what D sees (defectgan_model.py:200-203)
        whole = paired and not policy
        try:
            if whole:
                fakes_src, fakes_cls = self.netD(fakes)
            else:
                (fake_defects_src, fake_defects_cls), (fake_normals_src, fake_normals_cls) = \
                    self._netD_batched(diff_augment(fake_defects, policy), diff_augment(fake_normals, policy))
        finally:
            for p in d_params:
                p.requires_grad_(True)
        if whole:
            # Every loss group of the reference is a mean of per-pass means over equally sized halves of what the paired passes hold
            # as ONE tensor: the mean over the whole tensor is the same number (to the rounding of a differently ordered fp32 sum),
            # without slicing the passes' outputs apart (a zero fill, a copy and an add per slice in backward).
            gan_loss = self._cal_loss(fakes_src, 1.0, "bce")
            clf_loss = self._cal_loss(fakes_cls, head_labels.view_as(fakes_cls), self.clf_loss_type)
            rec_loss = self._cal_loss(recovers, reals, "l1")              # [recover_normals | recover_defects] vs [bg | df]
            sd_cyc_loss = self._cal_loss(probs, rec_probs, "l1")      # (--cycle_gan never pairs: _forks_generator_chains)
            sd_con_loss = self._mean([self._cal_loss(probs, None, "l1"), self._cal_loss(rec_probs, None, "l1")])
            out = (gan_loss, clf_loss, rec_loss, sd_cyc_loss, sd_con_loss)
            return out + (distill["latent"], distill["embed"]) if distill is not None else out

        gan_loss = [self._cal_loss(fake_defects_src, 1.0, "bce"), self._cal_loss(fake_normals_src, 1.0, "bce")]
        clf_loss = [self._cal_loss(fake_defects_cls, df_labels.view_as(fake_defects_cls), self.clf_loss_type),
                    self._cal_loss(fake_normals_cls, nm_labels.view_as(fake_normals_cls), self.clf_loss_type)]
        rec_loss = [self._cal_loss(recover_defects, df_data, "l1"), self._cal_loss(recover_normals, bg_data, "l1")]
        if self.opt.cycle_gan:                              # defectgan_model.py:222-227: no spatial-distribution losses
            zero = torch.zeros([], device=rec_loss[0].device)
            return self._mean(gan_loss), self._mean(clf_loss), self._mean(rec_loss), zero, zero
        sd_cyc_loss = [self._cal_loss(df_prob, rec_df_prob, "l1"), self._cal_loss(nm_prob, rec_nm_prob, "l1")]
        sd_con_loss = [self._cal_loss(df_prob, None, "l1"), self._cal_loss(nm_prob, None, "l1"),
                       self._cal_loss(rec_df_prob, None, "l1"), self._cal_loss(rec_nm_prob, None, "l1")]
        out = (self._mean(gan_loss), self._mean(clf_loss), self._mean(rec_loss), self._mean(sd_cyc_loss), self._mean(sd_con_loss))
        return out + (distill["latent"], distill["embed"]) if distill is not None else out      # (:238-244; not in the cycle_gan return)

    def _compute_discriminator_loss(self, bg_data, df_labels, df_data):
        """defectgan_model.py:251-292"""
        labels_in = df_labels
        nm_labels, df_labels = self._get_labels(df_labels)
        nm_feat, df_feat = self._style_feats(bg_data, nm_labels, df_labels, df_data)
        self.netG.clear_spade_cache()
        with torch.no_grad():
            # (--add_noise with an INJECTED noise source -- the parity tests: keep the reference's two calls so that the
            #  source is consumed exactly as the reference consumes it; with the device RNG one pass over both batches
            #  draws the same i.i.d. N(0,1) field, just in one call)
            injected_noise = getattr(self.opt, "add_noise", False) and ops.noise_source is not None
            if self.netG.training or os.environ.get("DEI2I_SPLIT_D") or injected_noise:
                fake_defects, _ = self.netG(bg_data, df_labels, df_feat)
                fake_normals, _ = self.netG(df_data, nm_labels, nm_feat)
            else:
                # netG is in eval mode here (defectgan_model.py:87-90): BatchNorm uses running statistics and SPADE's / AdaIN's
                # InstanceNorm is per sample, so one pass over both batches is the same function as two passes
                feats = None if nm_feat is None else torch.cat([df_feat, nm_feat], 0)
                both_labels = None
                if (ops.paired_passes and self._forks_generator_chains(bg_data, nm_feat, flag=True) and bg_data.shape == df_data.shape
                        and any(p.requires_grad for p in self.netG.parameters())):
                    # the G step of this iteration will run its passes paired over the same label sets: compute the SPADE class
                    # tables of both sets now, WITH their autograd history, and let that step find them (_paired_label_sets) --
                    # one table computation per iteration instead of two (20 small convs and as many launches of glue)
                    both_labels, tail_labels, _ = self._paired_label_sets(labels_in, nm_labels, df_labels)
                    with torch.enable_grad():
                        self.netG.prime_spade((both_labels, tail_labels))
                if both_labels is None:
                    both_labels = torch.cat([df_labels, nm_labels], 0)
                fakes, _ = self.netG(torch.cat([bg_data, df_data], 0), both_labels, feats)
                fake_defects, fake_normals = fakes.split([bg_data.shape[0], df_data.shape[0]])
        policy = getattr(self.opt, "diff_aug", "")            # defectgan_model.py:266-270: fakes first, then the real batches
        fake_defects, fake_normals = diff_augment(fake_defects.detach(), policy), diff_augment(fake_normals.detach(), policy)
        df_data, bg_data = diff_augment(df_data, policy), diff_augment(bg_data, policy)
        (fake_defects_src, _), (fake_normals_src, _), (real_defects_src, real_defects_cls), \
            (real_normals_src, real_normals_cls) = self._netD_batched(fake_defects, fake_normals, df_data, bg_data)
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
        if self.opt.style_norm_block_type == "adain":            # defectgan_model.py:310-312
            return self.netG(data, labels, self.netE(data, labels))
        if self.opt.style_norm_block_type == "sean":             # :304-306
            return self.netG(data, labels, self._get_style_embeds(labels))
        return self.netG(data, self._expand_seg(labels))

    def _get_style_embeds(self, labels):
        """defectgan_model.py:394-411: per sample ``num_embeds`` embeddings drawn (python's ``random.choices``, like the
        reference) from the list stored for its label tuple -- zeros when that list is empty; None with --sean_alpha 0."""
        if self.opt.sean_alpha == 0:
            return None
        num_embeds = self.opt.num_embeds
        embed_list = []
        for label in labels.reshape(labels.size(0), -1):
            tuple_label = tuple(label.int().tolist())
            if not self.embeddings[tuple_label]:
                mean_embed = torch.zeros(num_embeds, self.opt.embed_nc, device=self.opt.device)
            else:
                mean_embed = torch.stack(random.choices(self.embeddings[tuple_label], k=num_embeds))
            embed_list.append(mean_embed)
        return torch.stack(embed_list)

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

    def _style_feats(self, bg_data, nm_labels, df_labels, df_data):
        """-> (nm_label_feat, df_label_feat): netE(bg, nm_labels) then netE(df, df_labels) for the AdaIN decoder
        (defectgan_model.py:423-425), (None, None) for SPADE."""
        if self.opt.style_norm_block_type == "sean":             # :417-419: the normal labels' embeddings are drawn first
            return self._get_style_embeds(nm_labels), self._get_style_embeds(df_labels)
        if self.opt.style_norm_block_type != "adain":
            return None, None
        n = nm_labels.shape[0]
        return self.netE(bg_data, nm_labels.reshape(n, -1)), self.netE(df_data, df_labels.reshape(n, -1))
