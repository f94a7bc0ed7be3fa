"""DefectGanGenerator with the reference's constructor/forward signature and state_dict keys
(models/networks/generator.py:52-275), running on the HIP kernels."""
import math

import torch
from torch import nn

from .. import ops
from .architecture import BatchNorm2d, ConvBlock, DeConvBlock, NormConvBlock, NormResBlock, ResBlock
from .base_network import BaseNetwork


class DefectGanGenerator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        assert (opt.num_res & 1) == 0, "num_res must be even"
        self.opt = opt
        self.cycle_gan = opt.cycle_gan
        self.label_nc = opt.label_nc
        self.skip_conn = opt.skip_conn
        if opt.skip_conn:
            raise NotImplementedError("skip_conn: the reference's UnetBlock path does not run as written "
                                      "(architecture.py:451,504-513)")
        self.prec = ops.get_precision(getattr(opt, "compute_dtype", "bf16"))
        self.fp8 = ops.wants_fp8(getattr(opt, "compute_dtype", "bf16"))     # e4m3 forward GEMMs of the 3x3 convs

        crt_dim = opt.ngf
        self.stem = ConvBlock(opt.input_nc, crt_dim, kernel_size=(7, 7), padding="same", padding_mode="reflect",
                              norm_layer=BatchNorm2d, act_layer="leaky_relu", use_spectral=opt.use_spectral)
        conv_blk, de_conv_blk, enc_res_blk, dec_res_blk = [], [], [], []
        for _ in range(opt.num_scales):
            conv_blk.append(ConvBlock(crt_dim, crt_dim * 2, kernel_size=(4, 4), stride=(2, 2), padding=1,
                                      padding_mode="reflect", norm_layer=BatchNorm2d, act_layer="leaky_relu",
                                      use_spectral=opt.use_spectral))
            crt_dim *= 2
        for _ in range(opt.num_res // 2):
            enc_res_blk.append(ResBlock(crt_dim, crt_dim, kernel_size=(3, 3), stride=(1, 1), padding="same",
                                        padding_mode="reflect", norm_layer=BatchNorm2d, act_layer="leaky_relu",
                                        use_spectral=opt.use_spectral))
        for _ in range(opt.num_res // 2, opt.num_res):
            dec_res_blk.append(NormResBlock(opt.style_norm_block_type, opt.hidden_nc, opt.label_nc, crt_dim, crt_dim,
                                            style_distill=opt.style_distill, embed_nc=opt.embed_nc, kernel_size=(3, 3),
                                            stride=(1, 1), padding="same", padding_mode="reflect", up_scale=False,
                                            act_layer="relu", use_spectral=opt.use_spectral, add_noise=opt.add_noise))
        for _ in range(opt.num_scales):
            de_conv_blk.append(NormConvBlock(opt.style_norm_block_type, opt.hidden_nc, opt.label_nc, crt_dim, crt_dim // 2,
                                             style_distill=opt.style_distill, embed_nc=opt.embed_nc, kernel_size=(3, 3),
                                             stride=(1, 1), padding="same", padding_mode="reflect", up_scale=True,
                                             act_layer="relu", use_spectral=opt.use_spectral, add_noise=opt.add_noise))
            crt_dim //= 2
        self.enc_blk = nn.Sequential(*conv_blk)
        self.enc_res_blk = nn.Sequential(*enc_res_blk)
        self.dec_res_blk = nn.Sequential(*dec_res_blk)
        self.dec_blk = nn.Sequential(*de_conv_blk)
        self.foreground_head = DeConvBlock(crt_dim, 3, kernel_size=(3, 3), padding="same", padding_mode="reflect",
                                           up_scale=False, norm_layer=None, act_layer="tanh")
        self.distribution_head = DeConvBlock(crt_dim, 1, kernel_size=(3, 3), padding="same", padding_mode="reflect",
                                             up_scale=False, norm_layer=None, act_layer="sigmoid")
        self._head_dim = crt_dim
        self._packed_heads = ops.PackedWeights()
        self._packed_label_path = {}               # packed copies of the SPADE modules' concatenated first-conv filters, per group

    def forward(self, x, labels, style_feat=None):
        with ops.fp8_forward(self.fp8):
            out = self._forward(x, labels, style_feat)
        if not self.__dict__.get("_marked_table_modules"):           # once: the table modules the first forward did not reach never run
            for m in self._table_modules():
                if "_ran_class_mode" not in m.__dict__:
                    m._ran_class_mode = False
            self.__dict__["_marked_table_modules"] = True
        return out

    def _forward(self, x, labels, style_feat=None):
        assert isinstance(x, torch.Tensor), "x must be Original Images: Torch.Tensor"
        if labels.dim() == 2:
            labels = labels.reshape(labels.size(0), labels.size(1), 1, 1)
        feat = self.stem(ops.to_nhwc(x, self.prec))
        for enc_blk in self.enc_blk:
            feat = enc_blk(feat, labels)
        # out_stats: the next layer is a norm -- the producer leaves the statistics records of its output (ops._stats_of)
        n_enc, n_dec, n_up = len(self.enc_res_blk), len(self.dec_res_blk), len(self.dec_blk)
        for i, enc_res_blk in enumerate(self.enc_res_blk):
            feat = enc_res_blk(feat, labels, out_stats=(i == n_enc - 1))
        for i, dec_res_blk in enumerate(self.dec_res_blk):
            feat = dec_res_blk(feat, labels, style_feat, out_stats=True)
        for i, dec_blk in enumerate(self.dec_blk):
            feat = dec_blk(feat, labels, style_feat, out_stats=(i < n_up - 1))
        # generator.py:266-267 -- nan_to_num only when a NaN is present; device-side flag, no host sync
        with torch.no_grad():
            ops.nan_guard_(feat)
        # both heads as one 4-output conv (3 tanh channels + 1 sigmoid channel), then the compose kernel
        fg_w, pr_w = self.foreground_head.conv.weight, self.distribution_head.conv.weight
        if self.cycle_gan:
            # generator.py:272-273: (foreground, spatial_prob), no composition.  Two separate head convs: the losses of this mode
            # never read spatial_prob (defectgan_model.py:222-227), and the distribution head then gets NO gradient (None, as
            # in the reference) rather than the zeros a fused 4-output conv would hand it
            fg = ops.to_nchw(self.foreground_head.conv(feat), 3)
            pr = ops.to_nchw(self.distribution_head.conv(feat), 1)
            return torch.tanh(fg), torch.sigmoid(pr)
        w4 = torch.cat([fg_w, pr_w], 0)
        geom = ops.ConvGeom(self._head_dim, 4, 3, 1, 1, True, False)
        raw = ops.conv2d(feat, w4, None, self._packed_heads, geom, "none", sources=(fg_w, pr_w))
        output, spatial_prob = ops.compose(raw, x)
        return output, spatial_prob

    def prime_spade(self, label_tensors):
        """Batch the SPADE class-table computation of the label sets a loss graph is about to use (see SPADE.prime)."""
        mods = [m for m in self._table_modules() if hasattr(m, "prime")]
        if not mods:
            return
        if len(label_tensors) not in (1, 2) or not all(t.dim() == 4 and t.shape[2:] == (1, 1) for t in label_tensors):
            for m in mods:
                m.prime(label_tensors, self.prec)
            return
        # one concatenation and one 5x5 class image for all modules
        both = torch.cat(list(label_tensors), 0) if len(label_tensors) > 1 else label_tensors[0]
        seg = ops.to_nhwc(both, self.prec, size=(5, 5))
        # The first conv of every module's label path (normalization.py:17-19: label_nc -> hidden_nc, 3x3, ReLU) reads the SAME class
        # image: the modules that will run are served by ONE conv over their concatenated filters (each output channel of a conv is a
        # function of its own filter only), their slices of its output handed to them -- one launch instead of one per module,
        # forward, weight gradient and bias gradient alike.
        live = [m for m in mods if m.wants_prime(label_tensors, self.prec)]
        shared, ready = {}, {}
        convs = [m.mlp_shared[0] for m in live]
        if len(live) > 1 and all(type(c) is type(convs[0]) and c.out_channels == convs[0].out_channels and c.bias is not None
                                 and c.weight.shape == convs[0].weight.shape and c.effective_weight()[0] is c.weight for c in convs):
            # (the per-channel reduction kernels take 256 channel vectors at most: 2 048 channels in bf16, 1 024 in fp32)
            per = max(1, (256 * (8 if self.prec is ops.BF16 else 4)) // convs[0].out_channels)
            for lo in range(0, len(live), per):
                grp, cv = live[lo:lo + per], convs[lo:lo + per]
                if len(grp) < 2:
                    continue
                w_all = torch.cat([c.weight for c in cv], 0)
                b_all = torch.cat([c.bias for c in cv], 0)
                geom = ops.ConvGeom(cv[0].in_channels, w_all.shape[0], cv[0].kernel_size, 1, cv[0].padding, False, False)
                cache = self._packed_label_path.setdefault(lo, ops.PackedWeights())
                actv = ops.conv2d(seg, w_all, b_all, cache, geom, "relu", sources=tuple(c.weight for c in cv))
                second = [(m.mlp_gamma, m.mlp_beta) for m in grp]
                if ops.label_path_batched and ops.label_gamma_beta_supported(actv, cv[0].out_channels, second):
                    # ... and the second stage (hidden -> gamma | beta, per module on its slice of actv) as one launch for the group
                    tables = ops.label_gamma_beta(actv, cv[0].out_channels, second, self._packed_label_path.setdefault(("gb", lo), {}))
                    for m, gb in zip(grp, tables):
                        ready[id(m)] = gb
                else:
                    for m, part in zip(grp, actv.split(cv[0].out_channels, dim=-1)):
                        shared[id(m)] = part
        for m in mods:
            m.prime(label_tensors, self.prec, both, seg, shared.get(id(m)), ready.get(id(m)))

    def clear_spade_cache(self):
        """Drop the memoized SPADE gamma/beta tables (they carry autograd history: a table must not outlive the loss
        graph it was built in).  The model calls this at the start of every loss computation."""
        for m in self._table_modules():
            m._gb_cache.clear()

    def _table_modules(self):
        """The modules that memoize a gamma / beta table (walked a few times per step: the list is made once; the module tree is fixed
        after construction)."""
        mods = self.__dict__.get("_table_modules_list")
        if mods is None:
            mods = self.__dict__["_table_modules_list"] = [m for m in self.modules() if hasattr(m, "_gb_cache")]
        return mods

    def update_per_epoch(self, epoch):
        """generator.py:277-284"""
        super().update_per_epoch(epoch)
        alpha = (1 + math.cos(math.pi * epoch / self.opt.num_epochs)) / 2
        if self.opt.style_norm_block_type == "sean":
            if self.opt.sean_alpha is None:
                self.set_sean_alpha(alpha)
            if getattr(self.opt, "use_running_stats", False):
                self.update_stats()

    def _seans(self):
        from .architecture import SEAN
        return [m for m in self.modules() if isinstance(m, SEAN)]

    def set_sean_alpha(self, alpha):
        """generator.py:286-289"""
        for m in self._seans():
            m.set_alpha(alpha)

    def enable_sean_distill_loss(self, enable_distill_loss):
        """generator.py:291-294"""
        for m in self._seans():
            m.distill_loss = enable_distill_loss

    def get_sean_distill_loss(self):
        """generator.py:296-306: the terms every SEAN layer collected since distillation was enabled, averaged per kind
        (an empty list stays an empty list)"""
        out = {"latent": [], "embed": []}
        for m in self._seans():
            if m.distill_loss:
                for kind in out:
                    out[kind] += m.distill_loss[kind]
        for kind in out:
            if out[kind]:
                out[kind] = torch.stack(out[kind]).mean()
        return out

    def update_stats(self):
        """generator.py:308-311"""
        for m in self._seans():
            m.update_stats()

    @property
    def track_running_stats(self):
        """generator.py:313-323 (the first SEAN layer's flag; the setter writes all of them)"""
        for m in self._seans():
            return m.track_running_stats

    @track_running_stats.setter
    def track_running_stats(self, value):
        for m in self._seans():
            m.track_running_stats = value

    @property
    def inference_running_stats(self):
        """generator.py:325-335"""
        for m in self._seans():
            return m.inference_running_stats

    @inference_running_stats.setter
    def inference_running_stats(self, value):
        for m in self._seans():
            m.inference_running_stats = value
