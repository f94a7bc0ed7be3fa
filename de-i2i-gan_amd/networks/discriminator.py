"""DefectGanDiscriminator (models/networks/discriminator.py:49-98): PatchGAN of stride-2 4x4 reflect-pad convs +
LeakyReLU(0.2), no normalisation; src_clf 3x3 patch logits, cls_clf full-extent conv -> label logits."""
import numpy as np
import torch
from torch import nn

from .. import ops
from .architecture import ConvBlock
from .base_network import BaseNetwork


class DefectGanDiscriminator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.prec = ops.get_precision(getattr(opt, "compute_dtype", "bf16"))
        self.label_nc = opt.label_nc
        crt_dim = opt.ndf
        stem = ConvBlock(opt.input_nc, crt_dim, kernel_size=(4, 4), stride=(2, 2), padding=1, padding_mode="reflect",
                         norm_layer=None, act_layer="leaky_relu", use_spectral=opt.use_spectral)
        conv_blk = []
        for _ in range(opt.num_layers):
            conv_blk.append(ConvBlock(crt_dim, crt_dim * 2, kernel_size=(4, 4), stride=(2, 2), padding=1,
                                      padding_mode="reflect", norm_layer=None, act_layer="leaky_relu",
                                      use_spectral=opt.use_spectral))
            crt_dim *= 2
        kernel_size = int(opt.image_size // np.power(2, opt.num_layers + 1))
        self.enc_blk = nn.Sequential(stem, *conv_blk)
        self.cls_clf = ConvBlock(crt_dim, opt.label_nc, kernel_size=(kernel_size, kernel_size), norm_layer=None, act_layer=None)
        self.src_clf = ConvBlock(crt_dim, 1, kernel_size=(3, 3), stride=(1, 1), padding="same", padding_mode="reflect",
                                 norm_layer=None, act_layer=None)

    def forward(self, x):
        assert isinstance(x, torch.Tensor), "x must be Original Images: Torch.Tensor"
        feat = ops.to_nhwc(x, self.prec)
        for blk in self.enc_blk:
            feat = blk(feat)
        src_logits = ops.to_nchw(self.src_clf(feat), 1)
        cls_logits = ops.to_nchw(self.cls_clf(feat), self.label_nc)
        return src_logits, cls_logits.reshape((cls_logits.size(0), cls_logits.size(1)))
