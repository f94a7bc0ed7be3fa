"""StyleExtractor (models/networks/extractor.py:36-96): the network ``netE`` that feeds the AdaIN decoder variant
(``--style_norm_block_type adain``) its per-sample style feature (N, hidden_nc).

``sean_alpha == 0``: five Linear layers (ReLU between them) on ``[labels | noise]``, noise ~ N(0,1) of width
``latent_dim - label_nc`` drawn per call on the device (tests install ``ops.noise_source``) -- (N x 256) GEMMs of a few
thousand FLOPs: plain library GEMMs through torch, the plumbing the host side is allowed.  ``sean_alpha == 1``: the conv
encoder on the image (extractor.py:50-80) -- 7x7 stride-2 conv + LeakyReLU, log2(image_size) - 3 down-scaling ResBlocks
(3x3 convs, InstanceNorm2d, LeakyReLU, AvgPool2d) and a 4x4 valid conv to (N, hidden_nc, 1, 1) -- on the HIP kernels."""
import math

import torch
from torch import nn

from .. import ops
from .architecture import ConvBlock, ResBlock
from .base_network import BaseNetwork


class StyleExtractor(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        assert opt.image_size in (64, 128, 256, 512, 1024), "image size should be one of [64, 128, 256, 512, 1024]"
        num_blocks = int(math.log2(opt.image_size)) - 3
        max_dim = 256
        self.hidden_nc = opt.hidden_nc
        self.prec = ops.get_precision(getattr(opt, "compute_dtype", "bf16"))
        self.sean_alpha = opt.sean_alpha
        self.noise_dim = opt.latent_dim - opt.label_nc
        if opt.sean_alpha == 0:
            layers = [nn.Linear(opt.latent_dim, max_dim), nn.ReLU(inplace=True)]
            for _ in range(3):
                layers += [nn.Linear(max_dim, max_dim), nn.ReLU(inplace=True)]
            layers.append(nn.Linear(max_dim, opt.hidden_nc))
            self.shared = nn.Sequential(*layers)
        elif opt.sean_alpha == 1:
            crt_dim = opt.ndf
            blocks = [ConvBlock(opt.input_nc, crt_dim, kernel_size=(7, 7), stride=(2, 2), padding=3, padding_mode="reflect",
                                norm_layer=None, act_layer="leaky_relu", use_spectral=False)]
            for _ in range(num_blocks):
                new_dim = min(crt_dim * 2, max_dim)
                blocks.append(ResBlock(crt_dim, new_dim, kernel_size=(3, 3), stride=(1, 1), padding="same", padding_mode="reflect",
                                       norm_layer=nn.InstanceNorm2d, act_layer="leaky_relu", use_spectral=False, down_scale=True))
                crt_dim = new_dim
            blocks.append(ConvBlock(crt_dim, opt.hidden_nc, kernel_size=(4, 4), stride=(1, 1), padding=0, norm_layer=None,
                                    act_layer=None, use_spectral=False))
            self.shared = nn.Sequential(*blocks)
        else:
            raise NotImplementedError("sean_alpha should be 0 or 1")

    def forward(self, x, labels):
        if self.sean_alpha == 1:                             # extractor.py:92-93: the image encoder; labels are not read
            feat = ops.to_nhwc(x, self.prec)
            for blk in self.shared:
                feat = blk(feat)
            return ops.to_nchw(feat, self.hidden_nc)         # (N, hidden_nc, 1, 1), what AdaIN views as (N, hidden_nc)
        noise = ops.draw_noise((labels.size(0), self.noise_dim), x.device)          # extractor.py:89
        latent = torch.cat([labels.reshape(labels.size(0), -1).float(), noise.to(x.device).float()], dim=1)
        return self.shared(latent)
