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


WIDEST = 256            # widest layer of either form (extractor.py:41)


class StyleExtractor(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        if opt.image_size not in (64, 128, 256, 512, 1024):
            raise AssertionError("image size should be one of [64, 128, 256, 512, 1024]")
        self.sean_alpha, self.hidden_nc = opt.sean_alpha, opt.hidden_nc
        self.noise_dim = opt.latent_dim - opt.label_nc
        self.prec = ops.get_precision(getattr(opt, "compute_dtype", "bf16"))
        if self.sean_alpha == 0:                              # MLP on [labels | noise]: latent_dim -> 256 x 4 -> hidden_nc
            widths = [opt.latent_dim] + [WIDEST] * 4
            mlp = []
            for fan_in, fan_out in zip(widths[:-1], widths[1:]):
                mlp += [nn.Linear(fan_in, fan_out), nn.ReLU(inplace=True)]
            self.shared = nn.Sequential(*mlp, nn.Linear(WIDEST, opt.hidden_nc))
        elif self.sean_alpha == 1:                            # conv encoder on the image: halves the size down to 4 x 4, then 4x4 -> 1x1
            halvings = int(math.log2(opt.image_size)) - 3      # after the stride-2 stem
            chans = [opt.ndf]
            for _ in range(halvings):
                chans.append(min(2 * chans[-1], WIDEST))
            enc = [ConvBlock(opt.input_nc, chans[0], (7, 7), (2, 2), 3, "reflect", norm_layer=None, act_layer="leaky_relu")]
            enc += [ResBlock(c_in, c_out, (3, 3), (1, 1), "same", "reflect", norm_layer=nn.InstanceNorm2d, act_layer="leaky_relu",
                             down_scale=True) for c_in, c_out in zip(chans[:-1], chans[1:])]
            enc.append(ConvBlock(chans[-1], opt.hidden_nc, (4, 4), (1, 1), 0, norm_layer=None, act_layer=None))
            self.shared = nn.Sequential(*enc)
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
