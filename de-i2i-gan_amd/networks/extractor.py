"""StyleExtractor (models/networks/extractor.py:36-96): the network ``netE`` that feeds the AdaIN decoder variant
(``--style_norm_block_type adain``) its per-sample style feature (N, hidden_nc).

``sean_alpha == 0``: five Linear layers (ReLU between them) on ``[labels | noise]``, noise ~ N(0,1) of width
``latent_dim - label_nc`` drawn per call on the device (tests install ``ops.noise_source``) -- (N x 256) GEMMs of a few
thousand FLOPs: plain library GEMMs through torch, the plumbing the host side is allowed.  ``sean_alpha == 1`` (a conv
encoder of down-scaling InstanceNorm ResBlocks on the image) is not implemented."""
import math

import torch
from torch import nn

from .. import ops
from .base_network import BaseNetwork


class StyleExtractor(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        assert opt.image_size in (64, 128, 256, 512, 1024), "image size should be one of [64, 128, 256, 512, 1024]"
        _ = int(math.log2(opt.image_size)) - 3              # num_blocks of the conv variant
        max_dim = 256
        self.sean_alpha = opt.sean_alpha
        self.noise_dim = opt.latent_dim - opt.label_nc
        if opt.sean_alpha == 0:
            layers = [nn.Linear(opt.latent_dim, max_dim), nn.ReLU(inplace=True)]
            for _ in range(3):
                layers += [nn.Linear(max_dim, max_dim), nn.ReLU(inplace=True)]
            layers.append(nn.Linear(max_dim, opt.hidden_nc))
            self.shared = nn.Sequential(*layers)
        elif opt.sean_alpha == 1:
            raise NotImplementedError("StyleExtractor with sean_alpha == 1 (conv encoder of down-scaling ResBlocks, "
                                      "extractor.py:50-80) is not implemented")
        else:
            raise NotImplementedError("sean_alpha should be 0 or 1")

    def forward(self, x, labels):
        noise = ops.draw_noise((labels.size(0), self.noise_dim), x.device)          # extractor.py:89
        latent = torch.cat([labels.reshape(labels.size(0), -1).float(), noise.to(x.device).float()], dim=1)
        return self.shared(latent)
