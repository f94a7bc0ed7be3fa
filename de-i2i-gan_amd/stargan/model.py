"""The reference's stargan-v2 networks (stargan-v2/core/model.py) with its constructor signatures, forward signatures and
``state_dict`` keys (NCHW fp32 parameters), computed by the HIP ops: NHWC activations in the compute dtype between the image-side
boundaries.  ``nn.Conv2d`` / ``nn.Linear`` / ``nn.InstanceNorm2d(affine=True)`` children are plain parameter containers with the
reference's names; every block's arithmetic is in its ``forward``.

Scope: ``norm_type='adain'`` (SEAN blocks need the ViT feature extractor's downloaded weights, model.py:541-542) and ``w_hpf = 0`` (the
high-pass skip needs the FAN landmark network, model.py:725-729) -- both raise otherwise."""
import copy
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from .. import ops

SQRT1_2 = 1.0 / math.sqrt(2.0)


class _Conv(nn.Module):
    """nn.Conv2d(cin, cout, k, 1, pad[, bias]) parameter container + launcher (zero padding, stride 1: every conv of this model)"""

    def __init__(self, cin, cout, k, pad, bias=True):
        super().__init__()
        self.cin, self.cout, self.k, self.pad = cin, cout, k, pad
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self._packed = ops.PackedWeights()
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin * k * k)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act="none", up=False):
        geom = ops.ConvGeom(self.cin, self.cout, self.k, 1, self.pad, False, up)
        return ops.conv2d(x, self.weight, self.bias, self._packed, geom, act)


class _InAffine(nn.Module):
    """nn.InstanceNorm2d(C, affine=True) parameter container; forward = act(IN(x) * weight + bias)"""

    def __init__(self, c):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))

    def forward(self, x, act="leaky_relu"):
        n = x.shape[0]
        return ops.in_affine_act(x, (self.weight - 1.0).unsqueeze(0).expand(n, -1), self.bias.unsqueeze(0).expand(n, -1), act)


class ResBlk(nn.Module):
    """model.py:26-67: (shortcut + residual) / sqrt(2)"""

    def __init__(self, dim_in, dim_out, actv=None, normalize=False, downsample=False):
        super().__init__()
        self.normalize, self.downsample, self.learned_sc = normalize, downsample, dim_in != dim_out
        self.conv1 = _Conv(dim_in, dim_in, 3, 1)
        self.conv2 = _Conv(dim_in, dim_out, 3, 1)
        if normalize:
            self.norm1, self.norm2 = _InAffine(dim_in), _InAffine(dim_in)
        if self.learned_sc:
            self.conv1x1 = _Conv(dim_in, dim_out, 1, 0, bias=False)

    def forward(self, x):
        sc = self.conv1x1(x) if self.learned_sc else x
        if self.downsample:
            sc = ops.avgpool2(sc)
        h = self.norm1(x) if self.normalize else ops.leaky_relu(x)            # [IN] -> LeakyReLU (fused)
        if self.downsample or self.normalize:
            h = self.conv1(h)
            if self.downsample:
                h = ops.avgpool2(h)
            h = self.norm2(h) if self.normalize else ops.leaky_relu(h)
        else:
            h = self.conv1(h, "leaky_relu")                                  # nothing between conv1 and the activation: its epilogue
        return ops.scale(ops.add(sc, self.conv2(h)), SQRT1_2)


class AdaIN(nn.Module):
    """model.py:69-80; forward = act((1 + gamma) * IN(x) + beta), (gamma | beta) = fc(s)"""

    def __init__(self, style_dim, num_features):
        super().__init__()
        self.num_features = num_features
        self.fc = nn.Linear(style_dim, num_features * 2)

    def forward(self, x, s, act="leaky_relu"):
        h = nn.functional.linear(s.float(), self.fc.weight, self.fc.bias)       # (N, 2C): a library GEMM on a handful of rows
        return ops.in_affine_act(x, h[:, :self.num_features], h[:, self.num_features:], act)


class AdainResBlk(nn.Module):
    """model.py:83-123 with w_hpf = 0: (residual + shortcut) / sqrt(2); the nearest x2 upsample runs inside the conv that follows it"""

    def __init__(self, dim_in, dim_out, style_dim=64, w_hpf=0, actv=None, upsample=False):
        super().__init__()
        if w_hpf != 0:
            raise NotImplementedError("AdainResBlk: w_hpf > 0 (the FAN heat-map skip) is not built")
        self.upsample, self.learned_sc = upsample, dim_in != dim_out
        self.conv1 = _Conv(dim_in, dim_out, 3, 1)
        self.conv2 = _Conv(dim_out, dim_out, 3, 1)
        self.norm1, self.norm2 = AdaIN(style_dim, dim_in), AdaIN(style_dim, dim_out)
        if self.learned_sc:
            self.conv1x1 = _Conv(dim_in, dim_out, 1, 0, bias=False)

    def forward(self, x, s, labels=None):
        if self.learned_sc:
            sc = self.conv1x1(x, up=self.upsample)
        else:
            sc = ops.upsample2(x) if self.upsample else x
        h = self.conv1(self.norm1(x, s), up=self.upsample)
        h = self.conv2(self.norm2(h, s))
        return ops.scale(ops.add(h, sc), SQRT1_2)


class Generator(nn.Module):
    """model.py:321-382"""

    def __init__(self, img_size=256, style_dim=64, max_conv_dim=512, w_hpf=1, norm_type="adain", embed_nc=768, label_nc=3, hidden_nc=256,
                 compute_dtype="bf16"):
        super().__init__()
        if norm_type != "adain" or w_hpf != 0:
            raise NotImplementedError("stargan Generator: norm_type 'adain' with w_hpf 0 is built (SURVEY.md section 8f rank 4)")
        self.prec = ops.get_precision(compute_dtype)
        dim_in = 2 ** 14 // img_size
        self.img_size = img_size
        self.from_rgb = _Conv(3, dim_in, 3, 1)
        self.encode, self.decode = nn.ModuleList(), nn.ModuleList()
        self.to_rgb = nn.Sequential(_InAffine(dim_in), nn.Identity(), _Conv(dim_in, 3, 1, 0))          # IN affine, LeakyReLU (fused), conv 1x1
        for _ in range(int(np.log2(img_size)) - 4):
            dim_out = min(dim_in * 2, max_conv_dim)
            self.encode.append(ResBlk(dim_in, dim_out, normalize=True, downsample=True))
            self.decode.insert(0, AdainResBlk(dim_out, dim_in, style_dim, w_hpf=w_hpf, upsample=True))
            dim_in = dim_out
        for _ in range(2):
            self.encode.append(ResBlk(dim_in, dim_in, normalize=True))
            self.decode.insert(0, AdainResBlk(dim_in, dim_in, style_dim, w_hpf=w_hpf))
        self.decoder_len = len(self.decode)

    def forward(self, x, s, masks=None, labels=None, layer_split_index=None):
        if masks is not None or layer_split_index is not None:
            raise NotImplementedError("stargan Generator: masks / layer_split_index belong to the w_hpf > 0 and SEAN variants")
        h = self.from_rgb(ops.to_nhwc(x, self.prec))
        for block in self.encode:
            h = block(h)
        for block in self.decode:
            h = block(h, s, labels)
        return ops.to_nchw(self.to_rgb[2](self.to_rgb[0](h)), 3)


class MappingNetwork(nn.Module):
    """model.py:442-471: (N, latent) MLPs -- library GEMMs on a handful of rows (torch.nn.Linear), as in the reference"""

    def __init__(self, latent_dim=16, style_dim=64, num_domains=2):
        super().__init__()
        layers = [nn.Linear(latent_dim, 512), nn.ReLU()]
        for _ in range(3):
            layers += [nn.Linear(512, 512), nn.ReLU()]
        self.shared = nn.Sequential(*layers)
        self.unshared = nn.ModuleList()
        for _ in range(num_domains):
            self.unshared += [nn.Sequential(nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU(),
                                            nn.Linear(512, style_dim))]

    def forward(self, z, y):
        h = self.shared(z)
        out = torch.stack([layer(h) for layer in self.unshared], dim=1)
        return out[torch.arange(y.size(0), device=y.device), y]


class _Trunk(nn.Module):
    """conv3x3 -> ResBlk(downsample) x (log2(size) - 2) -> LeakyReLU -> conv4x4 valid -> LeakyReLU [-> conv1x1] with the reference's
    Sequential indices (model.py:474-492, 508-524)"""

    def __init__(self, img_size, max_conv_dim, head_out, compute_dtype):
        super().__init__()
        self.prec = ops.get_precision(compute_dtype)
        dim_in = 2 ** 14 // img_size
        blocks = [_Conv(3, dim_in, 3, 1)]
        for _ in range(int(np.log2(img_size)) - 2):
            dim_out = min(dim_in * 2, max_conv_dim)
            blocks += [ResBlk(dim_in, dim_out, downsample=True)]
            dim_in = dim_out
        self.nres = len(blocks) - 1
        blocks += [nn.Identity(), _Conv(dim_in, dim_in, 4, 0), nn.Identity()]
        if head_out:
            blocks += [_Conv(dim_in, head_out, 1, 0)]
        self.dim_out = dim_in
        self.blocks = nn.Sequential(*blocks)

    def features(self, x):
        b = self.blocks
        h = b[0](ops.to_nhwc(x, self.prec))
        for i in range(1, self.nres + 1):
            h = b[i](h)
        return b[self.nres + 2](ops.leaky_relu(h), "leaky_relu")                 # LeakyReLU -> conv4x4 -> LeakyReLU (the conv's epilogue)


class StyleEncoder(nn.Module):
    """model.py:474-505"""

    def __init__(self, img_size=256, style_dim=64, num_domains=2, max_conv_dim=512, compute_dtype="bf16"):
        super().__init__()
        trunk = _Trunk(img_size, max_conv_dim, 0, compute_dtype)
        self.shared, self._trunk = trunk.blocks, [trunk]
        self.unshared = nn.ModuleList([nn.Linear(trunk.dim_out, style_dim) for _ in range(num_domains)])

    def forward(self, x, y):
        trunk = self._trunk[0]
        h = ops.to_nchw(trunk.features(x), trunk.dim_out)
        h = h.view(h.size(0), -1)
        out = torch.stack([layer(h) for layer in self.unshared], dim=1)
        return out[torch.arange(y.size(0), device=y.device), y]


class Discriminator(nn.Module):
    """model.py:508-532"""

    def __init__(self, img_size=256, num_domains=2, max_conv_dim=512, compute_dtype="bf16"):
        super().__init__()
        trunk = _Trunk(img_size, max_conv_dim, num_domains, compute_dtype)
        self.main, self._trunk, self.num_domains = trunk.blocks, [trunk], num_domains

    def forward(self, x, y):
        trunk = self._trunk[0]
        out = ops.to_nchw(trunk.blocks[trunk.nres + 4](trunk.features(x)), self.num_domains)
        out = out.view(out.size(0), -1)
        return out[torch.arange(y.size(0), device=y.device), y]


def build_model(args):
    """model.py:694-731 (adain): -> (nets, nets_ema) as attribute namespaces; no DataParallel wrapper (one process per GPU)"""
    dt = getattr(args, "compute_dtype", "bf16")
    kw = dict(max_conv_dim=getattr(args, "max_conv_dim", 512))
    nets = SimpleNamespace(generator=Generator(args.img_size, args.style_dim, w_hpf=args.w_hpf, norm_type=args.norm_type, compute_dtype=dt, **kw),
                           mapping_network=MappingNetwork(args.latent_dim, args.style_dim, args.num_domains),
                           style_encoder=StyleEncoder(args.img_size, args.style_dim, args.num_domains, compute_dtype=dt, **kw),
                           discriminator=Discriminator(args.img_size, args.num_domains, compute_dtype=dt, **kw))
    nets_ema = SimpleNamespace(generator=copy.deepcopy(nets.generator), mapping_network=copy.deepcopy(nets.mapping_network),
                               style_encoder=copy.deepcopy(nets.style_encoder))
    return nets, nets_ema
