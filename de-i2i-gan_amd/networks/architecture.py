"""Building blocks with the reference's constructor signatures, module nesting and state_dict keys
(models/networks/architecture.py, models/networks/normalization.py), executing on the HIP kernels.

Blocks take and return NHWC activations of the compute dtype (see ``ops``); the NCHW fp32 boundary lives in the
Generator / Discriminator.  Parameters stay fp32 in the reference's OIHW / per-channel shapes.
"""
import logging
import math

import torch
from torch import nn

from .. import ops


def _pair_first(v):
    return v[0] if isinstance(v, (tuple, list)) else v


class Conv2d(nn.Module):
    """Parameter container + launcher for one nn.Conv2d of the reference (weight OIHW fp32, optional bias).
    The class name contains 'Conv' so BaseNetwork.init_weights dispatches on it like on nn.Conv2d."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, padding_mode="zeros", bias=True):
        super().__init__()
        k = _pair_first(kernel_size)
        if isinstance(kernel_size, (tuple, list)) and kernel_size[0] != kernel_size[1]:
            raise NotImplementedError("non-square kernels are not used by the reference hot path")
        s = _pair_first(stride)
        if padding == "same":
            if k % 2 == 0:
                raise ValueError("padding='same' needs an odd kernel")
            pad = (k - 1) // 2
        elif padding == "valid":
            pad = 0
        else:
            pad = int(_pair_first(padding))
        if padding_mode not in ("zeros", "reflect"):
            raise NotImplementedError(f"padding_mode [{padding_mode}] is not implemented")
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_channels, out_channels, k, s, pad
        self.padding_mode = padding_mode
        self._register_weight(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packed = ops.PackedWeights()
        self.reset_parameters()

    def _register_weight(self, w):
        self.weight = nn.Parameter(w)

    def effective_weight(self):
        """-> (weight tensor the conv uses, the parameters / buffers its packed copies depend on)"""
        return self.weight, (self.weight,)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self._parameters.get("weight_orig", self._parameters.get("weight")), a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_channels * self.kernel_size * self.kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def geom(self, up=False):
        g = self.__dict__.setdefault("_geoms", {}).get(up)
        if g is None:
            g = self._geoms[up] = ops.ConvGeom(self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
                                               self.padding_mode == "reflect" and self.padding > 0, up)
        return g

    def forward(self, x, act="none", up=False, stats=False):
        """``stats``: a BatchNorm / InstanceNorm reads the output next (its statistics come from the conv epilogue)."""
        w, sources = self.effective_weight()
        return ops.conv2d(x, w, self.bias, self._packed, self.geom(up), act, sources=sources, stats=stats)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, padding_mode={self.padding_mode}, bias={self.bias is not None}")


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1, affine, track_running_stats) parameter/buffer container."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)

    def forward(self, y, act="none", res=None, stats=False):
        return ops.batchnorm_act(y, self.weight, self.bias, self.running_mean, self.running_var, self.training, act, res,
                                 self.momentum, self.eps, self.num_batches_tracked, stats=stats)

    def fused_conv(self, y, act, conv, stats=False):
        """conv(act(self(y))) with this BatchNorm's apply + activation on ``conv``'s operand path when the kernels take the
        shape (ops.bn_act_conv: the normalised tensor is never written); the two-kernel formulation otherwise."""
        if type(conv) is Conv2d and conv.bias is None:     # (a spectral conv derives its weight per call: effective_weight() iterates u, v)
            w, geom = conv.weight, conv.geom(False)
            need_grad = torch.is_grad_enabled() and (y.requires_grad or w.requires_grad or self.weight.requires_grad)
            if ops.bn_act_conv_supported(y, self.weight, w, geom, need_grad):
                return ops.bn_act_conv(y, self.weight, self.bias, self.running_mean, self.running_var, self.training, act, w,
                                       conv._packed, geom, self.momentum, self.eps, self.num_batches_tracked, stats=stats)
        return conv(self(y, act), stats=stats)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


class Act(nn.Module):
    """Placeholder occupying the activation slot of the reference's nn.Sequential (keeps child indices and
    printing aligned); the activation itself is fused into the producing kernel."""

    def __init__(self, name):
        super().__init__()
        self.name = name

    def forward(self, x):
        return x

    def extra_repr(self):
        return str(self.name)


def get_act_layer(act_str):
    """architecture.py:12-26"""
    if act_str in ("leaky_relu", "relu", "sigmoid", "tanh"):
        return Act(act_str)
    if act_str is None:
        logging.info("create conv block without activation layer")
        return Act(None)
    raise NameError(f"activation layer named {act_str} not defined")


class SpectralConv2d(Conv2d):
    """torch.nn.utils.spectral_norm(nn.Conv2d(...)) as the reference applies it with ``--use_spectral``
    (architecture.py:68-72,109-112,238-239,338-341): parameter ``weight_orig``, buffers ``weight_u`` / ``weight_v`` (same
    state_dict keys), effective weight = weight_orig / sigma with sigma = u . (W v), W = weight_orig as a
    (Cout, Cin*k*k) matrix; in training mode every forward first runs one power iteration on (u, v) in place (no grad),
    in eval mode the stored vectors are used as they are.  Iteration, sigma and the scaled weight run as 5 (eval: 3)
    small HIP launches per forward and 2 per backward (``ops.spectral_weight``, csrc/spectral.hip; dW_orig =
    (G - sum(G . w_eff) u v^T) / sigma).  Like the reference's old-style hook, ``.weight`` is a derived tensor, so ``init_weights`` -- which writes
    into ``m.weight.data`` -- leaves ``weight_orig`` at nn.Conv2d's default initialisation."""
    EPS = 1e-12

    def _register_weight(self, w):
        self.weight_orig = nn.Parameter(w)
        cout = w.shape[0]
        self.register_buffer("weight_u", nn.functional.normalize(torch.randn(cout), dim=0, eps=self.EPS))
        self.register_buffer("weight_v", nn.functional.normalize(torch.randn(w[0].numel()), dim=0, eps=self.EPS))

    def _inspect_weight(self):
        """weight_orig / sigma with the stored (u, v), no iteration, plain torch: what ``.weight`` shows (init_weights
        writes into it, checkpoints and printing may look at it) -- never what a convolution uses."""
        wmat = self.weight_orig.flatten(1)
        return self.weight_orig / torch.dot(self.weight_u, torch.mv(wmat, self.weight_v))

    @property
    def weight(self):
        with torch.no_grad():
            return self._inspect_weight()

    def effective_weight(self):
        # the HIP kernels (csrc/spectral.hip); ops.spectral_weight refuses host tensors -- there is no CPU path
        sources = (self.weight_orig, self.weight_u, self.weight_v)
        frozen = not self.training and not (torch.is_grad_enabled() and self.weight_orig.requires_grad)
        if frozen:
            # eval mode, no gradient wanted: nothing iterates (u, v), so w / sigma only changes with the parameters --
            # keep it (and its packed copies) until a stamp moves (the D step's generator passes, D inside the G step)
            key = tuple(ops.PackedWeights._stamp(t) for t in sources)
            hit = self.__dict__.get("_frozen_weight")
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    hit = (key, ops.spectral_weight(self.weight_orig, self.weight_u, self.weight_v, False))
                self.__dict__["_frozen_weight"] = hit
            return hit[1], sources
        self.__dict__.pop("_frozen_weight", None)
        w = ops.spectral_weight(self.weight_orig, self.weight_u, self.weight_v, self.training)
        w._dei2i_per_call = True          # derived anew per forward: the packed-weight cache keys on this tensor too
        return w, sources


def make_conv(use_spectral, *args, **kw):
    return (SpectralConv2d if use_spectral else Conv2d)(*args, **kw)


def _is_batchnorm(norm_layer):
    return norm_layer is BatchNorm2d or norm_layer is nn.BatchNorm2d


def _is_instancenorm(norm_layer):
    return norm_layer is nn.InstanceNorm2d or norm_layer == "instance"


class NoiseInjection(nn.Module):
    """x + weight * noise with one N(0,1) value per pixel, shared by the channels (architecture.py:374-389, the
    'constant' weight type the blocks use).  Works on the NHWC activations: the noise is drawn as (N,1,H,W) like the
    reference (device RNG, or ``ops.noise_source`` when a test injects it); ``ops.noise_inject`` is one HIP launch forward
    and a deterministic two-launch reduction for the weight gradient backward (not fused into the conv epilogue yet)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1, 1, 1, 1))

    def forward(self, x, noise=None):
        n, h, w, _ = x.shape
        if noise is None:
            noise = ops.draw_noise((n, 1, h, w), x.device)
        return ops.noise_inject(x, self.weight, noise)


def _noise(add_noise):
    return NoiseInjection() if add_noise else Act(None)


def _reject(use_spectral=False, add_noise=False):
    if add_noise:
        raise NotImplementedError("add_noise on this block is not used by the reference (generator.py:230-241)")


class ConvBlock(nn.Module):
    """conv -> [BatchNorm2d | InstanceNorm2d(affine=False)] -> [act]  (architecture.py:79-118; the InstanceNorm form is the conv
    StyleExtractor's, extractor.py:50-80: no parameters, per-(n, c) statistics, ops.instance_norm_act)"""

    def __init__(self, f_in, f_out, kernel_size=(3, 3), stride=(1, 1), padding=0, padding_mode="zeros", bias=False,
                 norm_layer=None, act_layer=None, use_spectral=False):
        super().__init__()
        blocks = [make_conv(use_spectral, f_in, f_out, kernel_size, stride, padding, padding_mode, bias)]
        self._instance_norm = _is_instancenorm(norm_layer)
        self._has_norm = norm_layer is not None and not self._instance_norm
        if self._instance_norm:
            blocks.append(Act("instance_norm(affine=False)"))
        if self._has_norm:
            if not _is_batchnorm(norm_layer):
                raise NotImplementedError("ConvBlock: BatchNorm2d and InstanceNorm2d are the norms the reference uses here")
            blocks.append(BatchNorm2d(f_out))
        if act_layer not in (None, "leaky_relu", "relu"):
            raise NotImplementedError(f"ConvBlock activation [{act_layer}] is not fused")
        self._act = act_layer or "none"
        blocks.append(get_act_layer(act_layer))
        self.conv_block = nn.Sequential(*blocks)

    def forward(self, x, seg=None, res=None, out_stats=False):
        if self._instance_norm:
            assert res is None
            return ops.instance_norm_act(self.conv_block[0](x, stats=True), self._act)
        if self._has_norm:
            conv, bn = self.conv_block[0], self.conv_block[1]
            if res is None and self.folds_eval_bn():
                # eval-mode BatchNorm is a fixed per-channel affine: folded into the conv's weights, BN + activation run in the
                # conv's epilogue and the apply pass over its output is not launched (ops.fold_bn_weight)
                w_eff, b_eff = ops.fold_bn_weight(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
                return ops.conv2d(x, w_eff, b_eff, ops.PackedWeights(), conv.geom(False), self._act, sources=(w_eff,), stats=out_stats)
            y = conv(x, stats=bn.training)
            return bn(y, self._act, res, stats=out_stats)
        assert res is None
        return self.conv_block[0](x, self._act)

    def folds_eval_bn(self):
        """conv -> BatchNorm (running statistics) -> act as ONE conv launch: eval mode, no autograd graph, a plain conv without bias"""
        conv, bn = self.conv_block[0], self.conv_block[1]
        return (ops.fold_eval_bn and self._has_norm and not bn.training and not torch.is_grad_enabled() and type(conv) is Conv2d
                and conv.bias is None)


class DeConvBlock(nn.Module):
    """[upsample] -> conv -> act (architecture.py:29-76).  In the reference hot path only the two generator heads
    use it (up_scale=False, no norm, tanh / sigmoid); their convs are fused into one launch by the generator, so this
    module is a parameter container with the reference's key layout."""

    def __init__(self, f_in, f_out, kernel_size=(3, 3), stride=(1, 1), padding=0, padding_mode="zeros", bias=False,
                 up_scale=True, norm_layer=None, act_layer=None, use_spectral=False, add_noise=False):
        super().__init__()
        _reject(use_spectral, add_noise)
        if up_scale or norm_layer is not None:
            raise NotImplementedError("DeConvBlock with up_scale / norm is not on the reference hot path")
        self.de_conv_block = nn.Sequential(make_conv(use_spectral, f_in, f_out, kernel_size, stride, padding, padding_mode, bias),
                                           get_act_layer(act_layer))

    @property
    def conv(self):
        return self.de_conv_block[0]


class ResBlock(nn.Module):
    """x + BN(conv(LReLU(BN(conv(x)))))  (architecture.py:121-176); ``down_scale=True`` (the conv StyleExtractor's blocks,
    :157-168): ConvBlock -> AvgPool2d(2, 2) -> ConvBlock, plus the shortcut AvgPool2d(ConvBlock 1x1 (x)) -- same child indices
    (res_block.0 / .2, conv_s.0) as the reference's Sequentials."""

    def __init__(self, f_in, f_out, kernel_size=(3, 3), stride=(1, 1), padding=0, padding_mode="zeros", bias=False,
                 norm_layer=BatchNorm2d, act_layer="relu", use_spectral=False, down_scale=False):
        super().__init__()
        self.down_scale = bool(down_scale)
        blocks = [ConvBlock(f_in, f_in, kernel_size, stride, padding, padding_mode, bias, norm_layer, act_layer, use_spectral),
                  ConvBlock(f_in, f_out, kernel_size, stride, padding, padding_mode, bias, norm_layer, None, use_spectral)]
        if self.down_scale:
            blocks.insert(1, Act("avg_pool 2x2"))
            self.conv_s = nn.Sequential(ConvBlock(f_in, f_out, (1, 1), (1, 1), 0, padding_mode, False, norm_layer, None, use_spectral),
                                        Act("avg_pool 2x2"))
        self.res_block = nn.Sequential(*blocks)

    def forward(self, x, seg=None, out_stats=False):
        """``out_stats``: a norm layer (the decoder's first SPADE) reads the block's output next."""
        if self.down_scale:
            h = self.res_block[2](ops.avgpool2(self.res_block[0](x)))
            return ops.add(h, ops.avgpool2(self.conv_s[0](x)), stats=out_stats)
        first, second = self.res_block[0], self.res_block[1]
        if first._has_norm and second._has_norm and first.folds_eval_bn():
            # eval mode: the first BatchNorm + LeakyReLU in the first conv's epilogue; second conv -> BN + identity add (one kernel)
            y2 = second.conv_block[0](first(x), stats=False)
            return second.conv_block[1](y2, second._act, x, stats=out_stats)
        if first._has_norm and second._has_norm:
            # conv -> [BN + LeakyReLU on the second conv's operand path] -> conv -> BN + identity add (one kernel)
            y1 = first.conv_block[0](x, stats=True)
            y2 = first.conv_block[1].fused_conv(y1, first._act, second.conv_block[0], stats=True)
            return second.conv_block[1](y2, second._act, x, stats=out_stats)
        h = first(x)
        return second(h, res=x, out_stats=out_stats)           # identity add fused into the BatchNorm-apply kernel


class SPADE(nn.Module):
    """normalization.py:10-37.  ``forward`` returns relu(IN(x)*(1+gamma)+beta) -- every reference call site
    applies ReLU right after (architecture.py:244,346-347), so it is fused; a preceding nearest x2 upsample of x
    (architecture.py:203,241) is fused as well.

    gamma/beta come from two zero-padded 3x3 convs on the nearest-resized label map.  When the label map is 1x1
    (training, defectgan_model.py:385-392) gamma/beta take only 5x5 distinct values per (n,c) (position relative to
    the 2-pixel border), so the convs run on a 5x5 'border class' image and the modulate kernel indexes the table."""

    def __init__(self, label_nc, norm_nc, hidden_nc=128, kernel_size=(3, 3), padding="same", norm_layer=None):
        super().__init__()
        self.norm_nc, self.hidden_nc, self.label_nc = norm_nc, hidden_nc, label_nc
        self.param_free_norm = Act("instance_norm(affine=False)")
        self.mlp_shared = nn.Sequential(Conv2d(label_nc, hidden_nc, kernel_size, padding=padding), Act("relu"))
        self.mlp_gamma = Conv2d(hidden_nc, norm_nc, kernel_size, padding=padding)
        self.mlp_beta = Conv2d(hidden_nc, norm_nc, kernel_size, padding=padding)
        self._packed_gb = ops.PackedWeights()
        self._gb_cache = {}

    def _gamma_beta(self, segmap, prec, class_mode, h, w, seg=None, actv=None):
        """gamma|beta of the label map: (N,5,5,2C) border-class table (class mode) or (N,H,W,2C) dense.  ``seg``: the label map
        already resized and laid out (the same for every SPADE module of a generator: prime_spade makes it once); ``actv``: this
        module's ReLU(mlp_shared(seg)), computed by prime_spade's one conv over all modules' filters."""
        if actv is None:
            if seg is None:
                seg = ops.to_nhwc(segmap, prec, size=(5, 5) if class_mode else (h, w))
            actv = self.mlp_shared[0](seg, "relu")
        w_gb = torch.cat([self.mlp_gamma.weight, self.mlp_beta.weight], 0)
        b_gb = torch.cat([self.mlp_gamma.bias, self.mlp_beta.bias], 0)
        geom = ops.ConvGeom(self.hidden_nc, 2 * self.norm_nc, self.mlp_gamma.kernel_size, 1, self.mlp_gamma.padding, False, False)
        return ops.conv2d(actv, w_gb, b_gb, self._packed_gb, geom, "none", sources=(self.mlp_gamma.weight, self.mlp_beta.weight))

    def fused_conv(self, x, segmap, conv, up=False, skip=False, stats=False):
        """conv(self(x, segmap, up)) -- with skip: (that, x) -- the InstanceNorm apply + modulate + ReLU (+ upsample) on
        ``conv``'s operand path when the label map is constant and the kernels take the shape (ops.spade_conv); the
        two-kernel formulation otherwise."""
        prec = ops.precision_of(x)
        n, hs, ws, c = x.shape
        h, w = (2 * hs, 2 * ws) if up else (hs, ws)
        class_mode = segmap.shape[2] == 1 and segmap.shape[3] == 1 and h >= 4 and w >= 4
        if class_mode and isinstance(conv, Conv2d) and conv.bias is None:
            geom = conv.geom(up)
            params_grad = any(p.requires_grad for p in conv.parameters()) or self.mlp_gamma.weight.requires_grad
            need_grad = torch.is_grad_enabled() and (x.requires_grad or params_grad)
            mode = ops.spade_conv_supported(x, None, geom, need_grad)
            if mode is not None:
                self._ran_class_mode = True
                gb = self._class_table(segmap, prec, h, w)
                wt, sources = conv.effective_weight()         # (a spectral conv iterates u, v here: once per forward, like conv(...))
                return ops.spade_conv(x, gb, wt, conv._packed, geom, skip=skip, stats=stats, mode=mode, sources=sources)
        if skip:
            z, xs = self(x, segmap, up=up, skip=True)
            return conv(z, stats=stats), xs
        return conv(self(x, segmap, up=up), stats=stats)

    def _class_table(self, segmap, prec, h, w):
        # The class table depends only on (label map, this module's weights), not on x: the loss graphs call G several
        # times with the SAME label tensors (defectgan_model.py:185-190), so the table -- with its autograd history,
        # autograd sums the gradients of all its uses -- is computed once per (label tensor, weight state, grad mode).
        key = self._table_key(segmap, prec)
        hit = self._gb_cache.get(key)
        if hit is None and not torch.is_grad_enabled():      # a table that carries history serves a no-grad pass as well
            hit = self._gb_cache.get(self._table_key(segmap, prec, grad=True))
        if hit is None:
            gb = self._gamma_beta(segmap, prec, True, h, w)
            if len(self._gb_cache) >= 2:
                self._gb_cache.clear()
            self._gb_cache[key] = (segmap, gb)          # keep the label tensor alive so its id stays unique
            return gb
        return hit[1]

    def forward(self, x, segmap, up=False, skip=False):
        prec = ops.precision_of(x)
        n, hs, ws, c = x.shape
        h, w = (2 * hs, 2 * ws) if up else (hs, ws)
        class_mode = segmap.shape[2] == 1 and segmap.shape[3] == 1 and h >= 4 and w >= 4
        self._ran_class_mode = class_mode                 # prime() only serves modules that run (norm_s never does)
        if not class_mode:
            return ops.spade_relu(x, self._gamma_beta(segmap, prec, False, h, w), up, 0, skip=skip)
        return ops.spade_relu(x, self._class_table(segmap, prec, h, w), up, 1, skip=skip)

    def _table_key(self, segmap, prec, grad=None):
        params = (self.mlp_shared[0].weight, self.mlp_shared[0].bias, self.mlp_gamma.weight, self.mlp_gamma.bias,
                  self.mlp_beta.weight, self.mlp_beta.bias)
        return (id(segmap), segmap._version, prec.code, torch.is_grad_enabled() if grad is None else grad,
                tuple(ops.PackedWeights._stamp(p) + (p.requires_grad,) for p in params))

    def wants_prime(self, segmaps, prec):
        """Will prime() compute tables for these label tensors?  (a module that runs in class mode and does not hold them yet)"""
        return (getattr(self, "_ran_class_mode", True) and len(segmaps) in (1, 2)
                and all(s_.dim() == 4 and s_.shape[2] == 1 and s_.shape[3] == 1 for s_ in segmaps)
                and not all(self._table_key(sgm, prec) in self._gb_cache for sgm in segmaps))

    def prime(self, segmaps, prec, both=None, seg=None, actv=None, gb=None):
        """Compute the class tables of several (N,C,1,1) label tensors in ONE pass over their concatenation and memoize
        each tensor's slice: the same function as one pass per tensor (the table convs act per sample), with half the
        launches of the 5x5 table path (forward and backward) when a loss graph uses two label sets."""
        # (before the generator's first forward nobody knows which modules run -- norm_s of a block without a learned shortcut never
        #  does --: all are served then, so that the first step takes the same kernels as every later one; the generator marks the
        #  modules its first forward did not reach)
        if not getattr(self, "_ran_class_mode", True):
            return
        if len(segmaps) not in (1, 2) or any(s.dim() != 4 or s.shape[2] != 1 or s.shape[3] != 1 for s in segmaps):
            return
        if all(self._table_key(sgm, prec) in self._gb_cache for sgm in segmaps):
            return                                       # both tables are there already (same tensors, same parameter state)
        if gb is None:        # (else: prime_spade computed this module's table in the batched launch, ops.label_gamma_beta)
            gb = self._gamma_beta(both if both is not None else torch.cat(list(segmaps), 0), prec, True, 0, 0, seg=seg, actv=actv)
        self._gb_cache.clear()
        for sgm, part in zip(segmaps, ops.split_rows(gb, [sgm.shape[0] for sgm in segmaps])):
            self._gb_cache[self._table_key(sgm, prec)] = (sgm, part)


class AdaIN(nn.Module):
    """normalization.py:40-73 (denorm_type 'linear'): ``IN(x) * (1 + gamma) + beta`` with per-(n, c) gamma / beta from two
    Linear layers on the style feature (N, hidden_nc) of the StyleExtractor.  That is SPADE with a gamma / beta that does
    not vary over space, so it runs on the SPADE kernels: the (N, 5, 5, 2C) border-class table holds the same (gamma | beta)
    in all 25 classes (autograd sums the table's gradient back over them); ReLU and a preceding upsample are fused like in
    ``SPADE``.  The two Linear layers -- (N x hidden_nc) by (hidden_nc x C) -- go through torch's library GEMM."""

    def __init__(self, norm_nc, hidden_nc=None, norm_layer=None, denorm_type="linear"):
        super().__init__()
        if denorm_type != "linear":
            raise NotImplementedError("AdaIN: only denorm_type 'linear' is used by the reference's decoder blocks")
        assert hidden_nc is not None, "hidden_nc should be set when denorm_type is linear."
        self.norm_nc, self.hidden_nc, self.denorm_type = norm_nc, hidden_nc, denorm_type
        self.param_free_norm = Act("instance_norm(affine=False)")
        self.mlp_gamma = nn.Linear(hidden_nc, norm_nc)
        self.mlp_beta = nn.Linear(hidden_nc, norm_nc)

    def _class_table(self, style_feat, prec, c_stride):
        n = style_feat.size(0)
        assert style_feat.size(1) == self.hidden_nc, "The channel of style feature is not equal to hidden_nc."
        feat = style_feat.view(n, self.hidden_nc).float()
        gb = torch.cat([nn.functional.pad(self.mlp_gamma(feat), (0, c_stride - self.norm_nc)),
                        nn.functional.pad(self.mlp_beta(feat), (0, c_stride - self.norm_nc))], dim=1)     # (N, 2 * c_stride)
        return gb.to(prec.dtype).view(n, 1, 1, 2 * c_stride).expand(n, 5, 5, 2 * c_stride).contiguous()

    def fused_conv(self, x, style_feat, conv, up=False, skip=False, stats=False):
        """conv(self(x, style_feat, up)) -- with skip: (that, x) -- see SPADE.fused_conv"""
        prec = ops.precision_of(x)
        gb = self._class_table(style_feat, prec, x.shape[-1])
        if isinstance(conv, Conv2d) and conv.bias is None:
            geom = conv.geom(up)
            need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in conv.parameters()) or gb.requires_grad)
            mode = ops.spade_conv_supported(x, None, geom, need_grad)
            if mode is not None:
                wt, sources = conv.effective_weight()
                return ops.spade_conv(x, gb, wt, conv._packed, geom, skip=skip, stats=stats, mode=mode, sources=sources)
        if skip:
            z, xs = ops.spade_relu(x, gb, up, 1, skip=True)
            return conv(z, stats=stats), xs
        return conv(ops.spade_relu(x, gb, up, 1), stats=stats)

    def forward(self, x, style_feat, up=False, skip=False):
        return ops.spade_relu(x, self._class_table(style_feat, ops.precision_of(x), x.shape[-1]), up, 1, skip=skip)


def label_to_str(label):
    """utils/util.py:178-180: the indices of the set bits, joined by '-' ('' for the all-zero label)"""
    return "-".join(str(i) for i, v in enumerate(label) if v == 1)


class SEAN(nn.Module):
    """normalization.py:76-202: ``IN(x) * (1 + gamma) + beta`` with per-(n, c) gamma / beta from Linear layers on a mixed
    style code: ``latent = ReLU(Linear(labels))``; without style embeddings (``feat is None``: --sean_alpha 0) the code IS the
    latent; with embeddings ``feat`` (N, num_embeds, embed_nc) it is ``mean_e(ReLU(Linear(feat)) + latent)``, rows that come
    out all-zero replaced by the latent (:177-179).  Like AdaIN the modulation does not vary over space, so it runs on the
    SPADE kernels with a constant (N, 5, 5, 2C) table; the Linear layers ((N * num_embeds) x embed_nc x hidden_nc and smaller)
    go through torch's library GEMM.

    ``--use_running_stats`` (:104-125,162-176): while ``track_running_stats`` is on (the four generator passes of the G loss) every
    sample's mixed code is appended to the list of its label combination; ``update_stats`` (once per epoch) writes the lists'
    per-feature statistics into the ``mean_<labels>`` / ``std_<labels>`` buffers -- the reference stores the STANDARD DEVIATION in
    ``mean_*`` and the MEAN in ``std_*`` (:124 assigns ``mean[:], std[:] = new_std, new_mean``), and so does this, so that
    checkpoints interchange -- and ``inference_running_stats`` builds the code from a noise vector and those buffers instead of
    embeddings.  ``--style_distill`` (:127-143,181-190): while enabled (the G loss) every forward with embeddings takes
    KL(latent || mix) * 0.1 + KL(enc || mix) at temperature 4 against the detached mixed code and runs ITS OWN backward on the
    spot (gradients land in mlp_latent / mlp_shared before the main backward adds to them); the two terms are kept for logging.
    ``alpha`` is stored by ``set_alpha`` and, as in the reference, read by nothing."""

    def __init__(self, embed_nc, norm_nc, label_nc, hidden_nc=128, latent_dim=16, norm_layer=None, style_distill=False):
        super().__init__()
        self.style_distill = bool(style_distill)
        if self.style_distill:
            self._distill_loss = None
        self.latent_dim, self.label_nc, self.hidden_nc, self.norm_nc = latent_dim, label_nc, hidden_nc, norm_nc
        self.noise_dim = latent_dim - label_nc
        self.alpha = 1.0
        self.param_free_norm = Act("instance_norm(affine=False)")
        self.mlp_shared = nn.Sequential(nn.Linear(embed_nc, hidden_nc), nn.ReLU(inplace=True))
        self.mlp_gamma = nn.Linear(hidden_nc, norm_nc)
        self.mlp_beta = nn.Linear(hidden_nc, norm_nc)
        self.mlp_latent = nn.Sequential(nn.Linear(label_nc, hidden_nc), nn.ReLU(inplace=True))
        self.embeds = {}
        for bits in range(1 << label_nc):             # torch.cartesian_prod order: the first label is the slowest digit
            label = [(bits >> (label_nc - 1 - i)) & 1 for i in range(label_nc)]
            self.register_buffer("mean_" + label_to_str(label), torch.zeros(hidden_nc))
            self.register_buffer("std_" + label_to_str(label), torch.zeros(hidden_nc))
            self.embeds[tuple(label)] = []
        self.inference_running_stats = False
        self.track_running_stats = False
        self.num_embeds_tracked = 10000

    def set_alpha(self, alpha):
        self.alpha = alpha

    def update_stats(self):
        """normalization.py:111-125 (per epoch): statistics of the tracked codes of every label combination -> its two buffers
        (swapped, see the class docstring); the lists keep their newest ``num_embeds_tracked`` entries."""
        eps = 1e-5
        for label, tracked in self.embeds.items():
            if tracked:
                first = getattr(self, "mean_" + label_to_str(label))
                second = getattr(self, "std_" + label_to_str(label))
                feat = torch.stack(tracked, dim=0)
                new_std = (feat.var(dim=0) + eps).sqrt()
                new_mean = feat.mean(dim=0)
                first[:], second[:] = new_std, new_mean
                self.embeds[label] = tracked[-self.num_embeds_tracked:]

    @property
    def distill_loss(self):
        return self._distill_loss

    @distill_loss.setter
    def distill_loss(self, enable):
        """True: start collecting {'latent': [...], 'embed': [...]}; False: distillation off (normalization.py:131-143)"""
        self._distill_loss = {"latent": [], "embed": []} if enable else None

    def _class_table(self, cond, prec, c_stride):
        labels, feat = cond
        n = labels.size(0)
        labels = labels.reshape(n, -1).float()
        latent_code = self.mlp_latent(labels)
        if feat is None:
            mix_feat = latent_code
        elif self.inference_running_stats:            # feat: one noise vector (hidden_nc,) per sample
            rows = []
            for label, noise in zip(labels, feat):
                key = label_to_str(tuple(int(v.item()) for v in label))
                rows.append(noise.float() * getattr(self, "std_" + key) * 1.5 + getattr(self, "mean_" + key))
            mix_feat = torch.stack(rows, dim=0)
        else:
            enc_feat = self.mlp_shared(feat.float())
            mix_feat = enc_feat + latent_code.view(n, 1, -1)
            if mix_feat.dim() == 3:
                mix_feat = mix_feat.mean(dim=1)
            if self.track_running_stats:
                for label, row in zip(labels, mix_feat.clone().detach()):
                    key = tuple(int(v.item()) for v in label)
                    if row.dim() == 2:
                        self.embeds[key].extend(row)
                    else:
                        self.embeds[key].append(row)
            mask = (mix_feat == 0).all(dim=1).view(-1, 1)
            mix_feat = mix_feat * ~mask + latent_code * mask
            if self.style_distill and self._distill_loss is not None:
                from ..utils.distill import calc_kl_with_logits
                target = mix_feat.detach()
                latent_term = calc_kl_with_logits(latent_code, target, 4)
                embed_term = calc_kl_with_logits(enc_feat, target, 4)
                (latent_term * 0.1 + embed_term).backward(retain_graph=True)
                self._distill_loss["latent"].append(latent_term)
                self._distill_loss["embed"].append(embed_term)
        gb = torch.cat([nn.functional.pad(self.mlp_gamma(mix_feat), (0, c_stride - self.norm_nc)),
                        nn.functional.pad(self.mlp_beta(mix_feat), (0, c_stride - self.norm_nc))], dim=1)     # (N, 2 * c_stride)
        return gb.to(prec.dtype).view(n, 1, 1, 2 * c_stride).expand(n, 5, 5, 2 * c_stride).contiguous()

    fused_conv = AdaIN.fused_conv            # (cond = (labels, feat) instead of the style feature: only _class_table reads it)
    forward = AdaIN.forward


def _style_norm(style_norm_block_type, label_nc, f, hidden_nc, embed_nc=None, style_distill=False):
    if style_norm_block_type == "spade":
        return SPADE(label_nc, f, hidden_nc=hidden_nc)
    if style_norm_block_type == "adain":
        return AdaIN(f, hidden_nc=hidden_nc)
    if style_norm_block_type == "sean":
        assert embed_nc is not None, "embed_nc must be specified for SEAN"
        return SEAN(embed_nc, f, label_nc, hidden_nc=hidden_nc, style_distill=style_distill)
    raise NotImplementedError(f"style_norm_block_type [{style_norm_block_type}] is not implemented")


def _norm_cond(style_norm_block_type, labels, style_feat):
    """norm_forward of the decoder blocks (architecture.py:246-254,363-371): what the block's norm layer is given"""
    if style_norm_block_type == "adain":
        return style_feat
    if style_norm_block_type == "sean":
        return (labels, style_feat)
    return labels


class NormConvBlock(nn.Module):
    """up -> SPADE -> ReLU -> conv  (architecture.py:179-254, order at :241-245)"""

    def __init__(self, style_norm_block_type, hidden_nc, label_nc, f_in, f_out, style_distill=False, embed_nc=None,
                 kernel_size=(3, 3), stride=(1, 1), padding=0, padding_mode="zeros", bias=False, up_scale=False,
                 norm_layer=None, act_layer="relu", use_spectral=False, add_noise=False):
        super().__init__()
        if act_layer != "relu":
            raise NotImplementedError("NormConvBlock: the reference uses ReLU here")
        self.up_scale = up_scale
        self.up = Act("nearest x2" if up_scale else None)
        self.noise = _noise(add_noise)
        self.style_norm_block_type = style_norm_block_type
        self.norm = _style_norm(style_norm_block_type, label_nc, f_in, hidden_nc, embed_nc, style_distill)
        self.conv = make_conv(use_spectral, f_in, f_out, kernel_size, stride, padding, padding_mode, bias)
        self.act = get_act_layer(act_layer)

    def forward(self, x, labels, style_feat=None, out_stats=False):
        """``out_stats``: another norm layer reads the output next (the following NormConvBlock's SPADE)."""
        cond = _norm_cond(self.style_norm_block_type, labels, style_feat)
        return self.noise(self.norm.fused_conv(x, cond, self.conv, up=self.up_scale, stats=out_stats))


class NormResBlock(nn.Module):
    """x + conv_1(ReLU(SPADE_1(conv_0(ReLU(SPADE_0(x))))))  (architecture.py:260-371).  norm_s / conv_s exist as
    parameters (state_dict compatibility) but are never executed when up_scale=False (:352-357)."""

    def __init__(self, style_norm_block_type, hidden_nc, label_nc, f_in, f_out, style_distill=False, embed_nc=None,
                 kernel_size=(3, 3), stride=(1, 1), padding=0, padding_mode="zeros", bias=False, up_scale=False,
                 norm_layer=None, act_layer="relu", use_spectral=False, add_noise=False):
        super().__init__()
        if up_scale:
            raise NotImplementedError("NormResBlock(up_scale=True) is not on the reference hot path")
        if act_layer != "relu":
            raise NotImplementedError("NormResBlock: the reference uses ReLU here")
        self.up_scale = False
        self.up = Act("nearest x2")
        self.noise_0, self.noise_1 = _noise(add_noise), _noise(add_noise)
        f_mid = min(f_in, f_out)
        self.style_norm_block_type = style_norm_block_type
        self.norm_0 = _style_norm(style_norm_block_type, label_nc, f_in, hidden_nc, embed_nc, style_distill)
        self.norm_1 = _style_norm(style_norm_block_type, label_nc, f_mid, hidden_nc, embed_nc, style_distill)
        self.norm_s = _style_norm(style_norm_block_type, label_nc, f_in, hidden_nc, embed_nc, style_distill)
        self.act = get_act_layer(act_layer)
        self.conv_0 = make_conv(use_spectral, f_in, f_mid, kernel_size, stride, padding, padding_mode, bias)
        self.conv_1 = make_conv(use_spectral, f_mid, f_out, kernel_size, stride, padding, padding_mode, bias)
        self.conv_s = make_conv(use_spectral, f_in, f_out, kernel_size, stride, padding, padding_mode, bias)

    def forward(self, x, labels, style_feat=None, out_stats=False):
        # norm_0 hands x through (xs) so that the identity branch's gradient is added inside its backward kernel
        cond = _norm_cond(self.style_norm_block_type, labels, style_feat)
        if x.is_contiguous():
            h, xs = self.norm_0.fused_conv(x, cond, self.conv_0, skip=True, stats=True)
        else:
            h, xs = self.norm_0.fused_conv(x, cond, self.conv_0, stats=True), x
        h = self.noise_0(h)
        h = self.noise_1(self.norm_1.fused_conv(h, cond, self.conv_1))
        return ops.add(h, xs, stats=out_stats)


class MaskToken(nn.Module):
    """Learnable fill of the masked pixels of the MAE stage (architecture.py:392-418): ``imgs * masks + token * (1 - masks)``
    on the NCHW fp32 images (3 channels -- host-side plumbing in front of the generator, plain torch ops).
    Kinds: zero | mean (per-image channel mean of the kept pixels / mask_ratio, as the reference computes it) | scalar |
    vector | position | full; the last four own one parameter named ``mask_token``."""

    SHAPES = {"scalar": lambda c, s: (1, 1, 1, 1), "vector": lambda c, s: (1, c, 1, 1),
              "position": lambda c, s: (1, 1, s, s), "full": lambda c, s: (1, c, s, s)}

    def __init__(self, opt):
        super().__init__()
        self.mask_token_type = opt.mask_token_type
        self.mask_ratio = opt.mask_ratio
        if self.mask_token_type in ("zero", "mean"):
            self.mask_token = 0
        elif self.mask_token_type in self.SHAPES:
            self.mask_token = nn.Parameter(torch.zeros(self.SHAPES[self.mask_token_type](opt.input_nc, opt.image_size)))
        else:
            raise ValueError("Unknown mask token type: {}".format(self.mask_token_type))

    def forward(self, imgs, masks):
        kept = imgs * masks
        token = self.mask_token
        if self.mask_token_type == "mean":
            mean = kept.mean(dim=(2, 3)) / self.mask_ratio
            token = self.mask_token = mean.reshape(mean.size(0), mean.size(1), 1, 1)
        return kept + token * (1 - masks)
