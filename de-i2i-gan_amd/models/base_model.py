"""BaseModel: what the reference's trainers rely on from a model object (models/base_model.py:7-86) -- the ``networks``
mapping discovered from ``net*`` attributes, weight initialisation, checkpoint save / load per network, the mean-reduced
loss helper and the per-epoch hook."""
import torch.nn

from .. import ops
from ..networks import load_network, save_network

_LOSSES = {"bce": ops.bce_logits, "bce_logits": ops.bce_logits, "l1": ops.l1}      # the two the hot path uses, as HIP ops


class BaseModel:
    network_prefix = "net"

    def __init__(self, opt):
        self.opt = opt

    @property
    def networks(self):
        """{'G': self.netG, 'D': self.netD, ...}: every nn.Module attribute whose name starts with 'net', keyed by the rest
        of the name (trainers index optimizers by these keys)."""
        cut = len(self.network_prefix)
        return {attr[cut:] if attr.startswith(self.network_prefix) else attr: module
                for attr, module in vars(self).items()
                if attr.startswith(self.network_prefix) and isinstance(module, torch.nn.Module)}

    def _each_network(self):
        return self.networks.items()

    def init_weights(self):
        kind, gain = self.opt.init_type, self.opt.init_variance
        print(f"initialize model's parameters using {kind} with variance={gain}")
        for label, net in self._each_network():
            if label.endswith("_"):               # a trailing underscore marks a network that keeps its own weights
                continue
            net.init_weights(kind, gain)

    def save(self, epoch):
        for label, net in self._each_network():
            save_network(net, label, epoch, self.opt)

    def load(self, epoch):
        print(f"load model's weights from epoch {epoch}")
        for label, net in self._each_network():
            load_network(net, label, epoch, self.opt)

    def load_network(self, network_name, epoch):
        print(f"load net_{network_name}'s weights from epoch {epoch}")
        load_network(self.networks[network_name], network_name, epoch, self.opt)

    def __repr__(self):
        blocks = []
        for label, net in self._each_network():
            rule = "=" * 50 + f"{self.network_prefix + label:^8}" + "=" * 50 + "\n"
            blocks.append(rule + repr(net) + "\n" + rule)
        return "".join(blocks)

    def _cal_loss(self, logits, targets, loss_type):
        """base_model.py:68-80 -- mean-reduced; `targets` may be a python constant (all-ones / all-zeros labels)."""
        fn = _LOSSES.get(loss_type)
        if fn is None:
            raise ValueError(f"loss_type: {loss_type} is not on the MI355X hot path (bce | l1)")
        return fn(logits, targets)

    def update_per_epoch(self, epoch):
        for _, net in self._each_network():
            hook = getattr(net, "update_per_epoch", None)
            if hook is not None:
                hook(epoch)
