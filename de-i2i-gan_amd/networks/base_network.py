"""BaseNetwork -- the surface the reference's networks share (models/networks/base_network.py:5-65): ``device``,
``print_network``, ``init_weights(init_type, gain)``, ``update_per_epoch(epoch)``."""
import torch.nn as nn
from torch.nn import init

# init_type -> filler of a Conv* / Linear* weight (the reference's choices, base_network.py:36-51)
_WEIGHT_FILL = {
    "normal": lambda w, gain: init.normal_(w, 0.0, gain),
    "xavier": lambda w, gain: init.xavier_normal_(w, gain=gain),
    "xavier_uniform": lambda w, gain: init.xavier_uniform_(w, gain=1.0),
    "kaiming": lambda w, gain: init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: init.orthogonal_(w, gain=gain),
}


def _clear_bias(module):
    bias = getattr(module, "bias", None)
    if bias is not None:
        init.constant_(bias.data, 0.0)


def _init_module(module, init_type, gain):
    """One module of the tree, chosen by class NAME like the reference does (so wrappers and subclasses whose names
    contain 'Conv' / 'Linear' / 'BatchNorm2d' are caught the same way): BatchNorm2d weight ~ N(1, gain), Conv* / Linear*
    weight by ``init_type``, every bias 0."""
    kind = type(module).__name__
    if "BatchNorm2d" in kind:
        if getattr(module, "weight", None) is not None:
            init.normal_(module.weight.data, 1.0, gain)
        _clear_bias(module)
        return
    if not (("Conv" in kind or "Linear" in kind) and hasattr(module, "weight")):
        return
    if init_type == "none":
        module.reset_parameters()
    else:
        fill = _WEIGHT_FILL.get(init_type)
        if fill is None:
            raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
        fill(module.weight.data, gain)
    _clear_bias(module)


class BaseNetwork(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @property
    def device(self):
        return next(self.parameters()).device

    def print_network(self):
        millions = sum(p.numel() for p in self.parameters()) / 1e6
        print(f"Network [{type(self).__name__}] was created. Total number of parameters: {millions:.1f} million. "
              "To see the architecture, do print(network).")

    def init_weights(self, init_type="normal", gain=0.02):
        # whole tree first, then child networks that bring their own init_weights (base_network.py:27-61) -- the order
        # fixes the order of the RNG draws
        self.apply(lambda m: _init_module(m, init_type, gain))
        for child in self.children():
            if hasattr(child, "init_weights"):
                child.init_weights(init_type, gain)

    def train(self, mode: bool = True):
        """nn.Module.train, without its per-module recursion through ``__setattr__``: the trainers flip the generator between eval and
        train mode twice per step (defectgan_model.py:83-90), ~130 modules each time -- 1.6 ms of host time per step, on steps that are
        host-bound (MAE stage).  The module list is made once; the tree is fixed after construction."""
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        mods = self.__dict__.get("_all_modules_list")
        if mods is None:
            mods = self.__dict__["_all_modules_list"] = list(self.modules())
        for m in mods:
            m.__dict__["training"] = mode
        return self

    def update_per_epoch(self, epoch):
        """hook the trainers call once per epoch; nothing to do for the SPADE generator / discriminator"""
