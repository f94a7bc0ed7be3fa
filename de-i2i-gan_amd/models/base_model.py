"""BaseModel (models/base_model.py:7-86): network discovery by the 'net' attribute prefix, init/save/load, losses."""
import torch.nn

from .. import ops
from ..networks import load_network, save_network


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.network_prefix = "net"

    @property
    def networks(self):
        return {name.replace(self.network_prefix, ""): value for name, value in self.__dict__.items()
                if name.startswith(self.network_prefix) and isinstance(value, torch.nn.Module)}

    def init_weights(self):
        print(f"initialize model's parameters using {self.opt.init_type} with variance={self.opt.init_variance}")
        for network_name, network in self.networks.items():
            if not network_name.endswith("_"):
                network.init_weights(self.opt.init_type, self.opt.init_variance)

    def save(self, epoch):
        for network_name, network in self.networks.items():
            save_network(network, network_name, epoch, self.opt)

    def load(self, epoch):
        print(f"load model's weights from epoch {epoch}")
        for network_name, network in self.networks.items():
            load_network(network, network_name, epoch, self.opt)

    def load_network(self, network_name, epoch):
        print(f"load net_{network_name}'s weights from epoch {epoch}")
        load_network(self.networks[network_name], network_name, epoch, self.opt)

    def __repr__(self):
        model_repr = ""
        for network_name, network in self.networks.items():
            split_line = "=" * 50 + f"{self.network_prefix + network_name:^8}" + "=" * 50 + "\n"
            model_repr += split_line + repr(network) + "\n" + split_line
        return model_repr

    def _cal_loss(self, logits, targets, loss_type):
        """base_model.py:68-80 -- mean-reduced; `targets` may be a python constant (all-ones / all-zeros labels)."""
        if loss_type in ("bce", "bce_logits"):
            return ops.bce_logits(logits, targets)
        if loss_type == "l1":
            return ops.l1(logits, targets)
        raise ValueError(f"loss_type: {loss_type} is not on the MI355X hot path (bce | l1)")

    def update_per_epoch(self, epoch):
        for network in self.networks.values():
            if hasattr(network, "update_per_epoch"):
                network.update_per_epoch(epoch)
