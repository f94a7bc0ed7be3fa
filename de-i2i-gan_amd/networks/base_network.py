"""BaseNetwork -- same surface as the reference (models/networks/base_network.py:5-65)."""
import torch.nn as nn
from torch.nn import init


class BaseNetwork(nn.Module):
    def __init__(self):
        super().__init__()

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @property
    def device(self):
        return next(self.parameters()).device

    def print_network(self):
        num_params = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million. "
              "To see the architecture, do print(network)." % (type(self).__name__, num_params / 1000000))

    def init_weights(self, init_type="normal", gain=0.02):
        """Class-name-substring dispatch exactly as the reference does it (base_network.py:27-56): BatchNorm2d
        weight ~ N(1, gain), bias 0; Conv*/Linear* weights by `init_type`, bias 0."""

        def init_func(m):
            classname = m.__class__.__name__
            if classname.find("BatchNorm2d") != -1:
                if hasattr(m, "weight") and m.weight is not None:
                    init.normal_(m.weight.data, 1.0, gain)
                if hasattr(m, "bias") and m.bias is not None:
                    init.constant_(m.bias.data, 0.0)
            elif hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
                if init_type == "normal":
                    init.normal_(m.weight.data, 0.0, gain)
                elif init_type == "xavier":
                    init.xavier_normal_(m.weight.data, gain=gain)
                elif init_type == "xavier_uniform":
                    init.xavier_uniform_(m.weight.data, gain=1.0)
                elif init_type == "kaiming":
                    init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    init.orthogonal_(m.weight.data, gain=gain)
                elif init_type == "none":
                    m.reset_parameters()
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
                if hasattr(m, "bias") and m.bias is not None:
                    init.constant_(m.bias.data, 0.0)

        self.apply(init_func)
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights(init_type, gain)

    def update_per_epoch(self, epoch):
        """update network per epoch"""
        pass
