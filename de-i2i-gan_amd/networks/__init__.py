"""Checkpoint files of one network, named and filtered like the reference's (models/networks/__init__.py:4-23):
``<ckpt_dir>/<run name>/<epoch>_net_<label>.pth`` holds the plain ``state_dict``."""
import torch


def _checkpoint_path(opt, run_name, net_label, epoch):
    return opt.ckpt_dir / run_name / f"{epoch}_net_{net_label}.pth"


def save_network(net, net_label, epoch, opt):
    path = _checkpoint_path(opt, opt.name, net_label, epoch)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(net.state_dict(), path)


def _reference_key(key):
    # older checkpoints of the reference carry 'spade_' / 'sean_' in their keys; it strips both on load (:21)
    return key.replace("spade_", "").replace("sean_", "")


def load_network(net, net_label, epoch, opt):
    """Tensors only (``weights_only=True``); 'mlp_latent' entries are dropped and missing / unexpected keys tolerated
    (``strict=False``) exactly as the reference does (:21-22).  Returns the network on ``opt.device``."""
    stored = torch.load(_checkpoint_path(opt, opt.load_model_name, net_label, epoch), map_location="cpu", weights_only=True)
    state = {_reference_key(k): v for k, v in stored.items() if "mlp_latent" not in k}
    net.load_state_dict(state, strict=False)
    return net.to(opt.device, non_blocking=True)
