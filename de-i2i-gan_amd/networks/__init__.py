"""Checkpoint I/O with the reference's file naming and key handling (models/networks/__init__.py:4-23)."""
import torch


def save_network(net, net_label, epoch, opt):
    save_dir = opt.ckpt_dir / opt.name
    save_dir.mkdir(parents=True, exist_ok=True)
    torch.save(net.state_dict(), save_dir / f"{epoch}_net_{net_label}.pth")


def load_network(net, net_label, epoch, opt):
    save_path = opt.ckpt_dir / opt.load_model_name / f"{epoch}_net_{net_label}.pth"
    weights = torch.load(save_path, map_location="cpu", weights_only=True)
    # reference: strip 'spade_' / 'sean_' prefixes, drop 'mlp_latent' keys, strict=False (networks/__init__.py:21-22)
    weights = {k.replace("spade_", "").replace("sean_", ""): v for k, v in weights.items() if "mlp_latent" not in k}
    net.load_state_dict(weights, strict=False)
    return net.to(opt.device, non_blocking=True)
