"""DiffAugment (Zhao et al. 2020, "Differentiable Augmentation for Data-Efficient GAN Training") as the reference applies it
to the images the discriminator sees (utils/diffaug.py:9-76; call sites defectgan_model.py:200-203,266-270): policies
'color' (brightness, saturation, contrast), 'translation' (random shift by up to 1/8 of the side, zero fill) and 'cutout'
(a random half-size square set to zero), all differentiable w.r.t. the image.

NCHW fp32 images, plain torch ops (3-channel images in front of the discriminator: plumbing, not a hot kernel).  The
per-sample random numbers are drawn with the global torch RNG ON THE HOST in the reference's order -- a run seeded like
the reference's CPU path sees the reference's augmentations -- and uploaded (a few values per sample)."""
import torch


def _rand(n, x):
    return torch.rand(n, 1, 1, 1).to(device=x.device, dtype=x.dtype)


def _randint(lo, hi, n, x):
    return torch.randint(lo, hi, size=[n, 1, 1]).to(x.device)


def brightness(x):
    return x + (_rand(x.size(0), x) - 0.5)


def saturation(x):
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (_rand(x.size(0), x) * 2) + mean


def contrast(x):
    mean = x.mean(dim=[1, 2, 3], keepdim=True)
    return (x - mean) * (_rand(x.size(0), x) + 0.5) + mean


def translation(x, ratio=0.125):
    """out[n, :, i, j] = x[n, :, i + ty_n, j + tx_n] (zero outside), one integer shift pair per sample."""
    n, _, h, w = x.shape
    max_y, max_x = int(h * ratio + 0.5), int(w * ratio + 0.5)
    ty = _randint(-max_y, max_y + 1, n, x)
    tx = _randint(-max_x, max_x + 1, n, x)
    rows = torch.arange(h, device=x.device).view(1, h, 1) + ty          # source row of every output row, per sample
    cols = torch.arange(w, device=x.device).view(1, 1, w) + tx
    inside = ((rows >= 0) & (rows < h) & (cols >= 0) & (cols < w)).unsqueeze(1).to(x.dtype)
    rows, cols = rows.clamp(0, h - 1), cols.clamp(0, w - 1)
    batch = torch.arange(n, device=x.device).view(n, 1, 1)
    gathered = x.permute(0, 2, 3, 1)[batch, rows, cols]                   # (n, h, w, c)
    return gathered.permute(0, 3, 1, 2) * inside


def cutout(x, ratio=0.5):
    """zero a (ratio*h) x (ratio*w) window centred at a random pixel (the part of it inside the image)"""
    n, _, h, w = x.shape
    ch, cw = int(h * ratio + 0.5), int(w * ratio + 0.5)
    cy = _randint(0, h + (1 - ch % 2), n, x)
    cx = _randint(0, w + (1 - cw % 2), n, x)
    rows = torch.arange(h, device=x.device).view(1, h, 1)
    cols = torch.arange(w, device=x.device).view(1, 1, w)
    top, left = cy - ch // 2, cx - cw // 2
    hole = (rows >= top) & (rows < top + ch) & (cols >= left) & (cols < left + cw)
    return x * (~hole).unsqueeze(1).to(x.dtype)


POLICIES = {"color": (brightness, saturation, contrast), "translation": (translation,), "cutout": (cutout,)}


def diff_augment(x, policy=""):
    if not policy:
        return x
    for name in policy.split(","):
        if name not in POLICIES:
            raise KeyError(f"DiffAugment policy [{name}] is not defined (color | translation | cutout)")
        for fn in POLICIES[name]:
            x = fn(x)
    return x.contiguous()
