"""Random patch masks of the MAE pre-training stage (utils/util.py:48-71).  1 = pixel kept, 0 = pixel masked.

Host-side and tiny ((N,1,H,W) floats): drawn with the global torch RNG on the CPU, exactly like the reference, so a run
seeded like the reference sees the reference's masks (the order of the draws matters: row shift, column shift, then the
Bernoulli field)."""
import torch
import torch.nn.functional as F


def generate_mask(image_size, patch_size: int, mask_ratio: float) -> torch.Tensor:
    """(N, C, H, W) -> (N, 1, H, W): every patch_size x patch_size patch is kept with probability 1 - mask_ratio."""
    n, _, h, w = image_size
    keep = torch.bernoulli(torch.full((n, 1, h // patch_size, w // patch_size), 1.0 - mask_ratio))
    # every keep-bit becomes a patch_size x patch_size block (nearest upsample: 0.3 ms on the host; repeat_interleave and
    # expand + reshape cost 12 - 50 ms here)
    return F.interpolate(keep, scale_factor=patch_size, mode="nearest")


def generate_shifted_mask(image_size, patch_size: int, mask_ratio: float) -> torch.Tensor:
    """The patch grid starts at a random offset in [0, patch_size)^2: a mask one patch larger is drawn and cropped."""
    n, c, h, w = image_size
    row0 = int(torch.randint(low=0, high=patch_size, size=(1,)))
    col0 = int(torch.randint(low=0, high=patch_size, size=(1,)))
    big = generate_mask((n, c, h + patch_size, w + patch_size), patch_size, mask_ratio)
    return big[:, :, row0:row0 + h, col0:col0 + w]


def draw_shifted_mask(image_size, patch_size: int, mask_ratio: float):
    """The random part of generate_shifted_mask only -- (row shift, column shift, per-patch keep bits), same RNG draws in
    the same order -- so that the expansion to pixels can run on the device (expand_shifted_mask): the host then touches
    (h/p + 1) x (w/p + 1) values per image instead of h x w."""
    n, _, h, w = image_size
    row0 = int(torch.randint(low=0, high=patch_size, size=(1,)))
    col0 = int(torch.randint(low=0, high=patch_size, size=(1,)))
    keep = torch.bernoulli(torch.full((n, 1, (h + patch_size) // patch_size, (w + patch_size) // patch_size), 1.0 - mask_ratio))
    return row0, col0, keep


def expand_shifted_mask(keep: torch.Tensor, row0: int, col0: int, patch_size: int, h: int, w: int) -> torch.Tensor:
    """keep bits (on any device) -> the (N,1,h,w) pixel mask generate_shifted_mask would return for the same draws."""
    big = F.interpolate(keep, scale_factor=patch_size, mode="nearest")
    return big[:, :, row0:row0 + h, col0:col0 + w]
