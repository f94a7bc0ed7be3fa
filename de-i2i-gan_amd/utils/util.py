"""Small numeric helpers of the reference's utils/util.py that the hot path's optional branches use."""
import torch.nn.functional as F


def calc_kl_with_logits(p, q, temperature=4.0):
    """utils/util.py:109-119: KL between the temperature-softened distributions of two logit tensors (softmax over dim 1),
    'batchmean', scaled by temperature^2 -- the distillation terms of SEAN (--style_distill)."""
    return F.kl_div(F.log_softmax(q / temperature, dim=1), F.log_softmax(p / temperature, dim=1), reduction="batchmean",
                    log_target=True) * temperature * temperature
