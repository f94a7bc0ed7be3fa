"""The distillation term of SEAN's --style_distill (reference: utils/util.py:109-119, called from normalization.py:181-190)."""
import torch.nn.functional as F


def calc_kl_with_logits(p, q, temperature=4.0):
    """KL(softmax(p / T) || softmax(q / T)) summed over dim 1 and the remaining dims, divided by the size of dim 0 ('batchmean'),
    times T^2: both distributions are softened by the temperature before they are compared, and the T^2 factor keeps the
    gradient's scale independent of T.  Arguments are logits; broadcasting between them follows torch's rules (the SEAN caller
    compares (N, num_embeds, hidden) encoder features with an (N, hidden) target)."""
    log_target = F.log_softmax(p / temperature, dim=1)
    log_input = F.log_softmax(q / temperature, dim=1)
    kl = F.kl_div(log_input, log_target, reduction="batchmean", log_target=True)
    return kl * (temperature * temperature)
