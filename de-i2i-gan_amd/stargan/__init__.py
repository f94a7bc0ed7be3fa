"""stargan-v2 G/D train iteration (reference: stargan-v2/core/model.py, core/solver.py) on the HIP ops.
``--norm_type adain``, ``--w_hpf 0`` (the configuration of every documented AFHQ command)."""
from .model import Discriminator, Generator, MappingNetwork, StyleEncoder, build_model  # noqa: F401
from .solver import Solver, adv_loss, compute_d_loss, compute_g_loss, moving_average, r1_reg  # noqa: F401
