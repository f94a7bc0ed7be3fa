"""CPU: the drop-in boundary -- module classes construct from an `opt` with the reference's attribute names and
expose the reference's state_dict keys and shapes (manifest captured from the reference in the goldens); the C-ABI
library loads and exports every symbol include/dei2i_hip.h declares (no kernels are launched here)."""
import re
from pathlib import Path

import pytest
import torch

from helpers import load_golden, make_opt
from oracle import defectgan_oracle as O

REPO = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("name", ["t0_img32_b2", "t1_img64_b4"])
def test_state_dict_keys_match_reference(name):
    from de_i2i_gan_amd.networks.discriminator import DefectGanDiscriminator
    from de_i2i_gan_amd.networks.generator import DefectGanGenerator
    meta, arr, c, cfg = load_golden(name)
    opt = make_opt(c, "cpu")
    G, D = DefectGanGenerator(opt), DefectGanDiscriminator(opt)
    assert list(G.state_dict().keys()) == meta["G_keys"]
    assert list(D.state_dict().keys()) == meta["D_keys"]
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == O.generator_state_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == O.discriminator_state_shapes(cfg)
    # default-size manifest: 133 generator entries / 8 discriminator entries
    big = make_opt(dict(ngf=64, ndf=64, hidden_nc=128, num_layers=5, image_size=256, batch=1), "cpu")
    assert len(DefectGanDiscriminator(big).state_dict()) == 8


def test_init_weights_follows_reference_rules():
    from de_i2i_gan_amd.networks.generator import DefectGanGenerator
    meta, arr, c, cfg = load_golden("t0_img32_b2")
    G = DefectGanGenerator(make_opt(c, "cpu"))
    torch.manual_seed(0)
    G.init_weights("normal", 0.02)
    sd = G.state_dict()
    assert abs(sd["stem.conv_block.0.weight"].std().item() - 0.02) < 0.004
    assert abs(sd["stem.conv_block.1.weight"].mean().item() - 1.0) < 0.05          # BatchNorm weight ~ N(1, 0.02)
    assert sd["dec_blk.0.norm.mlp_gamma.bias"].abs().max().item() == 0             # biases zeroed
    assert sd["stem.conv_block.1.running_var"].eq(1).all()


def test_library_exports_every_declared_symbol():
    from de_i2i_gan_amd import _lib
    header = (REPO / "include" / "dei2i_hip.h").read_text()
    declared = set(re.findall(r"\b(dei2i_[a-z0-9_]+)\s*\(", header))
    declared -= {"dei2i_stream"}
    lib = _lib.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/dei2i_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dei2i_version() >= 100


def test_ops_refuse_cpu_tensors():
    from de_i2i_gan_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.to_nhwc(torch.zeros(1, 3, 4, 4), ops.F32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l1(torch.zeros(4))


def test_product_never_imports_oracle():
    for p in (REPO / "de-i2i-gan_amd").rglob("*.py"):
        assert "oracle" not in p.read_text(), p
