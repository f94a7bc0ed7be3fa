"""Shared test helpers: opt namespace with the reference's field names, formula weight fill, golden loading."""
import json
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from oracle import defectgan_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    meta = json.loads((GOLD / f"{name}.json").read_text())
    arr = np.load(GOLD / f"{name}.npz")
    c = meta["config"]
    cfg = O.Cfg(image_size=c["image_size"], ngf=c["ngf"], ndf=c["ndf"], num_layers=c["num_layers"], hidden_nc=c["hidden_nc"],
                num_scales=c.get("num_scales", 2), use_spectral=c.get("use_spectral", False),
                add_noise=c.get("add_noise", False), diff_aug=c.get("diff_aug", ""), cycle_gan=c.get("cycle_gan", False))
    O.NOISE_SOURCE = O.shape_noise if c.get("add_noise") else None      # the goldens' deterministic stand-in for N(0,1)
    return meta, arr, c, cfg


def make_opt(c, device, compute_dtype="f32", **over):
    """The attribute set the reference's DefectGanTrainer reads, plus compute_dtype."""
    opt = SimpleNamespace(
        model="defectgan", num_res=6, cycle_gan=c.get("cycle_gan", False), label_nc=6, skip_conn=False, ngf=c["ngf"], ndf=c["ndf"], input_nc=3,
        use_spectral=c.get("use_spectral", False), num_scales=c.get("num_scales", 2), style_norm_block_type="spade",
        hidden_nc=c["hidden_nc"], style_distill=False, embed_nc=768, add_noise=c.get("add_noise", False), num_layers=c["num_layers"], image_size=c["image_size"],
        batch_size=c["batch"], device=torch.device(device), is_train=True, clf_loss_type="bce", continue_training=False,
        load_model_name=None, init_type="normal", init_variance=0.02, phase="train", ckpt_dir=Path(tempfile.mkdtemp()),
        name="t", iters_per_epoch=10, num_epochs=-1, num_iters=100, lr=[2e-4], optimizer="adam", scheduler="step",
        lr_decay=5e-3, loss_weight=[2, 5, 5, 5, 1], num_critics=1, diff_aug=c.get("diff_aug", ""), sean_alpha=None, use_running_stats=False,
        save_latest_freq=10 ** 9, compute_dtype=compute_dtype)
    for k, v in over.items():
        setattr(opt, k, v)
    return opt


def formula_fill(net):
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(O.formula_tensor(k, tuple(v.shape)).to(v.device))


class record_kinks:
    """Context manager: record, in forward order, the branch the HIP path took at every piecewise-linear kink on the
    gradient path -- the input format of ``oracle.KinkTape`` (see the note above ``KINK_TAPE`` in the oracle).
    'relu' / 'leaky' entries hold the fused op's NHWC output, 'l1' entries hold sign(a - b)."""

    def __enter__(self):
        from de_i2i_gan_amd import _lib as L
        from de_i2i_gan_amd import ops
        self.tape, self._patched = [], []
        kind_of = {L.ACT_RELU: "relu", L.ACT_LRELU: "leaky"}

        def conv(a, out):                # (x, weight, bias, cache, sources, geom, act); the SPADE label-path ReLU is exact
            return (kind_of[a[6]], out, "conv") if a[6] == L.ACT_LRELU else None

        def bn(a, out):                  # (y, weight, bias, res, running_mean, running_var, training, momentum, eps, act, counter)
            return (kind_of[a[9]], out) if a[9] != L.ACT_NONE else None

        def spade(a, out):
            return ("relu", out)

        def l1(a, out):                  # (a, b); b None = zeros (sigmoid output: no sign change possible)
            return ("l1", torch.sign(a[0].detach().float() - a[1].detach().float())) if a[1] is not None else None

        for cls, pick in ((ops._Conv2d, conv), (ops._BatchNormAct, bn), (ops._SpadeRelu, spade), (ops._L1, l1)):
            orig = cls.apply

            def apply(*a, _orig=orig, _pick=pick):
                out = _orig(*a)
                first = out[0] if isinstance(out, tuple) else out        # (z, x handed through) of spade_relu(skip=True)
                if first.requires_grad:
                    r = _pick(a, first)
                    if r is not None:
                        self.tape.append((r[0], r[1].detach().double().cpu()) + tuple(r[2:]))
                return out
            cls.apply = apply
            self._patched.append(cls)
        return self.tape

    def __exit__(self, *exc):
        for cls in self._patched:
            del cls.apply                # fall back to torch.autograd.Function.apply
        return False


def unbatch_discriminator(tape, ncalls, n):
    """The product runs the discriminator ONCE on the concatenation of `ncalls` image batches of `n` (it has no batch
    statistics); the oracle follows the reference and calls it once per batch.  Re-order the discriminator's records
    (the fused conv+LeakyReLU sites, one run of them per D pass) into per-call order and drop the source tag."""
    out, i = [], 0
    while i < len(tape):
        if len(tape[i]) > 2 and tape[i][2] == "conv" and tape[i][1].shape[0] == ncalls * n:
            j = i
            while j < len(tape) and len(tape[j]) > 2 and tape[j][2] == "conv" and tape[j][1].shape[0] == ncalls * n:
                j += 1
            for c in range(ncalls):
                out.extend((k, t[c * n:(c + 1) * n]) for k, t, *_ in tape[i:j])
            i = j
        else:
            out.append(tuple(tape[i][:2]))
            i += 1
    return out


def unpair_generator(tape, n):
    """The product runs the G loss's four generator passes as TWO passes over 2 x batch (ops.paired_passes: heads [bg | df], then
    tails [fake_defects | fake_normals]); the oracle follows the reference's order bg -> fake_defects, fake_defects ->
    recover_normals, df -> fake_normals, fake_normals -> recover_defects.  Re-order the generator's records -- the leading run of
    untagged norm + activation sites over 2n samples, half of it per paired pass -- into that per-pass order."""
    j = 0
    while j < len(tape) and len(tape[j]) == 2 and tape[j][0] in ("relu", "leaky") and tape[j][1].shape[0] == 2 * n:
        j += 1
    if j == 0 or j % 2:
        return list(tape)
    heads, tails = tape[:j // 2], tape[j // 2:j]
    out = []
    for lo in (0, n):
        out.extend((k, t[lo:lo + n]) for k, t in heads)
        out.extend((k, t[lo:lo + n]) for k, t in tails)
    # the paired form takes its L1 losses over the whole tensors: the reconstruction site holds [recover_normals - bg | recover_defects -
    # df] (the reference evaluates the defects term first), the cycle site [df_prob - rec_df_prob | nm_prob - rec_nm_prob] (in order)
    rest, seen = [], 0
    for rec in tape[j:]:
        if rec[0] == "l1" and len(rec) == 2 and rec[1].shape[0] == 2 * n:
            halves = (rec[1][n:], rec[1][:n]) if seen == 0 else (rec[1][:n], rec[1][n:])
            rest.extend(("l1", h) for h in halves)
            seen += 1
        else:
            rest.append(rec)
    return out + rest
