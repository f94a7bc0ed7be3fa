"""ctypes binding of libdei2i_hip.so (C ABI: include/dei2i_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdei2i_hip.so")

BF16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1
PROF_GATHER_GEMM, PROF_WGRAD, PROF_HALO_CONV, PROF_HALO_FOLD = 0, 1, 2, 3


class ConvDesc(Structure):
    """struct dei2i_conv"""
    _fields_ = [("dtype", c_int), ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("CinS", c_int), ("CoutS", c_int), ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int),
                ("pad_mode", c_int), ("up", c_int)]


class ProDesc(Structure):
    """struct dei2i_pro: operand-path normalisation of a conv's input (fused conv + norm + act)"""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("n_stride", c_int), ("slope", c_float), ("ring", c_void_p)]


class EpiNormDesc(Structure):
    """struct dei2i_epi_norm: the backward reductions of the norm layer in front of a conv, from that conv's dgrad epilogue"""
    _fields_ = [("kind", c_int), ("up", c_int), ("act", c_int), ("group_images", c_int), ("x", c_void_p), ("mean", c_void_p),
                ("rstd", c_void_p), ("gb", c_void_p), ("a", c_void_p), ("b", c_void_p), ("partial", c_void_p)]


class LabelMod(Structure):
    """struct dei2i_label_mod: one SPADE module's entry in the batched label-path launches (csrc/label_path.hip)"""
    _fields_ = [("gamma_weight", c_void_p), ("beta_weight", c_void_p), ("gamma_bias", c_void_p), ("beta_bias", c_void_p),
                ("packed_fwd", c_void_p), ("packed_dgrad", c_void_p), ("gb", c_void_p), ("d_gamma_weight", c_void_p),
                ("d_beta_weight", c_void_p), ("d_gamma_bias", c_void_p), ("d_beta_bias", c_void_p), ("C", c_int), ("in_off", c_int),
                ("live", c_int), ("reserved", c_int)]


class AdamRec(Structure):
    """struct dei2i_adam_rec"""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64)]


_P = c_void_p
_CD = POINTER(ConvDesc)
_PD = POINTER(ProDesc)
_ED = POINTER(EpiNormDesc)

# name -> (restype, argtypes); every symbol include/dei2i_hip.h declares
SIGNATURES = {
    "dei2i_version": (c_int, []),
    "dei2i_init": (c_int, [c_int]),
    "dei2i_error_string": (c_char_p, [c_int]),
    "dei2i_set_option": (c_int, [c_char_p, c_int]),
    "dei2i_set_debug_buffer": (c_int, [c_void_p]),
    "dei2i_packed_fwd_elems": (c_size_t, [_CD]),
    "dei2i_wgrad_slab_elems": (c_size_t, [_CD]),
    "dei2i_packed_dgrad_elems": (c_size_t, [_CD]),
    "dei2i_pack_weight_fwd": (c_int, [_CD, _P, _P, _P]),
    "dei2i_pack_weight_dgrad": (c_int, [_CD, _P, _P, _P]),
    "dei2i_pack_weight_both": (c_int, [_CD, _P, _P, _P, _P]),
    "dei2i_unpack_wgrad": (c_int, [_CD, _P, _P, c_float, _P]),
    "dei2i_conv2d_out_shape": (None, [_CD, POINTER(c_int), POINTER(c_int)]),
    "dei2i_conv2d_dgrad_shape": (None, [_CD, POINTER(c_int), POINTER(c_int)]),
    "dei2i_conv2d_workspace_bytes": (c_size_t, [_CD]),
    "dei2i_conv2d_fwd": (c_int, [_CD, _P, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "dei2i_conv2d_dgrad": (c_int, [_CD, _P, _P, _P, _P, c_size_t, _P]),
    "dei2i_conv2d_dgrad_input": (c_int, [_CD, _P, _P, _P, _P, _P, c_size_t, _P]),
    "dei2i_conv2d_wgrad": (c_int, [_CD, _P, _P, _P, _P]),
    "dei2i_quantize_fp8": (c_int, [c_size_t, _P, c_float, _P, _P]),
    "dei2i_pack_weight_fwd_fp8": (c_int, [_CD, _P, _P, c_float, _P, _P, _P]),
    "dei2i_conv2d_fp8_supported": (c_int, [_CD]),
    "dei2i_conv2d_fwd_fp8": (c_int, [_CD, _P, _P, _P, _P, c_int, _P, _P]),
    "dei2i_conv2d_wgrad_oihw": (c_int, [_CD, _P, _P, _P, c_size_t, _P, c_int, _P]),
    "dei2i_fold_pad": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "dei2i_nchw_to_nhwc": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_nhwc_to_nchw": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_nchw_to_nhwc_resize": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_cast_from_f32": (c_int, [c_int, c_size_t, _P, _P, _P]),
    "dei2i_moments_chunks": (c_int, [c_int]),
    "dei2i_moments_partial": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_bn_finalize_train": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "dei2i_bn_finalize_eval": (c_int, [c_int, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "dei2i_in_finalize": (c_int, [c_int, c_int, c_int, _P, c_float, _P, _P, _P]),
    "dei2i_affine_act_fwd": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P, c_int, _P, _P, c_float, _P]),
    "dei2i_spade_act_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, c_float, _P]),
    "dei2i_act_bwd": (c_int, [c_int, c_size_t, _P, _P, c_int, _P, _P]),
    "dei2i_colsum_blocks": (c_int, [c_size_t]),
    "dei2i_colsum": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P]),
    "dei2i_bn_bwd_chunks": (c_int, [c_size_t]),
    "dei2i_bn_bwd_partial": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "dei2i_bn_bwd_apply": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "dei2i_spade_bwd_partial": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "dei2i_spade_bwd_apply": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "dei2i_compose_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "dei2i_compose_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "dei2i_nan_guard": (c_int, [c_int, c_size_t, _P, _P, _P]),
    "dei2i_bce_logits_fwd": (c_int, [c_size_t, _P, _P, c_float, _P, _P]),
    "dei2i_bce_logits_bwd": (c_int, [c_size_t, _P, _P, c_float, _P, _P, _P]),
    "dei2i_l1_fwd": (c_int, [c_size_t, _P, _P, _P, _P]),
    "dei2i_l1_bwd": (c_int, [c_size_t, _P, _P, _P, _P, _P, _P]),
    "dei2i_noise_fwd": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P, _P]),
    "dei2i_noise_bwd": (c_int, [c_int, c_size_t, c_int, _P, _P, _P, _P, c_int, _P]),
    "dei2i_spectral_scratch_floats": (c_size_t, [c_int, c_int]),
    "dei2i_spectral_fwd": (c_int, [c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "dei2i_spectral_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "dei2i_fold_bn_weight": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "dei2i_bn_finalize_train_groups": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    "dei2i_affine_act_groups_fwd": (c_int, [c_int, c_int, c_size_t, c_int, _P, _P, _P, _P, c_int, _P, _P, c_float, _P]),
    "dei2i_affine_act_stats_groups_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "dei2i_bn_bwd_partial_groups": (c_int, [c_int, c_int, c_size_t, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "dei2i_bn_bwd_apply_groups": (c_int, [c_int, c_int, c_size_t, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P]),
    "dei2i_label_gb_packed_elems": (c_size_t, [c_int, c_int]),
    "dei2i_label_gb_pack": (c_int, [_P, c_int, c_int, _P]),
    "dei2i_label_gb_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "dei2i_label_gb_dgrad": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "dei2i_label_gb_wgrad": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "dei2i_adam_step": (c_int, [_P, c_int, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float, _P]),
    "dei2i_sgd_rmsprop_step": (c_int, [_P, c_int, c_int64, c_int, c_float, c_float, c_float, c_float, _P]),
    "dei2i_prof_enable": (c_int, [c_int, c_int]),
    "dei2i_prof_collect": (c_int, [c_int, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "dei2i_prof_collect_timed": (c_int, [c_int, POINTER(c_int64), POINTER(c_double)]),
    "dei2i_conv2d_fused_supported": (c_int, [_CD, c_int]),
    "dei2i_conv2d_stats_chunks": (c_int, [_CD]),
    "dei2i_conv2d_fwd_fused": (c_int, [_CD, _P, _P, _P, c_int, _P, _PD, _P, _P]),
    "dei2i_conv2d_wgrad_pro_supported": (c_int, [_CD]),
    "dei2i_conv2d_wgrad_oihw_pro": (c_int, [_CD, _P, _P, _P, c_size_t, _P, c_int, _PD, _P]),
    "dei2i_ring_pixels": (c_size_t, [c_int, c_int]),
    "dei2i_conv2d_ring_supported": (c_int, [_CD]),
    "dei2i_conv2d_dgrad_norm_supported": (c_int, [_CD]),
    "dei2i_conv2d_dgrad_norm_chunks": (c_int, [_CD]),
    "dei2i_conv2d_dgrad_input_norm": (c_int, [_CD, _P, _P, _P, _ED, _P]),
    "dei2i_in_act_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P]),
    "dei2i_in_affine_act_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "dei2i_avgpool2_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_avgpool2_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dei2i_spade_bwd_border": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "dei2i_conv2d_fwd_ring": (c_int, [_CD, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "dei2i_affine_act_img_fwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P]),
    "dei2i_spade_prep": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "dei2i_in_finalize_chunks": (c_int, [c_int, c_int, c_int, c_int, _P, c_float, _P, _P, _P]),
    "dei2i_bn_finalize_train_chunks": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "dei2i_affine_act_stats_fwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "dei2i_launch_counts": (c_int, [POINTER(c_int64), c_int]),
    "dei2i_launch_counts_reset": (None, []),
    "dei2i_kernel_name": (c_char_p, [c_int]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; whichever copy is mapped first serves the whole process.  Import torch
    # first so this library shares torch's HIP runtime (streams, device pointers) instead of /opt/rocm's copy.
    import torch  # noqa: F401
    path = os.environ.get("DEI2I_LIB", LIB_PATH)     # a differently built copy of the SAME library, for same-box A/B timing
    if not os.path.exists(path):
        raise RuntimeError(
            f"libdei2i_hip.so not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the de-i2i-gan_amd ops.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def launch_counts(reset=False):
    """{kernel family name: host-side launch count since the last reset} of the MFMA kernels (dei2i_launch_counts)."""
    lib = load()
    buf = (c_int64 * 32)()
    n = lib.dei2i_launch_counts(buf, 32)
    out = {lib.dei2i_kernel_name(i).decode(): int(buf[i]) for i in range(n)}
    if reset:
        lib.dei2i_launch_counts_reset()
    return out


def check(rc, what=""):
    if rc != 0:
        msg = load().dei2i_error_string(int(rc))
        raise RuntimeError(f"dei2i {what} failed: code {rc} ({msg.decode() if msg else '?'})")
