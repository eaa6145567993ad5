"""ctypes binding of libcrdr_hip.so (the C ABI declared in include/crdr_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing, or an entry point
returns an error, this module raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# CRDR_HIP_LIB selects another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("CRDR_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "_lib", "libcrdr_hip.so"))

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "C", "OH", "OW", "OC", "kh", "kw", "stride", "pad", "transposed", "ldx", "ldy",
        "wrows", "wcols", "flags", "ldres", "ldg", "wlayout", "reserved", "ldpre", "ldmask")]


class ConvIO(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "x", "w", "y", "bias", "vec2", "res", "scale", "shift", "gx", "gt", "sig", "pre", "mask", "cs")]


class ColsumJob(C.Structure):
    _fields_ = [("cs", c_void_p), ("out_pre", c_void_p), ("out_post", c_void_p)] + [(n, C.c_int32) for n in (
        "rows", "ld", "C", "accumulate", "nslab", "cpad")] + [("scratch_off", C.c_int64)]


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "PH", "PW", "PC", "ldp", "QH", "QW", "QC", "ldq", "kh", "kw", "stride", "pad", "gI", "gJ",
        "accumulate", "algo")]


class EbwdDesc(C.Structure):
    _fields_ = [("M", C.c_int64)] + [(n, C.c_int32) for n in (
        "C", "flags", "lddout", "ldout", "lddz", "ldgres", "ldg")]


class WgradJob(C.Structure):
    _fields_ = [("slab", c_void_p), ("g", c_void_p)] + [(n, C.c_int32) for n in (
        "PC", "QC", "gI", "gJ", "T", "nsplit", "smallj", "accumulate", "gJtot", "reserved")]


class LinearGroup(C.Structure):  # mirrors struct crdr_linear_group
    _fields_ = [(n, c_void_p * 16) for n in ("w", "b", "y", "dy", "dw", "db")] + [("O", C.c_int32 * 16)]


class PackItem(C.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p)] + [(n, C.c_int32) for n in (
        "I", "J", "T", "rows", "cols", "mode", "srcJ", "dld")] + [("tstride", C.c_int64)]


class W4FilterItem(C.Structure):  # mirrors struct crdr_w4_filter_item (320 bytes)
    _fields_ = [("w", c_void_p * 16), ("u", c_void_p)] + [(n, C.c_int32) for n in (
        "G", "Cin", "Cout", "wrows", "wcols", "kchunks", "ntile", "nvar")] + [("widx", (C.c_int32 * 9) * 4), ("units", C.c_int64)]


class GdnDesc(C.Structure):
    _fields_ = [("M", C.c_int64)] + [(n, C.c_int32) for n in ("C", "ldx", "ldy", "inverse")] + [
        ("beta_min", C.c_float), ("reparam_offset", C.c_float)]


class GcDesc2(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "HW", "C", "ldy", "ldmu", "ldsigma", "ldyhat", "ldyhat2", "ldnoise", "ldlik",
                                         "ldgrad", "lddyhat", "Ctot", "c0")] + [
        ("scale_bound", C.c_float), ("likelihood_bound", C.c_float)]


class GcIO(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("y", "mu", "sigma", "noise", "philox", "yhat", "yhat2", "lik_noisy", "lik_quant",
                                        "bits_noisy", "bits_quant", "gbits", "dyhat", "dy", "dmu", "dsigma", "ws")] + [
        ("ws_bytes", C.c_size_t)]


class EbwdIO(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "dout", "out", "vec2", "scale", "shift", "gt", "sig", "dz", "gres", "dgt", "colsums", "dbias_accum")]


class GcDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "HW", "C", "ldy", "ldmu", "ldsigma", "ldyhat")] + [
        ("scale_bound", C.c_float), ("likelihood_bound", C.c_float)]


EPI_BIAS, EPI_RELU, EPI_LRELU, EPI_VEC2, EPI_RES, EPI_GATE, EPI_AFFINE, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64, 128
EPI_PREADD, EPI_RELUMASK, EPI_LRELUMASK, EPI_MASKOFF, EPI_COLSUM = 256, 512, 1024, 2048, 4096
CONV_NOSPLIT, CONV_BF16X3, WGRAD_BF16X3 = 8192, 16384, 1 << 16
CONV_BF16X6, WGRAD_BF16X6 = 32768, 1 << 17
WGRAD_SQUARE_Q = 1 << 18   # crdr_hip.h: the gathered operand enters squared (the GDN gamma gradient)
MAX_GROUP = 16
EB_PARAMS = 58

# name -> (restype, argtypes); every symbol include/crdr_hip.h declares
_I, _I64, _F, _D, _P, _SZ = C.c_int, C.c_int64, C.c_float, C.c_double, c_void_p, C.c_size_t
SIGNATURES = {
    "crdr_last_error": (C.c_char_p, []),
    "crdr_version": (_I, []),
    "crdr_arch": (C.c_char_p, []),
    "crdr_profile_enable": (None, [_I]),
    "crdr_profile_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "crdr_conv2d_num_configs": (_I, []),
    "crdr_conv2d_num_stream_configs": (_I, []),
    "crdr_conv2d_num_wino_configs": (_I, []),
    "crdr_conv2d_wgrad_num_configs": (_I, []),
    "crdr_conv2d_wgrad_num_wino_configs": (_I, []),
    "crdr_conv2d_workspace": (_SZ, [C.POINTER(ConvDesc)]),
    "crdr_conv2d_choose_algo": (_I, [C.POINTER(ConvDesc), _I]),
    "crdr_conv2d": (_I, [C.POINTER(ConvDesc), C.POINTER(ConvIO), _P, _SZ, _P]),
    "crdr_conv2d_flops": (_D, [C.POINTER(ConvDesc)]),
    "crdr_conv2d_colsum_layout": (_I, [C.POINTER(ConvDesc), _I, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crdr_colsum_slab_rows": (_I, []),
    "crdr_colsum_finish_batched": (_I, [_P, _P, _P, _P, _P, _P]),
    "crdr_conv2d_grouped_workspace": (_SZ, [C.POINTER(ConvDesc), _I]),
    "crdr_conv2d_grouped": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _SZ, _P]),
    "crdr_conv2d_filter_cache_bytes": (_SZ, [C.POINTER(ConvDesc), _I]),
    "crdr_conv2d_grouped_ex": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _SZ, _P, _SZ, _I, _P]),
    "crdr_conv2d_filter_item": (_I, [C.POINTER(ConvDesc), _I, C.POINTER(W4FilterItem)]),
    "crdr_w4_filters_batched": (_I, [_P, _P, _P, _P]),
    "crdr_conv2d_wgrad_grouped_workspace": (_SZ, [C.POINTER(WgradDesc), _I]),
    "crdr_conv2d_wgrad_partial_grouped": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _I, _P, _SZ, _P, _P]),
    "crdr_pack_weight_item": (_I, [C.POINTER(PackItem), _P]),
    "crdr_colsum_scatter": (_I, [_P, _I, _I64, _I, _I, _P, _I, _P, _SZ, _P]),
    "crdr_gauss_cond_fwd_workspace": (_SZ, [C.POINTER(GcDesc2)]),
    "crdr_gauss_cond_fwd2": (_I, [C.POINTER(GcDesc2), C.POINTER(GcIO), _P]),
    "crdr_gauss_cond_bwd2": (_I, [C.POINTER(GcDesc2), C.POINTER(GcIO), _P]),
    "crdr_gdn_workspace": (_SZ, [C.POINTER(GdnDesc), _I]),
    "crdr_gdn_fwd": (_I, [C.POINTER(GdnDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    "crdr_gdn_bwd": (_I, [C.POINTER(GdnDesc), _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "crdr_gauss_symbols": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _F, _I, _I, _I, _P, _P, _P]),
    "crdr_philox_fork": (_I, [_P, _P, C.c_uint64, _P]),
    "crdr_philox_uniform": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "crdr_conv2d_wgrad_workspace": (_SZ, [C.POINTER(WgradDesc)]),
    "crdr_conv2d_wgrad": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _SZ, _P]),
    "crdr_conv2d_wgrad_partial": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _SZ, C.POINTER(WgradJob), _P]),
    "crdr_wgrad_reduce_batched": (_I, [_P, _P, _P, _P]),
    "crdr_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "crdr_pack_weights_batched": (_I, [_P, _P, _P, _P]),
    "crdr_epilogue_bwd_workspace": (_SZ, [C.POINTER(EbwdDesc)]),
    "crdr_epilogue_bwd": (_I, [C.POINTER(EbwdDesc), C.POINTER(EbwdIO), _P, _SZ, _P]),
    "crdr_col2im_rgb": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "crdr_spectral_norm_fwd": (_I, [_P, _I, _I, _P, _P, _I, C.c_float, _P, _P, _P, _SZ, _P]),
    "crdr_spectral_norm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _SZ, _P]),
    "crdr_crop_flip_normalize": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "crdr_linear_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "crdr_linear_bwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P]),
    "crdr_linear_group_fwd": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "crdr_linear_group_bwd": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "crdr_affine": (_I, [_P, _I, _P, _P, _P, _I, _I64, _I, _P]),
    "crdr_colsum_workspace": (_SZ, [_I64, _I]),
    "crdr_colsum": (_I, [_P, _I, _I64, _I, _P, _I, _P, _SZ, _P]),
    "crdr_interp_ca_params": (_I, [_P, _P, _I, _I, _F, _P, _P, _P]),
    "crdr_interp_ca_params_bwd": (_I, [_P, _I, _I, _F, _P, _P, _P, _P, _P]),
    "crdr_bias_relu_slots": (_I, [_P, _I, _I64, _I, _I, _P, _P, _P]),
    "crdr_lrp": (_I, [_P, _I, _P, _I, _P, _I, _I64, _I, _P]),
    "crdr_lrp_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I64, _I, _P]),
    "crdr_gauss_cond_fwd": (_I, [C.POINTER(GcDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "crdr_gauss_cond_bwd": (_I, [C.POINTER(GcDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "crdr_entropy_bottleneck_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "crdr_entropy_bottleneck_bwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "crdr_eb_quantile_loss": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "crdr_reduce_workspace": (_SZ, [_I64]),
    "crdr_sqdiff_sum": (_I, [_P, _P, _I64, _P, _P, _SZ, _P]),
    "crdr_sqdiff_bwd": (_I, [_P, _P, _I64, _P, _F, _P, _P, _P]),
    "crdr_bce_diff_sum": (_I, [_P, _P, _I64, _F, _P, _P, _SZ, _P]),
    "crdr_bce_diff_bwd": (_I, [_P, _P, _I64, _F, _P, _F, _P, _P, _P]),
    "crdr_sqnorm": (_I, [_P, _I64, _P, _P, _SZ, _P]),
    "crdr_adam_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _P, _F, _P]),
    "crdr_adam_step_dyn": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _P, _P, _F, _P]),
    "crdr_maxpool3s2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "crdr_maxpool3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "crdr_lpips_layer_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "crdr_lpips_layer_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "crdr_pmf_to_quantized_cdf": (_I, [_P, _I, _I, _P]),
    "crdr_rans_encode_with_indexes": (_I64, [_P, _P, _I64, _P, _I, _P, _P, _I, _P, _I64]),
    "crdr_rans_decoder_create": (_P, []),
    "crdr_rans_decoder_destroy": (None, [_P]),
    "crdr_rans_decoder_set_stream": (_I, [_P, _P, _I64]),
    "crdr_rans_decoder_decode_stream": (_I, [_P, _P, _I64, _P, _I, _P, _P, _I, _P]),
    "crdr_rans_decode_with_indexes": (_I, [_P, _I64, _P, _I64, _P, _I, _P, _P, _I, _P]),
}


class CrdrHipError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 6) -> str:
    """Compile libcrdr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(["make", "-C", CSRC_DIR, f"-j{jobs}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise CrdrHipError("building libcrdr_hip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


_lock = threading.Lock()
_lib = None


def load():
    """Load the library (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise CrdrHipError(
                f"{LIB_PATH} not found: the HIP extension is required (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C crdr_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what: str = ""):
    if rc is not None and rc < 0:
        msg = load().crdr_last_error().decode("utf-8", "replace")
        raise CrdrHipError(f"{what}: rc={rc}: {msg}")
    return rc
