"""
ctypes binding of include/ddpm3d.h (csrc/libddpm3d.so, gfx950).

There is deliberately no fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  torch is used here only for device memory
and the current HIP stream.
"""

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDPM3D_LIB: developer override to A/B an experimental build of the same ABI
LIB_PATH = os.environ.get("DDPM3D_LIB") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "libddpm3d.so"))

IN_SAME, IN_POOL, IN_UP, IN_PLANAR2, IN_STRIDE2 = 0, 1, 2, 3, 4
RES_NONE, RES_SAME, RES_POOL, RES_UP = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1
OUT_NDHWC, OUT_NCDHW = 0, 1
F_LEARN_SIGMA, F_PREDICT_XSTART, F_CLIP = 1, 2, 4
NCOEF = 8
PREC_F32, PREC_F16X3, PREC_F16, PREC_F16X3_WZ, PREC_F16_WZ, PREC_BF16, PREC_BF16_WZ = 0, 1, 2, 3, 4, 5, 6
PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3, "f16": PREC_F16, "bf16": PREC_BF16}
# the Winograd-along-depth form of a mode (same arithmetic, 2/3 of the MFMAs), where one exists
WINOGRAD_OF = {PREC_F16X3: PREC_F16X3_WZ, PREC_F16: PREC_F16_WZ, PREC_BF16: PREC_BF16_WZ}
# ddpm3d_conv_desc.io_dtype bits: which activation tensors hold bf16
IO_SRC0_BF16, IO_SRC1_BF16, IO_OUT_BF16, IO_RES_BF16 = 1, 2, 4, 8
IO_HALF_IS_F16 = 16    # the flagged tensors hold IEEE f16 (the --use_fp16 storage), not bf16
ABI_VERSION = 12
# ddpm3d_conv_desc.kernel_hint bits (launch orders of identical arithmetic; tests and A/B measurements)
HINT_WSTAT_OFF, HINT_WSTAT_ON = 0x100, 0x200
HINT_SPLITK_SHIFT = 16      # bits 16..21: forced split factor (measurement only, tools/splitk_sweep.py)
HINT_WZ_ORDER_SHIFT = 12     # bits 12..14: tap issue order of the f16x3 Winograd-D kernel (A/B measurements)

_fp = C.c_void_p


class ConvDesc(C.Structure):
    """struct ddpm3d_conv_desc (field order is the ABI)."""
    _fields_ = [
        ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32), ("ksize", C.c_int32), ("in_mode", C.c_int32),
        ("src0", _fp), ("src1", _fp), ("C0", C.c_int32), ("C1", C.c_int32),
        ("aff_a", _fp), ("aff_b", _fp), ("act", C.c_int32), ("precision", C.c_int32),
        ("w_packed", _fp), ("bias", _fp), ("bias_stride_n", C.c_int32), ("res_mode", C.c_int32),
        ("res", _fp), ("out", _fp), ("out_layout", C.c_int32), ("stats_rows", C.c_int32),
        ("stats", _fp), ("workspace", _fp), ("workspace_bytes", C.c_size_t),
        ("kernel_hint", C.c_int32), ("in_bound_count", C.c_int32), ("in_bound", _fp),
        ("in_bound_stride", C.c_int32), ("io_dtype", C.c_int32),
    ]


class ConvWeights(C.Structure):
    """struct ddpm3d_conv_weights"""
    _fields_ = [("w_packed", _fp), ("w_packed_wz", _fp), ("bias", _fp), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("ksize", C.c_int32), ("precision", C.c_int32), ("precision_wz", C.c_int32)]


class Layer(C.Structure):
    """struct ddpm3d_layer"""
    _fields_ = [("kind", C.c_int32), ("updown", C.c_int32), ("heads", C.c_int32), ("film_off", C.c_int32),
                ("norm1_gamma", _fp), ("norm1_beta", _fp), ("norm2_gamma", _fp), ("norm2_beta", _fp),
                ("conv1", ConvWeights), ("conv2", ConvWeights), ("skip", ConvWeights)]


class UnetDesc(C.Structure):
    """struct ddpm3d_unet_desc"""
    _fields_ = [("n_layers", C.c_int32), ("layers", C.POINTER(Layer)),
                ("n_input_blocks", C.c_int32), ("input_block_layers", C.POINTER(C.c_int32)),
                ("n_middle_layers", C.c_int32),
                ("n_output_blocks", C.c_int32), ("output_block_layers", C.POINTER(C.c_int32)),
                ("first", ConvWeights), ("out_gamma", _fp), ("out_beta", _fp), ("out", ConvWeights),
                ("film", C.c_int32), ("planar", C.c_int32), ("in_channels", C.c_int32), ("cin_pad", C.c_int32),
                ("arithmetic", C.c_int32)]


LAYER_RES, LAYER_ATTN, LAYER_DOWNCONV, LAYER_UPCONV = 1, 2, 3, 4
UPDOWN = {None: 0, "down": 1, "up": 2}

EXPORTS = {
    # name: (restype, argtypes)
    "ddpm3d_abi_version": (C.c_int, []),
    "ddpm3d_last_error": (C.c_char_p, []),
    "ddpm3d_packed_weight_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpm3d_pack_conv_weight": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_conv_stats_rows": (C.c_int, [C.c_int] * 8),
    "ddpm3d_conv_workspace_bytes": (C.c_size_t, [C.c_int] * 8),
    "ddpm3d_conv3d": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "ddpm3d_conv_kernel_family": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, C.c_int]),
    "ddpm3d_conv_plan": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "ddpm3d_unet_plan_bytes": (C.c_size_t, [C.POINTER(UnetDesc), C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpm3d_unet_plan_create": (C.c_int, [C.POINTER(UnetDesc), C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t,
                                          C.POINTER(_fp)]),
    "ddpm3d_unet_forward": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, _fp, _fp]),
    "ddpm3d_unet_plan_destroy": (None, [_fp]),
    "ddpm3d_unet_last_error": (C.c_char_p, []),
    "ddpm3d_gn_finalize": (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_float, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "ddpm3d_absmax": (C.c_int, [_fp, _fp, C.c_int, C.c_size_t, _fp, _fp]),
    "ddpm3d_gn_stats_rows": (C.c_int, [C.c_int]),
    "ddpm3d_gn_stats": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_timestep_embedding": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ddpm3d_linear": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, _fp, C.c_int, _fp]),
    "ddpm3d_attention": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_attention_p": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int,
                                     _fp, _fp]),
    "ddpm3d_ncdhw_to_ndhwc": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_ndhwc_to_ncdhw": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_ncdhw_to_ndhwc_pad": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_subsample_hw2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ddpm3d_p_sample_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ddpm3d_ddim_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float,
                                   _fp, _fp, _fp]),
    "ddpm3d_add_embedding": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ddpm3d_pool_act": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp,
                                  C.c_int, _fp]),
    "ddpm3d_mfma_probe_flops_per_iter": (C.c_double, [C.c_int]),
    "ddpm3d_mfma_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
}

_lib = None


def load():
    """The loaded library (cached).  Raises if it is absent or of another ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "ddpm3d HIP library not found at %s -- build it with "
            "`make -C 3d-denoising-diffusion-model_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ddpm3d_abi_version() != ABI_VERSION:
        raise RuntimeError("libddpm3d ABI %d, binding expects %d" % (lib.ddpm3d_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


E_INVAL, E_LAUNCH, E_NOSUP, E_2BIG = -1, -2, -3, -4
PROBE_F16_32X32X16, PROBE_F16_16X16X32, PROBE_F32_32X32X2, PROBE_BF16_32X32X16, PROBE_BF16_16X16X32 = 0, 1, 2, 3, 4


class Ddpm3dError(RuntimeError):
    """A failed C-ABI call; `.code` is the DDPM3D_E* value it returned."""

    def __init__(self, code, msg):
        super().__init__("ddpm3d error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise Ddpm3dError(rc, load().ddpm3d_last_error().decode())


def conv_plan(desc):
    """(stats rows, workspace bytes, split factor over Cin) of ddpm3d_conv3d on this descriptor"""
    rows, ws, split = C.c_int(0), C.c_size_t(0), C.c_int(0)
    check(load().ddpm3d_conv_plan(C.byref(desc), C.byref(rows), C.byref(ws), C.byref(split)))
    return rows.value, ws.value, split.value


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_device(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s must live on the GPU: this package runs on HIP kernels only "
                           "(got %s)" % (what, getattr(t, "device", type(t))))
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("%s must be contiguous float32" % what)
