"""ctypes binding of libamphion_hip.so (C ABI in include/amphion_hip.h).

There is no CPU fallback anywhere in this package: if the library is missing, or
a tensor is not on a ROCm device, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMP_LIB_PATH: another build of the same ABI (same-box A/B of kernel changes, tools/gpu_r2_z.sh)
LIB_PATH = os.environ.get("AMP_LIB_PATH") or os.path.join(_HERE, "lib", "libamphion_hip.so")

AMP_MAX_STAGES = 8
AMP_MAX_KERNELS = 8
AMP_MAX_DILATIONS = 8

AMP_ARCH_HIFIGAN, AMP_ARCH_BIGVGAN, AMP_ARCH_HIFIGAN_VITS = 0, 1, 2
AMP_ACT_LRELU, AMP_ACT_SNAKE, AMP_ACT_SNAKEBETA = 0, 1, 2
AMP_PRECISION_F32, AMP_PRECISION_F16X3 = 0, 1
AMP_CONV_OPT_PAD_REFLECT, AMP_CONV_OPT_TANH = 1, 2
AMP_PAD_REPLICATE, AMP_PAD_ZEROS, AMP_PAD_REFLECT = 0, 1, 2
PRECISIONS = {"f32": AMP_PRECISION_F32, "fp32": AMP_PRECISION_F32, "f16x3": AMP_PRECISION_F16X3}


class AmpError(RuntimeError):
    """A libamphion_hip call returned a negative amp_status."""

    def __init__(self, status, message):
        super().__init__(f"libamphion_hip error {status}: {message}")
        self.status = status


class amp_gen_desc(ctypes.Structure):
    _fields_ = [
        ("arch", c_int32),
        ("n_in", c_int32),
        ("upsample_initial_channel", c_int32),
        ("n_stages", c_int32),
        ("upsample_rates", c_int32 * AMP_MAX_STAGES),
        ("upsample_kernel_sizes", c_int32 * AMP_MAX_STAGES),
        ("n_kernels", c_int32),
        ("resblock_kernel_sizes", c_int32 * AMP_MAX_KERNELS),
        ("n_dilations", c_int32 * AMP_MAX_KERNELS),
        ("resblock_dilation_sizes", (c_int32 * AMP_MAX_DILATIONS) * AMP_MAX_KERNELS),
        ("resblock_type", c_int32),
        ("activation", c_int32),
        ("snake_logscale", c_int32),
        ("gin_channels", c_int32),
    ]


class amp_mel_desc(ctypes.Structure):
    """``amp_mel_desc(n_fft, win_size, hop_size, n_mel, pad_mode, mag_eps, log_clip[, mel_bands_dev, ...])``: ``struct_size`` (the
    header's first field since amp_version 140) is filled in here."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_fft", c_int32),
        ("win_size", c_int32),
        ("hop_size", c_int32),
        ("n_mel", c_int32),
        ("pad_mode", c_int32),
        ("mag_eps", c_float),
        ("log_clip", c_float),
        ("mel_bands_dev", c_void_p),
        ("range_dev", c_void_p),
        ("range_host", c_void_p),
        ("range_reset_dev", c_void_p),
        ("range_seq", c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)


_SIGNATURES = {
    "amp_version": (c_int, []),
    "amp_last_error": (c_char_p, []),
    "amp_device_count": (c_int, []),
    "amp_set_precision": (c_int, [c_int]),
    "amp_get_precision": (c_int, []),
    "amp_gen_create": (c_int, [POINTER(amp_gen_desc), POINTER(c_void_p)]),
    "amp_gen_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "amp_gen_finalize": (c_int, [c_void_p]),
    "amp_gen_hop": (c_int, [c_void_p]),
    "amp_gen_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "amp_set_group_mb": (c_int, [c_int]),
    "amp_set_pair_strips": (c_int, [c_int]),
    "amp_range_check": (c_int, [c_void_p]),
    "amp_gen_range_check": (c_int, [c_void_p, c_void_p]),
    "amp_gen_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "amp_gen_forward_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "amp_gen_set_profiling": (c_int, [c_void_p, c_int]),
    "amp_gen_last_timing_ms": (c_int, [c_void_p, c_int, POINTER(c_float)]),
    "amp_gen_timing_ms": (c_int, [c_void_p, c_int, c_int, POINTER(c_float)]),
    "amp_gen_kernel_name": (c_int, [c_void_p, c_int, c_int, c_char_p, c_size_t]),
    "amp_gen_destroy": (None, [c_void_p]),
    "amp_conv_create": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, POINTER(c_void_p)]),
    "amp_conv_out_len": (c_int, [c_void_p, c_int]),
    "amp_conv_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_float, c_void_p, c_void_p]),
    "amp_conv_forward_strided": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_float, c_void_p, c_float, c_void_p, c_void_p]),
    "amp_conv_forward_ragged": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_float, c_void_p, c_float, c_void_p, c_void_p]),
    "amp_conv_forward_mrf": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "amp_apnet_polar": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_istft_same": (c_int, [POINTER(amp_mel_desc), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_layer_norm_c": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "amp_add_channel_bias": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "amp_layer_norm_c_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "amp_dds_seam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float,
                             c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "amp_dwconv_layer_norm_c": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "amp_rel_attention_strided": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_set_rel_attention_tiled": (c_int, [c_int]),
    "amp_set_resblock_streams": (c_int, [c_int]),
    "amp_gen_prepare_streams": (c_int, [c_void_p]),
    "amp_rel_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_dwconv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_spline_flow": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_spline_flow_proj": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_affine_reverse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_embed_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "amp_durations": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_expand_path": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "amp_expand_path_strided": (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "amp_gauss_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_void_p]),
    "amp_snake": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "amp_fir_upsample": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "amp_fir_filter": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "amp_wav_to_pcm16": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_void_p, ctypes.c_longlong, c_void_p]),
    "amp_conv_set_option": (c_int, [c_void_p, c_int, c_int]),
    "amp_pair_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "amp_resblock_forward": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "amp_set_resblock_fusion": (c_int, [c_int]),
    "amp_conv_destroy": (None, [c_void_p]),
    "amp_set_small_conv": (c_int, [c_int]),
    "amp_set_conv_blk": (c_int, [c_int]),
    "amp_set_conv_rg_fast": (c_int, [c_int]),
    "amp_set_pingpong": (c_int, [c_int]),
    "amp_set_conv_blk_narrow": (c_int, [c_int]),
    "amp_ampblock_forward": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_void_p]),
    "amp_set_ampblock_fusion": (c_int, [c_int]),
    "amp_conv_create_gated": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, POINTER(c_void_p)]),
    "amp_wn_forward": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int, c_int,
                               c_void_p, c_void_p, c_void_p]),
    "amp_wn_gate": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_int, c_int, c_int, c_void_p]),
    "amp_wn_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "amp_sequence_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "amp_coupling_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "amp_flip_channels": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "amp_posterior_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "amp_antialias_snake": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_istft_forward": (c_int, [POINTER(amp_mel_desc), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_mel_num_frames": (c_int, [POINTER(amp_mel_desc), c_int]),
    "amp_mel_init": (c_int, []),
    "amp_mel_forward_ragged": (c_int, [POINTER(amp_mel_desc), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "amp_mel_backward": (c_int, [POINTER(amp_mel_desc), c_void_p, c_int, c_int] + [c_void_p] * 11),
    "amp_mel_forward": (c_int, [POINTER(amp_mel_desc), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m amphion_amd.build` "
                "(the HIP path has no CPU fallback)"
            )
        # torch's own libamdhip64 must be the HIP runtime of the process: loading this library first would pull in
        # /opt/rocm's copy, and the two runtimes do not see each other's devices or allocations ("no HIP device visible"
        # from a process that imported this module before torch)
        import torch  # noqa: F401

        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def set_precision(name):
    """Arithmetic of the conv contractions for handles created from now on: "f16x3" (default, split-f16
    MFMA with fp32 accumulate) or "f32" (exact fp32 MFMA).  include/amphion_hip.h: amp_precision."""
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {name!r}")
    check(lib().amp_set_precision(PRECISIONS[name]))


def get_precision():
    return "f32" if lib().amp_get_precision() == AMP_PRECISION_F32 else "f16x3"


def check(status):
    if status < 0:
        raise AmpError(status, lib().amp_last_error().decode("utf-8", "replace"))
    return status


AMP_ERR_UNSUPPORTED = -4
AMP_ERR_RANGE = -6


def range_check(device=None):
    """Synchronise the current stream of ``device`` and raise ``AmpError`` (status AMP_ERR_RANGE) if an OP-LEVEL f16x3
    launch (``amp_conv_forward``, ``amp_pair_forward``) since the last check staged an activation beyond the split-f16
    operand range (|x| > 4094 or infinite): the output of that launch is not the fp32 reference's
    (``amp_range_check``).  Generators have their own flag: ``generator.check_range()``."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        check(lib().amp_range_check(current_stream_ptr(dev)))


def require_device_tensor(t, name="tensor"):
    import torch

    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the amphion_amd kernels run on a ROCm (MI355X) device only; "
            "there is no CPU fallback (move the model and its input to 'cuda')"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


def on_device(device):
    """``torch.cuda.device(device)`` when ``device`` is not the current one, else a no-op context: the guard object and its two
    device switches cost ~3 us per launch on a path that is ~9 us of host work per launch (VITS text side)."""
    import torch

    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NULL_CTX
    return torch.cuda.device(device)


def current_stream_ptr(device):
    """the raw hipStream_t of torch's current stream on ``device`` (the private accessor torch's own compiled-graph runtime uses:
    no Stream object per launch -- the text side of VITS is ~250 launches, all host-bound)"""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return c_void_p(raw(torch.cuda.current_device() if idx is None else idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
