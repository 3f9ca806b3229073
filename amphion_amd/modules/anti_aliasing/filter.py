"""Kaiser-windowed sinc low-pass filter (modules/anti_aliasing/filter.py:30-99): host-side filter design, and
LowPassFilter1d.forward on the depthwise FIR kernel ``amp_fir_filter`` (inside Activation1d the filtering is fused
into ``amp_antialias_snake`` instead)."""
import ctypes
import math

import torch
import torch.nn as nn

from amphion_amd import _lib

_PAD_MODES = {"replicate": _lib.AMP_PAD_REPLICATE, "constant": _lib.AMP_PAD_ZEROS, "reflect": _lib.AMP_PAD_REFLECT}


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """filter.py:30-61 -> [1, 1, kernel_size]"""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if even:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    filter_ = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filter_ /= filter_.sum()
    return filter_.view(1, 1, kernel_size)


class LowPassFilter1d(nn.Module):
    """filter.py:64-99 (buffer key ``filter``)."""

    def __init__(self, cutoff=0.5, half_width=0.6, stride: int = 1, padding: bool = True,
                 padding_mode: str = "replicate", kernel_size: int = 12):
        super().__init__()
        if cutoff < -0.0:
            raise ValueError("Minimum cutoff must be larger than zero.")
        if cutoff > 0.5:
            raise ValueError("A cutoff above 0.5 does not make sense.")
        self.kernel_size = kernel_size
        self.even = kernel_size % 2 == 0
        self.pad_left = kernel_size // 2 - int(self.even)
        self.pad_right = kernel_size // 2
        self.stride = stride
        self.padding = padding
        self.padding_mode = padding_mode
        self.register_buffer("filter", kaiser_sinc_filter1d(cutoff, half_width, kernel_size))

    def forward(self, x):  # x: [B, C, T]      (filter.py:92-99)
        x = _lib.require_device_tensor(x, "LowPassFilter1d input")
        B, C, T = x.shape
        if self.padding and self.padding_mode not in _PAD_MODES:
            raise NotImplementedError(f"padding_mode {self.padding_mode!r}: the HIP filter pads by replicate, constant or reflect")
        pl, pr = (self.pad_left, self.pad_right) if self.padding else (0, 0)
        mode = _PAD_MODES[self.padding_mode] if self.padding else _lib.AMP_PAD_REPLICATE
        t_out = (T + pl + pr - self.kernel_size) // self.stride + 1
        y = torch.empty((B, C, max(t_out, 0)), dtype=torch.float32, device=x.device)
        taps = self.filter.detach().reshape(-1).float().cpu().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().amp_fir_filter(
                ctypes.c_void_p(x.data_ptr()), B, C, T, ctypes.c_void_p(taps.data_ptr()), taps.numel(), self.stride, pl, pr,
                mode, ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(x.device)))
        return y
