"""UpSample1d / DownSample1d (modules/anti_aliasing/resample.py:17-65) with the reference's buffer keys
(``upsample.filter``, ``downsample.lowpass.filter``).  Inside Activation1d the resampling runs fused with the
activation (``amp_antialias_snake``); called on their own, the modules run the depthwise FIR kernels
``amp_fir_upsample`` / ``amp_fir_filter``."""
import ctypes

import torch
import torch.nn as nn

from amphion_amd import _lib

from .filter import LowPassFilter1d, kaiser_sinc_filter1d


class UpSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=None):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
        self.stride = ratio
        self.pad = self.kernel_size // ratio - 1
        self.pad_left = self.pad * self.stride + (self.kernel_size - self.stride) // 2
        self.pad_right = self.pad * self.stride + (self.kernel_size - self.stride + 1) // 2
        self.register_buffer(
            "filter", kaiser_sinc_filter1d(cutoff=0.5 / ratio, half_width=0.6 / ratio, kernel_size=self.kernel_size)
        )

    def forward(self, x):  # x: [B, C, T] -> [B, C, ratio * T]      (resample.py:36-45)
        x = _lib.require_device_tensor(x, "UpSample1d input")
        B, C, T = x.shape
        y = torch.empty((B, C, self.ratio * T), dtype=torch.float32, device=x.device)
        taps = self.filter.detach().reshape(-1).float().cpu().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().amp_fir_upsample(
                ctypes.c_void_p(x.data_ptr()), B, C, T, ctypes.c_void_p(taps.data_ptr()), taps.numel(), self.ratio,
                ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(x.device)))
        return y


class DownSample1d(nn.Module):
    def __init__(self, ratio=2, kernel_size=None):
        super().__init__()
        self.ratio = ratio
        self.kernel_size = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
        self.lowpass = LowPassFilter1d(cutoff=0.5 / ratio, half_width=0.6 / ratio, stride=ratio,
                                       kernel_size=self.kernel_size)

    def forward(self, x):  # resample.py:62-65
        return self.lowpass(x)
