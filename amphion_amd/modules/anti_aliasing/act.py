"""Anti-aliased activation (modules/anti_aliasing/act.py:12-36): up x2 -> Snake -> down x2, executed
as ONE fused gfx950 kernel (amp_antialias_snake) instead of the reference's three tensor ops.  Any other ratio /
kernel size (or an activation that is not Snake / SnakeBeta) runs as the three stand-alone HIP ops."""
import ctypes

import torch
import torch.nn as nn

from amphion_amd import _lib

from .resample import DownSample1d, UpSample1d


class Activation1d(nn.Module):
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12,
                 down_kernel_size: int = 12):
        super().__init__()
        self._fused = (up_ratio, down_ratio, up_kernel_size, down_kernel_size) == (2, 2, 12, 12)
        self.up_ratio = up_ratio
        self.down_ratio = down_ratio
        self.act = activation
        self.upsample = UpSample1d(up_ratio, up_kernel_size)
        self.downsample = DownSample1d(down_ratio, down_kernel_size)

    def forward(self, x):  # x: [B, C, T]
        x = _lib.require_device_tensor(x, "Activation1d input")
        if not (self._fused and hasattr(self.act, "alpha") and hasattr(self.act, "alpha_logscale")):
            return self.downsample(self.act(self.upsample(x)))       # act.py:32-34 as three launches
        B, C, T = x.shape
        y = torch.empty_like(x)
        alpha = self.act.alpha.detach().float().contiguous()
        beta = self.act.beta.detach().float().contiguous() if getattr(self.act, "has_beta", False) else None
        if alpha.device != x.device:
            raise RuntimeError(f"Activation1d parameters are on {alpha.device}, input on {x.device}")
        fu = self.upsample.filter.detach().reshape(-1).float().cpu().contiguous()
        fd = self.downsample.lowpass.filter.detach().reshape(-1).float().cpu().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().amp_antialias_snake(
                ctypes.c_void_p(x.data_ptr()), B, C, T, ctypes.c_void_p(alpha.data_ptr()),
                ctypes.c_void_p(beta.data_ptr()) if beta is not None else None, int(self.act.alpha_logscale),
                ctypes.c_void_p(fu.data_ptr()), ctypes.c_void_p(fd.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                _lib.current_stream_ptr(x.device)))
        return y
