"""Snake / SnakeBeta (modules/activation_functions/snake.py:13-122): same constructor arguments, parameter
names and init as the reference.  Inside BigVGAN the activation runs fused into the anti-aliased Activation1d HIP
kernel; ``forward`` on its own is the element-wise HIP kernel ``amp_snake`` (ROCm tensors only, no CPU fallback).
"""
import ctypes

import torch
from torch import nn
from torch.nn import Parameter

from amphion_amd import _lib


def _snake_forward(mod, x):
    x = _lib.require_device_tensor(x, type(mod).__name__ + " input")
    B, C, T = x.shape
    alpha = mod.alpha.detach().float().contiguous()
    beta = mod.beta.detach().float().contiguous() if mod.has_beta else None
    if alpha.device != x.device:
        raise RuntimeError(f"{type(mod).__name__} parameters are on {alpha.device}, input on {x.device}")
    if alpha.numel() != C:
        raise ValueError(f"{type(mod).__name__}({alpha.numel()}) applied to {C} channels")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().amp_snake(
            ctypes.c_void_p(x.data_ptr()), B, C, T, ctypes.c_void_p(alpha.data_ptr()),
            ctypes.c_void_p(beta.data_ptr()) if beta is not None else None, int(mod.alpha_logscale),
            ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(x.device)))
    return y


class Snake(nn.Module):
    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        if self.alpha_logscale:  # log scale alphas initialized to zeros (snake.py:44-45)
            self.alpha = Parameter(torch.zeros(in_features) * alpha)
        else:
            self.alpha = Parameter(torch.ones(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    has_beta = False

    def forward(self, x):  # snake.py:51-61   x + sin^2(a x) / a
        return _snake_forward(self, x)


class SnakeBeta(nn.Module):
    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        if self.alpha_logscale:  # snake.py:97-99
            self.alpha = Parameter(torch.zeros(in_features) * alpha)
            self.beta = Parameter(torch.zeros(in_features) * alpha)
        else:
            self.alpha = Parameter(torch.ones(in_features) * alpha)
            self.beta = Parameter(torch.ones(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    has_beta = True

    def forward(self, x):  # snake.py:110-122   x + sin^2(a x) / b
        return _snake_forward(self, x)
