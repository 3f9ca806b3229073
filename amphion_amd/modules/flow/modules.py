"""MI355X-native WN / ResidualCouplingLayer / Flip (modules/flow/modules.py:74-151,314-397): same
constructor arguments and parameter names, forward executed by gfx950 kernels (fused implicit-GEMM
convs + the element-wise kernels of amp_wn_gate / amp_wn_accumulate / amp_coupling_apply).

``x_mask`` is represented by the valid lengths (the reference builds it with
``sequence_mask(lengths)``, vits.py:146): pass ``x_lengths`` instead of the dense mask.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules import hip_ops
from amphion_amd.modules.hip_ops import HipConv1d


class WN(nn.Module):
    """modules/flow/modules.py:74-158"""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        if p_dropout != 0:
            raise NotImplementedError("inference path: dropout must be 0 (the reference uses p_dropout=0 here)")
        self.hidden_channels = hidden_channels
        self.kernel_size = (kernel_size,)
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.p_dropout = p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = HipConv1d(gin_channels, 2 * hidden_channels * n_layers, 1)
        for i in range(n_layers):
            dilation = dilation_rate**i
            padding = int((kernel_size * dilation - dilation) / 2)
            self.in_layers.append(HipConv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation,
                                            padding=padding))
            rs_ch = 2 * hidden_channels if i < n_layers - 1 else hidden_channels  # last one is not necessary
            self.res_skip_layers.append(HipConv1d(hidden_channels, rs_ch, 1))

    def forward(self, x, x_lengths=None, g=None, **kwargs):
        """x [B, H, T] (not modified) -> output [B, H, T] = sum of skips * mask (modules.py:126-151)."""
        x = _lib.require_device_tensor(x, "WN input").clone()
        B, H, T = x.shape
        lens = hip_ops.lens_tensor(x_lengths, x.device)
        output = torch.empty_like(x)
        cond = None
        if g is not None:
            cond = self.cond_layer(_lib.require_device_tensor(g, "g"))  # [B, 2H*n_layers, 1]
        x_in = torch.empty((B, 2 * H, T), dtype=torch.float32, device=x.device)
        acts = torch.empty_like(x)
        for i in range(self.n_layers):
            self.in_layers[i](x, out=x_in)
            g_l = cond[:, i * 2 * H:, 0] if cond is not None else None
            hip_ops.wn_gate(x_in, g_l, acts)
            rs = self.res_skip_layers[i](acts)
            hip_ops.wn_accumulate(x, output, rs, lens, first=(i == 0), last=(i == self.n_layers - 1))
        if lens is not None:
            hip_ops.sequence_mask_(output, lens)
        return output

    def remove_weight_norm(self):
        if self.gin_channels != 0:
            self.cond_layer.remove_weight_norm()
        for l in self.in_layers:
            l.remove_weight_norm()
        for l in self.res_skip_layers:
            l.remove_weight_norm()


class Flip(nn.Module):
    """modules/flow/modules.py:314-321"""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = hip_ops.flip_channels(_lib.require_device_tensor(x, "Flip input"))
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return x


class ResidualCouplingLayer(nn.Module):
    """modules/flow/modules.py:340-397 (mean_only=True as built by ResidualCouplingBlock, vits.py:100)."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        if not mean_only:
            raise NotImplementedError("only the mean_only=True layers of VITS are on the hot path")
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only
        self.pre = HipConv1d(self.half_channels, hidden_channels, 1, weight_norm=False)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout, gin_channels=gin_channels)
        self.post = HipConv1d(hidden_channels, self.half_channels * (2 - mean_only), 1, weight_norm=False)
        self.post.weight.data.zero_()  # modules.py:375-376
        self.post.bias.data.zero_()

    def forward(self, x, x_lengths=None, g=None, reverse=False):
        x = _lib.require_device_tensor(x, "coupling input").clone()
        B, C, T = x.shape
        lens = hip_ops.lens_tensor(x_lengths, x.device)
        h = self.pre(x, x_batch_stride=C * T, T=T)  # x0 = x[:, :half]
        if lens is not None:
            hip_ops.sequence_mask_(h, lens)
        h = self.enc(h, x_lengths, g=g)
        m = self.post(h)
        if lens is not None:
            hip_ops.sequence_mask_(m, lens)
        hip_ops.coupling_apply_(x, m, lens, reverse)
        if not reverse:
            return x, torch.zeros(B, dtype=x.dtype, device=x.device)  # logdet = sum(logs) = 0 (mean only)
        return x
