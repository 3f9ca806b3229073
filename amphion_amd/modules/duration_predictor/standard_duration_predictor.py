"""DurationPredictor (modules/duration_predictor/standard_duration_predictor.py:13-61), same arguments and parameter
names."""
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules import hip_ops
from amphion_amd.modules.base import LayerNorm
from amphion_amd.modules.hip_ops import HipConv1d


class DurationPredictor(nn.Module):
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, gin_channels=0):
        super().__init__()
        self.in_channels, self.filter_channels, self.kernel_size = in_channels, filter_channels, kernel_size
        self.p_dropout, self.gin_channels = p_dropout, gin_channels
        self.conv_1 = HipConv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2, weight_norm=False)
        self.norm_1 = LayerNorm(filter_channels)
        self.conv_2 = HipConv1d(filter_channels, filter_channels, kernel_size, padding=kernel_size // 2, weight_norm=False)
        self.norm_2 = LayerNorm(filter_channels)
        self.proj = HipConv1d(filter_channels, 1, 1, weight_norm=False)
        if gin_channels != 0:
            self.cond = HipConv1d(gin_channels, in_channels, 1, weight_norm=False)

    def forward(self, x, lens=None, g=None):
        """:40-61 in eval mode -> logw [B, 1, T]"""
        x = _lib.require_device_tensor(x, "DurationPredictor input")
        x = x.clone()
        if g is not None:
            hip_ops.add_channel_bias_(x, self.cond(_lib.require_device_tensor(g, "g")))   # [B, C, 1] broadcast over time
        if lens is not None:
            hip_ops.sequence_mask_(x, lens)
        x = self.norm_1(self.conv_1(x, slope_out=0.0))             # relu on store, then LayerNorm
        if lens is not None:
            hip_ops.sequence_mask_(x, lens)
        x = self.norm_2(self.conv_2(x, slope_out=0.0))
        if lens is not None:
            hip_ops.sequence_mask_(x, lens)
        x = self.proj(x)
        return hip_ops.sequence_mask_(x, lens) if lens is not None else x
