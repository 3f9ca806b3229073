"""LayerNorm over the channel axis of [B, C, T] (modules/base/base_module.py:11-24), same parameter names
(``gamma``, ``beta``).  HIP kernel: amp_layer_norm_c (vits_text.hip)."""
import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules import hip_ops


class LayerNorm(nn.Module):
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x, res=None, gelu=False, post=None, lens=None):
        """post + act(LN(x + res)): the fusions its callers need (Encoder: norm(x + y); DDSConv: x + gelu(norm(y))).  ``lens``:
        the result is zero beyond each item's length, whatever x / res hold there."""
        x = _lib.require_device_tensor(x, "LayerNorm input")
        return hip_ops.layer_norm_c(x, self.gamma.detach(), self.beta.detach(), res=res, post=post, eps=self.eps, gelu=gelu, lens=lens)
