"""Encoder / MultiHeadAttention / FFN of the VITS text encoder (modules/transformer/attentions.py:16-77,165-358,361-417):
same constructor arguments and parameter names; 1x1 / k-tap convs on the implicit-GEMM kernel, attention with
windowed relative-position embeddings in ``amp_rel_attention``.  Masks are the valid lengths (int32 [B] on the device).
GPU parity: tests/test_gpu_vits_infer.py."""
from __future__ import annotations

import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules import hip_ops
from amphion_amd.modules.base import LayerNorm
from amphion_amd.modules.hip_ops import HipConv1d


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=None, heads_share=True, block_length=None,
                 proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        if window_size is None or not heads_share or block_length is not None or proximal_bias:
            raise NotImplementedError("the HIP attention covers VITS' setup: relative window, shared heads, no block / proximal bias")
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.window_size = window_size
        self.k_channels = channels // n_heads
        self.conv_q = HipConv1d(channels, channels, 1, weight_norm=False)
        self.conv_k = HipConv1d(channels, channels, 1, weight_norm=False)
        self.conv_v = HipConv1d(channels, channels, 1, weight_norm=False)
        self.conv_o = HipConv1d(channels, out_channels, 1, weight_norm=False)
        rel_stddev = self.k_channels**-0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)
        if proximal_init:
            with torch.no_grad():
                self.conv_k.weight.copy_(self.conv_q.weight)
                self.conv_k.bias.copy_(self.conv_q.bias)
        # the three projections read the same x: one launch over their stacked rows (parameters stay under conv_q / conv_k / conv_v)
        self._qkv = hip_ops.MergedConv1d()

    def forward(self, x, c, lens=None):
        """self-attention only (c is x): attentions.py:222-230"""
        if c is not x:
            raise NotImplementedError("relative attention is self-attention (attentions.py:241-243)")
        qkv = self._qkv((self.conv_q, self.conv_k, self.conv_v), x)    # [B, 3C, T]: q | k | v, read in place by the attention kernel
        o = hip_ops.rel_attention_qkv(qkv, self.emb_rel_k.detach()[0].contiguous(), self.emb_rel_v.detach()[0].contiguous(), lens,
                                      self.n_heads, self.window_size)
        return self.conv_o(o)


class FFN(nn.Module):
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None, causal=False):
        super().__init__()
        if activation is not None or causal or kernel_size % 2 == 0:
            raise NotImplementedError("the HIP FFN covers VITS' setup: relu, 'same' padding, odd kernel")
        self.kernel_size = kernel_size
        pad = (kernel_size - 1) // 2
        self.conv_1 = HipConv1d(in_channels, filter_channels, kernel_size, padding=pad, weight_norm=False)
        self.conv_2 = HipConv1d(filter_channels, out_channels, kernel_size, padding=pad, weight_norm=False)

    def forward(self, x, lens=None, mask_output=True):
        """conv_2(relu(conv_1(x * mask)) * mask) * mask   (attentions.py:392-400); relu = leaky_relu(0) on store.  The two inner
        masks are taken by the convs (``lens``: input columns beyond an item's length count as zero); ``mask_output=False`` leaves
        the columns of the result beyond each length UNSPECIFIED for a caller that masks them itself (the Encoder's LayerNorm)."""
        if lens is None:
            return self.conv_2(self.conv_1(x, slope_out=0.0))
        h = self.conv_1(x, slope_out=0.0, lens=lens)
        y = self.conv_2(h, lens=lens)
        return hip_ops.sequence_mask_(y, lens) if mask_output else y


class Encoder(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0, window_size=4, **kwargs):
        super().__init__()
        self.hidden_channels, self.n_layers = hidden_channels, n_layers
        self.attn_layers, self.norm_layers_1 = nn.ModuleList(), nn.ModuleList()
        self.ffn_layers, self.norm_layers_2 = nn.ModuleList(), nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout, window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, lens=None):
        """attentions.py:64-76 in eval mode (dropout = identity)"""
        x = _lib.require_device_tensor(x, "Encoder input").clone()
        if lens is not None:
            hip_ops.sequence_mask_(x, lens)
        # Round 4: with ``lens`` each LayerNorm writes zero beyond an item's length, so x stays masked throughout (the reference's
        # x is unmasked between layers, but no valid column ever reads those: the attention masks keys, the FFN convs their input)
        # and the `* x_mask` launches of the FFN and of the end (:76) are gone -- 19 launches fewer per call at 6 layers.
        for i in range(self.n_layers):
            y = self.attn_layers[i](x, x, lens)
            x = self.norm_layers_1[i](x, res=y, lens=lens)
            y = self.ffn_layers[i](x, lens, mask_output=False)
            x = self.norm_layers_2[i](x, res=y, lens=lens)
        return x
