"""MI355X-native WN / ResidualCouplingLayer / Flip (modules/flow/modules.py:74-151,314-397): same
constructor arguments and parameter names, forward executed by gfx950 kernels (fused implicit-GEMM
convs + the element-wise kernels of amp_wn_gate / amp_wn_accumulate / amp_coupling_apply).

``x_mask`` is represented by the valid lengths (the reference builds it with
``sequence_mask(lengths)``, vits.py:146): pass ``x_lengths`` instead of the dense mask.
"""
from __future__ import annotations

import os

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
        # False: always the four unfused launches per layer -- cross-check / A-B switch (tests set it)
        self.fused = True         # two launches per layer (amp_wn_forward); tests set False for the four unfused ops
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

    def forward(self, x, x_lengths=None, g=None, owns_input=False, input_masked=True, mask_output=True, **kwargs):
        """x [B, H, T] (not modified) -> output [B, H, T] = sum of skips * mask (modules.py:126-151).
        Launch savers for a caller that owns the tensors on both sides (ResidualCouplingLayer): ``owns_input`` lets x be used as
        the working buffer (it is overwritten), ``input_masked=False`` says x is not zero beyond the lengths (the fused in-layer
        kernels mask it at staging anyway), ``mask_output=False`` leaves the output beyond the lengths UNSPECIFIED."""
        x = _lib.require_device_tensor(x, "WN input")
        if not owns_input:
            x = x.clone()
        B, H, T = x.shape
        lens = hip_ops.lens_tensor(x_lengths, x.device)
        output = torch.empty_like(x)
        cond = None
        if g is not None:
            cond = self.cond_layer(_lib.require_device_tensor(g, "g"))  # [B, 2H*n_layers, 1]
        acts = torch.empty_like(x)
        # fused path (f16x3): in_layers[i] + gate in one launch, res_skip_layers[i] + residual / skip update in another
        if self.fused and hip_ops.wn_fused(self.in_layers, self.res_skip_layers, x, cond, lens, output, acts):
            if lens is not None and mask_output:
                hip_ops.sequence_mask_(output, lens)
            return output
        if lens is not None and not input_masked:
            hip_ops.sequence_mask_(x, lens)             # the unfused ops read x densely
        x_in = torch.empty((B, 2 * H, T), dtype=torch.float32, device=x.device)
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

    def forward(self, x, x_lengths=None, g=None, reverse=False, owns_input=False):
        """``owns_input``: x is a private tensor of the caller (the output of the preceding Flip) and is updated in place."""
        x = _lib.require_device_tensor(x, "coupling input")
        if not owns_input:
            x = x.clone()
        B, C, T = x.shape
        lens = hip_ops.lens_tensor(x_lengths, x.device)
        # Round 4: the three `* x_mask` of :381-388 are taken by the kernels instead of three launches -- pre / post skip the tiles
        # beyond an utterance's end, WN's in-layers mask h at staging, and the coupling update selects on t < len.
        h = self.pre(x, x_batch_stride=C * T, T=T, lens=lens)  # x0 = x[:, :half]
        h = self.enc(h, lens, g=g, owns_input=True, input_masked=False, mask_output=False)
        m = self.post(h, lens=lens)
        hip_ops.coupling_apply_(x, m, lens, reverse)
        if not reverse:
            return x, torch.zeros(B, dtype=x.dtype, device=x.device)  # logdet = sum(logs) = 0 (mean only)
        return x


# ---- the pieces of the stochastic duration predictor ---------------------------
class DepthwiseConv1d(nn.Module):
    """Parameters of nn.Conv1d(C, C, K, groups=C, dilation=d, padding=(K*d - d)//2) under torch's key names; forward =
    ``amp_dwconv`` on x * mask."""

    def __init__(self, channels, kernel_size, dilation):
        super().__init__()
        ref = nn.Conv1d(channels, channels, kernel_size, groups=channels, dilation=dilation, padding=(kernel_size * dilation - dilation) // 2)
        self.weight = nn.Parameter(ref.weight.data.clone())   # [C, 1, K]
        self.bias = nn.Parameter(ref.bias.data.clone())
        self.dilation = dilation

    def forward(self, x, lens=None):
        return hip_ops.dwconv(x, self.weight.detach().contiguous(), self.bias.detach().contiguous(), lens, self.dilation)


class DDSConv(nn.Module):
    """Dilated and depth-separable convolution, modules/flow/modules.py:25-72."""

    def __init__(self, channels, kernel_size, n_layers, p_dropout=0.0):
        super().__init__()
        from amphion_amd.modules.base import LayerNorm

        self.channels, self.kernel_size, self.n_layers, self.p_dropout = channels, kernel_size, n_layers, p_dropout
        self.convs_sep, self.convs_1x1 = nn.ModuleList(), nn.ModuleList()
        self.norms_1, self.norms_2 = nn.ModuleList(), nn.ModuleList()
        for i in range(n_layers):
            self.convs_sep.append(DepthwiseConv1d(channels, kernel_size, kernel_size**i))
            self.convs_1x1.append(HipConv1d(channels, channels, 1, weight_norm=False))
            self.norms_1.append(LayerNorm(channels))
            self.norms_2.append(LayerNorm(channels))

    def forward(self, x, lens=None, mask_output=True):
        """:63-72 in eval mode (the optional ``x + g`` of :61-62 is fused into the conv that produces x by the callers).
        ``mask_output=False``: the caller masks (ConvFlow's fused projection + spline)."""
        x = _lib.require_device_tensor(x, "DDSConv input")
        z = None                                           # gelu(norm_1(conv_sep(x * mask))) of the NEXT layer when the seam kernel made it
        for i in range(self.n_layers):
            sep, n1 = self.convs_sep[i], self.norms_1[i]
            if z is not None:
                y = z
            elif sep.weight.shape[-1] == 3:                # conv(x * mask) -> norm -> gelu in one launch
                y = hip_ops.dwconv_layer_norm_c(x, sep.weight.detach().contiguous(), sep.bias.detach().contiguous(), sep.dilation,
                                                n1.gamma.detach(), n1.beta.detach(), lens=lens, eps=n1.eps, gelu=True)
            else:
                y = n1(sep(x, lens), gelu=True)
            y = self.convs_1x1[i](y)
            z = None
            if i + 1 < self.n_layers:                      # x + gelu(norm_2(y)) AND the next layer's conv_sep -> norm_1 -> gelu: one launch
                r = hip_ops.dds_seam(y, x, self.norms_2[i], self.convs_sep[i + 1], self.norms_1[i + 1], lens)
                if r is not None:
                    x, z = r
                    continue
            x = self.norms_2[i](y, gelu=True, post=x)      # x + gelu(norm(y))
        return hip_ops.sequence_mask_(x, lens) if (lens is not None and mask_output) else x


class Log(nn.Module):
    """modules/flow/modules.py:304-312 -- used only by the training direction of the duration predictor."""

    def forward(self, x, x_mask, reverse=False, **kwargs):
        raise NotImplementedError("Log flow: training direction only (inference-only kernels)")


class ElementwiseAffine(nn.Module):
    """modules/flow/modules.py:325-340 (reverse direction on the GPU)."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.m = nn.Parameter(torch.zeros(channels, 1))
        self.logs = nn.Parameter(torch.zeros(channels, 1))

    def forward(self, x, lens=None, reverse=False, **kwargs):
        if not reverse:
            raise NotImplementedError("ElementwiseAffine forward: training direction only")
        return hip_ops.affine_reverse(_lib.require_device_tensor(x, "ElementwiseAffine input"), self.m.detach().reshape(-1).contiguous(),
                                      self.logs.detach().reshape(-1).contiguous(), lens)


class ConvFlow(nn.Module):
    """modules/flow/modules.py:400-458 for in_channels = 2 (one conditioning, one transformed channel)."""

    def __init__(self, in_channels, filter_channels, kernel_size, n_layers, num_bins=10, tail_bound=5.0):
        super().__init__()
        if in_channels != 2:
            raise NotImplementedError("the HIP spline flow covers the duration predictor's 2-channel flows")
        self.in_channels, self.filter_channels, self.kernel_size, self.n_layers = in_channels, filter_channels, kernel_size, n_layers
        self.num_bins, self.tail_bound, self.half_channels = num_bins, tail_bound, in_channels // 2
        self.pre = HipConv1d(self.half_channels, filter_channels, 1, weight_norm=False)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers, p_dropout=0.0)
        self.proj = HipConv1d(filter_channels, self.half_channels * (num_bins * 3 - 1), 1, weight_norm=False)
        self.proj.weight.data.zero_()   # :419-420
        self.proj.bias.data.zero_()

    def forward(self, x, lens=None, g=None, reverse=False, flip_in=False, flip_out=False):
        """x [B, 2, T].  ``flip_in`` / ``flip_out`` fold the Flip layers before / after this flow into the spline kernel."""
        x = _lib.require_device_tensor(x, "ConvFlow input")
        B, _, T = x.shape
        # x0 = the conditioning channel: channel 0, or channel 1 when the preceding Flip is folded in
        x0 = x[:, 1:2] if flip_in else x[:, 0:1]
        h = self.pre(x0, res=g, x_batch_stride=x.stride(0), T=T)   # pre(x0) + g: DDSConv's "x + g" (:61-62) fused into the conv; x0 read in place
        pj = self.proj
        if self.convs.channels <= 256 and pj.k == 1 and pj.bias is not None:
            # proj + `* x_mask` + spline in one launch (the mask is a select on the lengths there, so DDSConv's own final mask launch
            # is not needed either)
            h = self.convs(h, lens, mask_output=False)
            return hip_ops.spline_flow_proj(x, h, pj.folded_weight().detach().reshape(pj.cout, pj.cin).contiguous(), pj.bias.detach(), lens,
                                            self.num_bins, self.filter_channels, self.tail_bound, inverse=reverse, flip_in=flip_in,
                                            flip_out=flip_out)
        h = self.convs(h, lens)
        h = self.proj(h)                                    # masked inside the spline kernel (h * x_mask, :428)
        return hip_ops.spline_flow(x, h, lens, self.num_bins, self.filter_channels, self.tail_bound, inverse=reverse,
                                   flip_in=flip_in, flip_out=flip_out)
