"""MelGAN generator (models/vocoders/gan/generator/melgan.py:34-100) on the gfx950 conv kernels.

Same constructor (``MelGAN(cfg)`` reading ``cfg.preprocess.n_mel`` and ``cfg.model.melgan.{ratios, ngf,
n_residual_layers}``), same ``nn.Sequential`` layout -- so ``state_dict`` keys are the reference's
``model.<idx>.{bias, weight_g, weight_v}``, ``model.<idx>.block.{2,4}.*``, ``model.<idx>.shortcut.*`` -- and the
same forward: every element-wise layer of the Sequential is folded into the conv that follows it:

    ReflectionPad1d(p) + Conv1d(padding=0)   -> one conv with mirrored out-of-range reads (AMP_CONV_OPT_PAD_REFLECT)
    LeakyReLU(0.2) before a conv             -> leaky-ReLU-on-load
    shortcut(x) + block(x)                   -> the block's last 1x1 conv accumulates onto the shortcut's output
    Tanh                                     -> tanh-on-store of the last conv
"""
from __future__ import annotations

import numpy as np
import torch.nn as nn

from amphion_amd.modules.hip_ops import HipConv1d

LRELU = 0.2  # melgan.py:38,41,69,91


class _Folded(nn.Module):
    """Placeholder for a parameter-free layer of the reference Sequential (keeps the indices, and with them
    the state_dict keys, identical); its arithmetic runs inside the neighbouring conv kernel."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return f"{self.what} (fused into the adjacent conv kernel)"


class ResnetBlock(nn.Module):
    """melgan.py:34-48: shortcut(x) + conv1x1(lrelu(conv3_dilated(reflect_pad(lrelu(x)))))."""

    def __init__(self, dim, dilation=1):
        super().__init__()
        self.block = nn.Sequential(
            _Folded("LeakyReLU(0.2)"),
            _Folded(f"ReflectionPad1d({dilation})"),
            HipConv1d(dim, dim, 3, dilation=dilation, padding=dilation, pad_mode="reflect"),
            _Folded("LeakyReLU(0.2)"),
            HipConv1d(dim, dim, 1),
        )
        self.shortcut = HipConv1d(dim, dim, 1)

    def forward(self, x):
        s = self.shortcut(x)
        t = self.block[2](x, slope_in=LRELU)
        return self.block[4](t, slope_in=LRELU, res=s, out=s)   # in place on the shortcut's output


class MelGAN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hp = cfg.model.melgan
        ratios = list(hp.ratios)
        self.hop_length = int(np.prod(ratios))
        mult = int(2 ** len(ratios))
        model = [_Folded("ReflectionPad1d(3)"),
                 HipConv1d(cfg.preprocess.n_mel, mult * hp.ngf, 7, padding=3, pad_mode="reflect")]
        for r in ratios:
            if r % 2:
                raise NotImplementedError("odd upsampling ratios need output_padding (melgan.py:75-76); the "
                                          "polyphase transposed-conv kernel covers kernel - stride even")
            model += [_Folded("LeakyReLU(0.2)"),
                      HipConv1d(mult * hp.ngf, mult * hp.ngf // 2, 2 * r, transposed=True, stride=r, padding=r // 2)]
            for j in range(hp.n_residual_layers):
                model += [ResnetBlock(mult * hp.ngf // 2, dilation=3**j)]
            mult //= 2
        model += [_Folded("LeakyReLU(0.2)"), _Folded("ReflectionPad1d(3)"),
                  HipConv1d(hp.ngf, 1, 7, padding=3, pad_mode="reflect", tanh=True), _Folded("Tanh")]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        """melgan.py:99-100: mel [B, n_mel, T] -> waveform [B, 1, T * prod(ratios)]."""
        pending_lrelu = False
        for layer in self.model:
            if isinstance(layer, _Folded):
                pending_lrelu = pending_lrelu or layer.what.startswith("LeakyReLU")
            elif isinstance(layer, ResnetBlock):
                x = layer(x)
            else:
                x = layer(x, slope_in=LRELU if pending_lrelu else 1.0)
                pending_lrelu = False
        return x
