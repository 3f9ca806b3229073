"""MI355X-native HiFi-GAN generators: same classes, constructor arguments, ``state_dict`` keys
and ``forward`` contract as models/vocoders/gan/generator/hifigan.py, with the whole forward
executed by hand-written gfx950 kernels through libamphion_hip.so.

    HiFiGAN(cfg).forward(x[B, n_mel, T])          -> [B, 1, T * prod(upsample_rates)]   (hifigan.py:151-219)
    HiFiGAN_vits(...).forward(x[B, C, T], g=None)  -> same                               (hifigan.py:376-449)

There is no CPU path: ``forward`` raises unless the module and its input are on a ROCm device.
"""
from __future__ import annotations

import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules.vocoder_blocks import get_padding

from ._engine import ConvParams, HipGenerator

LRELU_SLOPE = 0.1


class ResBlock1(nn.Module):
    """Parameter container of ResBlock1 (hifigan.py:17-106); executed inside the generator kernel chain."""

    def __init__(self, cfg, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.cfg = cfg
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation]
        )
        for c in self.convs1:  # self.convs1.apply(init_weights), hifigan.py:55
            _consume_init_normal(c)
        self.convs2 = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=1, padding=get_padding(kernel_size, 1)) for _ in dilation]
        )
        for c in self.convs2:  # hifigan.py:90
            _consume_init_normal(c)

    def remove_weight_norm(self):
        for l in self.convs1:
            l.remove_weight_norm()
        for l in self.convs2:
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    """Parameter container of ResBlock2 (hifigan.py:109-148)."""

    def __init__(self, cfg, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.cfg = cfg
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation]
        )
        for c in self.convs:  # hifigan.py:138
            _consume_init_normal(c)

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


def _consume_init_normal(conv: ConvParams):
    """``module.apply(init_weights)`` on a weight-normed conv draws N(0, 0.01) into the derived
    ``weight`` attribute only; mirror the RNG draw so seeded inits match the reference."""
    import torch

    if conv.has_weight_norm:
        torch.empty_like(conv.weight_v).normal_(0.0, 0.01)
    else:
        conv.weight.data.normal_(0.0, 0.01)


class ResBlock1_vits(ResBlock1):
    """hifigan.py:232-326 (same parameters, no cfg)."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__(None, channels, kernel_size, dilation)


class ResBlock2_vits(ResBlock2):
    """hifigan.py:330-373."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__(None, channels, kernel_size, dilation)


class HiFiGAN(HipGenerator):
    """Drop-in for models/vocoders/gan/generator/hifigan.py:151-229."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hp = cfg.model.hifigan
        self.num_kernels = len(hp.resblock_kernel_sizes)
        self.num_upsamples = len(hp.upsample_rates)
        c0 = hp.upsample_initial_channel
        self._amp_n_in = cfg.preprocess.n_mel
        self.conv_pre = ConvParams(cfg.preprocess.n_mel, c0, 7, padding=3)
        resblock = ResBlock1 if hp.resblock == "1" else ResBlock2

        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
            self.ups.append(
                ConvParams(c0 // (2**i), c0 // (2 ** (i + 1)), k, transposed=True, stride=u, padding=(k - u) // 2)
            )

        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for j, (k, d) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
                self.resblocks.append(resblock(self.cfg, ch, k, d))

        self.conv_post = ConvParams(ch, 1, 7, padding=3)
        for c in self.ups:  # self.ups.apply(init_weights); self.conv_post.apply(init_weights), hifigan.py:200-201
            _consume_init_normal(c)
        _consume_init_normal(self.conv_post)

    def _amp_desc(self):
        hp = self.cfg.model.hifigan
        return self._fill_desc(_lib.AMP_ARCH_HIFIGAN, self.cfg.preprocess.n_mel, hp.upsample_initial_channel,
                               hp.upsample_rates, hp.upsample_kernel_sizes, hp.resblock_kernel_sizes,
                               hp.resblock_dilation_sizes, hp.resblock)

    def forward(self, x):
        """hifigan.py:203-219, executed as one chain of gfx950 kernels."""
        return self._amp_forward(x)

    def remove_weight_norm(self):
        print("Removing weight norm...")
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()


class HiFiGAN_vits(HipGenerator):
    """Drop-in for models/vocoders/gan/generator/hifigan.py:376-449 (the VITS decoder)."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self._amp_n_in = initial_channel
        self._hp = dict(resblock=resblock, resblock_kernel_sizes=list(resblock_kernel_sizes),
                        resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
                        upsample_rates=list(upsample_rates), upsample_initial_channel=upsample_initial_channel,
                        upsample_kernel_sizes=list(upsample_kernel_sizes), gin_channels=gin_channels)
        c0 = upsample_initial_channel
        self.conv_pre = ConvParams(initial_channel, c0, 7, padding=3, weight_norm=False)
        rb = ResBlock1_vits if resblock == "1" else ResBlock2_vits

        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(
                ConvParams(c0 // (2**i), c0 // (2 ** (i + 1)), k, transposed=True, stride=u, padding=(k - u) // 2)
            )

        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for j, (k, d) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
                self.resblocks.append(rb(ch, k, d))

        self.conv_post = ConvParams(ch, 1, 7, padding=3, weight_norm=False, bias=False)
        for c in self.ups:  # hifigan.py:419
            _consume_init_normal(c)
        if gin_channels != 0:
            self.cond = ConvParams(gin_channels, c0, 1, weight_norm=False)

    def _amp_desc(self):
        hp = self._hp
        return self._fill_desc(_lib.AMP_ARCH_HIFIGAN_VITS, self._amp_n_in, hp["upsample_initial_channel"],
                               hp["upsample_rates"], hp["upsample_kernel_sizes"], hp["resblock_kernel_sizes"],
                               hp["resblock_dilation_sizes"], hp["resblock"], gin=hp["gin_channels"])

    def forward(self, x, g=None):
        """hifigan.py:424-443.  ``g``: optional [B, gin_channels, 1] speaker embedding."""
        if g is not None and not hasattr(self, "cond"):
            raise AttributeError("'HiFiGAN_vits' object has no attribute 'cond'")  # as the reference would
        return self._amp_forward(x, g)

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
