"""MI355X-native BigVGAN generator: drop-in for models/vocoders/gan/generator/bigvgan.py:232-341
(same constructor, ``state_dict`` keys incl. Snake parameters and the persistent anti-aliasing
filter buffers, same ``forward`` contract), executed by gfx950 kernels through libamphion_hip.so."""
from __future__ import annotations

import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules.activation_functions import Snake, SnakeBeta
from amphion_amd.modules.anti_aliasing import Activation1d
from amphion_amd.modules.vocoder_blocks import get_padding

from ._engine import ConvParams, HipGenerator
from .hifigan import _consume_init_normal

LRELU_SLOPE = 0.1


def _make_activations(cfg, channels, n, activation):
    if activation == "snake":
        return nn.ModuleList([Activation1d(activation=Snake(channels, alpha_logscale=cfg.model.bigvgan.snake_logscale))
                              for _ in range(n)])
    if activation == "snakebeta":
        return nn.ModuleList([Activation1d(activation=SnakeBeta(channels, alpha_logscale=cfg.model.bigvgan.snake_logscale))
                              for _ in range(n)])
    raise NotImplementedError(  # bigvgan.py:132-135
        "activation incorrectly specified. check the config file and look for 'activation'."
    )


class AMPBlock1(nn.Module):
    """Parameter container of AMPBlock1 (bigvgan.py:23-152)."""

    def __init__(self, cfg, channels, kernel_size=3, dilation=(1, 3, 5), activation=None):
        super().__init__()
        self.cfg = cfg
        self.convs1 = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation]
        )
        for c in self.convs1:
            _consume_init_normal(c)
        self.convs2 = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=1, padding=get_padding(kernel_size, 1)) for _ in dilation]
        )
        for c in self.convs2:
            _consume_init_normal(c)
        self.num_layers = len(self.convs1) + len(self.convs2)
        self.activations = _make_activations(cfg, channels, self.num_layers, activation)

    def remove_weight_norm(self):
        for l in self.convs1:
            l.remove_weight_norm()
        for l in self.convs2:
            l.remove_weight_norm()


class AMPBlock2(nn.Module):
    """Parameter container of AMPBlock2 (bigvgan.py:155-229)."""

    def __init__(self, cfg, channels, kernel_size=3, dilation=(1, 3), activation=None):
        super().__init__()
        self.cfg = cfg
        self.convs = nn.ModuleList(
            [ConvParams(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation]
        )
        for c in self.convs:
            _consume_init_normal(c)
        self.num_layers = len(self.convs)
        self.activations = _make_activations(cfg, channels, self.num_layers, activation)

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class BigVGAN(HipGenerator):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hp = cfg.model.bigvgan
        self.num_kernels = len(hp.resblock_kernel_sizes)
        self.num_upsamples = len(hp.upsample_rates)
        c0 = hp.upsample_initial_channel
        self._amp_n_in = cfg.preprocess.n_mel
        self.conv_pre = ConvParams(cfg.preprocess.n_mel, c0, 7, padding=3)
        resblock = AMPBlock1 if hp.resblock == "1" else AMPBlock2

        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
            self.ups.append(nn.ModuleList([
                ConvParams(c0 // (2**i), c0 // (2 ** (i + 1)), k, transposed=True, stride=u, padding=(k - u) // 2)
            ]))

        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for j, (k, d) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
                self.resblocks.append(resblock(cfg, ch, k, d, activation=hp.activation))

        if hp.activation == "snake":
            self.activation_post = Activation1d(activation=Snake(ch, alpha_logscale=hp.snake_logscale))
        elif hp.activation == "snakebeta":
            self.activation_post = Activation1d(activation=SnakeBeta(ch, alpha_logscale=hp.snake_logscale))
        else:
            raise NotImplementedError(  # bigvgan.py:300-303
                "activation incorrectly specified. check the config file and look for 'activation'."
            )
        self.conv_post = ConvParams(ch, 1, 7, padding=3)
        for l in self.ups:  # bigvgan.py:308-311
            for c in l:
                _consume_init_normal(c)
        _consume_init_normal(self.conv_post)

    def _amp_desc(self):
        hp = self.cfg.model.bigvgan
        act = _lib.AMP_ACT_SNAKE if hp.activation == "snake" else _lib.AMP_ACT_SNAKEBETA
        return self._fill_desc(_lib.AMP_ARCH_BIGVGAN, self.cfg.preprocess.n_mel, hp.upsample_initial_channel,
                               hp.upsample_rates, hp.upsample_kernel_sizes, hp.resblock_kernel_sizes,
                               hp.resblock_dilation_sizes, hp.resblock, activation=act, logscale=hp.snake_logscale)

    def forward(self, x):
        """bigvgan.py:313-331."""
        return self._amp_forward(x)

    def remove_weight_norm(self):
        print("Removing weight norm...")
        for l in self.ups:
            for l_i in l:
                l_i.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
