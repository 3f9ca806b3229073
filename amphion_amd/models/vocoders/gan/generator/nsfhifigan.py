"""NSF-HiFiGAN drop-in (models/vocoders/gan/generator/nsfhifigan.py:181-283) on the HiFi-GAN kernel chain.

Same constructor (``NSFHiFiGAN(cfg)`` reading ``cfg.model.nsfhifigan.*``, ``cfg.preprocess.{n_mel, sample_rate}``),
same ``state_dict`` keys (including ``m_source.l_linear.*`` and ``noise_convs.*``) and ``forward(x, f0)``.

What the reference computes.  Its forward builds the harmonic source and the per-stage ``noise_convs`` outputs,
but then overwrites them (nsfhifigan.py:266-271):

    x_source = self.noise_convs[i](har_source)
    length = min(x.shape[-1], x_source.shape[-1])
    x = x[:, :, :length]
    x_source = x[:, :, :length]        # <- the source is replaced by x itself
    x = x + x_source                   # == 2 * x

so the output does not depend on ``f0``, on the SineGen noise or on the ``noise_convs`` / ``m_source``
parameters: it is HiFi-GAN with the output of every transposed conv doubled (both lengths are ``T * prod(rates so
far)``, so the crop is a no-op).  This drop-in reproduces exactly that (SURVEY.md §8 f.2 asks for the quirk to be
kept): the doubling is folded into the transposed convs (weight_g and bias x2: exact), the source branch is not
executed, its parameters are kept for checkpoint compatibility.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from amphion_amd import _lib

from ._engine import ConvParams, HipGenerator
from .hifigan import ResBlock1, ResBlock2, _consume_init_normal


class SineGen(nn.Module):
    """Parameter-free holder (modules/neural_source_filter/sine_excitation.py:18-30); never executed, see above."""

    def __init__(self, fs, harmonic_num=0, amp=0.1, noise_std=0.003, voiced_threshold=0):
        super().__init__()
        self.amp, self.noise_std, self.harmonic_num = amp, noise_std, harmonic_num
        self.dim, self.fs, self.voice_threshold = harmonic_num + 1, fs, voiced_threshold


class SourceModuleHnNSF(nn.Module):
    """nsfhifigan.py:160-178: ``l_linear`` is a real parameter (kept in the state_dict)."""

    def __init__(self, fs, harmonic_num=0, amp=0.1, noise_std=0.003, voiced_threshold=0):
        super().__init__()
        self.amp, self.noise_std = amp, noise_std
        self.l_sin_gen = SineGen(fs, harmonic_num, amp, noise_std, voiced_threshold)
        self.l_linear = nn.Linear(harmonic_num + 1, 1)
        self.l_tanh = nn.Tanh()


class NSFHiFiGAN(HipGenerator):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hp = cfg.model.nsfhifigan
        self.num_kernels = len(hp.resblock_kernel_sizes)
        self.num_upsamples = len(hp.upsample_rates)
        self.m_source = SourceModuleHnNSF(fs=cfg.preprocess.sample_rate, harmonic_num=hp.harmonic_num)
        self.noise_convs = nn.ModuleList()
        c0 = hp.upsample_initial_channel
        self._amp_n_in = cfg.preprocess.n_mel
        self.conv_pre = ConvParams(cfg.preprocess.n_mel, c0, 7, padding=3)
        resblock = ResBlock1 if hp.resblock == "1" else ResBlock2
        self.ups = nn.ModuleList()
        rates = list(hp.upsample_rates)
        for i, (u, k) in enumerate(zip(rates, hp.upsample_kernel_sizes)):
            c_cur = c0 // (2 ** (i + 1))
            self.ups.append(ConvParams(c0 // (2**i), c_cur, k, transposed=True, stride=u, padding=(k - u) // 2))
            if i + 1 < len(rates):
                stride_f0 = int(np.prod(rates[i + 1:]))
                self.noise_convs.append(nn.Conv1d(1, c_cur, kernel_size=stride_f0 * 2, stride=stride_f0, padding=stride_f0 // 2))
            else:
                self.noise_convs.append(nn.Conv1d(1, c_cur, kernel_size=1))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch //= 2
            for k, d in zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes):
                self.resblocks.append(resblock(cfg, ch, k, d))
        self.conv_post = ConvParams(ch, 1, 7, padding=3)
        for c in self.ups:
            _consume_init_normal(c)
        _consume_init_normal(self.conv_post)
        self.upp = int(np.prod(rates))

    def _amp_desc(self):
        hp = self.cfg.model.nsfhifigan
        return self._fill_desc(_lib.AMP_ARCH_HIFIGAN, self.cfg.preprocess.n_mel, hp.upsample_initial_channel,
                               hp.upsample_rates, hp.upsample_kernel_sizes, hp.resblock_kernel_sizes,
                               hp.resblock_dilation_sizes, hp.resblock)

    def _amp_weights(self):
        """HiFi-GAN weights with ``x = x + x`` after every transposed conv folded in (see module docstring)."""
        for key, t in self.state_dict().items():
            if key.startswith(("m_source.", "noise_convs.")):
                continue                                        # never used by the reference's forward either
            if key.startswith("ups.") and key.rsplit(".", 1)[-1] in ("weight_g", "weight", "bias"):
                t = t * 2.0
            yield key, t

    def forward(self, x, f0=None):
        """nsfhifigan.py:258-283.  ``f0`` is accepted and (as in the reference, by its own overwrite) ignored."""
        if f0 is not None and not isinstance(f0, torch.Tensor):
            raise TypeError("f0 must be a tensor or None")
        if f0 is not None and f0.shape[-1] < x.shape[-1]:
            # the reference crops x to min(len(x), len(x_source)) after every transposed conv (:267-269): with fewer
            # f0 frames than mel frames its output is SHORTER and its right edge differs; that crop is not built here
            raise ValueError(f"NSFHiFiGAN: f0 has {f0.shape[-1]} frames but the mel has {x.shape[-1]}: the reference would "
                             "crop the waveform to the f0 length -- pad f0 (pad_f0_to_tensors) or trim the mel")
        return self._amp_forward(x)

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
