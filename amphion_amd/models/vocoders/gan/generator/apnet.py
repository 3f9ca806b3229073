"""APNet generator drop-in (models/vocoders/gan/generator/apnet.py:280-399) on the gfx950 kernels.

Same constructor (``APNet(cfg)`` reading ``cfg.model.apnet.*`` and ``cfg.preprocess.{n_mel, n_fft, hop_size,
win_size}``), same ``state_dict`` keys, same ``forward(mel) -> (logamp, pha, rea, imag, audio[B, 1, T*hop])``.

    amplitude / phase branches   input conv -> mean of 3 ResBlocks (hifigan-style pairs at FRAME rate, 512 ch)
                                 -> leaky_relu(0.01) -> output conv(s)           : the fused implicit-GEMM conv kernel
                                 (residual add and the MRF sum / mean fused into the conv epilogues)
    atan2 / exp / cos / sin      one element-wise kernel (amp_apnet_polar)
    ISTFT, "same" padding        per-frame inverse FFT in LDS + gather overlap-add (amp_istft_same)
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules.hip_ops import HipConv1d
from amphion_amd.modules.vocoder_blocks import get_padding

LRELU_SLOPE = 0.1


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _ResBlock(nn.Module):
    """ASPResBlock / PSPResBlock (apnet.py:107-192,195-277): identical to HiFi-GAN's ResBlock1."""

    def __init__(self, cfg, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.cfg = cfg
        self.convs1 = nn.ModuleList([HipConv1d(channels, channels, kernel_size, dilation=d,
                                               padding=get_padding(kernel_size, d)) for d in dilation])
        self.convs2 = nn.ModuleList([HipConv1d(channels, channels, kernel_size, dilation=1,
                                               padding=get_padding(kernel_size, 1)) for _ in dilation])

    def run(self, x, acc, mode, div, tmp_a, tmp_b):
        """acc <- MRF-combine(acc, resblock(x)) (mode 0: =, 1: +=, 2: (acc + .)/div); tmp_* are scratch."""
        L = _lib.lib()
        dev = x.device
        B, _, T = x.shape
        st = _lib.current_stream_ptr(dev)
        cur = x
        n = len(self.convs1)
        for p, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            xt = c1(cur, slope_in=LRELU_SLOPE, slope_out=LRELU_SLOPE, out=tmp_a)     # xt = lrelu(c1(lrelu(x)))
            if p + 1 < n:
                dst = tmp_b if cur is not tmp_b else x.new_empty(x.shape)
                c2(xt, res=cur, out=dst)                                             # x = c2(xt) + x
                cur = dst
            else:
                h = c2._ensure(dev)
                with torch.cuda.device(dev):
                    _lib.check(L.amp_conv_forward_mrf(h, _p(xt), B, T, 1.0, _p(cur), _p(acc), mode, float(div), st))
        return acc


class ASPResBlock(_ResBlock):
    pass


class PSPResBlock(_ResBlock):
    pass


class ISTFT(nn.Module):
    """apnet.py:16-101, "same" padding only (what APNet uses); ``forward(re, im, window)``."""

    def __init__(self, n_fft, hop_length, win_length, padding="same"):
        super().__init__()
        if padding != "same":
            raise NotImplementedError("only padding='same' (the APNet head) runs on the HIP path")
        self.padding, self.n_fft, self.hop_length, self.win_length = padding, n_fft, hop_length, win_length
        self._env = {}

    def forward(self, rea, imag, window):
        B, N, T = rea.shape
        dev = rea.device
        key = (T, str(dev))
        if key not in self._env:                     # overlap-added window^2 (apnet.py:88-95), host side, cached
            w2 = window.detach().double().cpu().numpy() ** 2
            env = np.zeros((T - 1) * self.hop_length + self.win_length, dtype=np.float64)
            for f in range(T):
                env[f * self.hop_length: f * self.hop_length + self.win_length] += w2
            self._env[key] = torch.from_numpy(env.astype(np.float32)).to(dev)
        env = self._env[key]
        frames = torch.empty((B, T, self.n_fft), device=dev)
        out = torch.empty((B, T * self.hop_length), device=dev)
        d = _lib.amp_mel_desc(self.n_fft, self.win_length, self.hop_length, 0, 1, 0.0, 0.0)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().amp_istft_same(ctypes.byref(d), _p(rea), _p(imag), B, T, _p(window), _p(env), _p(frames),
                                                 _p(out), _lib.current_stream_ptr(dev)))
        return out


class APNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hp = cfg.model.apnet
        pp = cfg.preprocess
        self.ASP_num_kernels = len(hp.ASP_resblock_kernel_sizes)
        self.PSP_num_kernels = len(hp.PSP_resblock_kernel_sizes)
        bins = pp.n_fft // 2 + 1
        self.ASP_input_conv = HipConv1d(pp.n_mel, hp.ASP_channel, hp.ASP_input_conv_kernel_size,
                                        padding=get_padding(hp.ASP_input_conv_kernel_size, 1))
        self.PSP_input_conv = HipConv1d(pp.n_mel, hp.PSP_channel, hp.PSP_input_conv_kernel_size,
                                        padding=get_padding(hp.PSP_input_conv_kernel_size, 1))
        self.ASP_ResNet = nn.ModuleList([ASPResBlock(cfg, hp.ASP_channel, k, d) for k, d in
                                         zip(hp.ASP_resblock_kernel_sizes, hp.ASP_resblock_dilation_sizes)])
        self.PSP_ResNet = nn.ModuleList([PSPResBlock(cfg, hp.PSP_channel, k, d) for k, d in
                                         zip(hp.PSP_resblock_kernel_sizes, hp.PSP_resblock_dilation_sizes)])
        self.ASP_output_conv = HipConv1d(hp.ASP_channel, bins, hp.ASP_output_conv_kernel_size,
                                         padding=get_padding(hp.ASP_output_conv_kernel_size, 1))
        self.PSP_output_R_conv = HipConv1d(hp.PSP_channel, bins, hp.PSP_output_R_conv_kernel_size,
                                           padding=get_padding(hp.PSP_output_R_conv_kernel_size, 1))
        self.PSP_output_I_conv = HipConv1d(hp.PSP_channel, bins, hp.PSP_output_I_conv_kernel_size,
                                           padding=get_padding(hp.PSP_output_I_conv_kernel_size, 1))
        self.iSTFT = ISTFT(pp.n_fft, hop_length=pp.hop_size, win_length=pp.win_size)
        self._window = {}

    def _branch(self, mel, conv_in, blocks):
        """input conv -> (rb0 + rb1 + ...)/n -> leaky_relu(0.01) is applied by the output convs on load."""
        x = conv_in(mel)
        acc = torch.empty_like(x)
        tmp_a, tmp_b = torch.empty_like(x), torch.empty_like(x)
        n = len(blocks)
        for j, rb in enumerate(blocks):
            mode = 0 if (n == 1 or j == 0) else (2 if j == n - 1 else 1)
            rb.run(x, acc, mode, n, tmp_a, tmp_b)
        return acc

    def forward(self, mel):
        """apnet.py:354-399."""
        mel = _lib.require_device_tensor(mel, "mel")
        dev = mel.device
        a = self._branch(mel, self.ASP_input_conv, self.ASP_ResNet)
        logamp = self.ASP_output_conv(a, slope_in=0.01)               # F.leaky_relu default slope (:364,:375)
        p = self._branch(mel, self.PSP_input_conv, self.PSP_ResNet)
        R = self.PSP_output_R_conv(p, slope_in=0.01)
        I = self.PSP_output_I_conv(p, slope_in=0.01)
        pha, rea, imag = torch.empty_like(R), torch.empty_like(R), torch.empty_like(R)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().amp_apnet_polar(_p(logamp), _p(R), _p(I), R.numel(), _p(pha), _p(rea), _p(imag),
                                                  _lib.current_stream_ptr(dev)))
        wkey = str(dev)
        if wkey not in self._window:
            self._window[wkey] = torch.hann_window(self.cfg.preprocess.win_size).to(dev)
        audio = self.iSTFT(rea, imag, self._window[wkey])
        return logamp, pha, rea, imag, audio.unsqueeze(1)
