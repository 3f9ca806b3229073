"""Tacotron-style STFT front end (utils/stft.py:19-278) on the gfx950 mel kernel.

``STFT.transform`` and ``TacotronSTFT.mel_spectrogram`` keep the reference signatures.  The
reference expresses the DFT as a strided conv1d with a [2*(n_fft/2+1), 1, n_fft] windowed Fourier
basis and hard-codes ``.cuda()``/``.cpu()`` (stft.py:167-172); here the same quantity comes from the
in-LDS FFT of ``amp_mel_forward`` (pad_mode 1: reflect-pad n_fft/2, F = L/hop + 1).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import mel as _mel


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    return torch.exp(x) / C


class STFT(torch.nn.Module):
    """utils/stft.py:19-222 (forward transform only; ``inverse``/griffin_lim are not on the vocoder
    inference path)."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        if window != "hann":
            raise NotImplementedError("only the reference's default 'hann' window is supported")
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length
        self.window = window
        self._cfg = SimpleNamespace(n_fft=filter_length, win_size=win_length, hop_size=hop_length)

    def transform(self, input_data):
        """stft.py:152-181 -> (magnitude, phase), each [B, n_fft/2+1, L/hop + 1]."""
        out = _mel._run(input_data, self._cfg, n_mel=0, pad_mode=1, mag_eps=0.0, log_clip=0.0,
                        want=("mag", "re", "im"), window=_mel._window(self._cfg, input_data.device))
        phase = torch.atan2(out["im"], out["re"])
        return out["mag"], phase

    def inverse(self, magnitude, phase):
        raise NotImplementedError("STFT.inverse (Griffin-Lim path) is outside the vocoder-inference hot path")

    def forward(self, input_data):
        raise NotImplementedError("STFT.forward needs inverse(); outside the vocoder-inference hot path")


class TacotronSTFT(torch.nn.Module):
    """utils/stft.py:225-278."""

    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        mel_basis = _mel.librosa_mel_fn(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer("mel_basis", torch.from_numpy(mel_basis).float())
        self._cfg = SimpleNamespace(n_fft=filter_length, win_size=win_length, hop_size=hop_length,
                                    n_mel=n_mel_channels)

    def spectral_normalize(self, magnitudes):
        return dynamic_range_compression(magnitudes)

    def spectral_de_normalize(self, magnitudes):
        return dynamic_range_decompression(magnitudes)

    def mel_spectrogram(self, y):
        """stft.py:259-278: y [B, L] in [-1, 1] -> (mel [B, n_mel, F], energy [B, F])."""
        assert torch.min(y.data) >= -1
        assert torch.max(y.data) <= 1
        basis = self.mel_basis.to(y.device).contiguous()
        out = _mel._run(y, self._cfg, n_mel=self.n_mel_channels, pad_mode=1, mag_eps=0.0, log_clip=1e-5,
                        want=("mel", "mag"), basis=basis, window=_mel._window(self._cfg, y.device))
        energy = torch.norm(out["mag"], dim=1)
        return out["mel"], energy
