"""Tacotron-style STFT front end (utils/stft.py:19-278) on the gfx950 mel / inverse-STFT kernels.

``STFT.transform/inverse/forward``, ``TacotronSTFT.mel_spectrogram``, ``window_sumsquare`` and
``griffin_lim`` keep the reference signatures.  The reference expresses the DFT as a strided conv1d with
a [2*(n_fft/2+1), 1, n_fft] windowed Fourier basis (and its pseudo-inverse with conv_transpose1d for the
way back) and hard-codes ``.cuda()``/``.cpu()`` (stft.py:167-172); here the same quantities come from the
in-LDS FFTs of ``amp_mel_forward`` (pad_mode 1: reflect-pad n_fft/2, F = L/hop + 1) and
``amp_istft_forward``.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import numpy as np
import torch

from amphion_amd import _lib

from . import mel as _mel


def _hann_periodic_f64(n):
    """scipy.signal.get_window("hann", n, fftbins=True) in float64 (stft.py:67,141)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def window_sumsquare(window, n_frames, hop_length, win_length, n_fft, dtype=np.float32, norm=None):
    """utils/stft.py:19-75 (librosa 0.6): sum-square envelope of the analysis window at hop ``hop_length``,
    length ``n_fft + hop_length * (n_frames - 1)``.  Host-side numpy like the reference (the inverse STFT
    recomputes it per call there; ``STFT.inverse`` caches it per frame count here)."""
    if window != "hann" or norm is not None:
        raise NotImplementedError("only window='hann', norm=None (what the reference's STFT uses)")
    if win_length is None:
        win_length = n_fft
    n = n_fft + hop_length * (n_frames - 1)
    sq = _hann_periodic_f64(win_length) ** 2
    lp = (n_fft - win_length) // 2
    sq = np.pad(sq, (lp, n_fft - win_length - lp))          # librosa.util.pad_center
    env = np.zeros(n, dtype=dtype)
    for i in range(n_frames):                                # same accumulation order / dtype as the reference
        s0 = i * hop_length
        env[s0:min(n, s0 + n_fft)] += sq[:max(0, min(n_fft, n - s0))]
    return env


def griffin_lim(magnitudes, stft_fn, n_iters=30):
    """utils/stft.py:78-95: random initial phase (numpy global RNG, like the reference), then ``n_iters``
    rounds of inverse -> transform keeping the given magnitudes."""
    angles = np.angle(np.exp(2j * np.pi * np.random.rand(*magnitudes.size())))
    angles = torch.from_numpy(angles.astype(np.float32)).to(magnitudes.device)
    signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    for _ in range(n_iters):
        _, angles = stft_fn.transform(signal)
        signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    return signal


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    return torch.exp(x) / C


class STFT(torch.nn.Module):
    """utils/stft.py:113-222."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        if window != "hann":
            raise NotImplementedError("only the reference's default 'hann' window is supported")
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length
        self.window = window
        self._cfg = SimpleNamespace(n_fft=filter_length, win_size=win_length, hop_size=hop_length)
        self._wss = {}

    def transform(self, input_data):
        """stft.py:152-181 -> (magnitude, phase), each [B, n_fft/2+1, L/hop + 1]."""
        out = _mel._run(input_data, self._cfg, n_mel=0, pad_mode=1, mag_eps=0.0, log_clip=0.0,
                        want=("mag", "re", "im"), window=_mel._window(self._cfg, input_data.device))
        phase = torch.atan2(out["im"], out["re"])
        return out["mag"], phase

    def inverse(self, magnitude, phase):
        """stft.py:183-222: (magnitude, phase) [B, n_fft/2+1, F] -> waveform [B, 1, hop * (F - 1)] (+ 1 sample for an odd filter_length)."""
        magnitude = _lib.require_device_tensor(magnitude, "magnitude")
        phase = _lib.require_device_tensor(phase, "phase")
        if magnitude.shape != phase.shape or magnitude.dim() != 3 or magnitude.shape[1] != self.filter_length // 2 + 1:
            raise ValueError(f"magnitude/phase must both be [B, {self.filter_length // 2 + 1}, F], got "
                             f"{tuple(magnitude.shape)} / {tuple(phase.shape)}")
        B, _, F = magnitude.shape
        dev = magnitude.device
        key = (F, str(dev))
        if key not in self._wss:
            env = window_sumsquare(self.window, F, hop_length=self.hop_length, win_length=self.win_length,
                                   n_fft=self.filter_length, dtype=np.float32)
            self._wss[key] = torch.from_numpy(env).to(dev)
        wss = self._wss[key]
        window = _mel._window(self._cfg, dev)
        frames = torch.empty((B, F, self.filter_length), device=dev)
        # the reference crops int(filter_length / 2) samples either side of n_fft + hop * (F - 1): one more sample survives for an odd length
        out = torch.empty((B, 1, self.hop_length * (F - 1) + (self.filter_length & 1)), device=dev)
        d = _lib.amp_mel_desc(self.filter_length, self.win_length, self.hop_length, 0, 1, 0.0, 0.0)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().amp_istft_forward(ctypes.byref(d), p(magnitude), p(phase), B, F, p(window), p(wss),
                                                    p(frames), p(out), _lib.current_stream_ptr(dev)))
        return out

    def forward(self, input_data):
        """stft.py:219-222: transform then inverse."""
        self.magnitude, self.phase = self.transform(input_data)
        return self.inverse(self.magnitude, self.phase)


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
