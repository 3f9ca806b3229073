"""MI355X-native mel / STFT front end: same function names, arguments and return shapes as
utils/mel.py, computed by one fused gfx950 kernel per call (frame -> window -> FFT in LDS -> |X| ->
mel filterbank -> log) through ``amp_mel_forward``.

No CPU fallback: the audio tensor must live on a ROCm device.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from amphion_amd import _lib

# caches keyed like the reference's module-level dicts (utils/mel.py:107-108) -- but keyed
# correctly, so the basis is NOT rebuilt on every call (the reference's `cfg.fmax not in mel_basis`
# test never hits because its keys are "<fmax>_<device>", utils/mel.py:132)
mel_basis = {}
hann_window = {}


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * (3.0 / 200.0)
    log_region = f >= 1000.0
    with np.errstate(divide="ignore", invalid="ignore"):
        logv = 15.0 + np.log(np.where(log_region, f, 1000.0) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(log_region, logv, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    log_region = m >= 15.0
    logv = 1000.0 * np.exp((np.where(log_region, m, 15.0) - 15.0) * (np.log(6.4) / 27.0))
    return np.where(log_region, logv, lin)


def librosa_mel_fn(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """Slaney-scale, Slaney-normalised triangular mel filterbank == librosa 0.9.1
    ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (called at utils/mel.py:66-72,133-139 and
    utils/stft.py:245-247).  float32 [n_mels, n_fft//2+1]."""
    if fmax is None:
        fmax = sr / 2.0
    bins = n_fft // 2 + 1
    nu = np.arange(bins, dtype=np.float64) * (float(sr) / n_fft)          # FFT bin centre frequencies
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    lo, ce, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    rising = (nu[None, :] - lo) / (ce - lo)
    falling = (hi - nu[None, :]) / (hi - ce)
    tri = np.clip(np.minimum(rising, falling), 0.0, None)
    tri *= 2.0 / (hi - lo)                                                  # Slaney area normalisation
    return tri.astype(np.float32)


def _basis_and_window(cfg, device):
    key = (cfg.sample_rate, cfg.n_fft, cfg.n_mel, float(cfg.fmin), None if cfg.fmax is None else float(cfg.fmax), str(device))
    if key not in mel_basis:
        mel = librosa_mel_fn(sr=cfg.sample_rate, n_fft=cfg.n_fft, n_mels=cfg.n_mel, fmin=cfg.fmin, fmax=cfg.fmax)
        mel_basis[key] = torch.from_numpy(mel).float().to(device).contiguous()
    return mel_basis[key], _window(cfg, device)


def _window(cfg, device):
    wkey = (cfg.win_size, cfg.n_fft, str(device))
    if wkey not in hann_window:
        w = torch.hann_window(cfg.win_size)                                 # periodic Hann (torch default)
        if cfg.win_size < cfg.n_fft:                                        # torch.stft centre-pads the window
            lp = (cfg.n_fft - cfg.win_size) // 2
            w = torch.nn.functional.pad(w, (lp, cfg.n_fft - cfg.win_size - lp))
        hann_window[wkey] = w.float().to(device).contiguous()
    return hann_window[wkey]


def _range_warning(y):
    # utils/mel.py:21-24: the reference prints when the audio leaves [-1, 1]
    mn, mx = torch.aminmax(y)
    if mn < -1.0:
        print("min value is ", mn)
    if mx > 1.0:
        print("max value is ", mx)


def _run(y, cfg, *, n_mel, pad_mode, mag_eps, log_clip, want=("mel",), basis=None, window=None, lengths=None):
    y = _lib.require_device_tensor(y, "audio")
    if y.dim() == 1:
        y = y.unsqueeze(0)
    if y.dim() != 2:
        raise ValueError(f"expected audio of shape [B, L], got {tuple(y.shape)}")
    B, Lh = y.shape
    d = _lib.amp_mel_desc(cfg.n_fft, cfg.win_size, cfg.hop_size, n_mel, pad_mode, mag_eps, log_clip)
    L = _lib.lib()
    F = L.amp_mel_num_frames(ctypes.byref(d), Lh)
    bins = cfg.n_fft // 2 + 1
    dev = y.device
    outs = {}
    if "mel" in want:
        outs["mel"] = torch.empty((B, n_mel, F), device=dev)
    for k in ("mag", "re", "im"):
        if k in want:
            outs[k] = torch.empty((B, bins, F), device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    lens = None
    if lengths is not None:
        lens = torch.as_tensor(lengths).to(device=dev, dtype=torch.int32).contiguous()
        if lens.numel() != B:
            raise ValueError(f"lengths must hold B={B} values")
        for t in outs.values():
            t.zero_()                      # frames beyond an utterance's own count are not written by the kernel
    with torch.cuda.device(dev):
        _lib.check(L.amp_mel_forward_ragged(ctypes.byref(d), ptr(y), ptr(lens), B, Lh, ptr(window), ptr(basis),
                                            ptr(outs.get("mel")), ptr(outs.get("mag")), ptr(outs.get("re")),
                                            ptr(outs.get("im")), _lib.current_stream_ptr(dev)))
    return outs


def num_frames(n_samples, cfg):
    """Frames ``extract_mel_features`` yields for ``n_samples`` (reflect pad (n_fft-hop)/2, center=False)."""
    pad = (cfg.n_fft - cfg.hop_size) // 2
    return (n_samples + 2 * pad - cfg.n_fft) // cfg.hop_size + 1


def extract_mel_features_batch(wavs, cfg, device=None):
    """``extract_mel_features`` (utils/mel.py:111-170) over a LIST of waveforms of different lengths in ONE kernel
    launch: the batch is zero-padded, every utterance is reflect-padded at its own end
    (``amp_mel_forward_ragged``), and each returned ``[n_mel, F_i]`` equals the one-file-at-a-time result of the
    reference's feature extraction (processors/acoustic_extractor.py:394-401)."""
    if len(wavs) == 0:
        return []
    device = device or (wavs[0].device if isinstance(wavs[0], torch.Tensor) and wavs[0].is_cuda else "cuda")
    lens = [int(w.shape[-1]) for w in wavs]
    batch = torch.zeros((len(wavs), max(lens)), dtype=torch.float32)
    for i, w in enumerate(wavs):
        batch[i, : lens[i]] = torch.as_tensor(w, dtype=torch.float32).reshape(-1).cpu()
    batch = batch.to(device)
    basis, window = _basis_and_window(cfg, batch.device)
    mel = _run(batch, cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=1e-9, log_clip=1e-5, basis=basis, window=window,
               lengths=lens)["mel"]
    return [mel[i, :, : num_frames(lens[i], cfg)] for i in range(len(wavs))]


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    """utils/mel.py:10-12"""
    return torch.log(torch.clamp(x, min=clip_val) * C)


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def extract_linear_features(y, cfg, center=False):
    """utils/mel.py:20-52: |STFT| with eps 1e-9 -> [bins, F] (batch dim squeezed when B == 1)."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    _range_warning(y)
    out = _run(y, cfg, n_mel=0, pad_mode=0, mag_eps=1e-9, log_clip=0.0, want=("mag",), window=_window(cfg, y.device))
    return torch.squeeze(out["mag"], 0)


def mel_spectrogram_torch(y, cfg, center=False):
    """utils/mel.py:55-104: log-mel with eps 1e-6 -> [B, n_mel, F]."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    _range_warning(y)
    basis, window = _basis_and_window(cfg, y.device)
    return _run(y, cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=1e-6, log_clip=1e-5, basis=basis, window=window)["mel"]


def extract_mel_features(y, cfg, center=False):
    """utils/mel.py:111-170: log-mel with eps 1e-9 -> [n_mel, F] (squeeze(0))."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    _range_warning(y)
    basis, window = _basis_and_window(cfg, y.device)
    out = _run(y, cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=1e-9, log_clip=1e-5, basis=basis, window=window)["mel"]
    return out.squeeze(0)


def extract_mel_features_tts(y, cfg, center=False, taco=False, _stft=None):
    """utils/mel.py:173-241."""
    if not taco:
        return extract_mel_features(y, cfg, center=center)
    audio = torch.clip(y, -1, 1)
    spec, _energy = _stft.mel_spectrogram(audio)
    return spec.squeeze(0)


def amplitude_phase_spectrum(y, cfg):
    """utils/mel.py:244-280 -> (log_amplitude, phase, real, imag), each [B, bins, F] (B squeezed when 1)."""
    out = _run(y, cfg, n_mel=0, pad_mode=0, mag_eps=0.0, log_clip=0.0, want=("mag", "re", "im"),
               window=_window(cfg, y.device))
    rea, imag, mag = out["re"], out["im"], out["mag"]
    if rea.size(0) == 1:
        rea, imag, mag = rea.squeeze(0), imag.squeeze(0), mag.squeeze(0)
    log_amplitude = torch.log(mag + 1e-5)
    phase = torch.atan2(imag, rea)
    return log_amplitude, phase, rea, imag
