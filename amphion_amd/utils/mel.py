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
mel_bands = {}     # data_ptr of a cached basis -> (basis, [n_mel, 2] int32 non-zero band per filter, on its device)


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
        mel_bands[mel_basis[key].data_ptr()] = (mel_basis[key], basis_bands(mel).to(device))
    return mel_basis[key], _window(cfg, device)


def basis_bands(basis):
    """[n_mel, 2] int32: first and one-past-last bin with a non-zero weight per filter (``amp_mel_desc.mel_bands_dev``):
    the triangular filters of librosa touch 2..40 of the 513 bins, the kernel sums only over them."""
    nz = np.asarray(basis) != 0
    lo = np.where(nz.any(1), nz.argmax(1), 0)
    hi = np.where(nz.any(1), nz.shape[1] - nz[:, ::-1].argmax(1), 0)
    return torch.from_numpy(np.stack([lo, hi], 1).astype(np.int32)).contiguous()


def _window(cfg, device):
    wkey = (cfg.win_size, cfg.n_fft, str(device))
    if wkey not in hann_window:
        w = torch.hann_window(cfg.win_size)                                 # periodic Hann (torch default)
        if cfg.win_size < cfg.n_fft:                                        # torch.stft centre-pads the window
            lp = (cfg.n_fft - cfg.win_size) // 2
            w = torch.nn.functional.pad(w, (lp, cfg.n_fft - cfg.win_size - lp))
        hann_window[wkey] = w.float().to(device).contiguous()
    return hann_window[wkey]


_pending_range = []      # (event, pinned [2] tensor) of earlier calls whose min / max have not been looked at yet


def _range_warning(y):
    """utils/mel.py:21-24: the reference prints when the audio leaves [-1, 1].  Reading min / max on the host would
    stall the stream on every call (the comparison needs the value); here the pair is copied to pinned memory behind
    the reduction and looked at by a LATER call (or ``flush_range_warnings()``), once its copy has landed."""
    flush_range_warnings(block=False)
    if not (isinstance(y, torch.Tensor) and y.is_cuda):
        return                                     # the kernel call that follows refuses CPU tensors
    mm = torch.stack(torch.aminmax(y.detach()))
    host = torch.empty(2, dtype=mm.dtype, pin_memory=True)
    host.copy_(mm, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(y.device))
    _pending_range.append((ev, host))


def flush_range_warnings(block=True):
    """Print the pending out-of-range notices (``block=True`` waits for the copies still in flight)."""
    while _pending_range:
        ev, host = _pending_range[0]
        if not ev.query():
            if not block:
                return
            ev.synchronize()
        _pending_range.pop(0)
        mn, mx = float(host[0]), float(host[1])
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
    bands = mel_bands.get(basis.data_ptr()) if basis is not None else None       # (basis kept alive, bands) of a cached basis
    d = _lib.amp_mel_desc(cfg.n_fft, cfg.win_size, cfg.hop_size, n_mel, pad_mode, mag_eps, log_clip,
                          bands[1].data_ptr() if bands is not None and bands[0] is basis else None)
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


class _LogMelFunction(torch.autograd.Function):
    """Differentiable ``extract_mel_features`` for the training-time mel loss (gan_vocoder_trainer.py:387-392): forward =
    the fused front-end kernel (linear mel energies + magnitude / real / imaginary spectra kept for the backward),
    backward = ``amp_mel_backward`` (filterbank transpose, d|X|, inverse real FFT per frame, overlap-add with the
    reflection padding folded back)."""

    @staticmethod
    def forward(ctx, y, cfg, mag_eps, log_clip):
        basis, window = _basis_and_window(cfg, y.device)
        out = _run(y.detach(), cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=mag_eps, log_clip=0.0, want=("mel", "mag", "re", "im"),
                   basis=basis, window=window)
        ctx.cfg, ctx.mag_eps, ctx.log_clip, ctx.shape = cfg, mag_eps, log_clip, tuple(y.shape)
        ctx.save_for_backward(out["mel"], out["mag"], out["re"], out["im"])
        return torch.log(torch.clamp(out["mel"], min=log_clip))          # dynamic_range_compression_torch, mel.py:10-12

    @staticmethod
    def backward(ctx, g):
        mel_lin, mag, re, im = ctx.saved_tensors
        cfg = ctx.cfg
        basis, window = _basis_and_window(cfg, mel_lin.device)
        B, n_mel, F = mel_lin.shape
        Lh = ctx.shape[-1]
        dev = mel_lin.device
        d = _lib.amp_mel_desc(cfg.n_fft, cfg.win_size, cfg.hop_size, n_mel, 0, ctx.mag_eps, ctx.log_clip, None)
        bins = cfg.n_fft // 2 + 1
        spec_ws = torch.empty(2 * B * bins * F, device=dev)
        frames_ws = torch.empty(B * F * cfg.n_fft, device=dev)
        gw = torch.empty((B, Lh), device=dev)
        g = g.to(torch.float32).contiguous()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().amp_mel_backward(ctypes.byref(d), None, B, Lh, ptr(window), ptr(basis), ptr(mel_lin), ptr(mag),
                                                   ptr(re), ptr(im), ptr(g), ptr(spec_ws), ptr(frames_ws), ptr(gw),
                                                   _lib.current_stream_ptr(dev)))
        return gw.reshape(ctx.shape), None, None, None


def extract_mel_features(y, cfg, center=False):
    """utils/mel.py:111-170: log-mel with eps 1e-9 -> [n_mel, F] (squeeze(0)).  Differentiable w.r.t. ``y`` (a GPU
    tensor that requires grad): the backward runs on ``amp_mel_backward``."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    _range_warning(y)
    if torch.is_grad_enabled() and isinstance(y, torch.Tensor) and y.requires_grad:
        y2 = _lib.require_device_tensor(y, "audio")
        out = _LogMelFunction.apply(y2 if y2.dim() == 2 else y2.unsqueeze(0), cfg, 1e-9, 1e-5)
        return out.squeeze(0)
    basis, window = _basis_and_window(cfg, y.device)
    out = _run(y, cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=1e-9, log_clip=1e-5, basis=basis, window=window)["mel"]
    return out.squeeze(0)


class mel_criterion(torch.nn.Module):
    """The generator's mel loss (gan_vocoder_trainer.py:368-395): 45 * L1 between the log-mel of the target audio
    ``y_gt`` [B, L] and of the prediction ``y_pred`` [B, 1, L], both through ``extract_mel_features``; the gradient
    reaches ``y_pred`` through the HIP backward."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.l1Loss = torch.nn.L1Loss(reduction="mean")

    def forward(self, y_gt, y_pred):
        if self.cfg.model.generator not in ("hifigan", "nsfhifigan", "bigvgan", "melgan", "codec", "apnet"):
            raise NotImplementedError
        y_gt_mel = extract_mel_features(y_gt, self.cfg.preprocess)
        y_pred_mel = extract_mel_features(y_pred.squeeze(1), self.cfg.preprocess)
        return self.l1Loss(y_gt_mel, y_pred_mel) * 45


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
