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
    # FFT bin centre frequencies AS librosa 0.9.1 (the reference's pin, env.sh:13) lays them out: fft_frequencies() = linspace(0, sr / 2,
    # 1 + n_fft // 2) -- k * sr / n_fft for even n_fft, a slightly stretched grid for odd ones (round 5: the odd lengths became reachable)
    nu = np.linspace(0.0, float(sr) / 2.0, bins, endpoint=True)
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


# ---- the reference's out-of-range notice (utils/mel.py:21-24) without a reduction pass of its own ----------------------
# The front-end kernel folds the extreme samples it reads anyway into a 3-word slot (amp_mel_desc.range_dev) and the call copies
# the slot to pinned memory behind itself; a LATER call (or flush_range_warnings()) prints what has landed.  Per device: a ring of
# slots, each re-initialised by the kernel of the call before it -- no extra launch, no host synchronisation, no allocation per call
# (round 2: torch.aminmax + stack + a fresh pinned tensor + copy + event = as much host time as the mel kernel takes on the GPU).
# One ring per (device, STREAM): "kernel k resets the slot of kernel k + 1" is an ordering argument that only holds along one stream.
# A call that fails after taking its slot re-initialises both slots itself (_range_abort); inside a stream capture the notice is
# skipped altogether (the ring's bookkeeping is host state a replay would not repeat, and a full ring would have to synchronise).
_RANGE_SLOTS = 64
_INIT = (int(np.float32(-1.0).view(np.int32)), int(np.float32(1.0).view(np.int32)), 0, 0)
_range_rings = {}


class _RangeRing:
    def __init__(self, device):
        self.dev = torch.tensor([_INIT] * _RANGE_SLOTS, dtype=torch.int32).to(device)           # [slots, 4] (16-B rows)
        self.host = torch.zeros((_RANGE_SLOTS, 4), dtype=torch.int32).pin_memory()
        self.host_np = self.host.numpy()
        self.dev_ptr, self.host_ptr = self.dev.data_ptr(), self.host.data_ptr()
        self.seq = 0                       # calls issued
        self.done = 0                      # calls whose slot has been looked at
        self.stream_dev = device

    def abort(self, seq):
        """call `seq` took its slot and then raised before (or instead of) launching: nobody reset the next slot, and its own may
        hold a partial fold -- re-initialise both on the current stream and stop waiting for its copy"""
        init = torch.tensor(_INIT, dtype=torch.int32)
        for k in (seq, seq + 1):
            self.dev[k % _RANGE_SLOTS].copy_(init, non_blocking=False)
        row = self.host_np[seq % _RANGE_SLOTS]
        row[0], row[1], row[2] = _INIT[0], _INIT[1], seq & 0x7FFFFFFF      # "landed, nothing seen"

    def next(self):
        """(range_dev, range_host, range_reset_dev, seq) for the next call."""
        if self.seq - self.done >= _RANGE_SLOTS - 1:            # ring full: the oldest copy must land before its slot is re-used
            self.flush(block=True)
        self.seq += 1
        i, n = self.seq % _RANGE_SLOTS, (self.seq + 1) % _RANGE_SLOTS
        return self.dev_ptr + 16 * i, self.host_ptr + 16 * i, self.dev_ptr + 16 * n, self.seq

    def flush(self, block):
        while self.done < self.seq:
            k = self.done + 1
            row = self.host_np[k % _RANGE_SLOTS]
            if int(row[2]) != (k & 0x7FFFFFFF):                 # copy of call k not landed yet
                if not block:
                    return
                torch.cuda.synchronize(self.stream_dev)
                if int(row[2]) != (k & 0x7FFFFFFF):             # still nothing after a device-wide wait: call k raised before its
                    self.done = k                               # launch (bad shape, short audio) -- there is no notice to wait for
                    continue
            self.done = k
            mn = float(np.int32(row[0]).view(np.float32))
            mx = float(np.int32(row[1]).view(np.float32))
            if mn < -1.0:
                print("min value is ", mn)
            if mx > 1.0:
                print("max value is ", mx)


def _ring_key(device):
    return (device, torch.cuda.current_stream(device).cuda_stream)


def _range_slot(device):
    if torch.cuda.is_current_stream_capturing():
        return None, None, None, 0                  # no notice from inside a capture (see above)
    key = _ring_key(device)
    ring = _range_rings.get(key)
    if ring is None:
        ring = _range_rings[key] = _RangeRing(device)
    ring.flush(block=False)
    return ring.next()


def _range_abort(device, seq):
    ring = _range_rings.get(_ring_key(device))
    if ring is not None and seq:
        ring.abort(seq)


def _range_warning(y):
    """Kept for callers of the round-2 name: the notice now rides on the front-end launch itself (``report_range=True``)."""
    flush_range_warnings(block=False)


def flush_range_warnings(block=True):
    """Print the pending out-of-range notices (``block=True`` waits for the copies still in flight)."""
    for ring in _range_rings.values():
        ring.flush(block)


# ---- per (config, device) launch plan: everything a call needs that does not depend on the audio ----------------------------
_plans = {}


class _Plan:
    __slots__ = ("basis", "window", "bands", "descs", "n_fft", "win_size", "hop_size", "bins")


def _plan(cfg, device):
    key = (cfg.sample_rate, cfg.n_fft, cfg.win_size, cfg.hop_size, cfg.n_mel, float(cfg.fmin),
           None if cfg.fmax is None else float(cfg.fmax), device)
    p = _plans.get(key)
    if p is None:
        p = _Plan()
        p.basis, p.window = _basis_and_window(cfg, device)
        p.bands = mel_bands[p.basis.data_ptr()][1]
        p.descs = {}
        p.n_fft, p.win_size, p.hop_size, p.bins = cfg.n_fft, cfg.win_size, cfg.hop_size, cfg.n_fft // 2 + 1
        if cfg.n_fft == 1024 and not torch.cuda.is_current_stream_capturing():
            # the n_fft = 1024 kernel's one-time set-up (a blocking 4.5-KB table upload), here and not at the first launch:
            # a later call may sit inside a stream capture
            with torch.cuda.device(device):
                _lib.check(_lib.lib().amp_mel_init())
        _plans[key] = p
    return p


def _run(y, cfg, *, n_mel, pad_mode, mag_eps, log_clip, want=("mel",), basis=None, window=None, lengths=None, report_range=False):
    y = _lib.require_device_tensor(y, "audio")
    if y.dim() == 1:
        y = y.unsqueeze(0)
    if y.dim() != 2:
        raise ValueError(f"expected audio of shape [B, L], got {tuple(y.shape)}")
    B, Lh = y.shape
    bands = mel_bands.get(basis.data_ptr()) if basis is not None else None       # (basis kept alive, bands) of a cached basis
    d = _lib.amp_mel_desc(cfg.n_fft, cfg.win_size, cfg.hop_size, n_mel, pad_mode, mag_eps, log_clip,
                          bands[1].data_ptr() if bands is not None and bands[0] is basis else None)
    seq_taken = 0
    if report_range:
        d.range_dev, d.range_host, d.range_reset_dev, d.range_seq = _range_slot(y.device)
        seq_taken = d.range_seq
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
        try:
            _lib.check(L.amp_mel_forward_ragged(ctypes.byref(d), ptr(y), ptr(lens), B, Lh, ptr(window), ptr(basis),
                                                ptr(outs.get("mel")), ptr(outs.get("mag")), ptr(outs.get("re")),
                                                ptr(outs.get("im")), _lib.current_stream_ptr(dev)))
        except Exception:
            _range_abort(dev, seq_taken)           # the slot this call took never got its kernel
            raise
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


def _logmel_fast(y, cfg, mag_eps):
    """log-mel [B, n_mel, F] of a device tensor: the two hot entry points (``extract_mel_features``, ``mel_spectrogram_torch``) as
    ONE ctypes call -- basis / window / band table / descriptor come from the per-(config, device) plan, the range notice rides on
    the launch (``_range_slot``)."""
    if y.dim() == 1:
        y = y.unsqueeze(0)
    if y.dim() != 2:
        raise ValueError(f"expected audio of shape [B, L], got {tuple(y.shape)}")
    dev = y.device
    p = _plan(cfg, dev)
    d = p.descs.get(mag_eps)
    if d is None:
        d = p.descs[mag_eps] = _lib.amp_mel_desc(p.n_fft, p.win_size, p.hop_size, cfg.n_mel, 0, mag_eps, 1e-5, p.bands.data_ptr())
    B, Lh = y.shape
    pad = (p.n_fft - p.hop_size) // 2
    if Lh <= pad:      # too short for the reflection padding: let the library say so (AmpError), through the general path
        return _run(y, cfg, n_mel=cfg.n_mel, pad_mode=0, mag_eps=mag_eps, log_clip=1e-5, basis=p.basis, window=p.window)["mel"]
    d.range_dev, d.range_host, d.range_reset_dev, d.range_seq = _range_slot(dev)
    F = (Lh + 2 * pad - p.n_fft) // p.hop_size + 1
    if not y.is_contiguous() or y.dtype != torch.float32:
        y = y.contiguous().float()
    out = torch.empty((B, cfg.n_mel, F), device=dev)
    with torch.cuda.device(dev):
        try:
            _lib.check(_lib.lib().amp_mel_forward_ragged(ctypes.byref(d), y.data_ptr(), None, B, Lh, p.window.data_ptr(), p.basis.data_ptr(),
                                                         out.data_ptr(), None, None, None, _lib.current_stream_ptr(dev)))
        except Exception:
            _range_abort(dev, d.range_seq)
            raise
    return out


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    """utils/mel.py:10-12"""
    return torch.log(torch.clamp(x, min=clip_val) * C)


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def extract_linear_features(y, cfg, center=False):
    """utils/mel.py:20-52: |STFT| with eps 1e-9 -> [bins, F] (batch dim squeezed when B == 1)."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    out = _run(y, cfg, n_mel=0, pad_mode=0, mag_eps=1e-9, log_clip=0.0, want=("mag",), window=_window(cfg, y.device), report_range=True)
    return torch.squeeze(out["mag"], 0)


def mel_spectrogram_torch(y, cfg, center=False):
    """utils/mel.py:55-104: log-mel with eps 1e-6 -> [B, n_mel, F]."""
    if center:
        raise NotImplementedError("center=True is never used by the reference callers")
    y = _lib.require_device_tensor(y, "audio")
    return _logmel_fast(y, cfg, 1e-6)


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
    y = _lib.require_device_tensor(y, "audio")
    if torch.is_grad_enabled() and y.requires_grad:
        out = _LogMelFunction.apply(y if y.dim() == 2 else y.unsqueeze(0), cfg, 1e-9, 1e-5)
        return out.squeeze(0)
    return _logmel_fast(y, cfg, 1e-9).squeeze(0)


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
