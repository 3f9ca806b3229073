"""CPU oracle for the Amphion vocoder-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``amphion_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

This is a *restatement* of the reference algorithm as plain functions over a
``state_dict`` (reference key names, Appendix A of SURVEY.md).  The reference
itself has no arithmetic of its own on this path: every op is a call into
torch (``env.sh:15`` pins torch==2.0.1; this image has 2.10) or librosa 0.9.1
(``env.sh:13``, absent here).  The restatement therefore calls the same
``torch.nn.functional`` primitives on CPU for the convolutions and writes the
rest (weight-norm fold, Snake, Kaiser-sinc filter, Slaney mel filterbank,
framing + DFT) out explicitly.

Parity pin: the reference has no tests / golden vectors, so the
oracle is pinned against outputs of the reference classes themselves, imported
in the build container by ``tests/golden/make_golden.py`` and committed under
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks them).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # models/vocoders/gan/generator/hifigan.py:14, bigvgan.py:17


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """modules/vocoder_blocks/gan_utils.py:12-13"""
    return int((kernel_size * dilation - dilation) / 2)


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||_2 over all dims but 0.

    Used on every conv of the generators (hifigan.py:23,157,176,199).  For
    ConvTranspose1d dim 0 is C_in.  No epsilon.
    """
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * (v / norm)


def conv_params(sd, prefix, dtype):
    """Return (weight, bias) for ``prefix`` accepting weight-normed or folded keys
    (vocoder_inference.py:270-332 loads either form; remove_weight_norm collapses
    weight_g/weight_v to weight)."""
    if prefix + ".weight" in sd:
        w = torch.as_tensor(sd[prefix + ".weight"]).to(dtype)
    else:
        g = torch.as_tensor(sd[prefix + ".weight_g"]).to(dtype)
        v = torch.as_tensor(sd[prefix + ".weight_v"]).to(dtype)
        w = fold_weight_norm(g, v)
    b = sd.get(prefix + ".bias", None)
    if b is not None:
        b = torch.as_tensor(b).to(dtype)
    return w, b


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


# ----------------------------------------------------------------------------
# HiFi-GAN
# ----------------------------------------------------------------------------
def resblock1(sd, prefix, x, k, dils, dtype, act=None):
    """ResBlock1.forward hifigan.py:93-100 (ResBlock1_vits :305-320 with x_mask=None).

    ``act`` = None -> leaky_relu(0.1); otherwise a callable(idx, x) for AMPBlock1
    (bigvgan.py:137-146: activations[2p] before convs1[p], [2p+1] before convs2[p]).
    """
    for p, d in enumerate(dils):
        w1, b1 = conv_params(sd, f"{prefix}.convs1.{p}", dtype)
        w2, b2 = conv_params(sd, f"{prefix}.convs2.{p}", dtype)
        xt = F.leaky_relu(x, LRELU_SLOPE) if act is None else act(2 * p, x)
        xt = F.conv1d(xt, w1, b1, dilation=d, padding=get_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE) if act is None else act(2 * p + 1, xt)
        xt = F.conv1d(xt, w2, b2, dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(sd, prefix, x, k, dils, dtype, act=None):
    """ResBlock2.forward hifigan.py:140-145 / AMPBlock2.forward bigvgan.py:218-224."""
    for p, d in enumerate(dils):
        w, b = conv_params(sd, f"{prefix}.convs.{p}", dtype)
        xt = F.leaky_relu(x, LRELU_SLOPE) if act is None else act(p, x)
        xt = F.conv1d(xt, w, b, dilation=d, padding=get_padding(k, d))
        x = xt + x
    return x


def hifigan_forward(sd, hp, mel, dtype=torch.float32, g=None, ups_key="ups.{i}"):
    """HiFiGAN.forward hifigan.py:203-219 and HiFiGAN_vits.forward hifigan.py:424-443.

    ``hp`` carries resblock, upsample_rates, upsample_kernel_sizes,
    resblock_kernel_sizes, resblock_dilation_sizes (cfg.model.hifigan.*,
    hifigan.py:155-199).  ``g`` is the optional VITS speaker condition.
    """
    x = torch.as_tensor(mel).to(dtype)
    rates = list(_cfg_get(hp, "upsample_rates"))
    uks = list(_cfg_get(hp, "upsample_kernel_sizes"))
    rks = list(_cfg_get(hp, "resblock_kernel_sizes"))
    rds = [list(d) for d in _cfg_get(hp, "resblock_dilation_sizes")]
    rb = resblock1 if str(_cfg_get(hp, "resblock")) == "1" else resblock2
    nk = len(rks)

    w, b = conv_params(sd, "conv_pre", dtype)
    x = F.conv1d(x, w, b, padding=3)  # :204
    if g is not None:  # :426-427
        wc, bc = conv_params(sd, "cond", dtype)
        x = x + F.conv1d(torch.as_tensor(g).to(dtype), wc, bc)
    for i, (u, k) in enumerate(zip(rates, uks)):
        x = F.leaky_relu(x, LRELU_SLOPE)  # :206
        w, b = conv_params(sd, ups_key.format(i=i), dtype)
        x = F.conv_transpose1d(x, w, b, stride=u, padding=(k - u) // 2)  # :207
        xs = None
        for j in range(nk):  # :208-213
            r = rb(sd, f"resblocks.{i * nk + j}", x, rks[j], rds[j], dtype)
            xs = r if xs is None else xs + r
        x = xs / nk  # :214
    x = F.leaky_relu(x)  # default slope 0.01, :215
    w, b = conv_params(sd, "conv_post", dtype)
    x = F.conv1d(x, w, b, padding=3)  # :216
    return torch.tanh(x)  # :217


# ----------------------------------------------------------------------------
# BigVGAN: Snake + anti-aliased activation
# ----------------------------------------------------------------------------
def kaiser_sinc_filter1d(cutoff, half_width, kernel_size, dtype=torch.float32):
    """modules/anti_aliasing/filter.py:30-61 (returns [kernel_size])."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if even:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.to(dtype)


def snake(x, alpha, beta=None, logscale=False):
    """Snake.forward snake.py:51-61 / SnakeBeta.forward snake.py:110-122."""
    a = alpha.reshape(1, -1, 1).to(x.dtype)
    b = a if beta is None else beta.reshape(1, -1, 1).to(x.dtype)
    if logscale:
        a = torch.exp(a)
        b = torch.exp(b) if beta is not None else a
    return x + (1.0 / (b + 0.000000001)) * torch.pow(torch.sin(x * a), 2)


def activation1d(x, alpha, beta=None, logscale=False, filt_up=None, filt_down=None):
    """Activation1d.forward act.py:31-36 with UpSample1d (resample.py:36-45) and
    DownSample1d/LowPassFilter1d (resample.py:62-65, filter.py:92-99); ratio 2,
    kernel 12."""
    ratio, ks = 2, 12
    C = x.shape[1]
    if filt_up is None:
        filt_up = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, ks, x.dtype)
    if filt_down is None:
        filt_down = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, ks, x.dtype)
    fu = filt_up.to(x.dtype).reshape(1, 1, ks).expand(C, -1, -1)
    fd = filt_down.to(x.dtype).reshape(1, 1, ks).expand(C, -1, -1)
    pad = ks // ratio - 1  # 5
    pad_left = pad * ratio + (ks - ratio) // 2  # 15
    pad_right = pad * ratio + (ks - ratio + 1) // 2  # 15
    y = F.pad(x, (pad, pad), mode="replicate")
    y = ratio * F.conv_transpose1d(y, fu, stride=ratio, groups=C)
    y = y[..., pad_left:-pad_right]
    y = snake(y, alpha, beta, logscale)
    y = F.pad(y, (ks // 2 - 1, ks // 2), mode="replicate")  # (5, 6)
    return F.conv1d(y, fd, stride=ratio, groups=C)


def upsample1d(x, ratio=2, kernel_size=None, filt=None):
    """UpSample1d.forward modules/anti_aliasing/resample.py:17-45, any ratio / kernel size."""
    ks = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size  # :20-22
    C = x.shape[1]
    if filt is None:
        filt = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, ks, x.dtype)
    pad = ks // ratio - 1  # :24
    pad_left = pad * ratio + (ks - ratio) // 2  # :25
    pad_right = pad * ratio + (ks - ratio + 1) // 2  # :26-28
    y = F.pad(x, (pad, pad), mode="replicate")  # :38
    y = ratio * F.conv_transpose1d(y, filt.to(x.dtype).reshape(1, 1, ks).expand(C, -1, -1), stride=ratio, groups=C)
    return y[..., pad_left:-pad_right]  # :42


def lowpass1d(x, filt, stride=1, padding=True, padding_mode="replicate"):
    """LowPassFilter1d.forward modules/anti_aliasing/filter.py:92-99 with a given [K] filter."""
    ks = filt.numel()
    C = x.shape[1]
    if padding:  # pad_left = k//2 - int(even), pad_right = k//2   (:78-80)
        x = F.pad(x, (ks // 2 - int(ks % 2 == 0), ks // 2), mode=padding_mode)
    return F.conv1d(x, filt.to(x.dtype).reshape(1, 1, ks).expand(C, -1, -1), stride=stride, groups=C)


def downsample1d(x, ratio=2, kernel_size=None, filt=None):
    """DownSample1d.forward resample.py:48-65: a LowPassFilter1d(0.5/ratio, 0.6/ratio) with stride = ratio."""
    ks = int(6 * ratio // 2) * 2 if kernel_size is None else kernel_size
    if filt is None:
        filt = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, ks, x.dtype)
    return lowpass1d(x, filt.reshape(-1), stride=ratio)


def bigvgan_forward(sd, hp, mel, dtype=torch.float32):
    """BigVGAN.forward bigvgan.py:313-331."""
    x = torch.as_tensor(mel).to(dtype)
    rates = list(_cfg_get(hp, "upsample_rates"))
    uks = list(_cfg_get(hp, "upsample_kernel_sizes"))
    rks = list(_cfg_get(hp, "resblock_kernel_sizes"))
    rds = [list(d) for d in _cfg_get(hp, "resblock_dilation_sizes")]
    snakebeta = _cfg_get(hp, "activation") == "snakebeta"
    if _cfg_get(hp, "activation") not in ("snake", "snakebeta"):
        raise NotImplementedError(  # bigvgan.py:132-135
            "activation incorrectly specified. check the config file and look for 'activation'."
        )
    logscale = bool(_cfg_get(hp, "snake_logscale"))
    is1 = str(_cfg_get(hp, "resblock")) == "1"
    nk = len(rks)

    def make_act(prefix):
        def act(idx, t):
            p = f"{prefix}.activations.{idx}"
            al = torch.as_tensor(sd[p + ".act.alpha"])
            be = torch.as_tensor(sd[p + ".act.beta"]) if snakebeta else None
            fu = torch.as_tensor(sd[p + ".upsample.filter"]).reshape(-1)
            fd = torch.as_tensor(sd[p + ".downsample.lowpass.filter"]).reshape(-1)
            return activation1d(t, al, be, logscale, fu, fd)

        return act

    w, b = conv_params(sd, "conv_pre", dtype)
    x = F.conv1d(x, w, b, padding=3)
    for i, (u, k) in enumerate(zip(rates, uks)):
        w, b = conv_params(sd, f"ups.{i}.0", dtype)  # nested ModuleList bigvgan.py:261-276
        x = F.conv_transpose1d(x, w, b, stride=u, padding=(k - u) // 2)  # no lrelu :316-318
        xs = None
        for j in range(nk):
            pre = f"resblocks.{i * nk + j}"
            fn = resblock1 if is1 else resblock2
            r = fn(sd, pre, x, rks[j], rds[j], dtype, act=make_act(pre))
            xs = r if xs is None else xs + r
        x = xs / nk
    al = torch.as_tensor(sd["activation_post.act.alpha"])
    be = torch.as_tensor(sd["activation_post.act.beta"]) if snakebeta else None
    fu = torch.as_tensor(sd["activation_post.upsample.filter"]).reshape(-1)
    fd = torch.as_tensor(sd["activation_post.downsample.lowpass.filter"]).reshape(-1)
    x = activation1d(x, al, be, logscale, fu, fd)  # :327
    w, b = conv_params(sd, "conv_post", dtype)
    x = F.conv1d(x, w, b, padding=3)
    return torch.tanh(x)


# ----------------------------------------------------------------------------
# Mel / STFT front end
# ----------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa 0.9.1 ``filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (htk=False,
    norm="slaney", dtype float32) -- called at utils/mel.py:66-72,133-139,199-205 and
    utils/stft.py:245-247.  librosa is not in the reference tree; this restates its
    published algorithm.  Returns float32 [n_mels, n_fft//2+1].
    """
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def hann_periodic(n, dtype=torch.float32):
    """torch.hann_window(n) default periodic=True (utils/mel.py:28,76,143)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * k / n)).to(dtype)


def _stft_center_false(y, n_fft, hop, win, dtype):
    """torch.stft(center=False, onesided, normalized=False) written as explicit
    framing * window -> rfft (utils/mel.py:153-164). y: [B, L'] -> [B, bins, F, 2]."""
    y = y.to(dtype)
    frames = y.unfold(-1, n_fft, hop)  # [B, F, n_fft]
    w = hann_periodic(win, dtype)
    if win < n_fft:  # torch.stft centre-pads the window to n_fft
        lp = (n_fft - win) // 2
        w = F.pad(w, (lp, n_fft - win - lp))
    spec = torch.fft.rfft(frames * w, n=n_fft, dim=-1)  # [B, F, bins]
    spec = spec.transpose(1, 2)
    return torch.view_as_real(spec)


def _reflect_pad(y, cfg):
    p = int((cfg.n_fft - cfg.hop_size) / 2)
    return F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)


def extract_linear_features(y, cfg, dtype=torch.float32):
    """utils/mel.py:20-52. y [B, L] -> [B, bins, F] squeezed on dim 0."""
    y = _reflect_pad(torch.as_tensor(y).to(dtype), cfg)
    spec = _stft_center_false(y, cfg.n_fft, cfg.hop_size, cfg.win_size, dtype)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.squeeze(spec, 0)


def _mel_common(y, cfg, eps, dtype):
    y = _reflect_pad(torch.as_tensor(y).to(dtype), cfg)
    spec = _stft_center_false(y, cfg.n_fft, cfg.hop_size, cfg.win_size, dtype)
    spec = torch.sqrt(spec.pow(2).sum(-1) + eps)
    basis = torch.from_numpy(
        mel_filterbank(cfg.sample_rate, cfg.n_fft, cfg.n_mel, cfg.fmin, cfg.fmax)
    ).to(dtype)
    spec = torch.matmul(basis, spec)
    return torch.log(torch.clamp(spec, min=1e-5))  # utils/mel.py:10-12


def extract_mel_features(y, cfg, dtype=torch.float32):
    """utils/mel.py:111-170 (eps 1e-9, squeeze(0))."""
    return _mel_common(y, cfg, 1e-9, dtype).squeeze(0)


def mel_spectrogram_torch(y, cfg, dtype=torch.float32):
    """utils/mel.py:55-104 (eps 1e-6, no squeeze)."""
    return _mel_common(y, cfg, 1e-6, dtype)


def amplitude_phase_spectrum(y, cfg, dtype=torch.float32):
    """utils/mel.py:244-280."""
    y = _reflect_pad(torch.as_tensor(y).to(dtype), cfg)
    st = _stft_center_false(y, cfg.n_fft, cfg.hop_size, cfg.win_size, dtype)
    if st.size(0) == 1:
        st = st.squeeze(0)
    rea, imag = st[..., 0], st[..., 1]
    log_amp = torch.log(torch.abs(torch.sqrt(rea.pow(2) + imag.pow(2))) + 1e-5)
    phase = torch.atan2(imag, rea)
    return log_amp, phase, rea, imag


def taco_stft_transform(y, filter_length, hop_length, win_length, dtype=torch.float32):
    """STFT.transform utils/stft.py:152-181: reflect-pad n_fft/2, conv1d with the
    windowed Fourier basis [2*cutoff, 1, n_fft], stride hop -> (magnitude, phase)."""
    y = torch.as_tensor(y).to(dtype)
    B, L = y.shape
    cutoff = filter_length // 2 + 1
    fb = np.fft.fft(np.eye(filter_length))
    fb = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])])  # stft.py:57-61
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)  # scipy get_window("hann", fftbins=True) :67
    lp = (filter_length - win_length) // 2  # librosa.util.pad_center :68
    w = np.pad(w, (lp, filter_length - win_length - lp))
    basis = torch.from_numpy((fb * w[None, :]).astype(np.float32)).to(dtype).unsqueeze(1)
    x = F.pad(y.view(B, 1, 1, L), (filter_length // 2, filter_length // 2, 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, basis, stride=hop_length, padding=0)
    re, im = ft[:, :cutoff, :], ft[:, cutoff:, :]
    mag = torch.sqrt(re**2 + im**2)
    phase = torch.atan2(im, re)
    return mag, phase


def window_sumsquare(n_frames, hop_length, win_length, n_fft):
    """utils/stft.py:19-75 with window="hann", norm=None, dtype=float32."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    win_sq = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)) ** 2   # scipy hann, fftbins=True
    lp = (n_fft - win_length) // 2
    win_sq = np.pad(win_sq, (lp, n_fft - win_length - lp))
    for i in range(n_frames):
        sample = i * hop_length
        x[sample:min(n, sample + n_fft)] += win_sq[:max(0, min(n_fft, n - sample))]
    return x


def taco_stft_inverse(magnitude, phase, filter_length, hop_length, win_length, dtype=torch.float32):
    """STFT.inverse utils/stft.py:183-217 with the bases of STFT.__init__ :122-147: conv_transpose1d with
    pinv(scale * [Re F; Im F]).T * window, / window_sumsquare where > tiny, * filter_length / hop, crop."""
    scale = filter_length / hop_length
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = int(filter_length / 2 + 1)
    fb = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])])
    inv = torch.FloatTensor(np.linalg.pinv(scale * fb).T[:, None, :])
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lp = (filter_length - win_length) // 2
    win = torch.from_numpy(np.pad(win, (lp, filter_length - win_length - lp))).float()
    inv = (inv * win).to(dtype)
    magnitude = torch.as_tensor(magnitude).to(dtype)
    phase = torch.as_tensor(phase).to(dtype)
    rec = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
    out = F.conv_transpose1d(rec, inv, stride=hop_length, padding=0)
    ws = window_sumsquare(magnitude.size(-1), hop_length, win_length, filter_length)
    nz = torch.from_numpy(np.where(ws > np.finfo(np.float32).tiny)[0])
    ws = torch.from_numpy(ws).to(dtype)
    out[:, :, nz] /= ws[nz]
    out *= float(filter_length) / hop_length
    out = out[:, :, int(filter_length / 2):]
    out = out[:, :, : -int(filter_length / 2)]
    return out


def taco_mel_spectrogram(y, filter_length, hop_length, win_length, n_mel, sr, fmin, fmax, dtype=torch.float32):
    """TacotronSTFT.mel_spectrogram utils/stft.py:259-278 -> (mel, energy)."""
    mag, _ = taco_stft_transform(y, filter_length, hop_length, win_length, dtype)
    energy = torch.norm(mag, dim=1)
    basis = torch.from_numpy(mel_filterbank(sr, filter_length, n_mel, fmin, fmax)).to(dtype)
    mel = torch.matmul(basis, mag)
    mel = torch.log(torch.clamp(mel, min=1e-5))  # spectral_normalize, stft.py:249-251
    return mel, energy


# ----------------------------------------------------------------------------
# configs used by BASELINE.json
# ----------------------------------------------------------------------------
def hifigan_v1_hp():
    """config/vits.json:36-71 (HiFi-GAN V1 hyper-parameters)."""
    return dict(
        resblock="1",
        upsample_rates=[8, 8, 2, 2],
        upsample_kernel_sizes=[16, 16, 4, 4],
        upsample_initial_channel=512,
        resblock_kernel_sizes=[3, 7, 11],
        resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    )


def hifigan_recipe_hp():
    """egs/vocoder/gan/hifigan/exp_config.json:14-45 (resblock "2" recipe net)."""
    return dict(
        resblock="2",
        upsample_rates=[8, 8, 4],
        upsample_kernel_sizes=[16, 16, 8],
        upsample_initial_channel=256,
        resblock_kernel_sizes=[3, 5, 7],
        resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]],
    )


def bigvgan_base_hp():
    """egs/vocoder/gan/bigvgan/exp_config.json:14-53."""
    hp = hifigan_v1_hp()
    hp.update(activation="snakebeta", snake_logscale=True)
    return hp


def preprocess_22k():
    """config/fs2.json:25-31."""
    return SimpleNamespace(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000)


def preprocess_24k():
    """config/vocoder.json:34-40."""
    return SimpleNamespace(sample_rate=24000, n_fft=1024, win_size=1024, hop_size=256, n_mel=100, fmin=0, fmax=12000)


# ----------------------------------------------------------------------------
# APNet
# ----------------------------------------------------------------------------
def apnet_recipe_hp():
    """egs/vocoder/gan/apnet/exp_config.json:16-29."""
    return dict(ASP_channel=512, ASP_resblock_kernel_sizes=[3, 7, 11], ASP_resblock_dilation_sizes=[[1, 3, 5]] * 3,
                ASP_input_conv_kernel_size=7, ASP_output_conv_kernel_size=7, PSP_channel=512,
                PSP_resblock_kernel_sizes=[3, 7, 11], PSP_resblock_dilation_sizes=[[1, 3, 5]] * 3,
                PSP_input_conv_kernel_size=7, PSP_output_R_conv_kernel_size=7, PSP_output_I_conv_kernel_size=7)


def apnet_istft_same(spec, n_fft, hop, win, window):
    """ISTFT.forward apnet.py:46-101, padding="same"."""
    pad = (win - hop) // 2
    B, N, T = spec.shape
    ifft = torch.fft.irfft(spec, n_fft, dim=1, norm="backward") * window[None, :, None]
    output_size = (T - 1) * hop + win
    y = F.fold(ifft, output_size=(1, output_size), kernel_size=(1, win), stride=(1, hop))[:, 0, 0, pad:-pad]
    wsq = window.square().expand(1, T, -1).transpose(1, 2)
    env = F.fold(wsq, output_size=(1, output_size), kernel_size=(1, win), stride=(1, hop)).squeeze()[pad:-pad]
    return y / env


def apnet_forward(sd, hp, pp, mel, dtype=torch.float32):
    """APNet.forward apnet.py:354-399 -> (logamp, pha, rea, imag, audio[B, 1, L])."""
    mel = torch.as_tensor(mel).to(dtype)

    def branch(prefix, nk_key):
        ks = list(_cfg_get(hp, f"{prefix}_resblock_kernel_sizes"))
        ds = [list(d) for d in _cfg_get(hp, f"{prefix}_resblock_dilation_sizes")]
        w, b = conv_params(sd, f"{prefix}_input_conv", dtype)
        k_in = int(_cfg_get(hp, f"{prefix}_input_conv_kernel_size"))
        x = F.conv1d(mel, w, b, padding=get_padding(k_in, 1))
        xs = None
        for j in range(len(ks)):
            r = resblock1(sd, f"{prefix}_ResNet.{j}", x, ks[j], ds[j], dtype)
            xs = r if xs is None else xs + r
        return F.leaky_relu(xs / len(ks))

    a = branch("ASP", None)
    w, b = conv_params(sd, "ASP_output_conv", dtype)
    logamp = F.conv1d(a, w, b, padding=get_padding(int(_cfg_get(hp, "ASP_output_conv_kernel_size")), 1))
    p = branch("PSP", None)
    w, b = conv_params(sd, "PSP_output_R_conv", dtype)
    R = F.conv1d(p, w, b, padding=get_padding(int(_cfg_get(hp, "PSP_output_R_conv_kernel_size")), 1))
    w, b = conv_params(sd, "PSP_output_I_conv", dtype)
    I = F.conv1d(p, w, b, padding=get_padding(int(_cfg_get(hp, "PSP_output_I_conv_kernel_size")), 1))
    pha = torch.atan2(I, R)
    rea = torch.exp(logamp) * torch.cos(pha)
    imag = torch.exp(logamp) * torch.sin(pha)
    spec = torch.view_as_complex(torch.cat((rea.unsqueeze(-1), imag.unsqueeze(-1)), -1))
    n_fft, hop, win = int(_cfg_get(pp, "n_fft")), int(_cfg_get(pp, "hop_size")), int(_cfg_get(pp, "win_size"))
    audio = apnet_istft_same(spec, n_fft, hop, win, torch.hann_window(win).to(dtype))
    return logamp, pha, rea, imag, audio.unsqueeze(1)


# ----------------------------------------------------------------------------
# NSF-HiFiGAN
# ----------------------------------------------------------------------------
def nsfhifigan_forward(sd, hp, mel, f0=None, dtype=torch.float32):
    """NSFHiFiGAN.forward models/vocoders/gan/generator/nsfhifigan.py:258-283.

    The harmonic source and ``noise_convs`` outputs are computed by the reference and then DISCARDED:
    ``x_source = x[:, :, :length]`` (:269) replaces them, so ``x = x + x_source`` doubles x and the result is
    independent of ``f0`` (and of SineGen's random draws).  ``length`` equals both lengths (T * prod(rates so
    far)), so the crops are no-ops.  The restatement therefore never touches m_source / noise_convs.
    """
    x = torch.as_tensor(mel).to(dtype)
    rates = list(_cfg_get(hp, "upsample_rates"))
    uks = list(_cfg_get(hp, "upsample_kernel_sizes"))
    rks = list(_cfg_get(hp, "resblock_kernel_sizes"))
    rds = [list(d) for d in _cfg_get(hp, "resblock_dilation_sizes")]
    rb = resblock1 if str(_cfg_get(hp, "resblock")) == "1" else resblock2
    nk = len(rks)
    w, b = conv_params(sd, "conv_pre", dtype)
    x = F.conv1d(x, w, b, padding=3)                                   # :260
    for i, (u, k) in enumerate(zip(rates, uks)):
        x = F.leaky_relu(x, LRELU_SLOPE)                               # :262
        w, b = conv_params(sd, f"ups.{i}", dtype)
        x = F.conv_transpose1d(x, w, b, stride=u, padding=(k - u) // 2)  # :263
        x = x + x                                                      # :266-271 (x_source := x)
        xs = None
        for j in range(nk):                                            # :272-277
            r = rb(sd, f"resblocks.{i * nk + j}", x, rks[j], rds[j], dtype)
            xs = r if xs is None else xs + r
        x = xs / nk                                                    # :278
    x = F.leaky_relu(x)                                                # :279
    w, b = conv_params(sd, "conv_post", dtype)
    return torch.tanh(F.conv1d(x, w, b, padding=3))                    # :280-281


# ----------------------------------------------------------------------------
# MelGAN
# ----------------------------------------------------------------------------
def melgan_recipe_hp():
    """egs/vocoder/gan/melgan/exp_config.json:14-17."""
    return dict(ratios=[8, 8, 2, 2], ngf=32, n_residual_layers=3)


def melgan_forward(sd, hp, mel, dtype=torch.float32):
    """MelGAN.forward models/vocoders/gan/generator/melgan.py:51-100 over the reference's Sequential keys
    (``model.<idx>...``): ReflectionPad1d(3) + WNConv1d(k7) ; per ratio r: LeakyReLU(0.2), WNConvTranspose1d(k=2r,
    stride r, padding r//2 + r%2, output_padding r%2), n_residual_layers x ResnetBlock(dilation 3^j) ;
    LeakyReLU(0.2), ReflectionPad1d(3), WNConv1d(ngf -> 1, k7), Tanh."""
    x = torch.as_tensor(mel).to(dtype)
    ratios = list(_cfg_get(hp, "ratios"))
    nres = int(_cfg_get(hp, "n_residual_layers"))
    idx = 0
    idx += 1                                                   # ReflectionPad1d(3)                  :56
    w, b = conv_params(sd, f"model.{idx}", dtype)
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), w, b)        # :57-62
    idx += 1
    for r in ratios:
        idx += 1                                               # LeakyReLU(0.2)                      :69
        w, b = conv_params(sd, f"model.{idx}", dtype)
        x = F.conv_transpose1d(F.leaky_relu(x, 0.2), w, b, stride=r, padding=r // 2 + r % 2, output_padding=r % 2)  # :70-77
        idx += 1
        for j in range(nres):                                  # ResnetBlock :34-48,80-83
            d = 3**j
            w1, b1 = conv_params(sd, f"model.{idx}.block.2", dtype)
            w2, b2 = conv_params(sd, f"model.{idx}.block.4", dtype)
            ws, bs = conv_params(sd, f"model.{idx}.shortcut", dtype)
            t = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (d, d), mode="reflect"), w1, b1, dilation=d)
            t = F.conv1d(F.leaky_relu(t, 0.2), w2, b2)
            x = F.conv1d(x, ws, bs) + t
            idx += 1
    idx += 2                                                   # LeakyReLU(0.2), ReflectionPad1d(3)  :91-92
    w, b = conv_params(sd, f"model.{idx}", dtype)
    x = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (3, 3), mode="reflect"), w, b)  # :93
    return torch.tanh(x)                                        # :94


# ----------------------------------------------------------------------------
# VITS posterior encoder + flow (BASELINE.json config 5)
# ----------------------------------------------------------------------------
def _j(prefix, name):
    return name if not prefix else f"{prefix}.{name}"


def sequence_mask(lengths, max_length):
    """utils/util.py:618-622"""
    lengths = torch.as_tensor(lengths)
    return torch.arange(max_length).unsqueeze(0) < lengths.unsqueeze(1)


def wn_forward(sd, prefix, x, x_mask, n_layers, hidden, kernel_size, dilation_rate, dtype, g=None):
    """WN.forward modules/flow/modules.py:126-151 (fused_add_tanh_sigmoid_multiply utils/util.py:602-609)."""
    output = torch.zeros_like(x)
    if g is not None:
        wc, bc = conv_params(sd, _j(prefix, "cond_layer"), dtype)
        g = F.conv1d(g, wc, bc)
    for i in range(n_layers):
        d = dilation_rate**i
        w, b = conv_params(sd, _j(prefix, f"in_layers.{i}"), dtype)
        x_in = F.conv1d(x, w, b, dilation=d, padding=int((kernel_size * d - d) / 2))
        g_l = g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :] if g is not None else torch.zeros_like(x_in)
        in_act = x_in + g_l
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        w, b = conv_params(sd, _j(prefix, f"res_skip_layers.{i}"), dtype)
        rs = F.conv1d(acts, w, b)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def posterior_encoder_forward(sd, prefix, y, y_lengths, noise, out_channels=192, hidden=192, n_layers=16, dtype=torch.float32, g=None):
    """PosteriorEncoder.forward models/tts/vits/vits.py:145-152 with the Gaussian noise passed in."""
    y = torch.as_tensor(y).to(dtype)
    x_mask = sequence_mask(y_lengths, y.size(2)).unsqueeze(1).to(dtype)
    w, b = conv_params(sd, _j(prefix, "pre"), dtype)
    x = F.conv1d(y, w, b) * x_mask
    x = wn_forward(sd, _j(prefix, "enc"), x, x_mask, n_layers, hidden, 5, 1, dtype, g=g)
    w, b = conv_params(sd, _j(prefix, "proj"), dtype)
    stats = F.conv1d(x, w, b) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    z = (m + torch.as_tensor(noise).to(dtype) * torch.exp(logs)) * x_mask
    return z, m, logs, x_mask


def coupling_block_forward(sd, prefix, x, x_mask, reverse=False, channels=192, hidden=192, n_flows=4, n_layers=4, dtype=torch.float32, g=None):
    """ResidualCouplingBlock.forward vits.py:105-112 with mean-only ResidualCouplingLayer
    (modules/flow/modules.py:379-397) and Flip (:314-321)."""
    x = torch.as_tensor(x).to(dtype)
    half = channels // 2

    def layer(idx, x):
        p = _j(prefix, f"flows.{idx}")
        x0, x1 = torch.split(x, [half, half], 1)
        w, b = conv_params(sd, f"{p}.pre", dtype)
        h = F.conv1d(x0, w, b) * x_mask
        h = wn_forward(sd, f"{p}.enc", h, x_mask, n_layers, hidden, 5, 1, dtype, g=g)
        w, b = conv_params(sd, f"{p}.post", dtype)
        m = F.conv1d(h, w, b) * x_mask
        logs = torch.zeros_like(m)
        if not reverse:
            x1 = m + x1 * torch.exp(logs) * x_mask
        else:
            x1 = (x1 - m) * torch.exp(-logs) * x_mask
        return torch.cat([x0, x1], 1)

    order = list(range(2 * n_flows))
    if reverse:
        order = order[::-1]
    for idx in order:
        x = layer(idx, x) if idx % 2 == 0 else torch.flip(x, [1])
    return x
