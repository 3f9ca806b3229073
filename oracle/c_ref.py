"""ctypes loader for the plain-C restatement oracle/c/vocoder_ref.c (TEST INFRASTRUCTURE)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvocoder_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "vocoder_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "c")], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def conv1d(x, w, b, dilation, padding):
    x, w, b = _f(x), _f(w), _f(b)
    B, cin, T = x.shape
    cout, _, k = w.shape
    y = np.empty((B, cout, T + 2 * padding - dilation * (k - 1)), np.float32)
    lib().ref_conv1d(_p(x), _p(w), _p(b), _p(y), B, cin, cout, T, k, dilation, padding)
    return y


def conv_transpose1d(x, w, b, stride, padding):
    x, w, b = _f(x), _f(w), _f(b)
    B, cin, T = x.shape
    _, cout, k = w.shape
    y = np.empty((B, cout, (T - 1) * stride - 2 * padding + k), np.float32)
    lib().ref_conv_transpose1d(_p(x), _p(w), _p(b), _p(y), B, cin, cout, T, k, stride, padding)
    return y


def fold_weight_norm(g, v):
    g, v = _f(g), _f(v)
    w = np.empty_like(v)
    lib().ref_fold_weight_norm(_p(g.reshape(-1)), _p(v), _p(w), v.shape[0], int(v.size // v.shape[0]))
    return w


def activation1d(x, a, b, fu, fd):
    x, a, b, fu, fd = _f(x), _f(a), _f(b), _f(fu), _f(fd)
    y = np.empty_like(x)
    B, C, T = x.shape
    lib().ref_activation1d(_p(x), _p(y), B, C, T, _p(a), _p(b), _p(fu), _p(fd))
    return y


def stft(wav, n_fft, hop, pad, window):
    wav, window = _f(wav), _f(window)
    B, L = wav.shape
    F = (L + 2 * pad - n_fft) // hop + 1
    re = np.empty((B, n_fft // 2 + 1, F), np.float32)
    im = np.empty_like(re)
    lib().ref_stft(_p(wav), B, L, n_fft, hop, pad, _p(window), _p(re), _p(im))
    return re, im


def logmel(re, im, basis, eps, clip):
    re, im, basis = _f(re), _f(im), _f(basis)
    B, bins, F = re.shape
    mel = np.empty((B, basis.shape[0], F), np.float32)
    lib().ref_logmel(_p(re), _p(im), B, bins, F, _p(basis), basis.shape[0], ctypes.c_float(eps), ctypes.c_float(clip), _p(mel))
    return mel


def pcm16(x):
    """ref_pcm16: fp32 -> int16 PCM (torchaudio 2.0.2 / libsox semantics), any shape."""
    x = _f(x)
    y = np.empty(x.shape, np.int16)
    lib().ref_pcm16(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y
