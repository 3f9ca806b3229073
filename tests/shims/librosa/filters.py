import numpy as np


def mel(sr=None, n_fft=None, n_mels=128, fmin=0.0, fmax=None, **kw):
    from transformers.audio_utils import mel_filter_bank

    if fmax is None:
        fmax = sr / 2
    m = mel_filter_bank(n_fft // 2 + 1, n_mels, float(fmin), float(fmax), sr, norm="slaney", mel_scale="slaney")
    return m.T.astype(np.float32)
