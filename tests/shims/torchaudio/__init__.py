"""Stand-in for torchaudio (absent): `save` writes PCM16 wav via the stdlib (utils/io.py:76),
`load` reads PCM16 wav."""
import wave

import numpy as np
import torch


def save(path, waveform, sample_rate, encoding="PCM_S", bits_per_sample=16, **kw):
    x = waveform.detach().cpu().float().numpy()
    if x.ndim == 2:
        x = x[0]
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(pcm.tobytes())


def load(path, **kw):
    with wave.open(str(path), "rb") as w:
        sr = w.getframerate()
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    return torch.from_numpy(x).unsqueeze(0), sr
