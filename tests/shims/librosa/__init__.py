"""Stand-in for librosa (absent): only what utils/mel.py / utils/stft.py / utils/audio.py import."""
from . import filters, util  # noqa: F401


def load(path, sr=None, **kw):
    import torchaudio

    x, s = torchaudio.load(path)
    return x[0].numpy(), s
