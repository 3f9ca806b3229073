"""Pin the oracle's front end at the transform lengths round 5 added -- 1920 (the n_fft of 24 of the reference's 38 JSON configs), 2048, odd 1001, prime
1021, 400 with a 320-sample window -- against outputs of the REAL reference (utils/mel.py:20-52,111-170, utils/stft.py:152-222), produced in the
build container by tests/golden/make_golden_nfft.py.  CPU-only."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
TAGS = ["n1920", "n2048", "n1001", "n1021", "n400w320"]


@pytest.fixture(scope="module")
def gn():
    return np.load(os.path.join(HERE, "golden", "golden_nfft.npz"))


def nfft_case(gn, tag):
    """(preprocess config, waveform batch [2, L]) of one golden case"""
    sr, nfft, hop, win, n_mel, fmin, fmax, L = [int(v) for v in gn[f"{tag}_cfg"]]
    pp = NS(sample_rate=sr, n_fft=nfft, hop_size=hop, win_size=win, n_mel=n_mel, fmin=fmin, fmax=None if fmax < 0 else fmax)
    y1 = torch.from_numpy(gn["wav_pcm16"].astype(np.float32) / 32768.0)[:L]
    return pp, torch.stack([y1, torch.roll(y1, 777) * 0.5])


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_mel_front_end_matches_reference(gn, tag):
    pp, y = nfft_case(gn, tag)
    mel = vo.extract_mel_features(y, pp).numpy()
    ref = gn[f"{tag}_mel"]
    assert mel.shape == ref.shape
    big = np.exp(ref) > 1e-3
    assert np.abs(mel - ref)[big].max() <= 1e-4
    assert np.abs(mel - ref).max() <= 1e-3
    lin = vo.extract_linear_features(y[:1], pp).numpy()
    assert lin.shape == gn[f"{tag}_linear"].shape
    assert np.abs(lin - gn[f"{tag}_linear"]).max() <= 2e-5 * max(1.0, np.abs(lin).max())
    la, ph, re, im = vo.amplitude_phase_spectrum(y, pp)
    scale = max(1.0, float(np.abs(gn[f"{tag}_re"]).max()))
    assert np.abs(re.numpy() - gn[f"{tag}_re"]).max() <= 2e-5 * scale
    assert np.abs(im.numpy() - gn[f"{tag}_im"]).max() <= 2e-5 * scale
    big = np.exp(gn[f"{tag}_logamp"]) > 1e-3
    assert np.abs(la.numpy() - gn[f"{tag}_logamp"])[big].max() <= 5e-2     # log of small bins amplifies rounding


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_stft_transform_and_inverse_match_reference(gn, tag):
    pp, y = nfft_case(gn, tag)
    mag, phase = vo.taco_stft_transform(y, pp.n_fft, pp.hop_size, pp.win_size)
    ref = gn[f"{tag}_stft_mag"]
    assert tuple(mag.shape) == ref.shape
    assert np.abs(mag.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    # the phase where the bin is not rounding noise, modulo 2 pi
    big = ref > 1e-2 * ref.max()
    d = np.angle(np.exp(1j * (phase.numpy() - gn[f"{tag}_stft_phase"])))
    assert np.abs(d[big]).max() <= 1e-3
    wav = vo.taco_stft_inverse(torch.from_numpy(gn[f"{tag}_inv_mag"]), torch.from_numpy(gn[f"{tag}_inv_phase"]), pp.n_fft, pp.hop_size, pp.win_size)
    ref = gn[f"{tag}_inv_wav"]
    assert tuple(wav.shape) == ref.shape
    assert np.abs(wav.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
