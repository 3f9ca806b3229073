"""CPU: the oracle's STFT.inverse / window_sumsquare restatement against golden vectors produced by the real
reference class (tests/golden/make_golden_istft.py), and the product-side host helper window_sumsquare."""
import os

import numpy as np
import pytest

from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gi():
    return np.load(os.path.join(HERE, "golden", "golden_istft.npz"))


@pytest.mark.parametrize("tag", ["n1024", "n512w400"])
def test_oracle_inverse_matches_reference(gi, tag):
    nfft, hop, win = [int(v) for v in gi[tag + "_cfg"]]
    w = vo.taco_stft_inverse(gi[tag + "_mag"], gi[tag + "_phase"], nfft, hop, win).numpy()
    assert w.shape == gi[tag + "_wav"].shape
    assert np.abs(w - gi[tag + "_wav"]).max() <= 1e-7


@pytest.mark.parametrize("tag", ["n1024", "n512w400"])
def test_window_sumsquare_matches_reference(gi, tag):
    from amphion_amd.utils.stft import window_sumsquare

    nfft, hop, win = [int(v) for v in gi[tag + "_cfg"]]
    F = gi[tag + "_mag"].shape[-1]
    assert np.array_equal(vo.window_sumsquare(F, hop, win, nfft), gi[tag + "_wss"])
    assert np.array_equal(window_sumsquare("hann", F, hop_length=hop, win_length=win, n_fft=nfft), gi[tag + "_wss"])
    with pytest.raises(NotImplementedError):
        window_sumsquare("hamming", F, hop, win, nfft)
