"""Pin the oracle restatement against golden vectors produced by the REAL reference
classes (tests/golden/make_golden.py).  CPU-only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 2e-6  # oracle and reference run the same torch CPU kernels; only op grouping differs


def _keys(name):
    with open(os.path.join(HERE, "golden", f"keys_{name}.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


@pytest.mark.parametrize(
    "name,shapes",
    [
        ("hifigan_v1", lambda: synth.hifigan_param_shapes(80, vo.hifigan_v1_hp())),
        ("hifigan_recipe", lambda: synth.hifigan_param_shapes(100, vo.hifigan_recipe_hp())),
        ("hifigan_vits_g0", lambda: synth.hifigan_param_shapes(192, vo.hifigan_v1_hp(), vits=True)),
        ("hifigan_vits_g256", lambda: synth.hifigan_param_shapes(192, vo.hifigan_v1_hp(), vits=True, gin_channels=256)),
        ("bigvgan_base", lambda: synth.bigvgan_param_shapes(100, vo.bigvgan_base_hp())),
    ],
)
def test_param_shapes_match_reference_state_dict(name, shapes):
    assert [(k, tuple(v)) for k, v in shapes().items()] == _keys(name)


@pytest.mark.parametrize("tag", ["b1_t8", "b2_t33", "b3_t1"])
def test_hifigan_v1(golden, tag):
    hp = vo.hifigan_v1_hp()
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
    with torch.no_grad():
        y = vo.hifigan_forward(sd, hp, golden[f"hifigan_v1_{tag}_mel"])
    ref = golden[f"hifigan_v1_{tag}_wav"]
    assert y.shape == ref.shape
    assert np.abs(y.numpy() - ref).max() <= TOL


def test_hifigan_recipe_resblock2(golden):
    hp = vo.hifigan_recipe_hp()
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(100, hp), 77)
    with torch.no_grad():
        y = vo.hifigan_forward(sd, hp, golden["hifigan_recipe_b2_t19_mel"])
    assert np.abs(y.numpy() - golden["hifigan_recipe_b2_t19_wav"]).max() <= TOL


@pytest.mark.parametrize("gin", [0, 256])
def test_hifigan_vits(golden, gin):
    hp = vo.hifigan_v1_hp()
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(192, hp, vits=True, gin_channels=gin), 4321)
    g = golden[f"hifigan_vits_g{gin}_g"] if gin else None
    with torch.no_grad():
        y = vo.hifigan_forward(sd, hp, golden[f"hifigan_vits_g{gin}_z"], g=g)
    assert np.abs(y.numpy() - golden[f"hifigan_vits_g{gin}_wav"]).max() <= TOL


@pytest.mark.parametrize("tag", ["b1_t8", "b2_t13"])
def test_bigvgan_base(golden, tag):
    hp = vo.bigvgan_base_hp()
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75)
    with torch.no_grad():
        y = vo.bigvgan_forward(sd, hp, golden[f"bigvgan_base_{tag}_mel"])
    assert np.abs(y.numpy() - golden[f"bigvgan_base_{tag}_wav"]).max() <= 5e-6


def test_bigvgan_small_ampblock2_snake_linear(golden):
    hp = dict(resblock="2", activation="snake", snake_logscale=False, upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4],
              upsample_initial_channel=64, resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]])
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(20, hp), seed=9, g_gain=0.75)
    for k in sd:
        if k.endswith(".alpha"):
            sd[k] = sd[k].abs() + 0.5
    with torch.no_grad():
        y = vo.bigvgan_forward(sd, hp, golden["bigvgan_small_mel"])
    assert np.abs(y.numpy() - golden["bigvgan_small_wav"]).max() <= 5e-6


def test_bigvgan_bad_activation_raises():
    hp = dict(vo.bigvgan_base_hp(), activation="gelu")
    with pytest.raises(NotImplementedError):
        vo.bigvgan_forward({}, hp, np.zeros((1, 100, 4), np.float32))


def test_kaiser_filter_and_activation1d(golden):
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    assert np.abs(f.numpy() - golden["act1d_filter"]).max() <= 1e-8
    x = torch.from_numpy(golden["act1d_x"])
    y = vo.activation1d(x, torch.from_numpy(golden["act1d_snakebeta_log_alpha"]), torch.from_numpy(golden["act1d_snakebeta_log_beta"]), True)
    assert np.abs(y.numpy() - golden["act1d_snakebeta_log_y"]).max() <= 1e-6
    y = vo.activation1d(x, torch.from_numpy(golden["act1d_snake_lin_alpha"]), None, False)
    assert np.abs(y.numpy() - golden["act1d_snake_lin_y"]).max() <= 1e-6
    s = vo.snake(x, torch.from_numpy(golden["act1d_snake_lin_alpha"]), None, False)
    assert np.abs(s.numpy() - golden["act1d_snake_lin_snake_only"]).max() <= 1e-6
    y1 = vo.activation1d(torch.from_numpy(golden["act1d_T1_x"]), torch.from_numpy(golden["act1d_snake_lin_alpha"]), None, False)
    assert np.abs(y1.numpy() - golden["act1d_T1_y"]).max() <= 1e-6


@pytest.mark.parametrize("tag,pp", [("22k", vo.preprocess_22k()), ("24k", vo.preprocess_24k())])
def test_mel_front_end(golden, tag, pp):
    # mel filterbank restatement vs the independent transformers implementation used by the golden run
    mb = vo.mel_filterbank(pp.sample_rate, pp.n_fft, pp.n_mel, pp.fmin, pp.fmax)
    assert np.abs(mb - golden[f"melbasis_{tag}"]).max() <= 1e-8
    y = torch.from_numpy(golden["wav_pcm16"].astype(np.float32) / 32768.0).unsqueeze(0)
    y2 = torch.stack([y[0], torch.roll(y[0], 777) * 0.5])
    assert np.abs(vo.extract_mel_features(y, pp).numpy() - golden[f"mel_{tag}_extract"]).max() <= 2e-4
    assert np.abs(vo.mel_spectrogram_torch(y2, pp).numpy() - golden[f"mel_{tag}_melspec_b2"]).max() <= 2e-4
    lin = vo.extract_linear_features(y, pp).numpy()
    assert np.abs(lin - golden[f"mel_{tag}_linear"]).max() <= 2e-5 * max(1.0, np.abs(lin).max())
    la, ph, re, im = vo.amplitude_phase_spectrum(y2, pp)
    scale = max(1.0, float(np.abs(golden[f"mel_{tag}_re"]).max()))
    assert np.abs(re.numpy() - golden[f"mel_{tag}_re"]).max() <= 2e-5 * scale
    assert np.abs(im.numpy() - golden[f"mel_{tag}_im"]).max() <= 2e-5 * scale
    assert np.abs(la.numpy() - golden[f"mel_{tag}_logamp"]).max() <= 5e-2  # log of tiny bins amplifies rounding
    mel, energy = vo.taco_mel_spectrogram(y2, pp.n_fft, pp.hop_size, pp.win_size, pp.n_mel, pp.sample_rate, pp.fmin, pp.fmax)
    assert np.abs(mel.numpy() - golden[f"taco_{tag}_mel"]).max() <= 2e-4
    assert np.abs(energy.numpy() - golden[f"taco_{tag}_energy"]).max() <= 2e-4 * max(1.0, float(energy.max()))


def test_oracle_jets_waveform_decoder():
    """The oracle on the JETS decoder architecture (jets.py:454-458: recipe HiFi-GAN, n_mel = attention_dim = 256)
    against golden vectors of the real reference class (tests/golden/make_golden_jets.py)."""
    import json

    g = np.load(os.path.join(HERE, "golden", "golden_jets.npz"))
    hp = vo.hifigan_recipe_hp()
    shapes = synth.hifigan_param_shapes(256, hp)
    with open(os.path.join(HERE, "golden", "keys_hifigan_jets.json")) as f:
        assert [(k, tuple(v)) for k, v in json.load(f)] == [(k, tuple(v)) for k, v in shapes.items()]
    sd = synth.synth_state_dict(shapes, 2024, g_gain=1.0)
    y = vo.hifigan_forward(sd, hp, torch.from_numpy(g["z"])).numpy()
    assert np.abs(y - g["wav"]).max() <= 2e-6
