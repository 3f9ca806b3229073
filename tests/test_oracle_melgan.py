"""CPU: the oracle's MelGAN restatement against golden vectors of the real reference class
(tests/golden/make_golden_melgan.py) and the parameter key/shape lists."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = dict(ratios=[4, 2], ngf=16, n_residual_layers=2)


@pytest.fixture(scope="module")
def gm():
    return np.load(os.path.join(HERE, "golden", "golden_melgan.npz"))


def _keys(name):
    with open(os.path.join(HERE, "golden", f"keys_{name}.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def test_param_shapes_match_reference():
    assert [(k, tuple(v)) for k, v in synth.melgan_param_shapes(80, vo.melgan_recipe_hp()).items()] == _keys("melgan")
    assert [(k, tuple(v)) for k, v in synth.melgan_param_shapes(20, SMALL).items()] == _keys("melgan_small")


@pytest.mark.parametrize("tag", ["b1_t12", "b2_t41", "b1_t4"])
def test_melgan_recipe(gm, tag):
    hp = vo.melgan_recipe_hp()
    sd = synth.synth_state_dict(synth.melgan_param_shapes(80, hp), 2024, g_gain=0.85)
    with torch.no_grad():
        y = vo.melgan_forward(sd, hp, gm[f"melgan_{tag}_mel"]).numpy()
    ref = gm[f"melgan_{tag}_wav"]
    assert y.shape == ref.shape and np.abs(y - ref).max() <= 2e-6


def test_melgan_small(gm):
    sd = synth.synth_state_dict(synth.melgan_param_shapes(20, SMALL), 7, g_gain=0.85)
    with torch.no_grad():
        y = vo.melgan_forward(sd, SMALL, gm["melgan_small_mel"]).numpy()
    assert np.abs(y - gm["melgan_small_wav"]).max() <= 2e-6


def test_module_state_dict_keys_match_reference():
    """The drop-in MelGAN exposes exactly the reference's state_dict keys/shapes (checkpoints load unchanged)."""
    from types import SimpleNamespace as NS

    from amphion_amd.models.vocoders.gan.generator.melgan import MelGAN

    m = MelGAN(NS(preprocess=NS(n_mel=80), model=NS(melgan=NS(**vo.melgan_recipe_hp()))))
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == _keys("melgan")
    with pytest.raises(NotImplementedError):
        MelGAN(NS(preprocess=NS(n_mel=80), model=NS(melgan=NS(ratios=[5, 2], ngf=8, n_residual_layers=1))))
