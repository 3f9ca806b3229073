"""CPU: NSF-HiFiGAN restatement vs golden vectors of the real reference class (tests/golden/make_golden_nsf.py,
which also asserts the reference output is independent of f0: nsfhifigan.py:269 overwrites the source)."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
HP = dict(vo.hifigan_v1_hp(), harmonic_num=8, upsample_initial_channel=128)


def _keys(name):
    with open(os.path.join(HERE, "golden", f"keys_{name}.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def test_param_shapes_and_module_keys_match_reference():
    from amphion_amd.models.vocoders.gan.generator.nsfhifigan import NSFHiFiGAN

    assert [(k, tuple(v)) for k, v in synth.nsfhifigan_param_shapes(80, HP).items()] == _keys("nsfhifigan")
    m = NSFHiFiGAN(NS(preprocess=NS(n_mel=80, sample_rate=22050), model=NS(nsfhifigan=NS(**HP))))
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == _keys("nsfhifigan")


@pytest.mark.parametrize("tag", ["b1_t9", "b2_t17"])
def test_oracle_matches_reference(tag):
    g = np.load(os.path.join(HERE, "golden", "golden_nsf.npz"))
    sd = synth.synth_state_dict(synth.nsfhifigan_param_shapes(80, HP), 99, g_gain=0.6)
    with torch.no_grad():
        y = vo.nsfhifigan_forward(sd, HP, g[f"nsf_{tag}_mel"], g[f"nsf_{tag}_f0"]).numpy()
    assert y.shape == g[f"nsf_{tag}_wav"].shape
    assert np.abs(y - g[f"nsf_{tag}_wav"]).max() <= 2e-6
