"""Pin the oracle's VITS restatement (posterior encoder + flow) to golden vectors from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(HERE, "golden", "golden_vits.npz"))


@pytest.mark.parametrize("gin", [0, 256])
def test_param_shapes(gin):
    for name, shapes in (("enc_q", synth.posterior_encoder_param_shapes(gin_channels=gin)),
                         ("flow", synth.coupling_block_param_shapes(gin_channels=gin))):
        with open(os.path.join(HERE, "golden", f"keys_vits_{name}_g{gin}.json")) as f:
            ref = [(k, tuple(s)) for k, s in json.load(f)]
        assert [(k, tuple(v)) for k, v in shapes.items()] == ref


@pytest.mark.parametrize("gin", [0, 256])
def test_posterior_encoder_and_flow(gv, gin):
    se = synth.synth_state_dict(synth.posterior_encoder_param_shapes(gin_channels=gin), 2468, g_gain=0.5)
    sf = synth.synth_state_dict(synth.coupling_block_param_shapes(gin_channels=gin), 1357, g_gain=0.5)
    t = f"vits_g{gin}_"
    g = torch.from_numpy(gv[t + "g"]) if gin else None
    with torch.no_grad():
        z, m, logs, mask = vo.posterior_encoder_forward(se, "", gv[t + "y"], gv[t + "lens"], gv[t + "noise"], g=g)
        z_p = vo.coupling_block_forward(sf, "", z, mask, g=g)
        z_hat = vo.coupling_block_forward(sf, "", z_p, mask, reverse=True, g=g)
    for name, val in (("z", z), ("m", m), ("logs", logs), ("z_p", z_p), ("z_hat", z_hat)):
        assert np.abs(val.numpy() - gv[t + name]).max() <= 2e-5, name
