#!/usr/bin/env python
"""Golden vectors for the JETS waveform decoder (models/tts/jets/jets.py:454-458,619): JETS builds
the registry's ``HiFiGAN`` from egs/vocoder/gan/hifigan/exp_config.json with ``preprocess.n_mel = attention_dim``
(256) and calls it on the up-sampled hidden states ``zs.transpose(1, 2)``.  Run in the build container only:

    python tests/golden/make_golden_jets.py

Instantiates the REAL reference class with that architecture, loads the seeded synthetic weights of oracle/synth.py
and writes golden_jets.npz (z [2, 256, 11] -> wav) next to this file."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import synth  # noqa: E402
from oracle import vocoder_oracle as vo  # noqa: E402

ATTENTION_DIM = 256   # jets.py:420


def main():
    mg.install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_recipe_hp()               # = egs/vocoder/gan/hifigan/exp_config.json model.hifigan
    cfg = mg.ns({"preprocess": {"n_mel": ATTENTION_DIM}, "model": {"hifigan": hp}})
    m = HiFiGAN(cfg)
    mg.dump_keys("hifigan_jets", m)
    mg.load_synth(m, synth.hifigan_param_shapes(ATTENTION_DIM, hp), 2024, 1.0)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, ATTENTION_DIM, 11, generator=g)
    with torch.no_grad():
        wav = m(z)
    np.savez_compressed(os.path.join(HERE, "golden_jets.npz"), z=z.numpy(), wav=wav.numpy())
    print("wav", tuple(wav.shape), float(wav.abs().max()))


if __name__ == "__main__":
    main()
