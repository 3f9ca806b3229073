#!/usr/bin/env python
"""Golden vectors for NSF-HiFiGAN from the REAL reference class (models/vocoders/gan/generator/nsfhifigan.py),
CPU, build container only:   python tests/golden/make_golden_nsf.py -> golden_nsf.npz, keys_nsfhifigan.json
Also records that the reference output does not depend on f0 (nsfhifigan.py:269 overwrites the source)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import synth  # noqa: E402
from oracle import vocoder_oracle as vo  # noqa: E402


def main():
    mg.install_stubs()
    torch.manual_seed(0)
    from models.vocoders.gan.generator.nsfhifigan import NSFHiFiGAN

    hp = dict(vo.hifigan_v1_hp(), harmonic_num=8, upsample_initial_channel=128)
    cfg = mg.ns({"preprocess": {"n_mel": 80, "sample_rate": 22050}, "model": {"nsfhifigan": hp}})
    m = NSFHiFiGAN(cfg)
    mg.dump_keys("nsfhifigan", m)
    mg.load_synth(m, synth.nsfhifigan_param_shapes(80, hp), 99, 0.6)
    out = {}
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for tag, (B, T, seed) in {"b1_t9": (1, 9, 0), "b2_t17": (2, 17, 1)}.items():
            mel = synth.synth_mel(B, 80, T, seed)
            f0 = torch.rand(B, T, generator=g) * 300 + 80
            y1 = m(mel, f0)
            y2 = m(mel, torch.zeros(B, T))          # different f0, different SineGen noise draws
            assert torch.equal(y1, y2), "reference output depends on f0?"
            out[f"nsf_{tag}_mel"] = mel.numpy()
            out[f"nsf_{tag}_f0"] = f0.numpy()
            out[f"nsf_{tag}_wav"] = y1.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_nsf.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()), float(v.std()))


if __name__ == "__main__":
    main()
