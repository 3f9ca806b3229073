#!/usr/bin/env python
"""Golden vectors for MelGAN from the REAL reference class (models/vocoders/gan/generator/melgan.py), run on
CPU in the build container (needs /root/reference):   python tests/golden/make_golden_melgan.py
-> tests/golden/golden_melgan.npz, keys_melgan.json"""
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
    from models.vocoders.gan.generator.melgan import MelGAN

    out = {}
    hp = vo.melgan_recipe_hp()
    cfg = mg.ns({"preprocess": {"n_mel": 80}, "model": {"melgan": hp}})
    m = MelGAN(cfg)
    mg.dump_keys("melgan", m)
    mg.load_synth(m, synth.melgan_param_shapes(80, hp), 2024, 0.85)
    with torch.no_grad():
        for tag, (B, T, seed) in {"b1_t12": (1, 12, 0), "b2_t41": (2, 41, 1), "b1_t4": (1, 4, 2)}.items():
            mel = synth.synth_mel(B, 80, T, seed)
            out[f"melgan_{tag}_mel"] = mel.numpy()
            out[f"melgan_{tag}_wav"] = m(mel).numpy()
    # a small net with different ratios / widths
    hp2 = dict(ratios=[4, 2], ngf=16, n_residual_layers=2)
    m2 = MelGAN(mg.ns({"preprocess": {"n_mel": 20}, "model": {"melgan": hp2}}))
    mg.dump_keys("melgan_small", m2)
    mg.load_synth(m2, synth.melgan_param_shapes(20, hp2), 7, 0.85)
    with torch.no_grad():
        mel = synth.synth_mel(2, 20, 23, 5)
        out["melgan_small_mel"] = mel.numpy()
        out["melgan_small_wav"] = m2(mel).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_melgan.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()), float(v.std()), float((np.abs(v) > 0.99).mean()))


if __name__ == "__main__":
    main()
