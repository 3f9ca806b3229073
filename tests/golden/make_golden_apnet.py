#!/usr/bin/env python
"""Golden vectors for APNet from the REAL reference class (models/vocoders/gan/generator/apnet.py), CPU, build
container only:   python tests/golden/make_golden_apnet.py -> golden_apnet.npz, keys_apnet.json"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import synth  # noqa: E402
from oracle import vocoder_oracle as vo  # noqa: E402

# a narrower net than the recipe's 512 channels keeps the fixture small; same structure
HP = dict(vo.apnet_recipe_hp(), ASP_channel=96, PSP_channel=64)
PP = dict(n_mel=80, n_fft=1024, hop_size=256, win_size=1024)


def main():
    mg.install_stubs()
    torch.manual_seed(0)
    from models.vocoders.gan.generator.apnet import APNet

    m = APNet(mg.ns({"preprocess": PP, "model": {"apnet": HP}}))
    mg.dump_keys("apnet", m)
    mg.load_synth(m, synth.apnet_param_shapes(80, 1024, HP), 321, 0.45)
    out = {}
    with torch.no_grad():
        for tag, (B, T, seed) in {"b1_t10": (1, 10, 0), "b2_t27": (2, 27, 1)}.items():
            mel = synth.synth_mel(B, 80, T, seed)
            logamp, pha, rea, imag, audio = m(mel)
            out[f"apnet_{tag}_mel"] = mel.numpy()
            for n, v in (("logamp", logamp), ("pha", pha), ("rea", rea), ("imag", imag), ("audio", audio)):
                out[f"apnet_{tag}_{n}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "golden_apnet.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()), float(v.std()))


if __name__ == "__main__":
    main()
