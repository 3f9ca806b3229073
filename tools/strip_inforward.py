#!/usr/bin/env python
"""In-forward A/B of the persistent strip conv kernel: BigVGAN-base (BASELINE configs[2], B = 32) and HiFi-GAN V1 (configs[1], B = 64) forwards with
amp_set_conv_strip(0 | 1 | 2) [x amp_set_conv_strip_steps], alternating in ONE process on one box; prints ms per forward and whether the outputs
are bit-identical to mode 0.      python tools/strip_inforward.py [--reps 10] [--rounds 3] [--steps 0 2 4]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from types import SimpleNamespace as NS
from amphion_amd import _lib
from amphion_amd.utils.synthetic import randomize_
import bench_configs as bc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, nargs="+", default=[0])
    ap.add_argument("--configs", nargs="+", default=["c3", "c2"])
    a = ap.parse_args()
    L = _lib.lib()
    for name in a.configs:
        if name == "c3":
            from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
            hp = dict(bc.V1, activation="snakebeta", snake_logscale=True)
            m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).to(bc.DEV).eval()
            mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).to(bc.DEV)
        else:
            _, m = bc.hifigan()
            mel = (torch.randn(64, 80, 256, generator=torch.Generator().manual_seed(0)) * 2 - 5).to(bc.DEV)
        variants = [(0, 0)] + [(mode, s) for mode in (1, 2) for s in a.steps]
        ref = None
        for r in range(a.rounds):
            for mode, s in variants:
                _lib.check(L.amp_set_conv_strip(mode)); _lib.check(L.amp_set_conv_strip_steps(s))
                with torch.no_grad():
                    y = m(mel)
                    ms = bc.timed(lambda: m(mel), a.reps)
                if ref is None:
                    ref = y.clone()
                print(f"{name} round {r} strip={mode} steps={s}: {ms:.3f} ms  bitwise={bool(torch.equal(y, ref))}", flush=True)
        _lib.check(L.amp_set_conv_strip(-1)); _lib.check(L.amp_set_conv_strip_steps(0))


if __name__ == "__main__":
    main()
