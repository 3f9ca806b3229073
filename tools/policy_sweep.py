#!/usr/bin/env python
"""In-forward sweep of the launch-policy switches on ONE box, alternating rounds, outputs compared bitwise with the default's:
    python tools/policy_sweep.py --config c3 [--steps 10] [--rounds 3]
Each variant is one `amp_set_*` call away from the default policy.  Tuning aid; not part of the product."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from amphion_amd import _lib
from amphion_amd.utils.synthetic import randomize_, synthetic_mel

V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)
VARIANTS = [("default", None, None), ("conv_blk_narrow=2", "amp_set_conv_blk_narrow", 2), ("conv_blk_narrow=0", "amp_set_conv_blk_narrow", 0),
            ("conv_blk=2", "amp_set_conv_blk", 2), ("conv_blk=0", "amp_set_conv_blk", 0), ("ampblock_fusion=3", "amp_set_ampblock_fusion", 3),
            ("ampblock_fusion=2", "amp_set_ampblock_fusion", 2), ("resblock_fusion=3", "amp_set_resblock_fusion", 3), ("conv_rg_fast=0", "amp_set_conv_rg_fast", 0),
            ("pingpong=0", "amp_set_pingpong", 0), ("small_conv=0", "amp_set_small_conv", 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c2", "c3"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    if a.config == "c3":
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
        hp = dict(V1, activation="snakebeta", snake_logscale=True)
        m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
        mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).cuda()
    else:
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
        m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
        mel = synthetic_mel(64, 80, 256, seed=0).cuda()
    L = _lib.lib()
    res = {}
    with torch.no_grad():
        ref = m(mel).clone()
        for _ in range(8):
            m(mel)
        torch.cuda.synchronize()
        for rnd in range(a.rounds):
            for name, fn, val in VARIANTS:
                if fn:
                    _lib.check(getattr(L, fn)(val))
                try:
                    same = bool(torch.equal(m(mel), ref))
                    for _ in range(2):
                        m(mel)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.steps):
                        m(mel)
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(name, []).append((e0.elapsed_time(e1) / a.steps, same))
                except Exception as e:  # noqa: BLE001
                    res.setdefault(name, []).append((float("nan"), str(e)[:60]))
                if fn:
                    _lib.check(getattr(L, fn)(-1))
    print("variant," + ",".join(f"round{r}_ms" for r in range(a.rounds)) + ",bitwise")
    for name, v in res.items():
        print(name + "," + ",".join(f"{ms:.3f}" for ms, _ in v) + "," + str(all(s is True for _, s in v)))


if __name__ == "__main__":
    main()
