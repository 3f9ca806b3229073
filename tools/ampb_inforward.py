#!/usr/bin/env python
"""In-forward A/B of the AMPBlock launch policies on ONE box: config 3 (BigVGAN-base, B = 32 x 100 x 256) under
amp_set_ampblock_fusion 0 (separate conv / act1d launches) / 1 (policy) / 2 (whole-AMPBlock kernel wherever built) / 3 (four-wave
tiles at C = 32), alternating, at thermal steady state -- per-resblock GPU time from the handle's own HIP events and the kernels each
resblock ran (amp_gen_kernel_name).      python tools/ampb_inforward.py [--steps 10] [--rounds 2] [--modes 0 1 2 3]
Tuning aid; not part of the product."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from amphion_amd import _lib
from amphion_amd.utils.synthetic import randomize_, synthetic_mel

V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, activation="snakebeta", snake_logscale=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--modes", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--narrow", type=int, nargs="+", default=None, help="instead of the fusion modes: amp_set_conv_blk_narrow values to alternate (fusion = policy)")
    a = ap.parse_args()
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
    m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**V1)))), 1234, g_gain=0.75).cuda().eval()
    mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).cuda()     # tools/bench_configs.py c3
    L = _lib.lib()
    with torch.no_grad():
        for _ in range(10):
            m(mel)
        torch.cuda.synchronize()
        res = {}
        for rnd in range(a.rounds):
            for mode in (a.narrow if a.narrow is not None else a.modes):
                if a.narrow is not None:
                    _lib.check(L.amp_set_conv_blk_narrow(mode))
                else:
                    _lib.check(L.amp_set_ampblock_fusion(mode))
                for _ in range(3):
                    m(mel)
                m.set_profiling(a.steps)
                for _ in range(a.steps):
                    m(mel)
                torch.cuda.synchronize()
                r = res.setdefault(mode, {"fwd": [], "rb": {}, "names": {}})
                r["fwd"].append(sum(m.last_timing_ms(0, b) for b in range(a.steps)) / a.steps)
                for i in range(4):
                    for j in range(3):
                        w = 100 + 16 * i + j
                        r["rb"].setdefault((i, j), []).append(sum(m.last_timing_ms(w, b) for b in range(a.steps)) / a.steps)
                        r["names"][(i, j)] = " | ".join(m.kernel_names(w, 0))
                m.set_profiling(0)
    _lib.check(L.amp_set_ampblock_fusion(-1))
    _lib.check(L.amp_set_conv_blk_narrow(-1))
    print("mode," + ",".join(f"fwd_round{r}" for r in range(a.rounds)))
    for mode, r in res.items():
        print(f"{mode}," + ",".join(f"{v:.3f}" for v in r["fwd"]))
    print("stage,resblock(k)," + ",".join(f"mode{mo}_ms" for mo in res) + ",kernels per mode")
    for i in range(4):
        for j in range(3):
            row = [f"{sum(res[mo]['rb'][(i, j)]) / a.rounds:.3f}" for mo in res]
            names = " ;; ".join(f"{mo}: {res[mo]['names'][(i, j)]}" for mo in res)
            print(f"{i},{V1['resblock_kernel_sizes'][j]}," + ",".join(row) + "," + names)


if __name__ == "__main__":
    main()
