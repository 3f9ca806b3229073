#!/usr/bin/env python
"""Does the 256-MB Infinity Cache pay for BigVGAN's HBM-bound launches (49 Activation1d passes at 0.62 of the HBM peak, 48 unfused convs)?
Config 3 (BigVGAN-base, B = 32 x 100 x 256) with the batch run depth-first in groups (amp_set_group_mb: every tensor of a group is written and read back
while it may still be in the cache), alternating with the ungrouped forward on ONE box; per-stage MRF time from the handle's HIP events.
    python tools/mall_probe.py [--steps 10] [--rounds 2] [--mb 0 700 350 175]       Tuning aid; not part of the product."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from amphion_amd import _lib
from amphion_amd.utils.synthetic import randomize_

V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--mb", type=int, nargs="+", default=[0, 700, 350, 175])
    a = ap.parse_args()
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
    hp = dict(V1, activation="snakebeta", snake_logscale=True)
    m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
    mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).cuda()
    L = _lib.lib()
    with torch.no_grad():
        ref = m(mel).clone()
        for _ in range(5):
            m(mel)
        torch.cuda.synchronize()
        print("group_mb,round,fwd_ms,wall_ms,stage0,stage1,stage2,stage3,bitwise")
        for rnd in range(a.rounds):
            for mb in a.mb:
                _lib.check(L.amp_set_group_mb(mb))
                y = m(mel)
                same = bool(torch.equal(y, ref))
                for _ in range(2):
                    m(mel)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    m(mel)
                e1.record(); torch.cuda.synchronize()
                wall = e0.elapsed_time(e1) / a.steps
                if mb == 0:
                    m.set_profiling(a.steps)
                    for _ in range(a.steps):
                        m(mel)
                    torch.cuda.synchronize()
                    fwd = sum(m.last_timing_ms(0, b) for b in range(a.steps)) / a.steps
                    stg = [sum(m.last_timing_ms(2 + i, b) for b in range(a.steps)) / a.steps for i in range(4)]
                    m.set_profiling(0)
                    print(f"{mb},{rnd},{fwd:.3f},{wall:.3f}," + ",".join(f"{v:.3f}" for v in stg) + f",{same}", flush=True)
                else:
                    print(f"{mb},{rnd},,{wall:.3f},,,,,{same}", flush=True)
    _lib.check(L.amp_set_group_mb(0))


if __name__ == "__main__":
    main()
