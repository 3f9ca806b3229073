#!/usr/bin/env python
"""Where concurrent resblock streams stop paying: HiFi-GAN V1 forward at B x 256 frames, sequential (amp_set_resblock_streams(0)) against
concurrent (1), eager, HIP-event timed.  Sets kRbStreamsMaxFrames (generator.hip).   python tools/streams_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from amphion_amd import _lib
from amphion_amd.utils.synthetic import synthetic_mel

cfg, m = bc.hifigan()
L = _lib.lib()
print(f"{'B':>3} {'frames':>7} {'sequential ms':>14} {'concurrent ms':>14} {'ratio':>6}")
with torch.no_grad():
    for B in [int(b) for b in os.environ.get("SWEEP_B", "1,2,4,8,12,16,24,32").split(",")]:
        mel = synthetic_mel(B, 80, 256, seed=5).to(bc.DEV)
        res = []
        for mode in (0, 1):
            _lib.check(L.amp_set_resblock_streams(mode))
            res.append(bc.timed(lambda: m(mel), 20))
        print(f"{B:3d} {B * 256:7d} {res[0]:14.3f} {res[1]:14.3f} {res[1] / res[0]:6.3f}", flush=True)
_lib.check(L.amp_set_resblock_streams(-1))
