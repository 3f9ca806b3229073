#!/usr/bin/env python
"""C3 (BigVGAN-base, B = 32 x 256 frames) with and without conv + Activation1d in one launch (amp_set_fuse_act), alternating."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
from amphion_amd import _lib  # noqa: E402
from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN  # noqa: E402

NS = bc.NS
hp = dict(bc.V1, activation="snakebeta", snake_logscale=True)
m = bc.randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).to(bc.DEV).eval()
mel = torch.randn(32, 100, 256, generator=torch.Generator().manual_seed(0)).to(bc.DEV)
outs = {}
for rep in range(3):
    for on in (0, 1):
        _lib.lib().amp_set_fuse_act(on)
        outs[on] = m(mel)
        print("fuse_act", on, "%.2f ms" % bc.timed(lambda: m(mel), 10), flush=True)
print("fused == unfused bitwise:", bool(torch.equal(outs[0], outs[1])))
