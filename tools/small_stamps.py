#!/usr/bin/env python
"""Phase timeline of conv_small3_kernel (the C = 256 stage of a single utterance: three whole-K convs, k = 11 / 7 / 3, in one grid) from clock
stamps of every wave.  Experiment build: apply profiles/negative_kernels/r5_conv_small_stamps.patch,
    AMP_BUILD_TAG=cs AMP_BUILD_FLAGS=-DCS_TIMING python -m amphion_amd.build
    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_cs.so python tools/small_stamps.py
Stamps: 0 entry, 1 x tile + first A fragments requested, 2 accumulators (bias / residual) ready, 3 x arrived, converted, in LDS, 4 barrier,
8 / 9 K loop after 2 / 4 groups of chunks, 5 K loop issued, 6 stores issued, 7 stores have left.  Tuning aid; not part of the product."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from amphion_amd import _lib  # noqa: E402
import bench_configs as bc  # noqa: E402
from amphion_amd.utils.synthetic import synthetic_mel  # noqa: E402

L = _lib.lib()
fn = getattr(L, "amp_debug_cs_stamps", None)
if fn is None:
    sys.exit("this library has no stamps (build with -DCS_TIMING, see the docstring)")
cfg, m = bc.hifigan()
mel = synthetic_mel(1, 80, 256, seed=5).cuda()
with torch.no_grad():
    for _ in range(6):
        m(mel)
torch.cuda.synchronize()
buf = np.zeros(8 * 512 * 4 * 16, dtype=np.int64)
assert fn(ctypes.c_void_p(buf.ctypes.data)) == 0
S = buf.reshape(8, 512, 4, 16)
for d, slots in ((1, ((0, 128, "k=11 NI=1"), (128, 256, "k=7"), (256, 384, "k=3"))), (3, ((0, 128, "k=11 NI=1"), (128, 256, "k=7"), (256, 384, "k=3"))),
                 (5, ((0, 64, "k=11 NI=2"), (64, 192, "k=7"), (192, 320, "k=3")))):
    A = S[d]
    live = A[:, 0, 0] > 0
    if not live.any():
        continue
    t0 = A[live][:, :, 0].min()
    print(f"# conv_small3 launch with dilation {d} of slot 0: {int(live.sum())} workgroups; span first entry -> last 'stores have left' "
          f"{A[live][:, :, 7].max() - t0} ticks (counters of different XCDs are not synchronised: indicative)")
    print(f"{'slot':12s} {'wgs':>4s} | {'start':>7s} {'->req':>7s} {'->acc':>7s} {'->lds':>7s} {'->bar':>7s} {'K 0-2':>7s} {'K 2-4':>7s} {'K rest':>7s} {'->st':>7s} {'->left':>7s} | {'life':>7s}   (median ticks per wave; start = entry after the launch's first entry)")
    for lo, hi, name in slots:
        W = A[lo:hi]
        W = W[W[:, 0, 0] > 0]
        if not len(W):
            continue
        def med(a, b):
            return float(np.median((W[:, :, b] - W[:, :, a])))
        start = float(np.median(W[:, :, 0] - t0))
        print(f"{name:12s} {len(W):4d} | {start:7.0f} {med(0, 1):7.0f} {med(1, 2):7.0f} {med(2, 3):7.0f} {med(3, 4):7.0f} {med(4, 8):7.0f} {med(8, 9):7.0f} {med(9, 5):7.0f} {med(5, 6):7.0f} {med(6, 7):7.0f} | {med(0, 7):7.0f}")
