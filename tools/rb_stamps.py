#!/usr/bin/env python
"""Phase timeline of rb_f16x3_kernel (a whole ResBlock1 per launch: 12 of the headline's 28 ms) from clock stamps of every wave.
Experiment build: apply profiles/negative_kernels/r5_rb_stamps.patch,
    AMP_BUILD_TAG=rbt AMP_BUILD_FLAGS=-DRB_TIMING python -m amphion_amd.build
    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_rbt.so python tools/rb_stamps.py
Stamps: 0 entry, 1 x tile in registers, 2 staged, 3 barrier; pair p: 4+8p c1's K loop issued, 5 barrier, 6 seam written, 7 barrier, 8 c2's K loop
issued, 9 residual fma (10 barrier, 11 x staged + barrier when another pair follows); 27 before the epilogue, 28 stores issued, 29 stores have left.
Tuning aid; not part of the product."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd import _lib  # noqa: E402

L = _lib.lib(); _lib.set_precision("f16x3")
_lib.check(L.amp_set_resblock_fusion(2))
st = _lib.current_stream_ptr(torch.device("cuda", 0))
p = lambda t: ctypes.c_void_p(t.data_ptr())


def conv(C, k, dil, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous(); b = torch.randn(C, generator=g) * 0.1
    h = ctypes.c_void_p()
    _lib.check(L.amp_conv_create(0, C, C, k, 1, dil, (k * dil - dil) // 2, p(w), p(b), ctypes.byref(h)))
    return h


B = 64
for C, k, T, waves, ni in ((64, 11, 32768, 8, 4), (64, 7, 32768, 8, 4), (32, 11, 65536, 8, 4), (32, 3, 65536, 4, 4), (128, 3, 16384, 8, 4)):
    fn = getattr(L, f"amp_debug_rb_stamps_kt{k}", None)
    if fn is None:
        sys.exit("this library has no stamps (build with -DRB_TIMING, see the docstring)")
    h1 = [conv(C, k, d, 10 + d) for d in (1, 3, 5)]; h2 = [conv(C, k, 1, 20 + d) for d in (1, 3, 5)]
    arr = lambda hs: (ctypes.c_void_p * len(hs))(*[h.value for h in hs])
    x = torch.randn(B, C, T, device="cuda") * 0.5; y = torch.empty_like(x)
    for lo, hi in ((0, 3), (0, 2)) if (C, k) == (64, 11) else ((0, 3),):
        a1, a2, n = arr(h1[lo:hi]), arr(h2[lo:hi]), hi - lo
        for _ in range(3):
            _lib.check(L.amp_resblock_forward(a1, a2, n, p(x), B, T, 0.1, p(y), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.amp_resblock_forward(a1, a2, n, p(x), B, T, 0.1, p(y), st)); e1.record(); torch.cuda.synchronize()
        buf = np.zeros(8192 * 8 * 32, dtype=np.int64)
        assert fn(ctypes.c_void_p(buf.ctypes.data)) == 0
        S = buf.reshape(8192, 8, 32)
        S = S[S[:, 0, 29] > 0][:, :waves, :]
        d = lambda a, b: float(np.median(S[:, :, b] - S[:, :, a]))
        nch = C // 16
        mf = nch * k * 3 * ni                       # MFMAs per conv per wave
        life = d(0, 29)
        kl = []; bar = 0.0; seam = 0.0; restage = 0.0
        for q in range(n):
            start1 = 3 if q == 0 else 11 + 8 * (q - 1)
            kl += [d(start1, 4 + 8 * q), d(7 + 8 * q, 8 + 8 * q)]
            bar += d(4 + 8 * q, 5 + 8 * q) + d(6 + 8 * q, 7 + 8 * q)
            seam += d(5 + 8 * q, 6 + 8 * q) + d(8 + 8 * q, 9 + 8 * q) if q < n - 1 else d(5 + 8 * q, 6 + 8 * q)
            if q < n - 1:
                bar += d(9 + 8 * q, 10 + 8 * q)
                restage += d(10 + 8 * q, 11 + 8 * q)
        last_c2 = 8 + 8 * (n - 1)
        print(f"rb C={C} k={k} T={T} B={B}, pairs [{lo}, {hi}): launch {e0.elapsed_time(e1) * 1e3:.0f} us, {len(S)} workgroups x {waves} waves; wave life {life:.0f} cycles\n"
              f"   load x + stage {d(0, 2):.0f} (+ barrier {d(2, 3):.0f}); K loops {' '.join(f'{v:.0f}' for v in kl)} = {sum(kl):.0f} ({sum(kl) / life:.0%}; {mf} MFMAs each: "
              f"{mf * 32} nominal, {mf * 35.8:.0f} at the microbenchmark's issue rate; two waves share a SIMD at C = 64 / 128, four at the 2-per-CU forms);\n"
              f"   barriers {bar:.0f} ({bar / life:.0%}), seams + residual fma {seam:.0f} ({seam / life:.0%}), re-staging x + barrier {restage:.0f} ({restage / life:.0%}), "
              f"epilogue {d(last_c2, 28):.0f} + stores leaving {d(28, 29):.0f} ({d(last_c2, 29) / life:.0%})", flush=True)
        # which waves are late?  K-loop duration of pair 0's c1 per wave index (wave = wm * WN + wn), and the spread inside a workgroup
        k0 = (S[:, :, 4] - S[:, :, 3]).astype(np.float64)
        arrive = (S[:, :, 4] - S[:, :, 4].min(axis=1, keepdims=True)).astype(np.float64)
        print("   c1 of pair 0, per wave index: K-loop cycles " + " ".join(f"{np.median(k0[:, w]):.0f}" for w in range(waves))
              + f"; arrival at the barrier after the first wave: " + " ".join(f"{np.median(arrive[:, w]):.0f}" for w in range(waves))
              + f"; last - first arrival, median over workgroups {np.median(arrive.max(axis=1)):.0f}", flush=True)
    for h in h1 + h2:
        L.amp_conv_destroy(h)
