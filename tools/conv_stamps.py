#!/usr/bin/env python
"""Phase timeline of conv_f16x3_kernel from in-kernel clock stamps (experiment build: copy profiles/negative_kernels/conv_f16x3_stamps.hip.txt over csrc/conv_f16x3.hip, then
AMP_BUILD_TAG=ct AMP_BUILD_FLAGS=-DCONV_TIMING python -m amphion_amd.build; run with AMP_LIB_PATH pointing at the ct library)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd import _lib  # noqa: E402

L = _lib.lib()
for (B, C, T, k, d) in ((32, 32, 65536, 3, 1), (32, 32, 65536, 7, 1), (32, 64, 32768, 3, 1), (32, 128, 16384, 3, 1), (32, 128, 16384, 11, 1)):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(C, C, k, generator=g) * 0.05
    bias = torch.randn(C, generator=g) * 0.1
    h = ctypes.c_void_p()
    _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(bias.data_ptr()), ctypes.byref(h)))
    x = torch.randn(B, C, T, device="cuda")
    y = torch.empty_like(x)
    st = _lib.current_stream_ptr(x.device)
    fn = getattr(L, f"amp_debug_conv_stamps_kt{k}", None)      # absent in a regular build: launch times only
    times = []
    for rep in range(3 if fn is not None else 30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.amp_conv_forward(h, ctypes.c_void_p(x.data_ptr()), B, T, 1.0, None, 1.0, ctypes.c_void_p(y.data_ptr()), st))
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3)
        if fn is None:
            continue
        buf = (ctypes.c_longlong * 32)()
        fn(buf)
        t = np.array(buf[:32], dtype=np.int64)
        nch = C // 16
        chunks = [(int(t[5 + 3 * c] - (t[4] if c == 0 else t[7 + 3 * (c - 1)])), int(t[6 + 3 * c] - t[5 + 3 * c]), int(t[7 + 3 * c] - t[6 + 3 * c])) for c in range(min(nch, 8))]
        print(f"C={C} k={k} T={T}: launch {a.elapsed_time(b) * 1e3:.0f} us; workgroup 7: total {t[30] - t[0]} ticks: setup {t[1] - t[0]}, issue loads {t[2] - t[1]}, "
              f"wait+stage0 {t[3] - t[2]}, barrier {t[4] - t[3]}, chunks (mfma+issue, stage next, barrier) {chunks}, epilogue {t[30] - t[29]}")
    if fn is None:
        times.sort()
        print(f"C={C} k={k} T={T}: launch median {times[len(times) // 2]:.1f} us (min {times[0]:.1f}) over {len(times)}")
    L.amp_conv_destroy(h)
