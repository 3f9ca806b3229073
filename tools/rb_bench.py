#!/usr/bin/env python
"""Whole-ResBlock1 kernel (amp_resblock_forward, csrc/rb_f16x3.hip) against the three fused pairs it replaces
(amp_pair_forward x 3) at the BASELINE configs[1] stage shapes, through the C ABI.
    python tools/rb_bench.py [--reps 5] [--C 32 64] [--k 3 7] [--modes 2 3]
Tuning aid; not part of the product."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd import _lib

SHAPES = {128: 16384, 64: 32768, 32: 65536}
DILS = (1, 3, 5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--C", type=int, nargs="+", default=[32, 64])
    ap.add_argument("--k", type=int, nargs="+", default=[3, 7, 11])
    ap.add_argument("--modes", type=int, nargs="+", default=[2, 3], help="amp_set_resblock_fusion modes to time (2: wide tiles, 3: four-wave tiles at C = 32)")
    a = ap.parse_args()
    _lib.set_precision("f16x3")
    L = _lib.lib()
    st = _lib.current_stream_ptr(torch.device("cuda", 0))
    print("C,k,T,kernel,ms,TFLOP/s,GB/s_alg")
    for C in a.C:
        T = SHAPES[C]
        x = torch.randn(a.batch, C, T, device="cuda")
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        for k in a.k:
            g = torch.Generator().manual_seed(1)
            h1, h2 = [], []
            for hs, ds in ((h1, DILS), (h2, (1, 1, 1))):
                for d in ds:
                    w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous()
                    b = torch.randn(C, generator=g) * 0.1
                    h = ctypes.c_void_p()
                    _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
                    hs.append(h)
            a1, a2 = (ctypes.c_void_p * 3)(*[h.value for h in h1]), (ctypes.c_void_p * 3)(*[h.value for h in h2])
            flop = 6 * 2.0 * C * C * k * a.batch * T
            byts = 2 * 4.0 * a.batch * C * T

            def pairs():
                bufs = [x, y, z]
                for p in range(3):
                    rc = L.amp_pair_forward(h1[p], h2[p], ctypes.c_void_p(bufs[p % 3].data_ptr()), a.batch, T, 0.1, ctypes.c_void_p(bufs[(p + 1) % 3].data_ptr()), st)
                    if rc:
                        return rc
                return 0

            def rb():
                return L.amp_resblock_forward(a1, a2, 3, ctypes.c_void_p(x.data_ptr()), a.batch, T, 0.1, ctypes.c_void_p(y.data_ptr()), st)

            runs = [("3 pairs (warm-up)", pairs, None), ("3 pairs", pairs, None)] + [(f"resblock mode {m}", rb, m) for m in a.modes]
            for name, fn, mode in runs:
                if mode is not None:
                    _lib.check(L.amp_set_resblock_fusion(mode))
                if fn() != 0:
                    print(f"{C},{k},{T},{name},unsupported,,"); continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.reps
                print(f"{C},{k},{T},{name},{ms:.3f},{flop / ms / 1e9:.1f},{byts * (3 if mode is None else 1) / ms / 1e6:.0f}", flush=True)
            for h in h1 + h2:
                L.amp_conv_destroy(h)
        del x, y, z
    L.amp_set_resblock_fusion(-1)


if __name__ == "__main__":
    main()
