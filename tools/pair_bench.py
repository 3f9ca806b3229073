#!/usr/bin/env python
"""Timing of the fused ResBlock-pair kernels through the C ABI (amp_pair_forward) at the BASELINE configs[1] stage
shapes, strip-mined vs per-tile kernel.   python tools/pair_bench.py [--reps 5] [--C 128 64] [--k 3 7 11]
Tuning aid; not part of the product."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd import _lib

SHAPES = {256: 2048, 128: 16384, 64: 32768, 32: 65536}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--C", type=int, nargs="+", default=[256, 128, 64, 32])
    ap.add_argument("--k", type=int, nargs="+", default=[3, 7, 11])
    ap.add_argument("--d", type=int, nargs="+", default=[1, 5])
    ap.add_argument("--modes", type=int, nargs="+", default=[-1, 0], help="-1 the launch policy (A-ring strips where it picks them), 0 per-tile kernel")
    a = ap.parse_args()
    _lib.set_precision("f16x3")
    L = _lib.lib()
    st = _lib.current_stream_ptr(torch.device("cuda", 0))
    print("C,k,dil,T,kernel,ms,TFLOP/s")
    for C in a.C:
        T = SHAPES[C]
        x = torch.randn(a.batch, C, T, device="cuda")
        y = torch.empty_like(x)
        for k in a.k:
            for d in a.d:
                g = torch.Generator().manual_seed(1)
                hs = []
                for dd in (d, 1):
                    w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous()
                    b = torch.randn(C, generator=g) * 0.1
                    h = ctypes.c_void_p()
                    _lib.check(L.amp_conv_create(0, C, C, k, 1, dd, (k * dd - dd) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
                    hs.append(h)
                for mode in a.modes:
                    _lib.check(L.amp_set_pair_strips(mode))
                    def go():
                        return L.amp_pair_forward(hs[0], hs[1], ctypes.c_void_p(x.data_ptr()), a.batch, T, 0.1, ctypes.c_void_p(y.data_ptr()), st)
                    if go() != 0:
                        print(f"{C},{k},{d},{T},{'policy' if mode else 'tile'},unsupported,"); continue
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        go()
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / a.reps
                    print(f"{C},{k},{d},{T},{'policy' if mode else 'tile'},{ms:.3f},{2 * 2.0 * C * C * k * a.batch * T / ms / 1e9:.1f}", flush=True)
                for h in hs:
                    L.amp_conv_destroy(h)
        del x, y
    L.amp_set_pair_strips(-1)


if __name__ == "__main__":
    main()
