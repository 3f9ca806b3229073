#!/usr/bin/env python
"""Same-process A/B of the persistent strip conv kernel (csrc/conv_strip_f16x3.hip) against the per-tile kernels it replaces, at the shapes
of BigVGAN-base's unfused AMPBlock convs (BASELINE configs[2]: B = 32) and HiFi-GAN V1's C = 256 stage (configs[1]: B = 64).

    python tools/strip_bench.py [--reps 20] [--steps 0 1 2 4 8] [--only c128]

Per case: mean launch time (HIP events on the launch stream, `reps` back-to-back launches) with amp_set_conv_strip(0) and with
amp_set_conv_strip(2) for every --steps value (0 = the policy's choice), the fraction of the f16x3 MFMA peak (838.9 TFLOP/s) and whether
the two outputs are bit-identical.  Tuning aid; not part of the product."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd import _lib

PEAK = 2516.6 / 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--steps", type=int, nargs="+", default=[0, 1, 2, 4, 8])
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    _lib.set_precision("f16x3")
    L = _lib.lib()
    st = _lib.current_stream_ptr(torch.device("cuda", 0))
    cases = []
    for C, T, B in ((128, 16384, 32), (64, 32768, 32), (256, 2048, 32), (256, 2048, 64)):
        for k in (3, 7, 11):
            for d, with_res in ((1, True), (5, False)):
                cases.append((C, T, B, k, d, with_res))
    if a.only:
        cases = [c for c in cases if f"c{c[0]}" == a.only]
    print("C,T,B,k,dil,res,old_us,old_frac," + ",".join(f"s{s}_us,s{s}_frac" for s in a.steps) + ",bitwise")
    for C, T, B, k, d, with_res in cases:
        g = torch.Generator().manual_seed(1)
        w = (torch.randn((C, C, k), generator=g) * (C * k) ** -0.5).contiguous()
        b = torch.randn(C, generator=g) * 0.1
        h = ctypes.c_void_p()
        _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                     ctypes.byref(h)))
        x = torch.randn(B, C, T, device="cuda")
        y = torch.empty(B, C, T, device="cuda")
        res = torch.randn(B, C, T, device="cuda") if with_res else None
        rp = ctypes.c_void_p(res.data_ptr()) if res is not None else None

        def go():
            _lib.check(L.amp_conv_forward(h, ctypes.c_void_p(x.data_ptr()), B, T, 1.0, rp, 1.0, ctypes.c_void_p(y.data_ptr()), st))

        def timed():
            go(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                go()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.reps * 1e3
        flop = 2.0 * C * C * k * B * T
        row = [C, T, B, k, d, int(with_res)]
        _lib.check(L.amp_set_conv_strip(0))
        us = timed(); y0 = y.clone()
        row += [f"{us:.1f}", f"{flop / us / 1e6 / PEAK:.3f}"]
        same = True
        _lib.check(L.amp_set_conv_strip(2))
        for s in a.steps:
            _lib.check(L.amp_set_conv_strip_steps(s))
            us = timed()
            same = same and bool(torch.equal(y, y0))
            row += [f"{us:.1f}", f"{flop / us / 1e6 / PEAK:.3f}"]
        _lib.check(L.amp_set_conv_strip_steps(0))
        _lib.check(L.amp_set_conv_strip(-1))
        row.append(int(same))
        print(",".join(str(v) for v in row), flush=True)
        L.amp_conv_destroy(h)
        del x, y, res


if __name__ == "__main__":
    main()
