#!/usr/bin/env python
"""Per-layer timing of the conv kernels at the BASELINE configs[1] shapes (B=64, T=256 mel frames).

    python tools/conv_bench.py [--precision f16x3 f32] [--reps 5]

For every distinct MRF conv shape of HiFi-GAN V1 (stage channels x kernel x dilation) and the four
transposed convs: mean kernel time (HIP events on the launch stream), algorithmic TFLOP/s and the
layer-wise-minimum HBM GB/s (read x [+ res] + write y).  Tuning aid; not part of the product."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd import _lib


def run(prec, reps, B, Tmel, only=""):
    _lib.set_precision(prec)
    L = _lib.lib()
    st = _lib.current_stream_ptr(torch.device("cuda", 0))
    rows = []
    stages = [(256, 8), (128, 64), (64, 128), (32, 256)]   # (C, T multiplier)
    cases = []
    for C, tm in stages:
        for k in (3, 7, 11):
            for d in (1, 5):
                cases.append(("conv", C, C, k, d, 1, Tmel * tm, d == 1))
    for cin, cout, k, u, tm in ((512, 256, 16, 8, 1), (256, 128, 16, 8, 8), (128, 64, 4, 2, 64), (64, 32, 4, 2, 128)):
        cases.append(("convT", cin, cout, k, 1, u, Tmel * tm, False))
    if only == "blk":   # the launches conv_blk_f16x3.hip covers: transposed convs, k = 3 at C = 256
        cases = [c for c in cases if c[0] == "convT" or (c[1] == 256 and c[3] == 3)]
    if only == "rg":    # launches with several row groups per x tile: the C = 256 stage, the stride-8 transposed convs
        cases = [c for c in cases if (c[0] == "convT" and c[5] == 8) or (c[0] == "conv" and c[1] == 256)]
    if only == "c128":  # the unfused C = 128 convs (BigVGAN's stage 1; HiFi-GAN runs these as fused pairs)
        cases = [c for c in cases if c[0] == "conv" and c[1] == 128]
    for kind, cin, cout, k, d, u, T, with_res in cases:
        g = torch.Generator().manual_seed(1)
        tr = kind == "convT"
        w = (torch.randn((cin, cout, k) if tr else (cout, cin, k), generator=g) * (cin * k / u) ** -0.5).contiguous()
        b = torch.randn(cout, generator=g) * 0.1
        h = ctypes.c_void_p()
        pad = (k - u) // 2 if tr else (k * d - d) // 2
        _lib.check(L.amp_conv_create(int(tr), cin, cout, k, u, d, pad, ctypes.c_void_p(w.data_ptr()),
                                     ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
        x = torch.randn(B, cin, T, device="cuda")
        Tout = L.amp_conv_out_len(h, T)
        y = torch.empty(B, cout, Tout, device="cuda")
        res = torch.randn(B, cout, Tout, device="cuda") if with_res else None
        rp = ctypes.c_void_p(res.data_ptr()) if res is not None else None

        def go():
            _lib.check(L.amp_conv_forward(h, ctypes.c_void_p(x.data_ptr()), B, T, 0.1, rp, 1.0, ctypes.c_void_p(y.data_ptr()), st))
        go(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            go()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * cin * cout * k * B * (Tout if not tr else T)
        byts = 4.0 * B * (cin * T + cout * Tout * (2 if with_res else 1))
        rows.append((prec, kind, cin, cout, k, d, u, T, with_res, ms, flop / ms / 1e9, byts / ms / 1e6))
        L.amp_conv_destroy(h)
        del x, y, res
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", nargs="+", default=["f16x3", "f32"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--only", default="", choices=["", "blk", "rg", "c128"])
    a = ap.parse_args()
    print("prec,kind,cin,cout,k,dil,stride,T_in,res,ms,TFLOP/s,GB/s(min-traffic)")
    for p in a.precision:
        for r in run(p, a.reps, a.batch, a.frames, a.only):
            print(",".join(str(v) if not isinstance(v, float) else f"{v:.3f}" for v in r))


if __name__ == "__main__":
    main()
