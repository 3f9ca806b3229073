#!/usr/bin/env python
"""The frame-rate convs of the VITS text side in isolation (B = 16, T = 100): mean time of a launch when launches of the SAME conv run
back to back (weights hot in L2) and when an unrelated 256 MB copy runs between them (weights cold, as inside a forward where 8 ms
of other kernels pass between two uses).  Tuning aid.     python tools/text_conv_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from amphion_amd.modules.hip_ops import HipConv1d

B, T = 16, 100
cases = [("FFN conv_1", 192, 768, 3), ("FFN conv_2", 768, 192, 3), ("q|k|v", 192, 576, 1), ("conv_o / 1x1", 192, 192, 1), ("proj", 192, 384, 1)]
junk_a = torch.empty(64 << 20, device="cuda")
junk_b = torch.empty(64 << 20, device="cuda")
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
print(f"{'conv':14s} {'hot us':>8s} {'cold us':>8s}")
for name, cin, cout, k in cases:
    torch.manual_seed(1)
    conv = HipConv1d(cin, cout, k, padding=(k - 1) // 2, weight_norm=False).cuda()
    x = torch.randn(B, cin, T, device="cuda")
    y = torch.empty(B, cout, T, device="cuda")
    for _ in range(3):
        conv(x, out=y, lens=lens)
    torch.cuda.synchronize()
    n = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        conv(x, out=y, lens=lens)
    b.record(); torch.cuda.synchronize()
    hot = a.elapsed_time(b) * 1e3 / n
    cold = 0.0
    m = 10
    for _ in range(m):
        junk_b.copy_(junk_a)
        a.record(); conv(x, out=y, lens=lens); b.record(); torch.cuda.synchronize()
        cold += a.elapsed_time(b) * 1e3 / m
    print(f"{name:14s} {hot:8.1f} {cold:8.1f}")
