#!/usr/bin/env python
"""Kernel timeline of ONE single-utterance forward (HiFi-GAN V1, B = 1) from a rocprofv3 --kernel-trace csv: per kernel start (us into
the forward), duration and the idle gap before it; totals.   python tools/latency_trace.py <dir>      (run under rocprofv3 with --run)
    rocprofv3 --kernel-trace --output-format csv -d out -o lt -- python tools/latency_trace.py --run [--frames 256]"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--run" in sys.argv:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import bench_configs as bc
    from amphion_amd.utils.synthetic import synthetic_mel
    T = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 256
    if "--bigvgan" in sys.argv:
        from types import SimpleNamespace as NS
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
        hp = dict(bc.V1, activation="snakebeta", snake_logscale=True)
        m = bc.randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
        mel = torch.randn(1, 100, T, generator=torch.Generator().manual_seed(0)).cuda()
    else:
        cfg, m = bc.hifigan()
        mel = synthetic_mel(1, 80, T, seed=5).cuda()
    with torch.no_grad():
        for _ in range(12):
            m(mel)
        torch.cuda.synchronize()
    sys.exit(0)
d = sys.argv[1]
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void amp::", "")[:58], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)))
ev.sort()
# the last forward: from the last conv_pre-like first kernel after a conv_post
ends = [i for i, e in enumerate(ev) if "conv_post" in e[2]]
lo, hi = ends[-2] + 1, ends[-1]
f0 = ev[lo][0]
busy, gaps, prev = 0, 0, None
print("start_us,dur_us,gap_before_us,workgroups,kernel")
for s, e, n, g, w in ev[lo:hi + 1]:
    gap = 0 if prev is None else s - prev
    print(f"{(s - f0) / 1e3:8.1f},{(e - s) / 1e3:7.1f},{gap / 1e3:6.1f},{g // max(w, 1):6d},{n}")
    busy += e - s; gaps += max(gap, 0); prev = e
print(f"# forward span {(ev[hi][1] - f0) / 1e3:.1f} us, kernels {hi - lo + 1}, busy {busy / 1e3:.1f} us, idle between kernels {gaps / 1e3:.1f} us")
