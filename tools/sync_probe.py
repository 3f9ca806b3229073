#!/usr/bin/env python
"""Why does a host that waits for EVERY forward see ~21 ms extra on some calls (list API 37 / 60 ms alternating)?  Forward +
synchronize per call, with a host-side idle gap between calls; prints wall time of launch+sync next to the GPU-event time.
    python tools/sync_probe.py            Tuning aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as bc
from amphion_amd.utils.synthetic import synthetic_mel


def h2d_variants():
    """Does a pageable host -> device copy of the batch in front of the forward (what synthesis_audios does) bring the
    alternating +21 ms back?  none | pageable .to(device) | pinned staging + non_blocking copy"""
    cfg, m = bc.hifigan()
    lens = torch.randint(60, 400, (64,), generator=torch.Generator().manual_seed(3)).tolist()
    T = max(lens)
    mel_host = synthetic_mel(64, 80, T, seed=0)                       # pageable
    mel_pin = torch.empty_like(mel_host).pin_memory(); mel_pin.copy_(mel_host)
    mel_dev0 = mel_host.cuda()
    ext = [min(T, l + 17) for l in lens]
    with torch.no_grad():
        for _ in range(3):
            m.forward_ragged(mel_dev0, ext)
        torch.cuda.synchronize()
        m.set_profiling(1)
        for name in ("none", "pageable", "pinned", "pageable+sync", "pageable_lens_only"):
            rows = []
            for call in range(12):
                t0 = time.perf_counter()
                if name == "none":
                    md = mel_dev0
                elif name == "pageable":
                    md = mel_host.to("cuda:0")
                elif name == "pinned":
                    md = mel_pin.to("cuda:0", non_blocking=True)
                elif name == "pageable+sync":
                    md = mel_host.to("cuda:0"); torch.cuda.synchronize()
                else:
                    md = mel_dev0
                out = m.forward_ragged(md, ext)
                m.check_range()
                t1 = time.perf_counter()
                rows.append(((t1 - t0) * 1e3, m.last_timing_ms(0)))
                time.sleep(0.003)
            print(f"h2d={name:18s}: wall " + " ".join(f"{w:5.1f}" for w, _ in rows) + "   | gpu fwd " + " ".join(f"{g:5.1f}" for _, g in rows[:3]), flush=True)
        m.set_profiling(0)


def main():
    if "--h2d" in sys.argv:
        return h2d_variants()
    cfg, m = bc.hifigan()
    mel = synthetic_mel(64, 80, 256, seed=0).cuda()
    lens = torch.randint(60, 256, (64,), generator=torch.Generator().manual_seed(3)).tolist()
    out_host = torch.empty(64 * 65536, dtype=torch.float32, pin_memory=True)
    with torch.no_grad():
        for _ in range(5):
            m(mel)
        torch.cuda.synchronize()
        m.set_profiling(1)
        for name, fn in (("dense", lambda: m(mel)), ("ragged", lambda: m.forward_ragged(mel, lens))):
            for gap_ms in (0.0, 1.0, 4.0, 12.0):
                for d2h in (False, True):
                    rows = []
                    for call in range(10):
                        t0 = time.perf_counter()
                        out = fn()
                        if d2h:
                            out_host[: out.numel()].view(out.shape).copy_(out, non_blocking=True)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        rows.append(((t1 - t0) * 1e3, m.last_timing_ms(0)))
                        if gap_ms:
                            time.sleep(gap_ms * 1e-3)
                    print(f"{name} gap={gap_ms:4.1f}ms d2h={int(d2h)}: wall " + " ".join(f"{w:5.1f}" for w, _ in rows) + "   | gpu " + " ".join(f"{g:5.1f}" for _, g in rows[:4]), flush=True)


if __name__ == "__main__":
    main()
