#!/usr/bin/env python
"""Same-box A/B of the mel front end between two builds of the library: run once per build (AMP_LIB_PATH selects it) and
compare the printed digests -- equal digests = bit-identical mel / magnitude outputs on every case -- and the timings.

    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_old.so python tools/mel_ab.py
    python tools/mel_ab.py
"""
import hashlib
import os
import sys
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd.utils.mel import mel_spectrogram_torch  # noqa: E402

DEV = torch.device("cuda:0")
CASES = {
    "22k/80": NS(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000),
    "24k/100": NS(sample_rate=24000, n_fft=1024, win_size=1024, hop_size=256, n_mel=100, fmin=0, fmax=12000),
    "44k/128": NS(sample_rate=44100, n_fft=1024, win_size=1024, hop_size=256, n_mel=128, fmin=0, fmax=None),
    "16k/40": NS(sample_rate=16000, n_fft=1024, win_size=1024, hop_size=160, n_mel=40, fmin=50, fmax=7600),
}


def timed(fn, reps, chunks=10):
    """(median, max) over `chunks` chunks of `reps` back-to-back calls, ms per call."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(chunks):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps)
    out.sort()
    return out[len(out) // 2], out[-1]


def main():
    print("library:", os.environ.get("AMP_LIB_PATH", "(default)"))
    for name, pp in CASES.items():
        for B, L in ((64, 65536), (3, 7001), (1, 66150)):
            wav = (torch.rand(B, L, generator=torch.Generator().manual_seed(B + L)) * 2 - 1).to(DEV)
            out = mel_spectrogram_torch(wav, pp)
            torch.cuda.synchronize()
            dig = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
            med, worst = timed(lambda: mel_spectrogram_torch(wav, pp), 40)
            print(f"{name:8s} B={B:3d} L={L:6d} out={tuple(out.shape)} sha1={dig} {med * 1e3:8.1f} us/call (slowest chunk {worst * 1e3:.1f})", flush=True)


if __name__ == "__main__":
    main()
