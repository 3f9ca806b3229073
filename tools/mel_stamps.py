#!/usr/bin/env python
"""Phase timeline of mel1024_kernel from in-kernel clock stamps.  Experiment builds only: the MEL_STAMP macros and the
amp_debug_mel_stamps export are in profiles/negative_kernels/mel1024_prefetch.hip.txt (copy it over csrc/mel.hip, then
AMP_BUILD_TAG=tm AMP_BUILD_FLAGS=-DMEL_TIMING python -m amphion_amd.build, and run with AMP_LIB_PATH pointing at the tm library)."""
import ctypes
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd import _lib  # noqa: E402
from amphion_amd.utils.mel import mel_spectrogram_torch  # noqa: E402

pp = NS(sample_rate=22050, n_fft=1024, win_size=1024, hop_size=256, n_mel=80, fmin=0, fmax=8000)
L = _lib.lib()
for B, n in ((3, 7001), (64, 65536)):
    wav = torch.rand(B, n, device="cuda") * 2 - 1
    for rep in range(3):
        mel_spectrogram_torch(wav, pp)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 64)()
        L.amp_debug_mel_stamps(buf)
        t = np.array(buf[:30], dtype=np.int64)
        wall = (t[29] - t[28]) * 10.0   # ns at 100 MHz
        print(f"B={B} rep{rep}: wave 0 of workgroup 0: {wall / 1e3:.2f} us, {t[27] - t[0]} ticks; prologue {t[1] - t[0]}; frames (load, fft, split, project):",
              [(int(t[3 + 6 * f] - t[2 + 6 * f]), int(t[4 + 6 * f] - t[3 + 6 * f]), int(t[5 + 6 * f] - t[4 + 6 * f]), int(t[6 + 6 * f] - t[5 + 6 * f])) for f in range(4)],
              "tail", int(t[27] - t[26]))
