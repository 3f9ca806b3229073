#!/usr/bin/env python
"""Golden vectors for STFT.inverse / window_sumsquare from the REAL reference class (utils/stft.py),
run on CPU in the build container (needs /root/reference):   python tests/golden/make_golden_istft.py
-> tests/golden/golden_istft.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + load_by_path)


def main():
    mg.install_stubs()
    import scipy.signal  # noqa: F401  (the reference imports get_window from scipy)
    stft_mod = mg.load_by_path("ref_utils_stft", os.path.join(mg.REF, "utils/stft.py"))
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, (nfft, hop, win, B, F) in {"n1024": (1024, 256, 1024, 2, 9), "n512w400": (512, 128, 400, 1, 21)}.items():
        st = stft_mod.STFT(nfft, hop, win)
        mag = torch.rand(B, nfft // 2 + 1, F, generator=g) * 3.0
        phase = (torch.rand(B, nfft // 2 + 1, F, generator=g) * 2 - 1) * np.pi
        with torch.no_grad():
            wav = st.inverse(mag, phase)
        out[f"{tag}_cfg"] = np.array([nfft, hop, win])
        out[f"{tag}_mag"] = mag.numpy()
        out[f"{tag}_phase"] = phase.numpy()
        out[f"{tag}_wav"] = wav.numpy()
        out[f"{tag}_wss"] = stft_mod.window_sumsquare("hann", F, hop_length=hop, win_length=win, n_fft=nfft, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_istft.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
