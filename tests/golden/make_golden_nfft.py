#!/usr/bin/env python
"""Golden vectors of the REAL reference front end at the transform lengths round 5 added: utils/mel.py's
extract_mel_features / extract_linear_features / amplitude_phase_spectrum (utils/mel.py:20-52,111-170) and utils/stft.py's STFT.transform /
STFT.inverse (utils/stft.py:152-222) at (n_fft, hop, win) = (1920, 480, 1920) -- the n_fft of 24 of the reference's 38 JSON configs, e.g.
egs/vocoder/vocos/emilia_singnet.json:15 --, (2048, 512, 2048), odd (1001, 143, 1001), prime (1021, 256, 1021) and (400, 100, 320), run on
CPU in the build container (needs /root/reference):

    python tests/golden/make_golden_nfft.py        -> tests/golden/golden_nfft.npz

The input is 4 800 samples of an in-tree speech clip (stored in the file) plus a rolled, attenuated copy as a second batch item."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + load_by_path)

CASES = {
    # tag: (sample_rate, n_fft, hop, win, n_mel, fmin, fmax)
    "n1920": (24000, 1920, 480, 1920, 128, 0, 12000),
    "n2048": (44100, 2048, 512, 2048, 128, 0, None),
    "n1001": (16000, 1001, 143, 1001, 40, 0, 8000),
    "n1021": (16000, 1021, 256, 1021, 40, 20, 7600),
    "n400w320": (16000, 400, 100, 320, 80, 0, 8000),
}


def main():
    mg.install_stubs()
    import scipy.signal  # noqa: F401  (the reference imports get_window from scipy)
    from types import SimpleNamespace as NS
    mel_mod = mg.load_by_path("ref_utils_mel", os.path.join(mg.REF, "utils/mel.py"))
    stft_mod = mg.load_by_path("ref_utils_stft", os.path.join(mg.REF, "utils/stft.py"))
    sr, pcm = mg.read_wav(os.path.join(mg.REF, "egs/tts/VALLE/prompt_examples/260_123440_000010_000004.wav"))
    pcm = pcm[12000 : 12000 + 4800]
    out = {"wav_pcm16": pcm}
    y1 = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
    g = torch.Generator().manual_seed(23)
    for tag, (srate, nfft, hop, win, n_mel, fmin, fmax) in CASES.items():
        pp = NS(sample_rate=srate, n_fft=nfft, hop_size=hop, win_size=win, n_mel=n_mel, fmin=fmin, fmax=fmax)
        L = (4800 // hop) * hop                       # a whole number of hops: frames == L / hop (utils/mel.py:145-164)
        y = torch.stack([y1[:L], torch.roll(y1[:L], 777) * 0.5])
        out[f"{tag}_cfg"] = np.array([srate, nfft, hop, win, n_mel, fmin, -1 if fmax is None else fmax, L])
        with torch.no_grad():
            mel_mod.mel_basis.clear()
            mel_mod.hann_window.clear()
            out[f"{tag}_mel"] = mel_mod.extract_mel_features(y, pp).numpy()
            out[f"{tag}_linear"] = mel_mod.extract_linear_features(y[:1], pp).numpy()
            la, ph, re, im = mel_mod.amplitude_phase_spectrum(y, pp)
            out[f"{tag}_logamp"] = la.numpy()
            out[f"{tag}_phase"] = ph.numpy()
            out[f"{tag}_re"] = re.numpy()
            out[f"{tag}_im"] = im.numpy()
            st = stft_mod.STFT(nfft, hop, win)
            # STFT.transform hard-codes .cuda() on its operands (utils/stft.py:167-172); this container has no GPU: the reference's own code
            # runs with Tensor.cuda() as the identity
            real_cuda = torch.Tensor.cuda
            torch.Tensor.cuda = lambda self, *a, **k: self
            try:
                mag, phase = st.transform(y)
            finally:
                torch.Tensor.cuda = real_cuda
            out[f"{tag}_stft_mag"] = mag.numpy()
            out[f"{tag}_stft_phase"] = phase.numpy()
            # an inverse from an ARBITRARY (magnitude, phase) pair: the phase of a real signal's transform is ill-conditioned where the
            # magnitude vanishes, so the inverse is pinned on its own inputs
            F = mag.shape[-1]
            imag = torch.rand(2, mag.shape[1], F, generator=g) * 3.0
            iphase = (torch.rand(2, mag.shape[1], F, generator=g) * 2 - 1) * np.pi
            out[f"{tag}_inv_mag"] = imag.numpy()
            out[f"{tag}_inv_phase"] = iphase.numpy()
            out[f"{tag}_inv_wav"] = st.inverse(imag, iphase).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_nfft.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)
    print("bytes", os.path.getsize(os.path.join(HERE, "golden_nfft.npz")))


if __name__ == "__main__":
    main()
