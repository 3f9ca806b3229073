#!/usr/bin/env python
"""BASELINE.json configs[0] fixture: 16 real speech / singing clips through the REAL reference functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_c1.py

The reference ships no LJSpeech; SURVEY.md Appendix E names the 16 in-tree PCM16 clips that stand in for it (3.0 .. 12.9 s,
103.7 s in all).  Each is read with ``wave``, scaled by 1/32768 (what ``librosa.load`` yields, utils/audio.py:26-42),
resampled 24 kHz -> 22.05 kHz (scipy.signal.resample_poly 147/160) IN FULL (round 3: round 2 cut the clips to 1.4 .. 2.8 s),
trimmed to a whole number of mel frames and re-quantised to PCM16 -- THAT int16 array is the test's input clip, so the GPU
test needs no resampler.  On it the real reference code gives

  * ``mel_i``  = utils/mel.py::extract_mel_features (config 22.05 kHz / 1024 / 256 / 80 mel, config/fs2.json:25-31),
  * ``wav_i``  = models/vocoders/gan/generator/hifigan.py::HiFiGAN (V1, the seeded synthetic weights of oracle/synth.py,
                 seed 1234) on mel_i, through models/vocoders/gan/gan_vocoder_inference.py::vocoder_inference, cropped
                 to frames * hop as VocoderInference.inference does (vocoder_inference.py:355-361) -- for ALL 16 clips.

Written to golden_c1.npz: pcm_i int16 in full; to keep the fixture at a few MB the reference outputs are stored DECIMATED --
mel_i = every 4th frame starting at frame i % 4 ([80, ceil((frames - i % 4) / 4)]), wav_i = every 8th sample starting at
sample i % 8 -- a kernel error cannot hide between samples (every frame / sample goes through the same code, and the
offsets sweep all residues of the kernels' 32-frame / 4-sample groupings over the 16 clips); frames_i holds the lengths."""
import os
import sys

import numpy as np
import torch
from scipy.signal import resample_poly

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs for the absent third-party packages, reference imports)

from oracle import synth  # noqa: E402
from oracle import vocoder_oracle as vo  # noqa: E402

REF = mg.REF
CLIPS = (["egs/tts/VALLE/prompt_examples/%s.wav" % n for n in ("260_123440_000010_000004", "5142_33396_000002_000004",
                                                                 "6829_68771_000027_000000", "7176_92135_000004_000000")]
         + ["models/svc/vevosing/wav/%s.wav" % n for n in ("adele", "breathy", "jaychou", "taiyizhenren", "vibrato")]
         + ["models/tts/metis/wav/%s.wav" % n for n in ("l2s/prompt", "tse/mix", "tse/prompt", "tts/prompt", "vc/prompt", "vc/source")]
         + ["models/vc/vevo/wav/mandarin_female.wav"])
HOP = 256
MEL_DECIM, WAV_DECIM = 4, 8


def main():
    mg.install_stubs()
    sys.path.insert(1, os.path.join(os.path.dirname(HERE), "shims"))   # json5 / ruamel / torchaudio stand-ins for utils/util.py
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mel_mod = mg.load_by_path("ref_utils_mel", os.path.join(REF, "utils/mel.py"))
    from models.vocoders.gan.gan_vocoder_inference import vocoder_inference
    from models.vocoders.gan.generator.hifigan import HiFiGAN

    pp = vo.preprocess_22k()
    hp = vo.hifigan_v1_hp()
    cfg = mg.ns({"preprocess": dict(vars(pp), extract_amplitude_phase=False), "model": {"hifigan": hp, "generator": "hifigan"}})
    model = HiFiGAN(cfg)
    mg.load_synth(model, synth.hifigan_param_shapes(80, hp), 1234, 1.0)
    out = {}
    total = 0
    assert len(CLIPS) == 16
    for i, rel in enumerate(CLIPS):
        sr, x = mg.read_wav(os.path.join(REF, rel))
        assert sr == 24000, (rel, sr)
        y = resample_poly(x.astype(np.float64) / 32768.0, 147, 160)
        frames = len(y) // HOP                                # the whole clip, to a whole number of frames
        y = y[:frames * HOP]
        pcm = np.clip(np.round(y * 32768.0), -32768, 32767).astype(np.int16)
        out[f"pcm_{i}"] = pcm
        out[f"frames_{i}"] = np.array(frames)
        total += frames
        wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
        with torch.no_grad():
            mel = mel_mod.extract_mel_features(wav.unsqueeze(0), cfg.preprocess)      # [80, frames]
        assert tuple(mel.shape) == (80, frames), mel.shape
        out[f"mel_{i}"] = mel.numpy().astype(np.float32)[:, i % MEL_DECIM::MEL_DECIM].copy()
        with torch.no_grad():
            audio = vocoder_inference(cfg, model, mel.unsqueeze(0), device=torch.device("cpu"))
        full = audio.squeeze(0).squeeze(0)[: frames * HOP].numpy().astype(np.float32)
        out[f"wav_{i}"] = full[i % WAV_DECIM::WAV_DECIM].copy()
        print(i, rel, frames, f"{frames * HOP / 22050:.1f} s", "wav peak", float(np.abs(full).max()), flush=True)
    print("total", total, "frames =", total * HOP / 22050, "s")
    path = os.path.join(HERE, "golden_c1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
