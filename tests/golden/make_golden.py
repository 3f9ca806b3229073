#!/usr/bin/env python
"""Generate golden vectors by running the REAL reference classes on CPU.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_golden.py

Imports the reference's HiFiGAN / HiFiGAN_vits / BigVGAN / Activation1d /
utils.mel / utils.stft read-only from /root/reference (with sys.modules stubs for
the absent third-party packages), loads the seeded synthetic
weights of oracle/synth.py, runs them in fp32 on CPU and writes small .npz / .json
fixtures next to this file.  The oracle (oracle/vocoder_oracle.py) and the HIP
path are both checked against these files.
"""
import importlib.util
import json
import os
import sys
import types
import wave
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import synth  # noqa: E402
from oracle import vocoder_oracle as vo  # noqa: E402


def install_stubs():
    for name in ["lhotse", "lhotse.dataset", "lhotse.dataset.collation", "lhotse.dataset.input_strategies", "lhotse.utils"]:
        sys.modules[name] = MagicMock()
    # librosa stub: the mel filterbank comes from an INDEPENDENT implementation
    # (transformers.audio_utils) so that the oracle's own restatement is cross-checked.
    from transformers.audio_utils import mel_filter_bank

    librosa = types.ModuleType("librosa")
    filters = types.ModuleType("librosa.filters")
    util = types.ModuleType("librosa.util")

    def mel(sr=None, n_fft=None, n_mels=128, fmin=0.0, fmax=None, **kw):
        if fmax is None:
            fmax = sr / 2
        m = mel_filter_bank(n_fft // 2 + 1, n_mels, float(fmin), float(fmax), sr, norm="slaney", mel_scale="slaney")
        return m.T.astype(np.float32)

    def pad_center(data, size=None, axis=-1, **kw):
        n = data.shape[axis]
        lp = (size - n) // 2
        pads = [(0, 0)] * data.ndim
        pads[axis] = (lp, size - n - lp)
        return np.pad(data, pads)

    filters.mel = mel
    util.pad_center = pad_center
    util.tiny = lambda x: np.finfo(np.float32).tiny
    util.normalize = lambda x, **kw: x
    librosa.filters = filters
    librosa.util = util
    sys.modules["librosa"] = librosa
    sys.modules["librosa.filters"] = filters
    sys.modules["librosa.util"] = util


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ns(d):
    return SimpleNamespace(**{k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def dump_keys(name, model):
    sd = model.state_dict()
    with open(os.path.join(HERE, f"keys_{name}.json"), "w") as f:
        json.dump([[k, list(v.shape)] for k, v in sd.items()], f)
    return sd


def load_synth(model, shapes, seed, gain):
    sd = synth.synth_state_dict(shapes, seed=seed, g_gain=gain)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "param key restatement differs from reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd)
    model.eval()
    return sd


def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        sr = w.getframerate()
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    return sr, x


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from models.vocoders.gan.generator.bigvgan import BigVGAN
    from models.vocoders.gan.generator.hifigan import HiFiGAN, HiFiGAN_vits
    from modules.activation_functions.snake import Snake, SnakeBeta
    from modules.anti_aliasing.act import Activation1d

    out = {}

    # ---- HiFi-GAN V1 (config 2 architecture) ---------------------------------
    hp = vo.hifigan_v1_hp()
    cfg = ns({"preprocess": {"n_mel": 80}, "model": {"hifigan": hp}})
    m = HiFiGAN(cfg)
    dump_keys("hifigan_v1", m)
    load_synth(m, synth.hifigan_param_shapes(80, hp), 1234, 1.0)
    with torch.no_grad():
        for tag, (B, T, seed) in {"b1_t8": (1, 8, 0), "b2_t33": (2, 33, 1), "b3_t1": (3, 1, 2)}.items():
            mel = synth.synth_mel(B, 80, T, seed)
            out[f"hifigan_v1_{tag}_mel"] = mel.numpy()
            out[f"hifigan_v1_{tag}_wav"] = m(mel).numpy()

    # ---- recipe HiFi-GAN (resblock "2", k5, dilation 12, rate 4) --------------
    hp = vo.hifigan_recipe_hp()
    cfg = ns({"preprocess": {"n_mel": 100}, "model": {"hifigan": hp}})
    m = HiFiGAN(cfg)
    dump_keys("hifigan_recipe", m)
    load_synth(m, synth.hifigan_param_shapes(100, hp), 77, 1.0)
    with torch.no_grad():
        mel = synth.synth_mel(2, 100, 19, 3)
        out["hifigan_recipe_b2_t19_mel"] = mel.numpy()
        out["hifigan_recipe_b2_t19_wav"] = m(mel).numpy()

    # ---- HiFiGAN_vits (config 5 decoder), with and without g ------------------
    hp = vo.hifigan_v1_hp()
    for gin in (0, 256):
        m = HiFiGAN_vits(192, "1", hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"], hp["upsample_rates"],
                         hp["upsample_initial_channel"], hp["upsample_kernel_sizes"], gin_channels=gin)
        dump_keys(f"hifigan_vits_g{gin}", m)
        load_synth(m, synth.hifigan_param_shapes(192, hp, vits=True, gin_channels=gin), 4321, 1.0)
        with torch.no_grad():
            gen = torch.Generator().manual_seed(5)
            z = torch.randn(2, 192, 9, generator=gen)
            out[f"hifigan_vits_g{gin}_z"] = z.numpy()
            if gin:
                g = torch.randn(2, gin, 1, generator=gen)
                out[f"hifigan_vits_g{gin}_g"] = g.numpy()
                out[f"hifigan_vits_g{gin}_wav"] = m(z, g=g).numpy()
            else:
                out[f"hifigan_vits_g{gin}_wav"] = m(z).numpy()

    # ---- BigVGAN-base (config 3 architecture) ---------------------------------
    hp = vo.bigvgan_base_hp()
    cfg = ns({"preprocess": {"n_mel": 100}, "model": {"bigvgan": hp}})
    m = BigVGAN(cfg)
    dump_keys("bigvgan_base", m)
    load_synth(m, synth.bigvgan_param_shapes(100, hp), 1234, 0.75)
    with torch.no_grad():
        for tag, (B, T, seed) in {"b1_t8": (1, 8, 0), "b2_t13": (2, 13, 1)}.items():
            gen = torch.Generator().manual_seed(seed)
            mel = torch.randn(B, 100, T, generator=gen)
            out[f"bigvgan_base_{tag}_mel"] = mel.numpy()
            out[f"bigvgan_base_{tag}_wav"] = m(mel).numpy()

    # ---- small BigVGAN: AMPBlock2, plain 'snake', linear-scale alpha ----------
    hp = dict(resblock="2", activation="snake", snake_logscale=False, upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4],
              upsample_initial_channel=64, resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]])
    cfg = ns({"preprocess": {"n_mel": 20}, "model": {"bigvgan": hp}})
    m = BigVGAN(cfg)
    dump_keys("bigvgan_small", m)
    shapes = synth.bigvgan_param_shapes(20, hp)
    sd = synth.synth_state_dict(shapes, seed=9, g_gain=0.75)
    for k in sd:  # linear-scale alpha must stay positive-ish
        if k.endswith(".alpha"):
            sd[k] = sd[k].abs() + 0.5
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        gen = torch.Generator().manual_seed(11)
        mel = torch.randn(2, 20, 21, generator=gen)
        out["bigvgan_small_mel"] = mel.numpy()
        out["bigvgan_small_wav"] = m(mel).numpy()

    # ---- Activation1d standalone ----------------------------------------------
    with torch.no_grad():
        gen = torch.Generator().manual_seed(21)
        x = torch.randn(2, 5, 37, generator=gen) * 2
        for name, actcls, logscale in (("snakebeta_log", SnakeBeta, True), ("snake_lin", Snake, False)):
            a = actcls(5, alpha_logscale=logscale)
            a.alpha.data = torch.randn(5, generator=gen) * 0.3 + (0.0 if logscale else 1.0)
            if hasattr(a, "beta"):
                a.beta.data = torch.randn(5, generator=gen) * 0.3 + (0.0 if logscale else 1.0)
                out[f"act1d_{name}_beta"] = a.beta.data.numpy()
            out[f"act1d_{name}_alpha"] = a.alpha.data.numpy()
            act = Activation1d(activation=a)
            out[f"act1d_{name}_y"] = act(x).numpy()
            out[f"act1d_{name}_snake_only"] = a(x).numpy()
        out["act1d_x"] = x.numpy()
        out["act1d_filter"] = act.upsample.filter.reshape(-1).numpy()
        # T=1 edge (replicate padding only)
        x1 = torch.randn(1, 5, 1, generator=gen)
        out["act1d_T1_x"] = x1.numpy()
        out["act1d_T1_y"] = act(x1).numpy()

    # ---- Mel / STFT front end on a real in-tree clip --------------------------
    mel_mod = load_by_path("ref_utils_mel", os.path.join(REF, "utils/mel.py"))
    stft_mod = load_by_path("ref_utils_stft", os.path.join(REF, "utils/stft.py"))
    sr, pcm = read_wav(os.path.join(REF, "egs/tts/VALLE/prompt_examples/260_123440_000010_000004.wav"))
    assert sr == 24000
    pcm = pcm[12000 : 12000 + 256 * 40]  # 40 frames of speech
    out["wav_pcm16"] = pcm
    y = torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(0)
    y2 = torch.stack([y[0], torch.roll(y[0], 777) * 0.5])  # batch of 2
    for tag, pp in (("22k", vo.preprocess_22k()), ("24k", vo.preprocess_24k())):
        with torch.no_grad():
            mel_mod.mel_basis.clear()
            out[f"mel_{tag}_extract"] = mel_mod.extract_mel_features(y, pp).numpy()
            out[f"mel_{tag}_melspec_b2"] = mel_mod.mel_spectrogram_torch(y2, pp).numpy()
            out[f"mel_{tag}_linear"] = mel_mod.extract_linear_features(y, pp).numpy()
            la, ph, re, im = mel_mod.amplitude_phase_spectrum(y2, pp)
            out[f"mel_{tag}_logamp"] = la.numpy()
            out[f"mel_{tag}_phase"] = ph.numpy()
            out[f"mel_{tag}_re"] = re.numpy()
            out[f"mel_{tag}_im"] = im.numpy()
            # TacotronSTFT hard-codes .cuda() in transform (stft.py:167-172): run its maths on CPU
            taco = stft_mod.TacotronSTFT(pp.n_fft, pp.hop_size, pp.win_size, pp.n_mel, pp.sample_rate, pp.fmin, pp.fmax)
            st = taco.stft_fn
            nb, nsmp = y2.shape
            inp = torch.nn.functional.pad(y2.view(nb, 1, 1, nsmp), (st.filter_length // 2, st.filter_length // 2, 0, 0), mode="reflect").squeeze(1)
            ft = torch.nn.functional.conv1d(inp, st.forward_basis, stride=st.hop_length, padding=0)
            cut = st.filter_length // 2 + 1
            mag = torch.sqrt(ft[:, :cut] ** 2 + ft[:, cut:] ** 2)
            melo = taco.spectral_normalize(torch.matmul(taco.mel_basis, mag))
            out[f"taco_{tag}_mag"] = mag.numpy()
            out[f"taco_{tag}_mel"] = melo.numpy()
            out[f"taco_{tag}_energy"] = torch.norm(mag, dim=1).numpy()
            out[f"melbasis_{tag}"] = mel_mod.mel_basis[str(pp.fmax) + "_cpu"].numpy()

    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "golden_v1.npz"))
    print(f"wrote golden_v1.npz ({sz/1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
