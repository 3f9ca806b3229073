"""Vocoder feature extraction on the MI355X front-end kernels.

Drop-in for the mel / energy / amplitude-phase / audio part of
``processors/acoustic_extractor.py:376-449`` (``extract_utt_acoustic_features_vocoder``) and the on-disk format
of ``utils/io.py:12-30`` (``save_feature``: ``<dataset_output>/<feature_dir>/<uid>.npy``; mel is float32
``[n_mel, T]``), plus a batched variant that runs a whole list of utterances through ONE mel-kernel launch
(the reference processes one file per call).  Pitch / uv / label extraction (pyworld, parselmouth, mu-law) stay
host-side concerns of the reference and are not part of the accelerated path.

Audio loading: the reference uses librosa (``utils/audio.py:26-42``), absent here; ``load_audio_torch`` below
reads PCM16 / float WAV with the standard library + scipy (resampling with ``scipy.signal.resample_poly``) and
applies the reference's ``max_mag`` normalisation rule.
"""
from __future__ import annotations

import os
import wave

import numpy as np
import torch

from amphion_amd.utils.mel import amplitude_phase_spectrum, extract_mel_features, extract_mel_features_batch


def save_feature(process_dir, feature_dir, item, feature, overrides=True):
    """utils/io.py:12-30."""
    process_dir = os.path.join(process_dir, feature_dir)
    os.makedirs(process_dir, exist_ok=True)
    out_path = os.path.join(process_dir, item + ".npy")
    if os.path.exists(out_path) and not overrides:
        return
    np.save(out_path, feature)


def load_audio_torch(wave_file, fs):
    """utils/audio.py:26-42 without librosa: mono float32 in [-1, 1] at ``fs`` -> (tensor [L], fs)."""
    with wave.open(wave_file, "rb") as w:
        sr, nch, sw = w.getframerate(), w.getnchannels(), w.getsampwidth()
        raw = w.readframes(w.getnframes())
    if sw == 2:
        x = np.frombuffer(raw, dtype=np.int16).astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype=np.int32).astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)                       # librosa.load(mono=True)
    if sr != fs:
        from math import gcd

        from scipy.signal import resample_poly

        g = gcd(int(sr), int(fs))
        x = resample_poly(x, int(fs) // g, int(sr) // g).astype(np.float32)
    audio = torch.from_numpy(np.ascontiguousarray(x))
    # utils/audio.py:33-41: integer-valued data are rescaled by their bit depth; in-range float audio is untouched
    max_mag = float(audio.abs().max()) if audio.numel() else 0.0
    if max_mag > (2**15):
        audio = audio / (2**31)
    elif max_mag > 1.01:
        audio = audio / (2**15)
    return audio, fs


def extract_utt_acoustic_features_vocoder(dataset_output, cfg, utt, wav_torch=None, device="cuda"):
    """processors/acoustic_extractor.py:376-449 for the features the vocoder path consumes.

    ``utt`` = {"Uid": ..., "Path": ...}; ``wav_torch`` optionally supplies the already-loaded waveform."""
    uid = utt["Uid"]
    pp = cfg.preprocess
    if getattr(pp, "extract_pitch", False) or getattr(pp, "extract_label", False):
        raise NotImplementedError("pitch / uv / label extraction are outside the accelerated vocoder path")
    with torch.no_grad():
        if wav_torch is None:
            wav_torch, _ = load_audio_torch(utt["Path"], pp.sample_rate)
        wav = wav_torch.detach().cpu().numpy()
        wav_dev = wav_torch.to(device)
        mel = None
        if getattr(pp, "extract_mel", False):
            mel = extract_mel_features(wav_dev.unsqueeze(0), pp)
            save_feature(dataset_output, pp.mel_dir, uid, mel.cpu().numpy())
        if getattr(pp, "extract_energy", False):
            if pp.energy_extract_mode != "from_mel" or mel is None:
                raise NotImplementedError("energy_extract_mode 'from_waveform' is outside the accelerated path")
            energy = (mel.exp() ** 2).sum(0).sqrt().cpu().numpy()
            save_feature(dataset_output, pp.energy_dir, uid, energy)
        if getattr(pp, "extract_amplitude_phase", False):
            log_amplitude, phase, real, imaginary = amplitude_phase_spectrum(wav_dev.unsqueeze(0), pp)
            for d, v in ((pp.log_amplitude_dir, log_amplitude), (pp.phase_dir, phase), (pp.real_dir, real),
                         (pp.imaginary_dir, imaginary)):
                save_feature(dataset_output, d, uid, v.cpu().numpy())   # the reference saves tensors; same bytes
        if getattr(pp, "extract_audio", False):
            save_feature(dataset_output, pp.audio_dir, uid, wav)


def extract_mel_features_dataset(dataset_output, cfg, utts, wavs=None, batch_size=64, device="cuda"):
    """Mel (+ audio) features of many utterances: ``batch_size`` waveforms per kernel launch instead of one file
    per call; files are identical to ``extract_utt_acoustic_features_vocoder``'s."""
    pp = cfg.preprocess
    for s in range(0, len(utts), batch_size):
        grp = utts[s:s + batch_size]
        ws = [wavs[s + i] if wavs is not None else load_audio_torch(u["Path"], pp.sample_rate)[0] for i, u in enumerate(grp)]
        mels = extract_mel_features_batch(ws, pp, device=device)
        for u, w, m in zip(grp, ws, mels):
            save_feature(dataset_output, pp.mel_dir, u["Uid"], m.cpu().numpy())
            if getattr(pp, "extract_audio", False):
                save_feature(dataset_output, pp.audio_dir, u["Uid"], torch.as_tensor(w).cpu().numpy())
