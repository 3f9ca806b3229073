"""Audio output side of the vocoder path, mirrored from utils/io.py:49-76 (``save_audio``).

The reference copies fp32 audio to the host and lets torchaudio convert it to 16-bit PCM while writing
(``torchaudio.save(..., encoding="PCM_S", bits_per_sample=16)``).  Here the conversion is a HIP kernel
(``amp_wav_to_pcm16``, bit-exact restatement of torchaudio 2.0.2 / libsox 14.4.2) run on the batch while it is
still in HBM, so the D2H copy -- and the multi-GPU gather, see ``amphion_amd.distributed`` -- move 2 bytes per
sample instead of 4, and the host only writes RIFF headers and bytes.  No CPU fallback: tensors must be on the GPU.
"""
from __future__ import annotations

import wave

import torch

from amphion_amd import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def wav_to_pcm16(wav: torch.Tensor, lengths=None) -> torch.Tensor:
    """fp32 ``[L]`` / ``[B, L]`` on a ROCm device -> int16 of the same shape on that device.

    ``lengths`` (samples per row) zeroes each row's tail: the crop ``[: l * hop_size]`` of
    models/vocoders/vocoder_inference.py:359 applied to a padded batch (the host then slices the rows).
    """
    if not wav.is_cuda:
        raise RuntimeError("wav_to_pcm16: the waveform must be on the GPU (the HIP path has no CPU fallback)")
    if wav.dtype != torch.float32:
        raise TypeError(f"wav_to_pcm16: expected float32, got {wav.dtype}")
    x = wav if wav.dim() == 2 else wav.reshape(1, -1)
    if x.stride(-1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    B, L = x.shape
    out = torch.empty((B, L), dtype=torch.int16, device=x.device)
    if B == 0 or L == 0:
        return out.reshape(wav.shape)
    lens = None
    if lengths is not None:
        lens = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1).to(x.device)
        if lens.numel() != B:
            raise ValueError(f"wav_to_pcm16: {lens.numel()} lengths for {B} rows")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().amp_wav_to_pcm16(_ptr(x), B, L, x.stride(0) if B > 1 else L, _ptr(lens), _ptr(out), L, stream))
    return out.reshape(wav.shape)


def _write_wav(path, pcm_bytes, fs, silence_len=0):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(fs))
        pad = b"\x00\x00" * silence_len
        w.writeframes(pad + pcm_bytes + pad)


def save_audio(path, waveform, fs, add_silence=False, turn_up=False, volume_peak=0.9):
    """utils/io.py:49-76 with the same arguments; ``waveform`` is a GPU tensor ``[L]``, ``[1, L]`` or ``[C, L]``.

    turn_up: scale so the peak reaches ``volume_peak`` (:60-63); add_silence: ``fs // 20`` zero samples at both
    ends (:65-69); more than one channel is averaged to mono (:74-76); then 16-bit PCM_S (:76).
    """
    w = torch.as_tensor(waveform)
    if not w.is_cuda:
        raise RuntimeError("save_audio: the waveform must be on the GPU (the HIP path has no CPU fallback)")
    w = w.to(torch.float32)
    if turn_up:
        ratio = volume_peak / torch.maximum(w.max(), w.min().abs())
        w = w * ratio
    if w.dim() == 2:
        w = w[0] if w.shape[0] == 1 else torch.mean(w, dim=0)
    pcm = wav_to_pcm16(w.contiguous())
    _write_wav(path, pcm.cpu().numpy().tobytes(), fs, fs // 20 if add_silence else 0)


def save_audios(paths, wavs: torch.Tensor, lengths, fs):
    """One padded batch ``[B, L]`` (GPU) -> B wav files: one conversion kernel and ONE D2H copy of int16 for the
    whole batch, rows cropped to ``lengths`` samples -- the per-utterance loop of vocoder_inference.py:355-370
    (crop, ``save_audio``) without B host conversions of fp32 audio."""
    if wavs.dim() == 3:
        wavs = wavs.squeeze(1)
    lengths = [int(l) for l in lengths]
    if len(paths) != wavs.shape[0] or len(lengths) != wavs.shape[0]:
        raise ValueError("save_audios: paths / lengths do not match the batch")
    pcm = wav_to_pcm16(wavs, lengths).cpu().numpy()
    for path, row, l in zip(paths, pcm, lengths):
        _write_wav(path, row[: min(l, row.shape[0])].tobytes(), fs)
