"""fp32 -> signed 16-bit PCM as the reference writes its wavs (TEST INFRASTRUCTURE: the checker, never the product).

Reference call site: ``save_audio`` utils/io.py:49-76 ends in
``torchaudio.save(path, waveform, fs, encoding="PCM_S", bits_per_sample=16)`` (called twice per utterance from
``VocoderInference.inference`` models/vocoders/vocoder_inference.py:361-370).

The arithmetic lives in third-party code that is NOT under /root/reference and not installed here:
torchaudio==2.0.2 (env.sh:15), sox_io backend, linked against libsox 14.4.2.  Restated from their published sources:

* torchaudio/csrc/sox/effects_chain.cpp (``tensor_input_drain``), float32 tensors: the chunk is converted to
  float64, multiplied by 2147483648., clamped to [INT32_MIN, INT32_MAX] and cast to int32 (truncation toward zero)
  -> ``sox_sample_t``.
* libsox sox.h ``SOX_SAMPLE_TO_SIGNED_16BIT`` = ``SOX_SAMPLE_TO_UNSIGNED(16, d) ^ 0x8000`` where
  ``SOX_SAMPLE_TO_UNSIGNED(16, d) = d > SOX_SAMPLE_MAX - (1 << 15) ? 0xffff : ((uint32)(d ^ 0x80000000) + (1 << 15)) >> 16``
  i.e. round half up to 16 bits with saturation at +32767; libsox adds no dither on this path (the sox CLI's
  automatic dither effect is not part of the library's write path).

PARITY UNPINNED: neither torchaudio nor libsox can be executed in this image, the reference holds no golden wavs,
so this restatement is checked only against hand-derived known answers (tests/test_oracle_pcm16.py).
"""
import numpy as np

INT32_MIN, INT32_MAX = -(2**31), 2**31 - 1


def float_to_sox_sample(x):
    """float32 -> sox_sample_t (int32): effects_chain.cpp tensor_input_drain, Float case."""
    v = np.asarray(x, dtype=np.float32).astype(np.float64) * 2147483648.0
    nan = np.isnan(v)
    v = np.clip(np.where(nan, 0.0, v), INT32_MIN, INT32_MAX)
    d = np.trunc(v).astype(np.int64)
    # a NaN survives torch's clamp_ and its cast to int32 is undefined; x86 (cvttsd2si) yields INT32_MIN
    return np.where(nan, INT32_MIN, d)


def sox_sample_to_pcm16(d):
    """sox.h SOX_SAMPLE_TO_SIGNED_16BIT on int64-held sox samples."""
    d = np.asarray(d, dtype=np.int64)
    u = np.where(d > INT32_MAX - (1 << 15), 0xFFFF, (((d ^ 0x80000000) & 0xFFFFFFFF) + (1 << 15)) >> 16)
    return ((u ^ 0x8000) & 0xFFFF).astype(np.uint16).view(np.int16)


def float_to_pcm16(x, lens=None):
    """[..., L] float32 -> int16; ``lens`` (samples per row of a [B, L] array) zeroes each row's tail, the crop
    ``[: l * hop_size]`` of vocoder_inference.py:359 applied to a padded batch."""
    x = np.asarray(x, dtype=np.float32)
    y = sox_sample_to_pcm16(float_to_sox_sample(x)).reshape(x.shape)
    if lens is not None:
        t = np.arange(x.shape[-1])[None, :]
        y = np.where(t < np.asarray(lens).reshape(-1, 1), y, np.int16(0)).astype(np.int16)
    return y
