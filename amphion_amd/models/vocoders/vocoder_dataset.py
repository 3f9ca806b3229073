"""Host-side batch assembly of the vocoder inference path, mirrored from models/vocoders/vocoder_dataset.py.

``VocoderCollator`` (:229-264) turns the per-utterance feature dicts of the dataset (``mel`` [n_mel, T] as stored on
disk, ``audio`` [T * hop], ``frame_pitch`` [T], ``target_len``) into the zero-padded batch the inference loop
consumes (``VocoderInference.inference`` models/vocoders/vocoder_inference.py:334-374 transposes ``mel`` back to
[B, n_mel, T] and crops every output to ``target_len * hop_size``).  Pure host logic; the padded batch is what
``forward_ragged`` / ``synthesis_audios`` take together with the lengths."""
from __future__ import annotations

import numpy as np
import torch


def _pad_stack(rows):
    """rows of shape [L_i, ...] -> [B, max L, ...] zero-padded (what pad_sequence(batch_first=True) gives)."""
    width = max(r.shape[0] for r in rows)
    out = torch.zeros((len(rows), width) + tuple(rows[0].shape[1:]), dtype=rows[0].dtype)
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out


class VocoderCollator(object):
    """Zero-pads model inputs and targets to the longest item of the batch (vocoder_dataset.py:229-264):

    ``target_len`` -> LongTensor [B] plus ``mask`` [B, T_max, 1] (ones over each item's frames);
    ``mel`` [n_mel, T] -> [B, T_max, n_mel] (time-major, as the reference collates it);
    every other key (``audio``, ``frame_pitch``, ...) -> padded along its first axis."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __call__(self, batch):
        out = {}
        for key in batch[0].keys():
            if key == "target_len":
                lens = [int(b["target_len"]) for b in batch]
                out["target_len"] = torch.tensor(lens, dtype=torch.long)
                out["mask"] = _pad_stack([torch.ones((n, 1), dtype=torch.long) for n in lens])
            elif key == "mel":
                out[key] = _pad_stack([torch.from_numpy(np.asarray(b[key])).T for b in batch])
            else:
                out[key] = _pad_stack([torch.from_numpy(np.asarray(b[key])) for b in batch])
        return out


def batch_to_generator_input(batch):
    """(mel [B, n_mel, T_max], lengths list) from a collated batch: the transpose of vocoder_inference.py:349 and the
    frame counts for ``generator.forward_ragged`` / the crop of :359."""
    return batch["mel"].transpose(-1, -2).contiguous(), [int(v) for v in batch["target_len"]]
