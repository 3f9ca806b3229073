"""Host helpers mirrored from utils/util.py that the vocoder inference path calls."""
import torch


def pad_mels_to_tensors(mels, batched=None):
    """utils/util.py:114-182 -- zero-pad a list of [n_mel, T_i] mels into batches of ``batched``
    (one batch when None); returns (tensors, mel_frames) exactly as the reference does."""
    tensors, mel_frames = [], []
    n = len(mels)
    step = n if batched is None else batched
    start = 0
    while start < n:
        group = mels[start:start + step]
        size = max(m.shape[-1] for m in group)
        tensor = torch.zeros(len(group), mels[0].shape[0], size)
        frame = torch.zeros(len(group), dtype=torch.int32)
        for i, m in enumerate(group):
            tensor[i, :, : m.shape[-1]] = m[:]
            frame[i] = m.shape[-1]
        tensors.append(tensor)
        mel_frames.append(frame)
        start += step
    return tensors, mel_frames
