"""Drop-in for models/vocoders/gan/gan_vocoder_inference.py: same two functions, same signatures.

``vocoder_inference`` is the reference's batched forward wrapper (:11-38).  ``synthesis_audios``
keeps the reference contract (:41-96: list of [n_mel, T_i] mels -> list of [T_i * hop] audios) but
runs each padded batch through the generator ONCE instead of one utterance at a time; the crop to
``frame * hop`` is the reference's.  Because every conv zero-pads its own input, an utterance that
sits in a zero-padded batch differs from its B=1 result only within the receptive field of the tail;
``exact=True`` (default) therefore groups utterances of equal length and falls back to per-length
batches so results are identical to the reference's per-utterance loop.
"""
import torch

from amphion_amd.utils.util import pad_mels_to_tensors


def vocoder_inference(cfg, model, mels, f0s=None, device=None, fast_inference=False):
    """gan_vocoder_inference.py:11-38.  mels [B, n_mel, T] -> audios [B, T*hop] on the CPU."""
    model.eval()
    with torch.no_grad():
        mels = mels.to(device)
        if f0s is not None:
            f0s = f0s.to(device)
        if f0s is None and not cfg.preprocess.extract_amplitude_phase:
            output = model.forward(mels)
        elif cfg.preprocess.extract_amplitude_phase:
            (_, _, _, _, output) = model.forward(mels)
        else:
            output = model.forward(mels, f0s)
        return output.squeeze(1).detach().cpu()


def synthesis_audios(cfg, model, mels, f0s=None, batch_size=None, fast_inference=False, exact=True):
    """gan_vocoder_inference.py:41-96."""
    device = next(model.parameters()).device
    if f0s is not None:
        raise NotImplementedError("f0-conditioned generators (NSF-HiFiGAN) are outside the HiFi-GAN/BigVGAN hot path")
    hop = model.cfg.preprocess.hop_size
    audios = [None] * len(mels)
    if exact:
        # one true batch per distinct length: bit-identical to the reference's B=1 loop
        by_len = {}
        for i, m in enumerate(mels):
            by_len.setdefault(int(m.shape[-1]), []).append(i)
        for T, idxs in by_len.items():
            step = len(idxs) if batch_size is None else batch_size
            for s in range(0, len(idxs), step):
                grp = idxs[s:s + step]
                batch = torch.stack([torch.as_tensor(mels[i]) for i in grp])
                out = vocoder_inference(cfg, model, batch, device=device, fast_inference=fast_inference)
                for r, i in enumerate(grp):
                    audios[i] = out[r][: T * hop]
        return audios
    mel_batches, mel_frames = pad_mels_to_tensors(mels, batch_size)
    k = 0
    for mel_batch, mel_frame in zip(mel_batches, mel_frames):
        out = vocoder_inference(cfg, model, mel_batch, device=device, fast_inference=fast_inference)
        for i in range(mel_batch.shape[0]):
            audios[k] = out[i][: int(mel_frame[i]) * hop]
            k += 1
    return audios
