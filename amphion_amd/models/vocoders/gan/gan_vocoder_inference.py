"""Drop-in for models/vocoders/gan/gan_vocoder_inference.py: same two functions, same signatures.

``vocoder_inference`` is the reference's batched forward wrapper (:11-38).  ``synthesis_audios``
keeps the reference contract (:41-96: list of [n_mel, T_i] mels -> list of [T_i * hop] audios) but
runs each padded batch through the generator ONCE (``forward_ragged``: the kernels pad every layer at
each utterance's own end) instead of one utterance at a time, with identical results.
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
    """gan_vocoder_inference.py:41-96: list of [n_mel, T_i] mels -> list of [T_i * hop] audios.

    ``exact=True`` (default): utterances are sorted by length, zero-padded into batches of ``batch_size``
    and run through ``forward_ragged`` -- every kernel pads at each utterance's own end, so each audio is
    bit-identical to the reference's one-utterance-at-a-time loop while the GPU sees true batches.
    ``exact=False`` is the reference's *batched* behaviour (`pad_mels_to_tensors` + crop): utterances
    shorter than their batch differ from their B=1 result inside the receptive field of the tail.
    """
    device = next(model.parameters()).device
    if f0s is not None:
        raise NotImplementedError("f0-conditioned generators (NSF-HiFiGAN) are outside the HiFi-GAN/BigVGAN hot path")
    hop = model.cfg.preprocess.hop_size
    audios = [None] * len(mels)
    if exact and not hasattr(model, "forward_ragged"):
        # generators without per-utterance lengths in their kernels (MelGAN): one true batch per distinct
        # length -- still bit-identical to the reference's B=1 loop
        by_len = {}
        for i, m in enumerate(mels):
            by_len.setdefault(int(m.shape[-1]), []).append(i)
        for T, idxs in by_len.items():
            step = len(idxs) if not batch_size else int(batch_size)
            for s in range(0, len(idxs), step):
                grp = idxs[s:s + step]
                batch = torch.stack([torch.as_tensor(mels[i], dtype=torch.float32).cpu() for i in grp])
                out = vocoder_inference(cfg, model, batch, device=device, fast_inference=fast_inference)
                for r, i in enumerate(grp):
                    audios[i] = out[r][: T * hop]
        return audios
    if exact:
        order = sorted(range(len(mels)), key=lambda i: int(mels[i].shape[-1]), reverse=True)
        step = len(order) if not batch_size else int(batch_size)
        model.eval()
        with torch.no_grad():
            for s in range(0, len(order), step):
                grp = order[s:s + step]
                lens = [int(mels[i].shape[-1]) for i in grp]
                Tmax = lens[0]
                n_mel = int(mels[grp[0]].shape[0])
                batch = torch.zeros((len(grp), n_mel, Tmax), dtype=torch.float32)
                for r, i in enumerate(grp):
                    batch[r, :, : lens[r]] = torch.as_tensor(mels[i], dtype=torch.float32)
                out = model.forward_ragged(batch.to(device), lens).squeeze(1).cpu()
                for r, i in enumerate(grp):
                    audios[i] = out[r, : lens[r] * hop].clone()
        return audios
    mel_batches, mel_frames = pad_mels_to_tensors(mels, batch_size)
    k = 0
    for mel_batch, mel_frame in zip(mel_batches, mel_frames):
        out = vocoder_inference(cfg, model, mel_batch, device=device, fast_inference=fast_inference)
        for i in range(mel_batch.shape[0]):
            audios[k] = out[i][: int(mel_frame[i]) * hop]
            k += 1
    return audios
