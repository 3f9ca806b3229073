"""Drop-in for models/vocoders/gan/gan_vocoder_inference.py: same two functions, same signatures.

``vocoder_inference`` is the reference's batched forward wrapper (:11-38).  ``synthesis_audios``
keeps the reference contract (:41-96: list of [n_mel, T_i] mels -> list of [T_i * hop] audios) and its
arithmetic, but runs each padded batch through the generator ONCE instead of one item at a time; with
``ragged=True`` it instead vocodes every utterance as if alone, in length-sorted true batches
(``forward_ragged``: the kernels pad every layer at each utterance's own end).
"""
import threading
import weakref

import torch

from amphion_amd import _lib
from amphion_amd.utils.util import few_host_threads, pad_mels_to_tensors


def _reference_range(model, run, device):
    """``run()`` with the fp32 reference's operand range.  The f16x3 kernels flag an activation beyond |x| = 4094 instead
    of producing the reference's value; the audio is about to be copied to the host anyway, so the flag is checked HERE,
    for this very batch: generators on the handle path fall back to their exact-fp32 kernels and repeat
    (``HipGenerator._amp_exact_range``); op-level models (MelGAN, APNet) raise ``AmpError`` -- never inf / NaN audio
    returned silently, never an error surfacing one batch late."""
    if hasattr(model, "_amp_exact_range"):
        return model._amp_exact_range(run)
    out = run()
    _lib.range_check(device)
    return out


_staging = weakref.WeakKeyDictionary()      # model -> grow-only pinned host buffer
_staging_lock = threading.Lock()


def _crops_to_host(model, out, lengths):
    """``[out[i, :lengths[i]] for i]`` as host tensors.  ``out`` [B, L] (device) goes to the host in ONE DMA into a pinned
    staging buffer kept on the model (grow-only) and the kept samples are cut out of it -- instead of ``out.cpu()``: a fresh
    pageable [B, L] tensor per call (26 MB for 64 utterances: mmap + first-touch page faults + a bounce-buffered copy, 3-4 ms
    against 0.5 ms for the pinned DMA) of which the padding is then thrown away."""
    n = out.numel()
    # (the buffer lives in a weak-keyed side table, not on the module: it must not travel with torch.save(model) / deepcopy, and two
    #  threads serving one model take turns on it)
    with _staging_lock:
        buf = _staging.get(model)
        if buf is None or buf.numel() < n or buf.dtype != out.dtype:
            buf = _staging[model] = torch.empty(n, dtype=out.dtype, pin_memory=True)
        host = buf[:n].view(out.shape)
        host.copy_(out, non_blocking=True)
        torch.cuda.current_stream(out.device).synchronize()
        return [host[i, : int(l)].clone() for i, l in enumerate(lengths)]


def _ragged(model, batch, lens):
    """forward_ragged, through the generator's cached hipGraph when the batch is small enough for one to pay (same bits)"""
    if hasattr(model, "_forward_graphed_view"):
        return model._forward_graphed_view(batch, lens)     # copied to the host by the caller before the next forward
    return model.forward_ragged(batch, lens)


def vocoder_inference(cfg, model, mels, f0s=None, device=None, fast_inference=False):
    """gan_vocoder_inference.py:11-38.  mels [B, n_mel, T] -> audios [B, T*hop] on the CPU."""
    model.eval()
    with torch.no_grad():
        mels = mels.to(device)
        if f0s is not None:
            f0s = f0s.to(device)

        def run():
            if f0s is None and not cfg.preprocess.extract_amplitude_phase:
                if hasattr(model, "_forward_graphed_view"):      # small batches of a repeated shape: a cached hipGraph (same bits)
                    return model._forward_graphed_view(mels)     # .cpu() below, before any other forward
                return model.forward(mels)
            if cfg.preprocess.extract_amplitude_phase:
                return model.forward(mels)[4]
            return model.forward(mels, f0s)

        output = _reference_range(model, run, mels.device)
        return output.squeeze(1).detach().cpu()


def synthesis_audios(cfg, model, mels, f0s=None, batch_size=None, fast_inference=False, ragged=False):
    """gan_vocoder_inference.py:41-96: list of [n_mel, T_i] mels -> list of [T_i * hop] audios.

    Default (``ragged=False``) = the reference's arithmetic: ``pad_mels_to_tensors`` zero-pads the list, in order,
    into batches of ``batch_size``; the reference then runs every (padded) item through the generator ONE AT A
    TIME and crops to ``frames * hop`` -- here each padded batch is ONE forward, which gives bit-identical audio
    because items of a batch never interact; the part of an item's zero padding that lies more than a receptive
    field behind its last frame cannot reach the samples that are kept and is not computed.  (Note what that
    arithmetic means: an utterance shorter than its batch carries the zero mel frames behind it through the network, so the last receptive field of its audio
    depends on the batch it was put in.)

    ``ragged=True``: utterances are sorted by length and run through ``forward_ragged`` -- every kernel pads at
    each utterance's own end, so each audio equals that utterance vocoded ALONE, independent of batching.
    """
    with few_host_threads():      # the padding / cropping copies below must not wake torch's whole OpenMP pool (utils/util.py)
        return _synthesis_audios(cfg, model, mels, f0s, batch_size, fast_inference, ragged)


def _synthesis_audios(cfg, model, mels, f0s, batch_size, fast_inference, ragged):
    device = next(model.parameters()).device
    hop = model.cfg.preprocess.hop_size
    audios = [None] * len(mels)
    if ragged:
        if f0s is not None or not hasattr(model, "forward_ragged"):
            raise NotImplementedError("ragged=True needs a generator with forward_ragged and no f0 input")
        order = sorted(range(len(mels)), key=lambda i: int(mels[i].shape[-1]), reverse=True)
        step = len(order) if not batch_size else int(batch_size)
        model.eval()
        with torch.no_grad():
            for s in range(0, len(order), step):
                grp = order[s:s + step]
                lens = [int(mels[i].shape[-1]) for i in grp]
                n_mel = int(mels[grp[0]].shape[0])
                batch = torch.zeros((len(grp), n_mel, lens[0]), dtype=torch.float32)
                for r, i in enumerate(grp):
                    batch[r, :, : lens[r]] = torch.as_tensor(mels[i], dtype=torch.float32)
                batch = batch.to(device)
                out = _reference_range(model, lambda: _ragged(model, batch, lens), device).squeeze(1)
                for i, a in zip(grp, _crops_to_host(model, out, [l * hop for l in lens])):
                    audios[i] = a
        return audios
    mels = [torch.as_tensor(m, dtype=torch.float32).cpu() for m in mels]
    mel_batches, mel_frames = pad_mels_to_tensors(mels, batch_size)
    k = 0
    for mel_batch, mel_frame in zip(mel_batches, mel_frames):
        f0_batch = None
        if f0s is not None:   # pad_f0_to_tensors (utils/util.py:83-111): zero-padded [B, T] per batch
            f0_batch = torch.zeros((mel_batch.shape[0], mel_batch.shape[-1]), dtype=torch.float32)
            for i in range(mel_batch.shape[0]):
                f = torch.as_tensor(f0s[k + i], dtype=torch.float32).reshape(-1).cpu()
                f0_batch[i, : f.shape[0]] = f[: mel_batch.shape[-1]]
        if (f0_batch is None and hasattr(model, "receptive_frames") and hasattr(model, "forward_ragged")
                and not getattr(cfg.preprocess, "extract_amplitude_phase", False)):
            # same samples, less work: item i only keeps [: frames_i * hop], and those depend on nothing further than a
            # receptive field beyond frame frames_i -- the zero frames behind that (and everything the generator would
            # compute from them) are skipped by running the padded batch as a ragged one with lengths frames_i + RF
            # (tiles beyond an utterance's length exit at once).  Bit-identical to the padded forward on what is kept
            # (tests/test_gpu_inference_api.py).
            T = int(mel_batch.shape[-1])
            rf = model.receptive_frames()
            ext = [min(T, int(f) + rf) for f in mel_frame]
            model.eval()
            with torch.no_grad():
                mel_dev = mel_batch.to(device)
                out = _reference_range(model, lambda: _ragged(model, mel_dev, ext), device).squeeze(1)
            for a in _crops_to_host(model, out, [int(f) * hop for f in mel_frame]):
                audios[k] = a
                k += 1
            continue
        out = vocoder_inference(cfg, model, mel_batch, f0s=f0_batch, device=device, fast_inference=fast_inference)
        for i in range(mel_batch.shape[0]):
            audios[k] = out[i][: int(mel_frame[i]) * hop]
            k += 1
    return audios
