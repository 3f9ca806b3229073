"""Drop-in side of models/vocoders/vocoder_inference.py: the three registries the reference's
drivers dispatch through (:39-75), ``load_nnvocoder`` (:397-457), ``tensorize`` (:460-468),
``synthesis`` (:471-515) and the batch loop of ``VocoderInference.inference`` (:334-374, ``inference_batches``),
bound to the MI355X generators.

``install_into_reference(mod)`` overwrites the ``hifigan`` / ``bigvgan`` / ``melgan`` entries of the reference
module's own dicts, which is how ``bins/vocoder/inference.py`` runs unchanged on the HIP path
(see amphion_amd/integration and INTEGRATION.md).
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

from amphion_amd.models.vocoders.gan import gan_vocoder_inference
from amphion_amd.models.vocoders.gan.generator import apnet, bigvgan, hifigan, melgan, nsfhifigan

_vocoders = {
    "apnet": apnet.APNet,
    "bigvgan": bigvgan.BigVGAN,
    "hifigan": hifigan.HiFiGAN,
    "melgan": melgan.MelGAN,
    "nsfhifigan": nsfhifigan.NSFHiFiGAN,
}

# Forward call for the generalized Inferencer (vocoder_inference.py:52-62)
_vocoder_forward_funcs = {
    "apnet": gan_vocoder_inference.vocoder_inference,
    "bigvgan": gan_vocoder_inference.vocoder_inference,
    "hifigan": gan_vocoder_inference.vocoder_inference,
    "melgan": gan_vocoder_inference.vocoder_inference,
    "nsfhifigan": gan_vocoder_inference.vocoder_inference,
}

# APIs for other tasks, e.g. SVC, TTS, TTA (vocoder_inference.py:65-75)
_vocoder_infer_funcs = {
    "apnet": gan_vocoder_inference.synthesis_audios,
    "bigvgan": gan_vocoder_inference.synthesis_audios,
    "hifigan": gan_vocoder_inference.synthesis_audios,
    "melgan": gan_vocoder_inference.synthesis_audios,
    "nsfhifigan": gan_vocoder_inference.synthesis_audios,
}


def install_into_reference(ref_module):
    """Point the reference module's registries at the MI355X implementations."""
    for name in _vocoders:
        ref_module._vocoders[name] = _vocoders[name]
        ref_module._vocoder_forward_funcs[name] = _vocoder_forward_funcs[name]
        ref_module._vocoder_infer_funcs[name] = _vocoder_infer_funcs[name]
    return ref_module


def _strip_module_prefix(sd, model_sd):
    """from_multi_gpu handling of vocoder_inference.py:312-327 / :421-435."""
    out = dict(model_sd)
    for k, v in sd.items():
        kk = k.split("module.")[-1]
        if kk in model_sd and v.shape == model_sd[kk].shape:
            out[kk] = v
    return out


def load_nnvocoder(cfg, vocoder_name, weights_file, from_multi_gpu=False):
    """vocoder_inference.py:397-457.  ``weights_file``: a legacy .pt (``generator_state_dict``) or a
    checkpoint folder holding ``pytorch_model.bin`` / ``model.safetensors``."""
    print("Loading Vocoder from Weights file: {}".format(weights_file))
    if vocoder_name not in _vocoders:
        raise KeyError(f"'{vocoder_name}' is not on the MI355X hot path (supported: {sorted(_vocoders)})")
    model = _vocoders[vocoder_name](cfg)
    if not os.path.isdir(weights_file):
        ckpt = torch.load(weights_file, map_location="cpu")
        if vocoder_name in ("bigvgan", "hifigan", "melgan", "nsfhifigan"):   # :409-437
            sd = ckpt["generator_state_dict"]
            if from_multi_gpu:
                sd = _strip_module_prefix(sd, model.state_dict())
        else:                                                                # every other vocoder, :438-446
            sd = ckpt["state_dict"]
        model.load_state_dict(sd)
    else:
        root = os.path.join(weights_file, "checkpoint")
        if not os.path.isdir(root):
            root = weights_file
        ls = [str(i) for i in Path(root).glob("*") if "audio" not in str(i) and os.path.isdir(str(i))]
        if ls:
            ls.sort(key=lambda x: int(x.split("_")[-3].split("-")[-1]), reverse=True)
            root = ls[0]
        if os.path.exists(os.path.join(root, "model.safetensors")):
            from safetensors.torch import load_file

            sd = load_file(os.path.join(root, "model.safetensors"))
        else:
            sd = torch.load(os.path.join(root, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(_strip_module_prefix(sd, model.state_dict()))
    if torch.cuda.is_available():
        model = model.cuda()
    return model.eval()


def inference_batches(cfg, model, batches, uids, output_dir, test_batch_size=None):
    """The batch loop of ``VocoderInference.inference`` (vocoder_inference.py:334-374) over collated batches
    (``VocoderCollator``: ``mel`` [B, T_max, n_mel], ``audio`` [B, T_max * hop], ``target_len`` [B], optional
    ``frame_pitch``): forward function of the registry on ``mel.transpose(-1, -2)``, chunk into items, crop prediction
    and ground truth to ``target_len * hop_size``, write ``pred/<uid>.wav`` and ``gt/<uid>.wav`` as 16-bit PCM.

    ``uids`` is the metadata order (``test_dataset.metadata[i * batch_size + j]["Uid"]``, :362).  Returns the list of
    cropped predictions (GPU tensors) in that order.  One conversion kernel + one D2H copy of int16 per batch
    (``utils.io.save_audios``) instead of per-item fp32 copies."""
    from amphion_amd.utils.io import save_audios

    os.makedirs(os.path.join(output_dir, "pred"), exist_ok=True)
    os.makedirs(os.path.join(output_dir, "gt"), exist_ok=True)
    fwd = _vocoder_forward_funcs[cfg.model.generator]
    device = next(model.parameters()).device
    hop, fs = cfg.preprocess.hop_size, cfg.preprocess.sample_rate
    preds = []
    k = 0
    from amphion_amd.utils.util import few_host_threads

    with few_host_threads():      # host-side copies of the loop (batch tensors, PCM files): a few threads, not the whole OpenMP pool
        for batch in batches:
            kw = {"f0s": batch["frame_pitch"].float()} if getattr(cfg.preprocess, "use_frame_pitch", False) else {}
            audio_pred = fwd(cfg, model, batch["mel"].transpose(-1, -2), device=device, **kw)
            lens = [int(l) * hop for l in batch["target_len"]]
            names = [str(u) for u in uids[k:k + len(lens)]]
            k += len(lens)
            pred = (audio_pred.squeeze(1) if audio_pred.dim() == 3 else audio_pred).to(device)   # the forward func returns CPU audio (:38)
            save_audios([os.path.join(output_dir, "pred", n + ".wav") for n in names], pred, lens, fs)
            save_audios([os.path.join(output_dir, "gt", n + ".wav") for n in names], batch["audio"].to(device), lens, fs)
            preds.extend(row[:l] for row, l in zip(pred, lens))
    return preds


def tensorize(data, device, n_samples):
    """vocoder_inference.py:460-468"""
    assert type(data) == list
    if n_samples:
        data = data[:n_samples]
    return [torch.as_tensor(x, device=device) for x in data]


def synthesis(cfg, vocoder_weight_file, n_samples, pred, f0s=None, batch_size=64, fast_inference=False):
    """vocoder_inference.py:471-515.  pred: list of numpy [T_i, n_mel] -> list of [T_i * hop] audios."""
    vocoder_name = cfg.model.generator
    print("Synthesis audios using {} vocoder...".format(vocoder_name))
    vocoder = load_nnvocoder(cfg, vocoder_name, weights_file=vocoder_weight_file, from_multi_gpu=True)
    device = next(vocoder.parameters()).device
    mels_pred = tensorize([p.T for p in pred], device, n_samples)
    print("For predicted mels, #sample = {}...".format(len(mels_pred)))
    return _vocoder_infer_funcs[vocoder_name](cfg, vocoder, mels_pred, f0s=f0s, batch_size=batch_size,
                                              fast_inference=fast_inference)
