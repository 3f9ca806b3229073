"""BASELINE.json configs[0] on the GPU: the 16 real clips of SURVEY.md Appendix E IN FULL (3.0 .. 12.9 s, 103.6 s in all)
through the whole C1 pipeline -- wav -> HIP mel front end -> HiFi-GAN V1 -> crop -> 16-bit PCM wav files -- with
``inference.batch_size = 1`` (the recipe default) and as one padded batch, against golden vectors of the REAL reference
functions (tests/golden/make_golden_c1.py: utils/mel.py::extract_mel_features and the reference HiFiGAN through
gan_vocoder_inference.vocoder_inference on EVERY clip; stored as every 4th mel frame / every 8th sample with a per-clip
offset).  Needs no reference tree on the GPU box."""
import os
import wave

import numpy as np
import pytest
import torch
from types import SimpleNamespace as NS

from oracle import pcm16 as pcm_oracle
from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_c1.npz"))
N_CLIPS, HOP, MEL_DECIM, WAV_DECIM = 16, 256, 4, 8
FRAMES = [int(G[f"frames_{i}"]) for i in range(N_CLIPS)]


def _cfg():
    pp = vo.preprocess_22k()
    return NS(preprocess=NS(**vars(pp), extract_amplitude_phase=False, use_frame_pitch=False),
              model=NS(generator="hifigan", hifigan=NS(**vo.hifigan_v1_hp())), inference=NS(batch_size=1))


def _model(cfg):
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    m = HiFiGAN(cfg)
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, vo.hifigan_v1_hp()), seed=1234, g_gain=1.0))
    return m.cuda().eval()


def _clips():
    return [torch.from_numpy(G[f"pcm_{i}"].astype(np.float32) / 32768.0) for i in range(N_CLIPS)]


def test_mel_of_every_clip_matches_the_reference_function(capsys):
    """extract_mel_features (utils/mel.py:111-170) on real audio: log-mel within 1e-4 wherever the mel energy is above
    1e-3 (the log amplifies rounding below that), 1e-3 everywhere."""
    from amphion_amd.utils import mel as M

    cfg = _cfg()
    worst_big, worst_all = 0.0, 0.0
    for i, wav in enumerate(_clips()):
        got = M.extract_mel_features(wav.unsqueeze(0).cuda(), cfg.preprocess).cpu().numpy()
        assert got.shape == (80, FRAMES[i])
        got = got[:, i % MEL_DECIM::MEL_DECIM]                 # the golden holds every 4th frame from frame i % 4
        ref = G[f"mel_{i}"]
        assert got.shape == ref.shape
        d = np.abs(got - ref)
        big = np.exp(ref) > 1e-3
        worst_big = max(worst_big, float(d[big].max()))
        worst_all = max(worst_all, float(d.max()))
    with capsys.disabled():
        print(f"\n[c1] log-mel max |err| over 16 clips: {worst_big:.2e} where mel > 1e-3, {worst_all:.2e} everywhere")
    assert worst_big <= 1e-4
    assert worst_all <= 1e-3


def _run_loop(tmp_path, batch_size):
    from amphion_amd.models.vocoders.vocoder_dataset import VocoderCollator
    from amphion_amd.models.vocoders.vocoder_inference import inference_batches
    from amphion_amd.utils import mel as M

    cfg = _cfg()
    model = _model(cfg)
    wavs = _clips()
    # the dataset items of vocoder_dataset.py:25-147: mel [n_mel, T] as extracted, audio [T * hop], target_len
    items = []
    for w in wavs:
        mel = M.extract_mel_features(w.unsqueeze(0).cuda(), cfg.preprocess).cpu().numpy()
        items.append({"mel": mel, "audio": w.numpy(), "target_len": mel.shape[1]})
    coll = VocoderCollator(cfg)
    batches = [coll(items[i:i + batch_size]) for i in range(0, N_CLIPS, batch_size)]
    uids = [f"clip{i:02d}" for i in range(N_CLIPS)]
    preds = inference_batches(cfg, model, batches, uids, str(tmp_path), test_batch_size=batch_size)
    return cfg, wavs, uids, preds


def _read_pcm(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        return w.getframerate(), np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)


def test_c1_pipeline_batch_size_1(tmp_path, capsys):
    """wav -> mel -> HiFi-GAN -> crop -> PCM16 files for all 16 clips, one utterance per batch
    (VocoderInference.inference, vocoder_inference.py:334-374)."""
    cfg, wavs, uids, preds = _run_loop(tmp_path, 1)
    worst = 0.0
    for i in range(N_CLIPS):
        p = preds[i].cpu()
        assert p.shape == (FRAMES[i] * HOP,)
        assert torch.isfinite(p).all()
        # the reference generator on the REFERENCE mel; ours ran on the HIP mel of the same clip.  Every clip, every 8th sample.
        worst = max(worst, float((p[i % WAV_DECIM::WAV_DECIM] - torch.from_numpy(G[f"wav_{i}"])).abs().max()))
        fs, pcm = _read_pcm(os.path.join(tmp_path, "pred", uids[i] + ".wav"))
        assert fs == 22050
        assert np.array_equal(pcm, pcm_oracle.float_to_pcm16(p.numpy()))          # bit-exact PCM of OUR fp32 audio
        fs, gt = _read_pcm(os.path.join(tmp_path, "gt", uids[i] + ".wav"))
        assert np.array_equal(gt, G[f"pcm_{i}"])                                   # in-range PCM16 survives the round trip
    with capsys.disabled():
        print(f"\n[c1] wav -> HIP mel -> HIP HiFi-GAN vs reference mel -> reference HiFiGAN: max |err| {worst:.2e} (all 16 clips, "
              f"{sum(FRAMES) * HOP / 22050:.1f} s of audio)")
    assert worst <= 1e-4


def test_c1_pipeline_one_padded_batch_matches_reference_semantics(tmp_path):
    """batch_size = 16: the collator zero-pads every mel to the longest clip and the loop crops each output to
    target_len * hop -- an item's tail then depends on the padding inside its last receptive field, exactly as in the
    reference; away from the tail it equals the batch-size-1 run."""
    _, _, _, p1 = _run_loop(tmp_path / "b1", 1)
    _, _, _, p16 = _run_loop(tmp_path / "b16", 16)
    for i in range(N_CLIPS):
        a, b = p1[i].cpu(), p16[i].cpu()
        assert a.shape == b.shape
        keep = a.shape[0] - 24 * HOP                 # receptive field of HiFi-GAN V1 < 24 frames
        assert torch.equal(a[:keep], b[:keep])
    longest = max(range(N_CLIPS), key=lambda i: FRAMES[i])
    assert torch.equal(p1[longest].cpu(), p16[longest].cpu())            # the longest clip is never padded
