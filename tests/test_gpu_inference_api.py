"""GPU: the drop-in inference API (vocoder_inference / synthesis_audios / synthesis / load_nnvocoder)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]


def _cfg():
    hp = vo.hifigan_v1_hp()
    return NS(preprocess=NS(n_mel=80, hop_size=256, sample_rate=22050, extract_amplitude_phase=False),
              model=NS(generator="hifigan", hifigan=NS(**hp))), hp


def test_vocoder_inference_and_synthesis_audios(tmp_path):
    from amphion_amd.models.vocoders import vocoder_inference as vi
    from amphion_amd.models.vocoders.gan.gan_vocoder_inference import synthesis_audios, vocoder_inference

    cfg, hp = _cfg()
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
    torch.save({"generator_state_dict": {"module." + k: v for k, v in sd.items()}}, tmp_path / "g.pt")
    model = vi.load_nnvocoder(cfg, "hifigan", str(tmp_path / "g.pt"), from_multi_gpu=True)
    assert next(model.parameters()).is_cuda

    mel = synth.synth_mel(2, 80, 21, seed=3)
    out = vocoder_inference(cfg, model, mel, device="cuda")
    assert out.device.type == "cpu" and tuple(out.shape) == (2, 21 * 256)
    with torch.no_grad():
        ref = vo.hifigan_forward(sd, hp, mel).squeeze(1)
    assert (out - ref).abs().max().item() <= 1e-4

    # list API, default = the reference's arithmetic (gan_vocoder_inference.py:41-96): every utterance is
    # zero-padded to the longest of ITS batch (list order, batch_size 2), vocoded, cropped
    Ts = (5, 9, 5, 14)
    mels = [synth.synth_mel(1, 80, T, seed=10 + T)[0] for T in Ts]
    auds = synthesis_audios(cfg, model, [m.cuda() for m in mels], batch_size=2)
    assert [a.shape[0] for a in auds] == [T * 256 for T in Ts]
    with torch.no_grad():
        for i, (m, a) in enumerate(zip(mels, auds)):
            Tpad = max(Ts[2 * (i // 2)], Ts[2 * (i // 2) + 1])
            padded = torch.zeros(1, 80, Tpad)
            padded[0, :, : m.shape[-1]] = m
            r = vo.hifigan_forward(sd, hp, padded)[0, 0, : m.shape[-1] * 256]
            assert (a - r).abs().max().item() <= 1e-4
    # ragged=True: every utterance as if vocoded alone, whatever the batching
    auds2 = synthesis_audios(cfg, model, [m.cuda() for m in mels], batch_size=4, ragged=True)
    assert [a.shape[0] for a in auds2] == [T * 256 for T in Ts]
    with torch.no_grad():
        for m, a in zip(mels, auds2):
            r = vo.hifigan_forward(sd, hp, m.unsqueeze(0))[0, 0]
            assert (a - r).abs().max().item() <= 1e-4
    assert (auds2[3] - auds[3]).abs().max().item() <= 1e-4          # the longest item of a batch is never padded

    pred = [m.numpy().T for m in mels]                               # synthesis() takes [T, n_mel] arrays
    auds3 = vi.synthesis(cfg, str(tmp_path / "g.pt"), None, pred, batch_size=2)   # same batching => same audio
    for a, b in zip(auds, auds3):
        assert torch.equal(a, b)


@pytest.mark.parametrize("arch", ["hifigan", "bigvgan"])
def test_forward_ragged_is_bit_identical_to_per_utterance_forward(arch):
    """amp_gen_forward_ragged: every kernel pads at each utterance's own end, so a zero-padded batch gives
    exactly the audio of the reference's one-utterance-at-a-time loop (gan_vocoder_inference.py:74-96)."""
    hp = vo.hifigan_v1_hp() if arch == "hifigan" else vo.bigvgan_base_hp()
    if arch == "hifigan":
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN as Net
        n_mel = 80
        sd = synth.synth_state_dict(synth.hifigan_param_shapes(n_mel, hp), 1234)
        cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256), model=NS(hifigan=NS(**hp)))
    else:
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN as Net
        n_mel = 100
        sd = synth.synth_state_dict(synth.bigvgan_param_shapes(n_mel, hp), 1234, g_gain=0.75)
        cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256), model=NS(bigvgan=NS(**hp)))
    m = Net(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    lens = [23, 7, 1, 16, 23]
    mels = [synth.synth_mel(1, n_mel, T, seed=40 + i)[0] for i, T in enumerate(lens)]
    batch = torch.zeros(len(lens), n_mel, max(lens))
    for i, (mel, T) in enumerate(zip(mels, lens)):
        batch[i, :, :T] = mel
    with torch.no_grad():
        out = m.forward_ragged(batch.cuda(), lens).cpu()
        for i, (mel, T) in enumerate(zip(mels, lens)):
            solo = m(mel.unsqueeze(0).cuda()).cpu()
            assert torch.equal(out[i, 0, : T * 256], solo[0, 0]), (arch, i, T)
    with pytest.raises(ValueError):
        m.forward_ragged(batch.cuda(), [23, 7, 0, 16, 24])


def test_hipgraph_capture_replays_bit_identically():
    """generator.capture(): one hipGraphLaunch instead of 51 kernel launches, same bits as the eager forward."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
    m = m.cuda().eval()
    replay, static_in, static_out = m.capture(1, 40)
    for seed in (1, 2):
        mel = synth.synth_mel(1, 80, 40, seed=seed).cuda()
        with torch.no_grad():
            eager = m(mel).clone()
        static_in.copy_(mel)
        out = replay()
        torch.cuda.synchronize()
        assert out is static_out and torch.equal(out, eager)


def test_hipgraph_capture_survives_other_shapes_and_refuses_stale_weights():
    """The captured graph owns its workspace (an eager forward of a larger shape re-allocates the module's scratch
    without touching it) and is tied to the packed weights: after a parameter change replay() raises instead of
    reading freed memory."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    mel = synth.synth_mel(1, 80, 40, seed=3).cuda()
    with torch.no_grad():
        eager = m(mel).clone()
    replay, static_in, static_out = m.capture(1, 40)
    with torch.no_grad():
        m(synth.synth_mel(3, 80, 200, seed=4).cuda())       # larger (B, T): the module's workspace is re-allocated
        junk = torch.full((64, 1024, 1024), 7.0, device="cuda")   # recycle whatever the allocator freed
    static_in.copy_(mel)
    out = replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    del junk
    with torch.no_grad():
        m.conv_post.bias.add_(0.25)                           # weights changed -> handle rebuilt on the next forward
    with pytest.raises(RuntimeError, match="stale"):
        replay()
    with torch.no_grad():
        m(mel)                                                # rebuilds the handle
    with pytest.raises(RuntimeError, match="stale"):
        replay()
    replay2, static_in2, _ = m.capture(1, 40)
    static_in2.copy_(mel)
    out2 = replay2()
    torch.cuda.synchronize()
    with torch.no_grad():
        assert torch.equal(out2, m(mel))


def test_forward_graphed_buckets_replay_bit_identically_and_follow_the_weights():
    """forward_graphed: the second call of a (B, 32-frame bucket) captures a ragged graph, later calls of ANY length in the bucket
    replay it -- bit-identical to the eager forward of each utterance; a weight change drops the cache instead of replaying stale
    weights; big batches and grad-requiring inputs stay eager."""
    import pickle

    from amphion_amd.models.vocoders.gan.generator import _engine
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
    m = m.cuda().eval()
    with torch.no_grad():
        for T in (40, 37, 64, 33, 40):                         # one bucket (64 frames): eager, capture, replay, replay, replay
            mel = synth.synth_mel(1, 80, T, seed=T).cuda()
            got = m.forward_graphed(mel)
            assert got.shape == (1, 1, T * 256) and torch.equal(got, m(mel)), T
        cache = _engine._graph_caches[m]
        assert isinstance(cache[(1, 64, "cuda:0")], tuple)     # captured
        # ragged batch through the same mechanism
        mel = synth.synth_mel(3, 80, 50, seed=9).cuda()
        lens = [50, 21, 3]
        for _ in range(3):
            got = m.forward_graphed(mel, lens)
        want = m.forward_ragged(mel, lens)
        for i, n in enumerate(lens):
            assert torch.equal(got[i, :, : n * 256], want[i, :, : n * 256])
        # weights change: the next call must not replay the old graph
        mel = synth.synth_mel(1, 80, 40, seed=1).cuda()
        before = m.forward_graphed(mel).clone()
        m.conv_post.bias.add_(0.25)
        after = m.forward_graphed(mel)
        assert torch.equal(after, m(mel)) and not torch.equal(after, before)
        assert torch.equal(m.forward_graphed(mel), after) and torch.equal(m.forward_graphed(mel), after)   # re-captured, replayed
        # a full batch stays eager (nothing cached for it)
        big = synth.synth_mel(8, 80, 256, seed=2).cuda()
        assert torch.equal(m.forward_graphed(big), m(big))
        assert not any(isinstance(k, tuple) and k[0] == 8 for k in _engine._graph_caches[m])
    # the cache is a weak-keyed side table, not module state
    assert not any("graph" in k for k in vars(m)) and len(pickle.dumps(m.state_dict())) > 0


@pytest.mark.parametrize("arch", ["hifigan", "bigvgan"])
def test_list_api_skips_dead_padding_without_changing_a_bit(arch):
    """synthesis_audios (default, the reference's pad-then-crop arithmetic) runs a padded batch as a ragged one with
    lengths frames + receptive_frames(): what lies further behind an item's last frame cannot reach the samples that are
    kept.  The audios equal the crops of the plain padded forward bit for bit -- also with the receptive field cut to
    the bone (receptive_frames() - 4 would not do: checked as a sanity bound on the margin)."""
    from amphion_amd.models.vocoders.gan.gan_vocoder_inference import synthesis_audios

    if arch == "hifigan":
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN as Net
        hp, n_mel = vo.hifigan_v1_hp(), 80
        sd = synth.synth_state_dict(synth.hifigan_param_shapes(n_mel, hp), 1234)
        cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256, extract_amplitude_phase=False), model=NS(hifigan=NS(**hp)))
    else:
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN as Net
        hp, n_mel = vo.bigvgan_base_hp(), 100
        sd = synth.synth_state_dict(synth.bigvgan_param_shapes(n_mel, hp), 1234, g_gain=0.75)
        cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256, extract_amplitude_phase=False), model=NS(bigvgan=NS(**hp)))
    m = Net(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    Ts = [150, 31, 90, 7, 150, 64, 1, 120]
    mels = [synth.synth_mel(1, n_mel, T, seed=40 + i)[0] for i, T in enumerate(Ts)]
    auds = synthesis_audios(cfg, m, mels, batch_size=8)
    batch = torch.zeros(8, n_mel, 150)
    for i, mel in enumerate(mels):
        batch[i, :, : Ts[i]] = mel
    with torch.no_grad():
        full = m(batch.cuda()).squeeze(1).cpu()
        rf = m.receptive_frames()
        assert 12 <= rf <= 40
        for i, T in enumerate(Ts):
            assert torch.equal(auds[i], full[i, : T * 256]), (arch, i, T)
        # the margin is not generous by accident: with 4 frames the kept samples DO change
        short = m.forward_ragged(batch.cuda(), [min(150, T + 4) for T in Ts]).squeeze(1).cpu()
        assert any(not torch.equal(short[i, : T * 256], full[i, : T * 256]) for i, T in enumerate(Ts) if T + 4 < 150)


def test_launch_manifest_states_every_launch_of_a_forward(tmp_path):
    """AMP_LAUNCH_MANIFEST=<file> (read once per process, so a child process): every launch of a forward appends
    kernel<template arguments> \\t workgroups \\t algorithmic GFLOP \\t MB \\t what -- the names are the ones amp_gen_kernel_name reports, the conv
    FLOPs of one HiFi-GAN V1 forward add up to SURVEY.md's 2 398 848 FLOP per output sample (conv_pre .. conv_post, weights aside)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    man = tmp_path / "manifest.tsv"
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from types import SimpleNamespace as NS\n"
        "from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN\n"
        "from amphion_amd.utils.synthetic import randomize_, synthetic_mel\n"
        "hp = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,\n"
        "          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)\n"
        "m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256, sample_rate=22050), model=NS(hifigan=NS(**hp)))), 1234).cuda().eval()\n"
        "mel = synthetic_mel(16, 80, 64, seed=0).cuda()\n"
        "m.set_profiling(1)\n"
        "with torch.no_grad(): m(mel)\n"
        "torch.cuda.synchronize()\n"
        "print('|'.join(sorted({n for i in range(4) for j in range(3) for n in m.kernel_names(100 + 16 * i + j, 0)})))\n" % root)
    env = dict(os.environ, AMP_LAUNCH_MANIFEST=str(man))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    reported = set(r.stdout.strip().splitlines()[-1].split("|"))
    rows = [l.rstrip("\n").split("\t") for l in open(man)]
    assert rows and all(len(p) == 5 for p in rows)
    names = {p[0] for p in rows}
    assert reported <= names, (reported - names)                       # every resblock kernel the handle reports is in the manifest
    gflop = sum(float(p[2]) for p in rows)
    samples = 16 * 64 * 256
    assert abs(gflop * 1e9 / samples - 2398848) <= 0.002 * 2398848, gflop * 1e9 / samples
    assert all(int(p[1]) > 0 and float(p[3]) >= 0.0 for p in rows)
