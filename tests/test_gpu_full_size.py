"""BASELINE.json's FULL sizes (configs[1] HiFi-GAN V1 B=64 x 80 x 256, configs[2] BigVGAN-base B=32 x 100 x 256,
configs[4] VITS B=16 x 513 x 256), where the CPU oracle is too slow for the whole batch: size-independent
properties -- every item of the batch equals that item vocoded alone BIT FOR BIT (different grid sizes pick
different kernel variants: full-width tiles and fused pairs for the batch, half-width tiles for one utterance),
a batch equals its halves, a truncated input reproduces the prefix outside the receptive field of the cut --
plus the oracle on FOUR items of each batch (<= 1e-4 max-abs, BASELINE.json north_star)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
TOL = 1e-4
RF_FRAMES = 24      # > receptive half-width of HiFi-GAN V1 in mel frames (conv_pre 3 + stage-0 MRF 7.5 + the rest < 5)


def test_config2_hifigan_v1_full_size():
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    mel = synth.synth_mel(64, 80, 256, seed=0)
    with torch.no_grad():
        full = m(mel.cuda())
        assert tuple(full.shape) == (64, 1, 65536)
        assert torch.isfinite(full).all() and full.abs().max().item() <= 1.0
        for i in (0, 31, 63):
            assert torch.equal(full[i], m(mel[i:i + 1].cuda())[0]), f"item {i} alone differs from the batch"
        assert torch.equal(full[:32], m(mel[:32].cuda()))
        keep = (128 - RF_FRAMES) * 256
        assert torch.equal(m(mel[:, :, :128].cuda())[..., :keep], full[..., :keep])
        probe = [0, 21, 42, 63]                      # round 4: four items of the batch through the CPU oracle (round 3: two)
        ref = vo.hifigan_forward(sd, hp, mel[probe])
    err = (full[probe].cpu() - ref).abs().max().item()
    print(f"config 2 full size: |hip - oracle| on items {probe} = {err:.2e}")
    assert err <= TOL


@pytest.mark.parametrize("arch", ["hifigan", "bigvgan"])
def test_tile_order_is_bitwise_neutral(arch):
    """Ping-pong tile order (every other conv / pair / activation launch walks its tiles backwards, amp_set_pingpong) changes
    no bit of the waveform: tiles are independent (no atomics, x and y never alias across tiles)."""
    from amphion_amd import _lib

    if arch == "hifigan":
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

        hp = vo.hifigan_v1_hp()
        m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
        m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
        mel = synth.synth_mel(24, 80, 256, seed=5).cuda()
    else:
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN

        hp = vo.bigvgan_base_hp()
        m = BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp))))
        m.load_state_dict(synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75))
        mel = synth.synth_mel(12, 100, 200, seed=6).cuda()
    m = m.cuda().eval()
    L = _lib.lib()
    outs = {}
    try:
        with torch.no_grad():
            for on in (0, 1):
                _lib.check(L.amp_set_pingpong(on))
                outs[on] = m(mel).clone()
                outs[(on, "again")] = m(mel).clone()      # the launch parity differs between successive forwards
    finally:
        _lib.check(L.amp_set_pingpong(-1))
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[(1, "again")]) and torch.equal(outs[0], outs[(0, "again")])


def test_config3_bigvgan_base_full_size():
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN

    hp = vo.bigvgan_base_hp()
    m = BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp))))
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    mel = synth.synth_mel(32, 100, 256, seed=3)
    with torch.no_grad():
        full = m(mel.cuda())
        assert tuple(full.shape) == (32, 1, 65536) and torch.isfinite(full).all()
        for i in (0, 31):
            assert torch.equal(full[i], m(mel[i:i + 1].cuda())[0]), f"item {i} alone differs from the batch"
        assert torch.equal(full[16:], m(mel[16:].cuda()))
        probe = [0, 10, 21, 31]                      # round 4: four items (round 3: one); the batch runs the whole-AMPBlock kernel,
        ref = vo.bigvgan_forward(sd, hp, mel[probe])   # the single items above the separate launches -- bit-identical, checked above
    err = (full[probe].cpu() - ref).abs().max().item()
    print(f"config 3 full size: |hip - oracle| on items {probe} = {err:.2e}")
    assert err <= TOL


def test_config5_vits_full_size():
    from amphion_amd.models.tts.vits.vits import SynthesizerTrnDecodePath

    hp = vo.hifigan_v1_hp()
    net = SynthesizerTrnDecodePath(513, 192, 192, "1", hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"],
                                   hp["upsample_rates"], hp["upsample_initial_channel"], hp["upsample_kernel_sizes"])
    se = synth.synth_state_dict(synth.posterior_encoder_param_shapes(), 2468, g_gain=0.5)
    sf = synth.synth_state_dict(synth.coupling_block_param_shapes(), 1357, g_gain=0.5)
    sdec = synth.synth_state_dict(synth.hifigan_param_shapes(192, hp, vits=True), 4321)
    net.load_state_dict({**{"enc_q." + k: v for k, v in se.items()}, **{"flow." + k: v for k, v in sf.items()},
                         **{"dec." + k: v for k, v in sdec.items()}})
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(7)
    y = torch.rand(16, 513, 256, generator=gen)
    noise = torch.randn(16, 192, 256, generator=gen)
    lens = torch.full((16,), 256)
    with torch.no_grad():
        o, _, (z, z_p, z_hat) = net.reconstruct(y.cuda(), lens, noise=noise.cuda())
        assert tuple(o.shape) == (16, 1, 65536) and torch.isfinite(o).all()
        for i in (0, 15):
            oi, _, (zi, _, zhi) = net.reconstruct(y[i:i + 1].cuda(), lens[:1], noise=noise[i:i + 1].cuda())
            assert torch.equal(z[i], zi[0]) and torch.equal(z_hat[i], zhi[0]) and torch.equal(o[i], oi[0]), i
        probe = [0, 5, 10, 15]                       # round 4: four items (round 3: one)
        rz, _, _, rmask = vo.posterior_encoder_forward(se, "", y[probe], lens[probe], noise[probe])
        rzh = vo.coupling_block_forward(sf, "", vo.coupling_block_forward(sf, "", rz, rmask), rmask, reverse=True)
        ro = vo.hifigan_forward(sdec, hp, rzh * rmask)
    assert (z[probe].cpu() - rz).abs().max().item() <= TOL
    assert (z_hat[probe].cpu() - rzh).abs().max().item() <= TOL
    err = (o[probe].cpu() - ro).abs().max().item()
    print(f"config 5 full size: |hip - oracle| on items {probe} = {err:.2e}")
    assert err <= TOL


def test_config4_global_batch_on_one_gpu_equals_its_shards():
    """BASELINE configs[3]'s GLOBAL batch (512 x 80 x 256) on ONE GPU: every stage tensor of the late stages is 4.3 GB (byte offsets beyond
    2^32) -- the [512, 1, 65536] result equals the eight 64-item shards the ranks of an 8-GPU run would vocode, bit for bit (the
    gathered tensor == N single-GPU runs), and the workspace stays far inside the 288 GB."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
    m = m.cuda().eval()
    mel = synth.synth_mel(512, 80, 256, seed=11).cuda()
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        big = m(mel)
        assert big.shape == (512, 1, 65536) and torch.isfinite(big).all()
        for r in range(8):
            shard = m(mel[64 * r:64 * (r + 1)].contiguous())
            assert torch.equal(shard, big[64 * r:64 * (r + 1)]), f"shard {r} differs from the rows of the 512-item forward"
    assert torch.cuda.max_memory_allocated() < 64 * 2**30
