"""GPU parity: the whole-AMPBlock kernel (csrc/ampb_f16x3.hip: BigVGAN's AMPBlock1 -- three (Activation1d, dilated conv,
Activation1d, conv, + x) iterations, bigvgan.py:137-146 -- in ONE launch, the activations evaluated in registers) against the
twelve launches it replaces bit for bit, against the oracle ops in fp64, and inside the BigVGAN generator with the kernel forced on
vs off (dense and ragged)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _params(C, k, n):
    ws1 = [_rand(C, C, k, seed=10 + p, scale=(C * k) ** -0.5) for p in range(n)]
    bs1 = [_rand(C, seed=20 + p, scale=0.1) for p in range(n)]
    ws2 = [_rand(C, C, k, seed=30 + p, scale=(C * k) ** -0.5) for p in range(n)]
    bs2 = [_rand(C, seed=40 + p, scale=0.1) for p in range(n)]
    alphas = _rand(2 * n, C, seed=50, scale=0.3)
    betas = _rand(2 * n, C, seed=51, scale=0.3)
    return ws1, bs1, ws2, bs2, alphas, betas


def _ref64(ws1, bs1, ws2, bs2, alphas, betas, x, dils):
    """AMPBlock1.forward bigvgan.py:137-146 in fp64 (SnakeBeta, log-scale parameters)"""
    x = x.double()
    for i, (w1, b1, w2, b2, d) in enumerate(zip(ws1, bs1, ws2, bs2, dils)):
        k = w1.shape[2]
        xt = vo.activation1d(x, alphas[2 * i].double(), betas[2 * i].double(), True)
        xt = F.conv1d(xt, w1.double(), b1.double(), dilation=d, padding=(k * d - d) // 2)
        xt = vo.activation1d(xt, alphas[2 * i + 1].double(), betas[2 * i + 1].double(), True)
        xt = F.conv1d(xt, w2.double(), b2.double(), padding=(k - 1) // 2)
        x = xt + x
    return x


@pytest.fixture
def fusion():
    """amp_set_ampblock_fusion for one call sequence: 2 = the kernel wherever it is built (any grid), 3 = four-wave tiles."""
    from amphion_amd import _lib

    _lib.set_precision("f16x3")

    def use(mode):
        _lib.check(_lib.lib().amp_set_ampblock_fusion(mode))
    yield use
    _lib.check(_lib.lib().amp_set_ampblock_fusion(-1))


CASES = [
    # C, k, dilations, B, T        (W = 1024 at C = 32 in mode 2, 512 in mode 3; W = 512 at C = 64; halo 44 / 68 / 92 columns)
    (32, 3, (1, 3, 5), 2, 3000),
    (32, 3, (1, 3, 5), 1, 936),       # exactly one 1024-column tile (NT = 936)
    (32, 3, (1, 3, 5), 1, 940),       # one tile + 4 columns
    (32, 7, (1, 3, 5), 2, 2500),
    (32, 11, (1, 3, 5), 2, 2000),
    (32, 5, (1, 3, 5), 3, 776),
    (32, 7, (1, 3, 5), 1, 8),         # T smaller than the receptive field
    (32, 3, (1, 3, 5), 1, 4),         # T = 4: both ends inside the edge zone
    (32, 3, (2, 6), 2, 1500),         # two pairs, other dilations
    (32, 7, (1,), 2, 1100),           # one pair
    (64, 3, (1, 3, 5), 2, 1300),
    (64, 7, (1, 3, 5), 1, 900),
    (64, 11, (1, 3, 5), 1, 800),
    (64, 3, (1, 3, 5), 1, 12),
    # large launches: every CU busy, thousands of tiles
    (32, 3, (1, 3, 5), 300, 2100),
    (32, 11, (1, 3, 5), 40, 4000),
    (64, 7, (1, 3, 5), 150, 1400),
]


@pytest.mark.parametrize("C,k,dils,B,T", CASES)
@pytest.mark.parametrize("mode", [2, 3])
def test_ampblock_bitwise_equals_the_launches_it_replaces(fusion, C, k, dils, B, T, mode):
    from hip_helpers import ampblock_forward

    if mode == 3 and C != 32:
        pytest.skip("the four-wave tiles exist at C = 32 only")
    fusion(mode)
    n = len(dils)
    ws1, bs1, ws2, bs2, al, be = _params(C, k, n)
    x = _rand(B, C, T, seed=B + T, scale=1.5)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    ref = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=dils, fused=False)
    y = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=dils, fused=True)
    assert torch.isfinite(y).all()
    if not torch.equal(y, ref):
        bad = (y != ref).nonzero()
        raise AssertionError(f"{bad.shape[0]} of {y.numel()} differ; first {bad[0].tolist()} last {bad[-1].tolist()}; "
                             f"max |d| {(y - ref).abs().max().item():.3e}; columns {sorted(set(bad[:, 2].tolist()))[:24]}")


_FUZZ_OFF = int(__import__("os").environ.get("AMP_FUZZ_OFFSET", "0"))      # other seeds for a soak run (tests/test_gpu_fuzz.py)


@pytest.mark.parametrize("seed", range(_FUZZ_OFF, _FUZZ_OFF + 10))
def test_ampblock_random_shapes_bitwise(fusion, seed):
    """Seeded random (C, k, dilations, B, T, MRF mode, tile form): the one-launch block == the launches it replaces, bit for bit -- every tap count
    the kernel is built for (k = 5 runs on a ring of three taps since round 5), every MRF mode (the running sum is read four float4 at a time), one
    to three pairs, T from one quad to several tiles."""
    import random

    from hip_helpers import ampblock_forward

    rng = random.Random(7000 + seed)
    C = rng.choice([32, 32, 64])
    k = rng.choice([3, 5, 7, 11])
    n = rng.choice([1, 2, 3, 3])
    dils = tuple(rng.choice([1, 2, 3, 5]) for _ in range(n))
    if (k - 1) // 2 * max(dils) > 32:
        dils = tuple(min(d, 3) for d in dils)
    B = rng.choice([1, 2, 3, 7])
    T = 4 * rng.choice([1, 2, 3, 9, 64, 117, 234, 235, 400, 999])
    mode = rng.choice([2, 3]) if C == 32 else 2
    mrf = rng.choice([0, 1, 2])
    fusion(mode)
    ws1, bs1, ws2, bs2, al, be = _params(C, k, n)
    x = _rand(B, C, T, seed=seed, scale=1.5)
    y0 = _rand(B, C, T, seed=seed + 1, scale=0.7) if mrf else None
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    kw = dict(dilations=dils, mode=mrf, div=3.0, y0=y0)
    ref = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, fused=False, **kw)
    y = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, fused=True, **kw)
    assert torch.isfinite(y).all()
    assert torch.equal(y, ref), f"C={C} k={k} dils={dils} B={B} T={T} tile mode {mode} MRF mode {mrf}: max |d| {(y - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("C,k,dils,B,T", [(32, 3, (1, 3, 5), 2, 2000), (32, 11, (1, 3, 5), 1, 1200), (64, 7, (1, 3, 5), 2, 700),
                                           (32, 7, (1, 3, 5), 1, 8)])
def test_ampblock_vs_fp64_reference(fusion, C, k, dils, B, T):
    from hip_helpers import ampblock_forward

    fusion(2)
    ws1, bs1, ws2, bs2, al, be = _params(C, k, len(dils))
    x = _rand(B, C, T, seed=3, scale=1.5)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    y = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=dils)
    ref = _ref64(ws1, bs1, ws2, bs2, al, be, x, dils)
    err = (y.double() - ref).abs().max().item()
    print(f"C={C} k={k} |hip - fp64| = {err:.2e} (scale {ref.abs().max().item():.1f})")
    assert err <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mrf_mode", [1, 2])
@pytest.mark.parametrize("C,k", [(32, 7), (64, 3)])
def test_ampblock_mrf_modes_and_snake(fusion, C, k, mrf_mode):
    """the MRF accumulate / mean of the generator in the last conv's accumulator start (bigvgan.py:320-327), and plain Snake
    (beta = alpha, linear scale)"""
    from hip_helpers import ampblock_forward

    fusion(2)
    dils = (1, 3, 5)
    ws1, bs1, ws2, bs2, al, _ = _params(C, k, 3)
    al = al.abs() + 0.5
    x = _rand(2, C, 1600, seed=8, scale=1.2)
    y0 = _rand(2, C, 1600, seed=9)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    kw = dict(dilations=dils, mode=mrf_mode, div=3.0, y0=y0)
    ref = ampblock_forward(ws1, bs1, ws2, bs2, al, None, False, f, f, x, fused=False, **kw)
    y = ampblock_forward(ws1, bs1, ws2, bs2, al, None, False, f, f, x, fused=True, **kw)
    assert torch.equal(y, ref)


def test_ampblock_large_alpha_is_still_bitwise(fusion):
    """alpha = exp(4) = 54.6 and exp(9) = 8 103: |alpha * u| far beyond 1e5 -- one evaluation path for every argument, in every kernel"""
    from hip_helpers import ampblock_forward

    fusion(2)
    ws1, bs1, ws2, bs2, al, be = _params(32, 3, 3)
    al = al.clone()
    al[3, 5], al[0, 17], al[4, 30] = 4.0, 9.0, 12.5
    x = _rand(2, 32, 1400, seed=77, scale=1.5)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    ref = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=(1, 3, 5), fused=False)
    y = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=(1, 3, 5), fused=True)
    assert torch.isfinite(y).all() and torch.equal(y, ref)


def test_ampblock_refuses_what_it_does_not_cover(fusion):
    from amphion_amd import _lib
    from hip_helpers import ampblock_forward

    fusion(2)
    ws1, bs1, ws2, bs2, al, be = _params(32, 3, 3)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    with pytest.raises(_lib.AmpError):      # rows that are not whole float4
        ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, _rand(1, 32, 1001), dilations=(1, 3, 5))
    fusion(0)
    with pytest.raises(_lib.AmpError):
        ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, _rand(1, 32, 1000), dilations=(1, 3, 5))


def _bigvgan(hp, n_mel, sd):
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN

    cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256), model=NS(bigvgan=NS(**hp)))
    m = BigVGAN(cfg)
    m.load_state_dict(sd)
    return m.cuda().eval()


def test_bigvgan_with_and_without_the_kernel_dense_and_ragged(fusion, golden):
    """BigVGAN-base: forward with the whole-AMPBlock kernel wherever it is built == forward on separate launches, bit for bit --
    a dense batch and a ragged one (every utterance's own end goes through the kernel's edge path) -- and the reference's golden."""
    hp = vo.bigvgan_base_hp()
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75)
    m = _bigvgan(hp, 100, sd)
    gen = torch.Generator().manual_seed(3)
    mel = torch.randn(3, 100, 40, generator=gen)
    lens = torch.tensor([300, 211, 97, 300, 5, 160, 299, 64], dtype=torch.int32)
    melr = torch.randn(8, 100, 300, generator=gen)
    for i, l in enumerate(lens):
        melr[i, :, l:] = 0
    outs = {}
    for mode in (2, 0, 3):
        fusion(mode)
        with torch.no_grad():
            outs[mode] = (m(mel.cuda()).cpu(), m.forward_ragged(melr.cuda(), lens.cuda()).cpu(),
                          m(torch.from_numpy(golden["bigvgan_base_b2_t13_mel"]).cuda()).cpu())
    for mode in (2, 3):
        assert torch.equal(outs[mode][0], outs[0][0]), mode
        for i, l in enumerate(lens):
            assert torch.equal(outs[mode][1][i, :, : l * 256], outs[0][1][i, :, : l * 256]), (mode, i)
    assert (outs[2][2].numpy() - golden["bigvgan_base_b2_t13_wav"]).__abs__().max() <= 1e-4


def test_bigvgan_full_size_policy_equals_separate_launches(fusion):
    """BASELINE configs[2] shape (B = 32, T = 256): the default policy (whole-AMPBlock kernel in the late stages) against the separate
    launches, bit for bit over the whole batch."""
    hp = vo.bigvgan_base_hp()
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75)
    m = _bigvgan(hp, 100, sd)
    mel = _rand(32, 100, 256, seed=5).cuda()
    fusion(1)
    with torch.no_grad():
        y1 = m(mel)
    fusion(0)
    with torch.no_grad():
        y0 = m(mel)
    assert torch.equal(y1, y0)


@pytest.mark.parametrize("arch", ["bigvgan", "hifigan"])
def test_ragged_forward_ignores_what_lies_beyond_an_utterance(fusion, arch):
    """Beyond a ragged utterance's end the scratch tensors hold whatever the workspace held -- here NaN, on purpose.  Every layer takes
    its input as zero / replicated there, so the valid samples must not change (a zero FACTOR instead of a select would let NaN * 0
    through the conv taps)."""
    if arch == "bigvgan":
        hp = vo.bigvgan_base_hp()
        m = _bigvgan(hp, 100, synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75))
        n_mel = 100
    else:
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

        hp = vo.hifigan_v1_hp()
        m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
        m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
        m = m.cuda().eval()
        n_mel = 80
    fusion(2)
    from amphion_amd import _lib
    _lib.check(_lib.lib().amp_set_resblock_fusion(2))
    try:
        gen = torch.Generator().manual_seed(11)
        lens = torch.tensor([120, 37, 88, 120, 3, 64], dtype=torch.int32)
        mel = torch.randn(6, n_mel, 120, generator=gen)
        for i, l in enumerate(lens):
            mel[i, :, l:] = 0
        with torch.no_grad():
            y0 = m.forward_ragged(mel.cuda(), lens.cuda()).cpu()
            ws = m._amp_ws
            ws[: ws.numel() // 4 * 4].view(torch.float32).fill_(float("nan"))
            y1 = m.forward_ragged(mel.cuda(), lens.cuda()).cpu()
    finally:
        _lib.check(_lib.lib().amp_set_resblock_fusion(-1))
    for i, l in enumerate(lens):
        assert torch.isfinite(y1[i, :, : l * 256]).all(), i
        assert torch.equal(y1[i, :, : l * 256], y0[i, :, : l * 256]), i
