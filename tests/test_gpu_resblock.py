"""GPU parity: the whole-ResBlock1 kernel (csrc/rb_f16x3.hip: three (dilated, dilation-1) pairs in ONE launch, x and the
residual never leaving the CU) against the chain of fused pairs bit for bit, against the oracle ops (hifigan.py:93-100), and
inside the generators (HiFi-GAN V1 with the kernel forced on vs off, dense and ragged)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _weights(C, k, n):
    ws1 = [_rand(C, C, k, seed=10 + p, scale=(C * k) ** -0.5) for p in range(n)]
    bs1 = [_rand(C, seed=20 + p, scale=0.1) for p in range(n)]
    ws2 = [_rand(C, C, k, seed=30 + p, scale=(C * k) ** -0.5) for p in range(n)]
    bs2 = [_rand(C, seed=40 + p, scale=0.1) for p in range(n)]
    return ws1, bs1, ws2, bs2


def _ref(ws1, bs1, ws2, bs2, x, dils, slope):
    """ResBlock1.forward hifigan.py:93-100 in fp64"""
    x = x.double()
    for w1, b1, w2, b2, d in zip(ws1, bs1, ws2, bs2, dils):
        k = w1.shape[2]
        xt = F.conv1d(F.leaky_relu(x, slope), w1.double(), b1.double(), dilation=d, padding=(k * d - d) // 2)
        xt = F.conv1d(F.leaky_relu(xt, slope), w2.double(), b2.double(), padding=(k - 1) // 2)
        x = xt + x
    return x


@pytest.fixture
def fusion():
    """amp_set_resblock_fusion for one call sequence: 2 = the kernel wherever it is built (any grid), 3 = four-wave tiles."""
    from amphion_amd import _lib

    def use(mode):
        _lib.check(_lib.lib().amp_set_resblock_fusion(mode))
    yield use
    _lib.check(_lib.lib().amp_set_resblock_fusion(-1))


CASES = [
    # C, k, dilations, B, T
    (32, 3, (1, 3, 5), 2, 3000),      # three 1000-column tiles per item (W = 1024, RH = 12)
    (32, 3, (1, 3, 5), 1, 1000),      # exactly one tile
    (32, 3, (1, 3, 5), 1, 1001),      # one tile + 1 column
    (32, 7, (1, 3, 5), 2, 2500),      # RH = 36
    (32, 5, (1, 3, 5), 3, 777),
    (32, 7, (1, 3, 5), 1, 7),         # T smaller than the receptive field
    (32, 3, (1, 3, 5), 1, 1),         # T = 1
    (32, 3, (2, 6), 2, 1500),         # two pairs, other dilations
    (32, 7, (1,), 2, 1100),           # one pair
    (64, 3, (1, 3, 5), 2, 1300),      # W = 512
    (64, 7, (1, 3, 5), 1, 900),
    (64, 5, (1, 3, 5), 2, 488),       # exactly one tile at k = 5?  (W - 2 * 24 = 464: one tile + 24)
    (64, 3, (1, 3, 5), 1, 3),
    (32, 11, (1, 3, 5), 2, 2000),     # RH = 60 (not in the policy: 23-31 % of every tile is halo)
    (64, 11, (1, 3, 5), 1, 800),
    (128, 3, (1, 3, 5), 2, 700),      # C = 128: 256-column tiles, 16 guard columns
    (128, 5, (1, 3, 5), 1, 300),
    # large launches (thousands of tiles, every wave of the chip busy; these shapes also exercised the strip-walking variant of
    # the kernel that was measured slower and not kept, profiles/negative_kernels/rb_strip_f16x3.hip.txt)
    (32, 3, (1, 3, 5), 600, 2100),
    (32, 7, (1, 3, 5), 40, 5000),
    (32, 11, (1, 3, 5), 70, 4000),
    (64, 11, (1, 3, 5), 33, 3000),
    (64, 7, (1, 3, 5), 300, 1400),
    (128, 3, (1, 3, 5), 300, 800),
    (32, 5, (2, 6), 520, 1500),
    # round 6: C = 64, k <= 5 as four waves x 256 columns (two workgroups per CU) -- mode 3 and the policy; thousands of tiles, a cut of T that leaves a 1-column last tile
    (64, 3, (1, 3, 5), 200, 2089),
    (64, 5, (1, 3, 5), 40, 3000),
]


@pytest.mark.parametrize("C,k,dils,B,T", CASES)
@pytest.mark.parametrize("mode", [2, 3])
def test_resblock_kernel_equals_pairs_bitwise(C, k, dils, B, T, mode, fusion):
    """Per output element the whole-resblock kernel runs the operations of the fused pairs in the same order; the x that a
    pair would have written to HBM and read back is the fp32 value the kernel keeps in registers -> the same bits, for both
    tile forms and every cut of the time axis."""
    from amphion_amd import _lib
    from hip_helpers import resblock_forward

    _lib.set_precision("f16x3")
    ws1, bs1, ws2, bs2 = _weights(C, k, len(dils))
    x = _rand(B, C, T, seed=5)
    y_pairs = resblock_forward(ws1, bs1, ws2, bs2, x, dilations=dils, fused=False)
    fusion(mode)
    y_rb = resblock_forward(ws1, bs1, ws2, bs2, x, dilations=dils, fused=True)
    assert not torch.isnan(y_rb).any()
    assert torch.equal(y_rb, y_pairs)


@pytest.mark.parametrize("C,k,dils,B,T", [CASES[0], CASES[3], CASES[9], CASES[10]])
def test_resblock_kernel_vs_oracle(C, k, dils, B, T, fusion):
    from amphion_amd import _lib
    from hip_helpers import resblock_forward

    _lib.set_precision("f16x3")
    ws1, bs1, ws2, bs2 = _weights(C, k, len(dils))
    x = _rand(B, C, T, seed=6)
    fusion(2)
    y = resblock_forward(ws1, bs1, ws2, bs2, x, dilations=dils, fused=True)
    ref = _ref(ws1, bs1, ws2, bs2, x, dils, 0.1)
    err = (y.double() - ref).abs().max().item()
    print(f"\n[rb] C={C} k={k} T={T}: max |err| vs fp64 = {err:.2e} (|ref| max {ref.abs().max().item():.2f})")
    assert err <= 5e-6 * max(1.0, ref.abs().max().item())


def test_resblock_policy_falls_back_on_small_grids(fusion):
    """Mode 1 (default) admits the kernel only when its grid fills the chip; the op-level entry then says UNSUPPORTED (the
    generator runs the pairs: same bits)."""
    from amphion_amd import _lib
    from hip_helpers import resblock_forward

    _lib.set_precision("f16x3")
    ws1, bs1, ws2, bs2 = _weights(32, 3, 3)
    fusion(1)
    with pytest.raises(_lib.AmpError) as e:
        resblock_forward(ws1, bs1, ws2, bs2, _rand(1, 32, 2000, seed=1), dilations=(1, 3, 5), fused=True)
    assert e.value.status == _lib.AMP_ERR_UNSUPPORTED


V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
          resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)


@pytest.mark.parametrize("mode", [2, 3])
def test_generator_with_resblock_kernel_equals_pairs(mode, fusion):
    """HiFi-GAN V1 (stages of 256 / 128 / 64 / 32 channels): forward and ragged forward with the whole-resblock kernel forced
    on for every block it covers == the same forwards on fused pairs only, bit for bit (MRF accumulate / mean modes, tile
    seams at the items' ends, utterances shorter than a tile)."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    from amphion_amd.utils.synthetic import randomize_, synthetic_mel

    m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
    mel = synthetic_mel(3, 80, 37, seed=3).cuda()
    lens = [37, 20, 3]
    with torch.no_grad():
        fusion(0)
        a = m(mel).cpu()
        ar = m.forward_ragged(mel, lens).cpu()
        fusion(mode)
        b = m(mel).cpu()
        br = m.forward_ragged(mel, lens).cpu()
    assert torch.equal(a, b)
    for i, n in enumerate(lens):
        assert torch.equal(ar[i, :, : n * 256], br[i, :, : n * 256])


def test_low_yield_resblock_runs_as_two_launches_with_the_same_bits(fusion, tmp_path):
    """Round 5 (generator.hip rb_split): a whole-resblock tile that keeps less than 80 % of its columns (C = 64, k = 11: 392 of 512) runs as pairs
    [0, 2) + [2, 3) once the launch has 1 024+ workgroups.  Same bits as the fused pairs (dense and ragged), and the launch manifest shows the
    two launches (4 convs, then 2 convs with the MRF sum) where a smaller batch shows one (6 convs)."""
    import os, subprocess, sys
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    from amphion_amd.utils.synthetic import randomize_, synthetic_mel

    # one up-sampling stage of 64 channels: T = 4 x 1 600 = 6 400 columns, 17 tiles per item; B = 64 -> 1 088 workgroups
    hp = dict(resblock="1", upsample_rates=[4], upsample_kernel_sizes=[8], upsample_initial_channel=128, resblock_kernel_sizes=[3, 11],
              resblock_dilation_sizes=[[1, 3, 5]] * 2)
    m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=4), model=NS(hifigan=NS(**hp)))), 1234).cuda().eval()
    mel = synthetic_mel(64, 80, 1600, seed=3).cuda()
    lens = [1600 - 23 * (i % 40) for i in range(64)]
    with torch.no_grad():
        fusion(0)
        a = m(mel).cpu()
        ar = m.forward_ragged(mel, lens).cpu()
        fusion(1)
        b = m(mel).cpu()
        br = m.forward_ragged(mel, lens).cpu()
    assert torch.equal(a, b)
    for i, n in enumerate(lens):
        assert torch.equal(ar[i, :, : n * 4], br[i, :, : n * 4])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\nfrom types import SimpleNamespace as NS\n"
            "from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN\n"
            "from amphion_amd.utils.synthetic import randomize_, synthetic_mel\n"
            "hp = %r\n"
            "m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=4), model=NS(hifigan=NS(**hp)))), 1234).cuda().eval()\n"
            "with torch.no_grad(): m(synthetic_mel(int(sys.argv[1]), 80, 1600, seed=3).cuda())\n"
            "torch.cuda.synchronize()\n" % (root, hp))
    for B, want in ((64, {"4 convs", "2 convs +sum"}), (32, {"6 convs +sum"})):
        man = tmp_path / f"manifest_{B}.tsv"
        r = subprocess.run([sys.executable, "-c", code, str(B)], capture_output=True, text=True, env=dict(os.environ, AMP_LAUNCH_MANIFEST=str(man), AMP_RB_FUSION="1", AMP_PRECISION="f16x3"), timeout=600)   # the POLICY's split, whatever the caller's environment says
        assert r.returncode == 0, r.stderr[-2000:]
        work = {l.rstrip("\n").split("\t")[4].split(": ")[1] for l in open(man) if "whole ResBlock C=64 k=11" in l}
        assert work == want, (B, work)


@pytest.mark.parametrize("arch", ["hifigan", "bigvgan"])
def test_concurrent_resblock_streams_bitwise(arch):
    """The resblocks of a stage on concurrent streams, their accumulating launches chained by events (amp_set_resblock_streams) ==
    the sequential chain, bit for bit: dense, ragged, and replayed from a captured hipGraph (fork / join by events)."""
    from amphion_amd import _lib
    from amphion_amd.utils.synthetic import randomize_, synthetic_mel

    L = _lib.lib()
    if arch == "hifigan":
        from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
        m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
        n_mel, frames = 80, 37
    else:
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
        hp = dict(V1, activation="snakebeta", snake_logscale=True, upsample_initial_channel=256)
        m = randomize_(BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp)))), 1234, g_gain=0.75).cuda().eval()
        n_mel, frames = 100, 19
    mel = synthetic_mel(3, n_mel, frames, seed=3).cuda()
    lens = [frames, frames // 2, 3]
    try:
        with torch.no_grad():
            _lib.check(L.amp_set_resblock_streams(0))
            a = m(mel).cpu()
            ar = m.forward_ragged(mel, lens).cpu()
            _lib.check(L.amp_set_resblock_streams(1))
            b = m(mel).cpu()
            br = m.forward_ragged(mel, lens).cpu()
            b2 = m(mel).cpu()                      # the side buffers are reused: a second forward gives the same
            replay, static_in, static_out = m.capture(3, frames)
            static_in.copy_(mel)
            replay()
            torch.cuda.synchronize()
            c = static_out.cpu()
    finally:
        _lib.check(L.amp_set_resblock_streams(-1))
    assert torch.equal(a, b) and torch.equal(a, b2) and torch.equal(a, c)
    for i, n in enumerate(lens):
        assert torch.equal(ar[i, :, : n * 256], br[i, :, : n * 256])


def test_resblock_streams_policy_is_small_launches_only():
    """-1 (default): concurrent while B * T <= 4096 frames -- the workspace the library asks for tells which form a shape gets."""
    from amphion_amd import _lib
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    from amphion_amd.utils.synthetic import randomize_, synthetic_mel

    L = _lib.lib()
    m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
    with torch.no_grad():
        m(synthetic_mel(1, 80, 8, seed=1).cuda())
    h = m._amp_handle
    try:
        _lib.check(L.amp_set_resblock_streams(0))
        seq_small, seq_big = L.amp_gen_workspace_bytes(h, 1, 256), L.amp_gen_workspace_bytes(h, 64, 256)
        _lib.check(L.amp_set_resblock_streams(-1))
        assert L.amp_gen_workspace_bytes(h, 1, 256) > seq_small          # 4 more buffers: R, TMP of two more resblocks
        assert L.amp_gen_workspace_bytes(h, 64, 256) == seq_big          # a full batch: sequential
        assert L.amp_set_resblock_streams(2) < 0
    finally:
        _lib.check(L.amp_set_resblock_streams(-1))


def test_horizontal_pairs_kernel_is_exercised_and_bitwise():
    """One utterance of HiFi-GAN V1: with concurrent resblocks on, stages 1-3 run as pair3_kernel launches (the k = 11 / 7 / 3 pairs of one
    dilation in ONE grid, pair3_f16x3.hip) + the MRF-mean launch.  Same bits as the sequential chain, dense and ragged, and the library
    reports the kernel it ran."""
    import ctypes

    from amphion_amd import _lib
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    from amphion_amd.utils.synthetic import randomize_, synthetic_mel

    L = _lib.lib()
    m = randomize_(HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**V1)))), 1234).cuda().eval()
    mel = synthetic_mel(2, 80, 96, seed=11).cuda()
    lens = [96, 41]
    try:
        with torch.no_grad():
            _lib.check(L.amp_set_resblock_streams(0))
            a, ar = m(mel).cpu(), m.forward_ragged(mel, lens).cpu()
            _lib.check(L.amp_set_resblock_streams(1))
            b, br = m(mel).cpu(), m.forward_ragged(mel, lens).cpu()
    finally:
        _lib.check(L.amp_set_resblock_streams(-1))
    assert torch.equal(a, b)
    for i, n in enumerate(lens):
        assert torch.equal(ar[i, :, : n * 256], br[i, :, : n * 256])
