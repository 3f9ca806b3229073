"""GPU parity of the kernels that exist for the f16x3 arithmetic only (run once, under f16x3): the whole-K frame-rate conv
kernel (csrc/conv_small_f16x3.hip) against the pipelined one bit for bit, and the row-blocked kernel of the transposed /
k = 3 convs (csrc/conv_blk_f16x3.hip) against the pipelined one bit for bit."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f16x3():
    from amphion_amd import _lib

    _lib.set_precision("f16x3")
    yield


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---- frame-rate convs: the whole-K kernel (csrc/conv_small_f16x3.hip) against the pipelined one -------------
SMALL_CASES = [
    # cin, cout, k, dilation, B, T   (grids under 384 workgroups, Cin <= 256, M > 64)
    (192, 384, 5, 1, 2, 19),     # WN in-layer (VITS)
    (192, 384, 1, 1, 3, 200),    # WN res_skip
    (96, 192, 1, 1, 2, 37),      # coupling pre
    (192, 96, 1, 1, 2, 64),      # coupling post: M not a multiple of 128
    (256, 256, 3, 2, 1, 257),    # 16 chunks, dilation 2, ragged last tile
    (100, 130, 5, 3, 2, 75),     # Cin not a multiple of 16, M not a multiple of 32
    (17, 65, 3, 1, 1, 1),        # T = 1
    (256, 256, 11, 3, 1, 2048),  # the C = 256 stage of ONE utterance (hifigan.py:93-100 unfused): 128 x 32 tiles
    (256, 256, 11, 5, 1, 2048),  # dilation 5: 50-column halo -> 128 x 64 tiles
    (256, 256, 7, 1, 2, 700),
]


@pytest.mark.parametrize("cin,cout,k,d,B,T", SMALL_CASES)
def test_small_conv_bitwise(cin, cout, k, d, B, T):
    """Same chunk / tap / (hh, hl, lh) order per output element -> the same bits as conv_f16x3.hip, with and without the
    fused prologue / epilogue; and within the conv tolerance of the oracle."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    w = _rand(cout, cin, k, seed=1, scale=(cin * k) ** -0.5)
    b = _rand(cout, seed=2, scale=0.1)
    x = _rand(B, cin, T, seed=3)
    res = _rand(B, cout, T, seed=4)
    pad = (k * d - d) // 2
    L = _lib.lib()
    outs = {}
    try:
        for on in (1, 0):
            _lib.check(L.amp_set_small_conv(on))
            outs[on] = (conv_forward(w, b, x, dilation=d, padding=pad),
                        conv_forward(w, b, x, dilation=d, padding=pad, slope_in=0.1, res=res, slope_out=0.2),
                        conv_forward(w, None, x, dilation=d, padding=pad))
    finally:
        _lib.check(L.amp_set_small_conv(1))
    for a, c in zip(outs[1], outs[0]):
        assert torch.equal(a, c)
    ref = F.conv1d(x, w, b, dilation=d, padding=pad)
    assert (outs[1][0] - ref).abs().max().item() <= 2e-5
    ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=pad) + res, 0.2)
    assert (outs[1][1] - ref).abs().max().item() <= 2e-5


# ---- transposed convs / k = 3 convs with M % 256 == 0 on 512+ workgroups: the row-blocked kernel ------------------------
BLK_CASES = [
    # transposed, cin, cout, k, stride, dilation, B, T
    (True, 64, 32, 16, 8, 1, 24, 2100),    # M = 256, 4 chunks (two per round in mode 2), float4 polyphase stores
    (True, 48, 64, 16, 8, 1, 24, 1000),    # M = 512: two row groups; 3 chunks (odd: one per round in either mode)
    (True, 32, 64, 8, 4, 1, 24, 2101),     # stride 4
    (True, 32, 128, 4, 2, 1, 24, 2100),    # stride 2: the 8-byte store path (hifigan.py's last two up-sampling layers)
    (True, 40, 256, 6, 3, 1, 16, 1000),    # odd stride: generic scatter; Cin not a multiple of 16
    (True, 32, 32, 15, 8, 1, 24, 2100),    # k < 2 * stride: the second tap of the last phase is zero
    (True, 32, 32, 24, 8, 1, 24, 2690),    # k = 3 * stride: three taps -> the KT = 3 kernel with dstep = -1
    (False, 256, 256, 3, 1, 1, 8, 8200),   # the C = 256 stage's k = 3 convs (hifigan.py:93-100), ragged last tile
    (False, 256, 256, 3, 1, 5, 8, 8200),
    (False, 100, 256, 3, 1, 3, 8, 8321),   # Cin not a multiple of 16
    (False, 64, 512, 3, 1, 1, 4, 8200),    # two row groups
    (False, 256, 256, 7, 1, 3, 8, 8200),   # mode 3: the A-fragment ring form for k = 7 / 11 (the C = 256 stage)
    (False, 256, 256, 11, 1, 5, 8, 8200),
    (False, 48, 256, 11, 1, 1, 8, 8300),   # three chunks
    (False, 16, 512, 7, 1, 1, 4, 8200),    # ONE chunk: the ring's look-ahead into the "next chunk" reads the pad / next row block
]


@pytest.mark.parametrize("tr,cin,cout,k,s,d,B,T", BLK_CASES)
def test_blocked_conv_bitwise(tr, cin, cout, k, s, d, B, T):
    """Same accumulator start, chunk / tap / (hh, hl, lh) order and epilogue per output element -> the same bits as
    conv_f16x3.hip in every mode (0 pipelined, 1 one chunk per staging round, 2 two); within the conv tolerance of
    torch's fp32 conv."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    w = _rand(*((cin, cout, k) if tr else (cout, cin, k)), seed=1, scale=(cin * k / s) ** -0.5)
    b = _rand(cout, seed=2, scale=0.1)
    x = _rand(B, cin, T, seed=3)
    pad = (k - s) // 2 if tr else (k * d - d) // 2
    kw = dict(transposed=tr, stride=s, dilation=d, padding=pad)
    res = None if tr else _rand(B, cout, T, seed=4)
    L = _lib.lib()
    outs = {}
    try:
        for mode in (0, 1, 2, 3):
            _lib.check(L.amp_set_conv_blk(mode))
            outs[mode] = [conv_forward(w, b, x, **kw), conv_forward(w, None, x, slope_in=0.1, slope_out=0.2, **kw)]
            if res is not None:
                outs[mode].append(conv_forward(w, b, x, slope_in=0.1, res=res, **kw))
    finally:
        _lib.check(L.amp_set_conv_blk(-1))
    for mode in (1, 2, 3):
        for a, c in zip(outs[mode], outs[0]):
            assert torch.isfinite(a).all()
            assert torch.equal(a, c), f"mode {mode} differs from the pipelined kernel"
    ref = F.conv_transpose1d(x, w, b, stride=s, padding=pad) if tr else F.conv1d(x, w, b, dilation=d, padding=pad)
    err = (outs[2][0] - ref).abs().max().item()
    assert err <= 2e-5, err


RG_CASES = [
    # transposed, cin, cout, k, stride, dilation, B, T   (several row groups per x tile)
    (True, 64, 64, 16, 8, 1, 24, 1000),     # blocked kernel, 2 row groups of 256
    (True, 32, 48, 16, 8, 1, 3, 700),       # pipelined kernel, M = 384: 3 row groups of 128
    (False, 256, 256, 7, 1, 3, 8, 4100),    # the C = 256 stage (hifigan.py:93-100): 2 row groups of 128, ragged last tile
    (False, 64, 200, 11, 1, 1, 2, 777),     # M = 200: the last row group is partial
    (False, 192, 384, 5, 1, 1, 2, 64),      # small grid: half-width tiles / whole-K kernel
]


@pytest.mark.parametrize("tr,cin,cout,k,s,d,B,T", RG_CASES)
def test_row_group_order_is_bitwise_neutral(tr, cin, cout, k, s, d, B, T):
    """Row group as the fastest grid index (one XCD's L2 serves x to all row groups of a tile) vs the 2-D grid."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    w = _rand(*((cin, cout, k) if tr else (cout, cin, k)), seed=1, scale=(cin * k / s) ** -0.5)
    b = _rand(cout, seed=2, scale=0.1)
    x = _rand(B, cin, T, seed=3)
    pad = (k - s) // 2 if tr else (k * d - d) // 2
    kw = dict(transposed=tr, stride=s, dilation=d, padding=pad)
    L = _lib.lib()
    outs = {}
    try:
        for on in (0, 1):
            _lib.check(L.amp_set_conv_rg_fast(on))
            outs[on] = conv_forward(w, b, x, slope_in=0.1, **kw)
    finally:
        _lib.check(L.amp_set_conv_rg_fast(-1))
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1])
    ref = (F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=pad) if tr
           else F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=pad))
    assert (outs[1] - ref).abs().max().item() <= 2e-5


NARROW_CASES = [
    # cin, cout, k, dilation, B, T      (128 output rows: two waves along the columns, 256-column workgroup tiles; 64 rows: pipelined)
    (128, 128, 3, 1, 8, 8200),
    (128, 128, 3, 5, 8, 8200),
    (128, 128, 7, 3, 8, 8200),
    (128, 128, 11, 5, 8, 8200),
    (128, 128, 11, 1, 6, 8193),        # unaligned rows, ragged last tile
    (100, 128, 7, 1, 8, 8200),         # Cin not a multiple of 16
    (64, 64, 3, 1, 8, 16500),
    (64, 64, 7, 5, 8, 16500),
    (64, 64, 11, 3, 8, 16500),
    (16, 64, 11, 1, 8, 16500),         # ONE chunk
]


@pytest.mark.parametrize("cin,cout,k,d,B,T", NARROW_CASES)
def test_narrow_blocked_conv_bitwise(cin, cout, k, d, B, T):
    """Round 4: the row-blocked kernel with two waves along the columns for 128-row convs (BigVGAN's unpaired AMPBlock convs):
    amp_set_conv_blk_narrow 0 (pipelined kernel) / 1 (policy: k >= 7) / 2 (every tap count) give the same bits, incl. residual and
    leaky-ReLU on load / store."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    w = _rand(cout, cin, k, seed=1, scale=(cin * k) ** -0.5)
    b = _rand(cout, seed=2, scale=0.1)
    x = _rand(B, cin, T, seed=3)
    res = _rand(B, cout, T, seed=4)
    pad = (k * d - d) // 2
    L = _lib.lib()
    outs = {}
    try:
        for mode in (0, 1, 2):
            _lib.check(L.amp_set_conv_blk_narrow(mode))
            outs[mode] = [conv_forward(w, b, x, dilation=d, padding=pad), conv_forward(w, b, x, dilation=d, padding=pad, res=res),
                          conv_forward(w, None, x, dilation=d, padding=pad, slope_in=0.1, slope_out=0.2)]
    finally:
        _lib.check(L.amp_set_conv_blk_narrow(-1))
    for mode in (1, 2):
        for a, c in zip(outs[mode], outs[0]):
            assert torch.isfinite(a).all() and torch.equal(a, c), f"narrow mode {mode} differs from the pipelined kernel"
    ref = F.conv1d(x, w, b, dilation=d, padding=pad)
    assert (outs[2][0] - ref).abs().max().item() <= 2e-5


def test_blocked_conv_switch_rejects_bad_mode():
    from amphion_amd import _lib

    assert _lib.lib().amp_set_conv_blk(7) != 0
    assert _lib.lib().amp_set_conv_blk(-1) == 0




# ---- square Conv1d at C = 64 / 128 / 256 (unfused AMPBlock convs, the C = 256 stage): the persistent strip kernel --------
STRIP_CASES = [
    # C, k, dilation, B, T
    (128, 7, 1, 6, 2100),     # BigVGAN stage 1 (bigvgan.py:137-146): 2 x 2 waves, 192-column steps, ragged last tile
    (128, 7, 3, 6, 2100),
    (128, 11, 5, 4, 1900),    # 50-column receptive field
    (128, 11, 1, 3, 191),     # ONE partial tile per item
    (128, 3, 1, 6, 2100),     # two chunks per staging round
    (128, 3, 5, 6, 1537),
    (128, 5, 2, 4, 1000),
    (64, 7, 1, 6, 4100),      # 1 x 4 waves, 384-column steps
    (64, 11, 3, 4, 3000),
    (64, 3, 3, 4, 3000),
    (64, 11, 1, 2, 383),
    (256, 3, 1, 6, 1000),     # 4 x 1 waves, 96-column steps (hifigan.py:93-100 at C = 256)
    (256, 7, 5, 6, 1000),
    (256, 11, 3, 4, 777),
    (256, 11, 1, 1, 1),       # T = 1
]


@pytest.mark.parametrize("C,k,d,B,T", STRIP_CASES)
def test_strip_conv_bitwise(C, k, d, B, T):
    """Same accumulator start, chunk / tap / (hh, hl, lh) order and epilogue per output element -> the same bits as the per-tile
    kernels, however many column tiles a strip walks (1: only the prologue / flush path, 2, 5: the software-pipelined boundary); within the
    conv tolerance of torch's fp32 conv."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    w = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5)
    b = _rand(C, seed=2, scale=0.1)
    x = _rand(B, C, T, seed=3)
    res = _rand(B, C, T, seed=4)
    pad = (k * d - d) // 2
    kw = dict(dilation=d, padding=pad)
    L = _lib.lib()

    def run():
        return [conv_forward(w, b, x, **kw), conv_forward(w, b, x, slope_in=0.1, res=res, slope_out=0.2, **kw),
                conv_forward(w, None, x, res=res, **kw), conv_forward(w, None, x, slope_in=0.1, **kw)]
    try:
        _lib.check(L.amp_set_conv_strip(0))
        base = run()
        outs = {}
        _lib.check(L.amp_set_conv_strip(2))
        for steps in (1, 2, 5):
            _lib.check(L.amp_set_conv_strip_steps(steps))
            outs[steps] = run()
    finally:
        _lib.check(L.amp_set_conv_strip(-1))
        _lib.check(L.amp_set_conv_strip_steps(0))
    for steps, o in outs.items():
        for i, (a, c) in enumerate(zip(o, base)):
            assert torch.isfinite(a).all(), (steps, i)
            assert torch.equal(a, c), f"strip kernel, {steps} steps per strip, variant {i}: differs from the per-tile kernels"
    ref = F.conv1d(x, w, b, dilation=d, padding=pad)
    assert (outs[2][0] - ref).abs().max().item() <= 2e-5


def _strip_generator(arch):
    from types import SimpleNamespace as NS
    from oracle import synth
    if arch == "bigvgan":
        from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN
        hp = vo.bigvgan_base_hp()
        m = BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp))))
        m.load_state_dict(synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75))
        return m.cuda().eval(), 100
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN
    hp = vo.hifigan_v1_hp()
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234))
    return m.cuda().eval(), 80


@pytest.mark.parametrize("arch", ["bigvgan", "hifigan"])
def test_generators_with_and_without_the_strip_kernel(arch):
    """Whole generators, dense and ragged: every square conv the strip kernel covers on it (mode 2: also the MRF running-sum convs and the
    ragged tails) == the per-tile kernels, bit for bit.  The whole-block fusions are switched off so that the convs run unfused."""
    from amphion_amd import _lib

    m, n_mel = _strip_generator(arch)
    L = _lib.lib()
    gen = torch.Generator().manual_seed(7)
    mel = torch.randn(3, n_mel, 40, generator=gen)
    lens = torch.tensor([150, 211, 97, 5, 160, 64], dtype=torch.int32)
    melr = torch.randn(6, n_mel, 211, generator=gen)
    for i, l in enumerate(lens):
        melr[i, :, l:] = 0
    outs = {}
    try:
        _lib.check(L.amp_set_ampblock_fusion(0))
        _lib.check(L.amp_set_resblock_fusion(0))
        _lib.check(L.amp_set_pair_strips(0))
        for mode, steps in ((0, 0), (2, 0), (2, 3)):
            _lib.check(L.amp_set_conv_strip(mode))
            _lib.check(L.amp_set_conv_strip_steps(steps))
            with torch.no_grad():
                outs[(mode, steps)] = (m(mel.cuda()).cpu(), m.forward_ragged(melr.cuda(), lens.cuda()).cpu())
    finally:
        _lib.check(L.amp_set_conv_strip(-1))
        _lib.check(L.amp_set_conv_strip_steps(0))
        _lib.check(L.amp_set_ampblock_fusion(-1))
        _lib.check(L.amp_set_resblock_fusion(-1))
        _lib.check(L.amp_set_pair_strips(-1))
    for key in ((2, 0), (2, 3)):
        assert torch.isfinite(outs[key][0]).all()
        assert torch.equal(outs[key][0], outs[(0, 0)][0]), key
        for i, l in enumerate(lens):
            assert torch.equal(outs[key][1][i, :, : l * 256], outs[(0, 0)][1][i, :, : l * 256]), (key, i)
