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


