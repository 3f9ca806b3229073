"""GPU parity of the kernels that exist for the f16x3 arithmetic only (run once, under f16x3): the whole-K frame-rate conv
kernel (csrc/conv_small_f16x3.hip) against the pipelined one bit for bit, and conv + Activation1d in one launch
(csrc/conv_f16x3.hip, ACT variant) against the two launches bit for bit."""
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


# ---- a2(c1(.)) of an AMPBlock in one launch (csrc/conv_f16x3.hip, ACT variant) ----------------------------------
FUSED_ACT_CASES = [
    # C, k, dilation, B, T     (launches of >= 384 full-width tiles; T chosen so that the last tile is ragged / exact /
    #                           one sample long, rows 16-B aligned and not; the switch counts tiles of NT columns)
    (128, 7, 3, 4, 12400),     # 128 x 128 tiles (WM = 4), 112 outputs per tile
    (128, 11, 5, 4, 112 * 111 + 1),   # last tile holds ONE sample; 128-column staged halo
    (256, 3, 1, 2, 112 * 111),        # two row groups; T an exact number of tiles
    (64, 7, 1, 4, 24803),      # 64 x 256 tiles (2 x 2 waves), 240 outputs per tile, unaligned rows
    (32, 11, 3, 4, 50000),     # 32 x 512 tiles (1 x 4 waves): a row spans two waves
    (32, 3, 5, 8, 30000),
]


@pytest.mark.parametrize("C,k,d,B,T", FUSED_ACT_CASES)
def test_conv_act_fused_is_bitwise_the_two_launches(C, k, d, B, T):
    """The fused epilogue runs act1d_kernel's operation sequence on the conv's fp32 output tile: the result is the two
    launches' bit for bit -- interior tiles, both utterance ends (replicate padding of the up-sampler's input and of the
    Snake output), large Snake arguments (libm path) -- and within the activation's tolerance of the oracle."""
    from hip_helpers import conv_act_forward

    g = torch.Generator().manual_seed(C + k + T)
    w = torch.randn(C, C, k, generator=g) * (C * k) ** -0.5
    b = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, C, T, generator=g) * 1.5
    al = torch.randn(C, generator=g) * 0.3
    be = torch.randn(C, generator=g) * 0.3
    al[1] = 12.5                                            # exp(12.5) * |u| > 1e5: the libm sine path of that channel
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    y_f = conv_act_forward(w, b, x, al, be, True, f, f, dilation=d, fused=True)
    y_u = conv_act_forward(w, b, x, al, be, True, f, f, dilation=d, fused=False)
    assert torch.isfinite(y_f).all()
    assert torch.equal(y_f, y_u)
    keep = [c for c in range(C) if c != 1]
    ref = vo.activation1d(torch.nn.functional.conv1d(x[:1], w, b, dilation=d, padding=(k * d - d) // 2), al, be, True)
    assert (y_f[:1, keep] - ref[:, keep]).abs().max().item() <= 2e-5
