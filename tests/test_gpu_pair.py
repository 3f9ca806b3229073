"""GPU parity: the fused ResBlock1-pair kernel (pair_f16x3.hip) vs the oracle ops, hifigan.py:93-100."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _ref(w1, b1, w2, b2, x, d, slope):
    k = w1.shape[2]
    xt = F.conv1d(F.leaky_relu(x, slope), w1, b1, dilation=d, padding=(k * d - d) // 2)
    xt = F.conv1d(F.leaky_relu(xt, slope), w2, b2, padding=(k - 1) // 2)
    return xt + x


PAIR_CASES = [
    # C, k, dilation, B, T
    (128, 3, 1, 2, 300),
    (128, 7, 3, 1, 517),
    (128, 11, 5, 2, 86),      # exactly one tile
    (128, 11, 5, 1, 87),      # one tile + 1 column
    (64, 3, 5, 2, 1000),
    (64, 7, 1, 1, 129),
    (64, 11, 3, 3, 401),
    (32, 3, 3, 1, 2050),
    (32, 7, 5, 2, 777),
    (32, 11, 1, 1, 5),        # T smaller than any halo
    (32, 5, 6, 2, 333),       # k5 (recipe net), dilation 6
    (128, 5, 2, 1, 64),
    (64, 11, 5, 1, 1),        # T = 1
]


@pytest.mark.parametrize("C,k,d,B,T", PAIR_CASES)
def test_pair_matches_oracle(C, k, d, B, T):
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    _lib.set_precision("f16x3")
    w1 = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5)
    b1 = _rand(C, seed=2, scale=0.1)
    w2 = _rand(C, C, k, seed=3, scale=(C * k) ** -0.5)
    b2 = _rand(C, seed=4, scale=0.1)
    x = _rand(B, C, T, seed=5)
    ref = _ref(w1.double(), b1.double(), w2.double(), b2.double(), x.double(), d, 0.1)
    ref32 = _ref(w1, b1, w2, b2, x, d, 0.1)
    y = pair_forward(w1, b1, w2, b2, x, dilation=d)
    assert y.shape == ref.shape
    err = (y.double() - ref).abs().max().item()
    base = (ref32.double() - ref).abs().max().item()
    print(f"C={C} k={k} d={d}: |hip-f64|={err:.2e}  |torch32-f64|={base:.2e}")
    assert err <= 5e-6


def test_pair_unsupported_shapes_are_reported():
    from amphion_amd import _lib
    from amphion_amd._lib import AmpError
    from hip_helpers import pair_forward

    _lib.set_precision("f16x3")
    C, k = 48, 3   # channels not covered
    w = _rand(C, C, k, seed=1, scale=0.1)
    b = _rand(C, seed=2)
    with pytest.raises(AmpError):
        pair_forward(w, b, w, b, _rand(1, C, 40), dilation=1)
    C, k = 32, 11  # receptive field too wide for the staged tile: (k-1)*d = 100 > 64
    w = _rand(C, C, k, seed=1, scale=0.1)
    b = _rand(C, seed=2)
    with pytest.raises(AmpError):
        pair_forward(w, b, w, b, _rand(1, C, 400), dilation=10)
    _lib.set_precision("f32")
    try:
        C, k = 32, 3
        w = _rand(C, C, k, seed=1, scale=0.1)
        b = _rand(C, seed=2)
        with pytest.raises(AmpError):
            pair_forward(w, b, w, b, _rand(1, C, 40), dilation=1)
    finally:
        _lib.set_precision("f16x3")
