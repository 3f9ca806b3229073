"""GPU parity: the fused ResBlock1-pair kernels (strip-mined pair_strip_f16x3.hip, the default, and the per-tile
pair_f16x3.hip) vs the oracle ops, hifigan.py:93-100, and against each other / the unfused conv sequence bit for bit."""
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
# shapes that make a workgroup walk several steps of a strip (many more columns than 512 workgroups x one step)
STRIP_CASES = [
    # C, k, dilation, B, T
    (128, 11, 5, 40, 2000),
    (128, 3, 1, 64, 1100),
    (64, 7, 3, 70, 1500),
    (32, 11, 1, 33, 5000),
    (128, 7, 5, 1, 70000),    # one long utterance
]


# launches of 512+ strips at C = 128, k in {7, 11}: the policy's A-ring strips (pair_strip_f16x3.hip, wide = 3: 64 x 128-column wave
# tiles, one 256-column step per strip, four where 1 024+ such strips remain); smaller launches of the same shapes fall back to the per-tile kernel
RING_CASES = [
    # C, k, dilation, B, T
    (128, 11, 5, 64, 2100),
    (128, 7, 3, 64, 4100),
    (128, 11, 1, 48, 6000),
    # 1 024+ four-step strips (round 5: generator.hip strip_geometry): the headline's stage shape, and one whose last strip is a ragged tail
    (128, 11, 3, 64, 16384),
    (128, 7, 1, 70, 15000),
]


def _pair_inputs(C, k, B, T):
    w1 = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5)
    b1 = _rand(C, seed=2, scale=0.1)
    w2 = _rand(C, C, k, seed=3, scale=(C * k) ** -0.5)
    b2 = _rand(C, seed=4, scale=0.1)
    return w1, b1, w2, b2, _rand(B, C, T, seed=5)


@pytest.fixture
def strips():
    """amp_set_pair_strips for one call sequence: True = the launch policy (A-ring strips at C = 128, k >= 7 when the launch fills the chip),
    False = the per-tile kernel everywhere."""
    from amphion_amd import _lib

    def use(on):
        _lib.check(_lib.lib().amp_set_pair_strips(-1 if on else 0))
    yield use
    _lib.check(_lib.lib().amp_set_pair_strips(-1))


@pytest.mark.parametrize("C,k,d,B,T", [c for c in PAIR_CASES + STRIP_CASES if c[0] in (64, 128)] + RING_CASES)
def test_policy_kernel_equals_tile_kernel_bitwise(C, k, d, B, T, strips):
    """Whatever kernel the per-shape policy picks (amp_set_pair_strips(-1): per-tile kernel, or the A-ring strips at C = 128,
    k in {7, 11} when the launch fills the chip) gives the bits of the per-tile kernel."""
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    _lib.set_precision("f16x3")
    w1, b1, w2, b2, x = _pair_inputs(C, k, B, T)
    y_tile = pair_forward(w1, b1, w2, b2, x, dilation=d)            # strips(False) state of the fixture = per-tile
    strips(False)
    y_tile = pair_forward(w1, b1, w2, b2, x, dilation=d)
    _lib.check(_lib.lib().amp_set_pair_strips(-1))
    y_pol = pair_forward(w1, b1, w2, b2, x, dilation=d)
    assert not torch.isnan(y_pol).any()
    assert torch.equal(y_pol, y_tile)


@pytest.mark.parametrize("seed", range(6))
def test_four_step_strips_random_shapes_bitwise(seed, strips):
    """Seeded random launches large enough for the four-step strips (B x ceil(T / (4 * 256 - (k - 1))) >= 1 024; AMP_FUZZ_OFFSET moves the
    seeds): the policy's result == the per-tile kernel's, bit for bit, including strips that end in a ragged tail and T that is no
    multiple of anything."""
    import os
    import random
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    rng = random.Random(4242 + seed + int(os.environ.get("AMP_FUZZ_OFFSET", "0")))
    k = rng.choice([7, 11])
    d = rng.choice([1, 3, 5])
    len4 = 4 * 256 - (k - 1)
    spi = rng.randint(9, 20)
    T = (spi - 1) * len4 + rng.randint(1, len4)
    B = (1024 + spi - 1) // spi + rng.randint(0, 6)
    assert B * ((T + len4 - 1) // len4) >= 1024
    _lib.set_precision("f16x3")
    w1, b1, w2, b2, x = _pair_inputs(128, k, B, T)
    strips(False)
    y_tile = pair_forward(w1, b1, w2, b2, x, dilation=d)
    _lib.check(_lib.lib().amp_set_pair_strips(-1))
    y_pol = pair_forward(w1, b1, w2, b2, x, dilation=d)
    assert not torch.isnan(y_pol).any()
    assert torch.equal(y_pol, y_tile), (k, d, B, T)


@pytest.mark.parametrize("C,k,d,B,T", STRIP_CASES + RING_CASES)
def test_strip_partition_invariance_and_oracle(C, k, d, B, T, strips):
    """The launch plan depends on (B, T): a batch that fills the chip walks the A-ring strips (C = 128, k >= 7), a single item the per-tile
    kernel.  Every item of the batch must equal that item run alone bit for bit, and the probed items must match the fp64 oracle."""
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    _lib.set_precision("f16x3")
    strips(True)
    w1, b1, w2, b2, x = _pair_inputs(C, k, B, T)
    y = pair_forward(w1, b1, w2, b2, x, dilation=d)
    assert not torch.isnan(y).any()
    for b in sorted({0, B // 2, B - 1}):
        y1 = pair_forward(w1, b1, w2, b2, x[b:b + 1], dilation=d)
        assert torch.equal(y[b:b + 1], y1), f"item {b}"
        Tp = min(T, 6000)   # oracle on a prefix: exact for the first Tp - receptive field columns
        ref = _ref(w1.double(), b1.double(), w2.double(), b2.double(), x[b:b + 1, :, :Tp].double(), d, 0.1)
        keep = Tp if Tp == T else Tp - (k - 1) * (d + 1)
        assert (y1[..., :keep].double() - ref[..., :keep]).abs().max().item() <= 5e-6


@pytest.mark.parametrize("C,k,d,B,T", PAIR_CASES)
def test_pair_matches_oracle(C, k, d, B, T, strips):
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    _lib.set_precision("f16x3")
    strips(False)
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
