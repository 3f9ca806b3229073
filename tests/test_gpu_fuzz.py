"""GPU parity, randomized: seeded random conv / transposed-conv / fused-pair shapes and random small generator
architectures against the fp64 oracle ops.  Complements the hand-picked cases of test_gpu_conv / _pair /
_generator (odd channel counts, ragged tiles, every tap count and both kernel families)."""
import os
import random
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
# AMP_FUZZ_OFFSET=<n>: the same tests on other seeds (a soak run: `for o in 100 200 300; do AMP_FUZZ_OFFSET=$o pytest tests/test_gpu_fuzz.py; done`)
_OFF = int(os.environ.get("AMP_FUZZ_OFFSET", "0"))


def _rand(*shape, gen, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 12))
def test_random_conv1d(seed):
    from hip_helpers import conv_forward

    rng = random.Random(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    cin, cout = rng.choice([1, 3, 17, 32, 48, 80, 130, 200]), rng.choice([2, 31, 32, 64, 96, 129, 256])
    k = rng.choice([1, 2, 3, 4, 5, 7, 9, 11])
    d = rng.choice([1, 2, 3, 5, 9]) if k > 1 else 1
    if (k - 1) * d > 120:
        d = 1
    B, T = rng.choice([1, 2, 3]), rng.choice([1, 2, 31, 100, 257, 1000])
    w = _rand(cout, cin, k, gen=g, scale=(cin * k) ** -0.5)
    b = _rand(cout, gen=g, scale=0.1) if rng.random() < 0.8 else None
    x = _rand(B, cin, T, gen=g)
    pad = rng.choice([0, (k * d - d) // 2, (k - 1) * d]) if T + 0 > (k - 1) * d else (k - 1) * d
    slope_in = rng.choice([1.0, 0.1])
    slope_out = rng.choice([1.0, 0.2])
    res = None
    ref = F.conv1d(F.leaky_relu(x.double(), slope_in), w.double(), None if b is None else b.double(), dilation=d, padding=pad)
    if ref.shape[-1] <= 0:
        pytest.skip("empty output")
    if rng.random() < 0.5:
        res = _rand(*ref.shape, gen=g)
        ref = ref + res.double()
    ref = F.leaky_relu(ref, slope_out)
    y = conv_forward(w, b, x, dilation=d, padding=pad, slope_in=slope_in, res=res, slope_out=slope_out)
    assert y.shape == ref.shape
    assert (y.double() - ref).abs().max().item() <= 2e-5, (cin, cout, k, d, B, T, pad)   # fp32 chains up to K = 2200


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_conv_transpose1d(seed):
    from hip_helpers import conv_forward

    rng = random.Random(2000 + seed)
    g = torch.Generator().manual_seed(seed)
    cin, cout = rng.choice([8, 24, 64, 128, 200]), rng.choice([4, 16, 40, 64])
    u = rng.choice([2, 3, 4, 5, 8])
    k = u + 2 * rng.choice([0, 1, 2, u // 2 + 1]) if u % 2 == 0 else u + 2 * rng.choice([0, 1, 2])
    if (k + u - 1) // u > 11:
        k = 2 * u
    B, T = rng.choice([1, 2]), rng.choice([1, 5, 64, 300])
    w = _rand(cin, cout, k, gen=g, scale=(cin * k / u) ** -0.5)
    b = _rand(cout, gen=g, scale=0.1)
    x = _rand(B, cin, T, gen=g)
    pad = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=pad)
    y = conv_forward(w, b, x, transposed=True, stride=u, padding=pad, slope_in=0.1)
    assert y.shape == ref.shape
    assert (y.double() - ref).abs().max().item() <= 2e-5, (cin, cout, k, u, B, T)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 6))
def test_random_generator_architecture(seed):
    """Random small HiFi-GAN configs (rates, kernel sizes, resblock type, dilations) vs the fp64 oracle."""
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    rng = random.Random(3000 + seed)
    n_stages = rng.choice([1, 2, 3])
    rates = [rng.choice([2, 4, 8]) for _ in range(n_stages)]
    hp = dict(resblock=rng.choice(["1", "2"]), upsample_rates=rates, upsample_kernel_sizes=[2 * r for r in rates],
              upsample_initial_channel=rng.choice([64, 128, 256]),
              resblock_kernel_sizes=rng.sample([3, 5, 7, 11], rng.choice([1, 2, 3])), resblock_dilation_sizes=None)
    nd = 3 if hp["resblock"] == "1" else 2
    hp["resblock_dilation_sizes"] = [sorted(rng.sample([1, 2, 3, 5], nd)) for _ in hp["resblock_kernel_sizes"]]
    n_mel = rng.choice([20, 80])
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(n_mel, hp), 500 + seed, g_gain=0.9)
    m = HiFiGAN(NS(preprocess=NS(n_mel=n_mel), model=NS(hifigan=NS(**hp))))
    m.load_state_dict(sd)
    m = m.cuda().eval()
    mel = synth.synth_mel(rng.choice([1, 2]), n_mel, rng.choice([3, 17, 50]), seed=seed)
    with torch.no_grad():
        y = m(mel.cuda()).cpu()
        ref = vo.hifigan_forward(sd, hp, mel, dtype=torch.float64)
    err = (y.double() - ref).abs().max().item()
    print(hp, f"err {err:.2e}")
    assert err <= 1e-4
