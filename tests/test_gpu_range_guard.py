"""The f16x3 kernels stage activations as hi + lo f16 pairs after an exact x16: |x| up to 4094 is representable, the
fp32 reference (F.conv1d) has no such limit (a NaN input is not this guard's business: it propagates to the output as
it does through the reference).  An activation beyond the range must never give silently wrong audio: every
f16x3 kernel raises a per-device flag (amp_range_check / AMP_ERR_RANGE), a later forward reports it without
synchronising, and ``forward_exact_range`` repeats the call on the exact-fp32 MFMA kernels."""
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


@pytest.fixture(autouse=True)
def _f16x3_and_clean_flag():
    from amphion_amd import _lib

    _lib.set_precision("f16x3")
    try:
        _lib.range_check()
    except _lib.AmpError:
        pass
    yield
    try:
        _lib.range_check()
    except _lib.AmpError:
        pass


@pytest.mark.parametrize("peak,flagged", [(3000.0, False), (4094.0, False), (5e3, True), (1e5, True), (float("inf"), True), (float("-inf"), True)])
def test_conv_flags_operands_beyond_the_f16_range(peak, flagged):
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    cin, cout, k, T = 64, 64, 7, 300
    w = _rand(cout, cin, k, seed=1, scale=(cin * k) ** -0.5)
    b = _rand(cout, seed=2, scale=0.1)
    x = _rand(1, cin, T, seed=3)
    x[0, 5, 100] = peak
    y = conv_forward(w, b, x, padding=3)
    if flagged:
        with pytest.raises(_lib.AmpError) as e:
            _lib.range_check()
        assert e.value.status == _lib.AMP_ERR_RANGE
        _lib.range_check()                       # the check cleared the flag
    else:
        _lib.range_check()
        ref = F.conv1d(x.double(), w.double(), b.double(), padding=3)
        assert (y.double() - ref).abs().max().item() <= 5e-6 * ref.abs().max().item()


def test_pair_seam_is_guarded_too():
    """conv1's output is re-staged inside the fused pair (the seam): a large xt is caught there."""
    from amphion_amd import _lib
    from hip_helpers import pair_forward

    C, k = 64, 3
    w1 = _rand(C, C, k, seed=1, scale=40.0)                  # xt = c1(lrelu(x)) ~ 40 * sqrt(C k) * |x|: thousands
    b = _rand(C, seed=2, scale=0.1)
    w2 = _rand(C, C, k, seed=3, scale=(C * k) ** -0.5)
    x = _rand(1, C, 200, seed=5, scale=30.0)
    pair_forward(w1, b, w2, b, x, dilation=1)
    with pytest.raises(_lib.AmpError) as e:
        _lib.range_check()
    assert e.value.status == _lib.AMP_ERR_RANGE


def _hifigan():
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = dict(vo.hifigan_v1_hp(), upsample_initial_channel=128)
    m = HiFiGAN(NS(preprocess=NS(n_mel=80, hop_size=256), model=NS(hifigan=NS(**hp))))
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd, hp, "hifigan"


def _bigvgan():
    from amphion_amd.models.vocoders.gan.generator.bigvgan import BigVGAN

    hp = dict(vo.bigvgan_base_hp(), upsample_initial_channel=128)
    m = BigVGAN(NS(preprocess=NS(n_mel=100, hop_size=256), model=NS(bigvgan=NS(**hp))))
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd, hp, "bigvgan"


@pytest.mark.parametrize("make", [_hifigan, _bigvgan])
@pytest.mark.parametrize("peak", [5e3, 1e5])
def test_generator_reports_and_heals(make, peak):
    """A mel spike that drives conv_pre's output past the f16 range: check_range() raises, the NEXT forward refuses
    (no synchronisation needed), and forward_exact_range() returns the fp32 reference's audio."""
    from amphion_amd import _lib

    m, sd, hp, kind = make()
    n_mel = 80 if kind == "hifigan" else 100
    mel = synth.synth_mel(1, n_mel, 12, seed=3)
    mel[0, :, 6] = peak                                   # one frame far out of range (x 16 > 65504 after conv_pre)
    fwd = vo.hifigan_forward if kind == "hifigan" else vo.bigvgan_forward
    ref = fwd(sd, hp, mel, dtype=torch.float64)
    ref32 = fwd(sd, hp, mel, dtype=torch.float32)
    with torch.no_grad():
        m(mel.cuda())
        with pytest.raises(_lib.AmpError) as e:
            m.check_range()
        assert e.value.status == _lib.AMP_ERR_RANGE
        m.check_range()                                   # cleared
        # lazy report: run, let the flag copy land, the following forward refuses instead of launching
        m(mel.cuda())
        torch.cuda.synchronize()
        with pytest.raises(_lib.AmpError) as e:
            m(synth.synth_mel(1, n_mel, 12, seed=4).cuda())
        assert e.value.status == _lib.AMP_ERR_RANGE
        ok = m(synth.synth_mel(1, n_mel, 12, seed=4).cuda())   # the report cleared it: normal service again
        assert torch.isfinite(ok).all()
        m.check_range()
        with pytest.warns(RuntimeWarning, match="exact-fp32"):
            y = m.forward_exact_range(mel.cuda())
        assert torch.isfinite(y).all()
        err, base = (y.cpu().double() - ref).abs().max().item(), (ref32.double() - ref).abs().max().item()
        print(f"[range] {kind} peak {peak:g}: |hip f32 - f64| = {err:.2e}, the reference's own fp32 run: {base:.2e}")
        # activations of 1e4..1e6: fp32 itself is no longer 1e-4-exact, and at |alpha u| ~ 1e5 one ulp of u moves the phase by ~0.01 rad --
        # two correct fp32 evaluations differ by a few times the reference's own distance to fp64 (measured 3.1-4.1 x over rounds 2-4)
        assert err <= max(1e-4, 6 * base)
        y2 = m.forward_exact_range(mel.cuda())            # stays on the fp32 kernels, no second warning path
        assert torch.equal(y, y2)


def test_in_range_inputs_never_flag():
    from amphion_amd import _lib

    m, sd, hp, _ = _hifigan()
    with torch.no_grad():
        for seed in range(3):
            m(synth.synth_mel(2, 80, 20, seed=seed).cuda())
    _lib.range_check()


def test_op_level_and_generator_flags_do_not_mix():
    """Op-level launches report to the per-device word (amp_range_check), a generator to its own (check_range / the lazy
    refusal): an out-of-range amp_conv_forward must not make an unrelated generator refuse its next forward."""
    from amphion_amd import _lib
    from hip_helpers import conv_forward

    m, _, _, _ = _hifigan()
    w = _rand(64, 64, 3, seed=1, scale=0.1)
    x = _rand(1, 64, 50, seed=2)
    x[0, 0, 10] = 1e5
    conv_forward(w, None, x, padding=1)                   # raises the device word
    with torch.no_grad():
        for seed in range(3):
            m(synth.synth_mel(1, 80, 12, seed=seed).cuda())
            torch.cuda.synchronize()
    m.check_range()                                       # the generator's own word is clean
    with pytest.raises(_lib.AmpError):
        _lib.range_check()                                # the op-level word still holds the conv's report
