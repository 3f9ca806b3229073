"""GPU parity: whole generators through the nn.Module drop-ins (-> C ABI -> HIP kernels) against the
committed golden vectors of the real reference and against the CPU oracle on seeded inputs.
Tolerance: 1e-4 max-abs on the tanh output (BASELINE.json north_star)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
TOL = 1e-4


def _hifigan(hp, n_mel, seed):
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    cfg = NS(preprocess=NS(n_mel=n_mel, hop_size=256), model=NS(hifigan=NS(**hp)))
    m = HiFiGAN(cfg)
    sd = synth.synth_state_dict(synth.hifigan_param_shapes(n_mel, hp), seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


@pytest.mark.parametrize("tag", ["b1_t8", "b2_t33", "b3_t1"])
def test_hifigan_v1_golden(golden, tag):
    m, _ = _hifigan(vo.hifigan_v1_hp(), 80, 1234)
    mel = torch.from_numpy(golden[f"hifigan_v1_{tag}_mel"]).cuda()
    with torch.no_grad():
        y = m(mel).cpu().numpy()
    ref = golden[f"hifigan_v1_{tag}_wav"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= TOL


def test_hifigan_recipe_resblock2_golden(golden):
    m, _ = _hifigan(vo.hifigan_recipe_hp(), 100, 77)
    with torch.no_grad():
        y = m(torch.from_numpy(golden["hifigan_recipe_b2_t19_mel"]).cuda()).cpu().numpy()
    assert np.abs(y - golden["hifigan_recipe_b2_t19_wav"]).max() <= TOL


@pytest.mark.parametrize("B,T", [(1, 257), (3, 7), (2, 64)])
def test_hifigan_v1_vs_oracle(B, T):
    hp = vo.hifigan_v1_hp()
    m, sd = _hifigan(hp, 80, 1234)
    mel = synth.synth_mel(B, 80, T, seed=B * 100 + T)
    with torch.no_grad():
        y = m(mel.cuda()).cpu()
        ref = vo.hifigan_forward(sd, hp, mel)
        ref64 = vo.hifigan_forward(sd, hp, mel, dtype=torch.float64)
    err = (y - ref).abs().max().item()
    err64 = (y.double() - ref64).abs().max().item()
    base64 = (ref.double() - ref64).abs().max().item()
    print(f"B={B} T={T}: |hip-oracle32|={err:.2e} |hip-oracle64|={err64:.2e} |oracle32-oracle64|={base64:.2e}")
    assert err <= TOL


def test_remove_weight_norm_and_reload_give_same_output():
    hp = vo.hifigan_v1_hp()
    m, sd = _hifigan(hp, 80, 1234)
    mel = synth.synth_mel(1, 80, 12, seed=5).cuda()
    with torch.no_grad():
        y0 = m(mel).clone()
        m.remove_weight_norm()
        y1 = m(mel).clone()
        assert "conv_pre.weight" in m.state_dict()
        folded = {k: v.clone() for k, v in m.state_dict().items()}
        m2, _ = _hifigan(hp, 80, 999)      # different weights ...
        m2.load_state_dict(folded)          # ... replaced by the folded checkpoint
        y2 = m2(mel)
    # the handle folds g*v/||v|| with a double-precision norm, torch's remove_weight_norm in fp32:
    # weights differ by <= 1 ulp, which shows up as a few 1e-6 on the waveform
    assert (y0 - y1).abs().max().item() <= 2e-5
    assert (y0 - y2).abs().max().item() <= 2e-5
    assert torch.equal(y1, y2)


def test_forward_is_deterministic_and_batch_independent():
    hp = vo.hifigan_v1_hp()
    m, _ = _hifigan(hp, 80, 1234)
    mel = synth.synth_mel(4, 80, 20, seed=8).cuda()
    with torch.no_grad():
        a = m(mel).clone()
        b = m(mel).clone()
        c = torch.cat([m(mel[:2]), m(mel[2:])])
    assert torch.equal(a, b)
    assert torch.equal(a, c)


def test_batch_groups_bound_the_workspace_and_keep_the_bits():
    """amp_set_group_mb(n): the batch walks the generator in depth-first groups of about n MB of stage tensors (a bound on the workspace; off by
    default).  Items are independent, so any grouping gives the bits of the one-group forward -- here groups of 1-2 items of a batch of 5."""
    from amphion_amd import _lib

    hp = vo.hifigan_v1_hp()
    m, _ = _hifigan(hp, 80, 1234)
    mel = synth.synth_mel(5, 80, 40, seed=9).cuda()
    L = _lib.lib()
    with torch.no_grad():
        whole = m(mel).clone()
        try:
            _lib.check(L.amp_set_group_mb(2))
            grouped = m(mel).clone()
        finally:
            _lib.check(L.amp_set_group_mb(0))
    assert torch.equal(whole, grouped)
    with pytest.raises(_lib.AmpError):
        _lib.check(L.amp_set_group_mb(-1))


@pytest.mark.parametrize("gin", [0, 256])
def test_hifigan_vits_golden(golden, gin):
    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN_vits

    hp = vo.hifigan_v1_hp()
    m = HiFiGAN_vits(192, "1", hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"], hp["upsample_rates"],
                     hp["upsample_initial_channel"], hp["upsample_kernel_sizes"], gin_channels=gin)
    m.load_state_dict(synth.synth_state_dict(synth.hifigan_param_shapes(192, hp, vits=True, gin_channels=gin), 4321))
    m = m.cuda().eval()
    z = torch.from_numpy(golden[f"hifigan_vits_g{gin}_z"]).cuda()
    g = torch.from_numpy(golden[f"hifigan_vits_g{gin}_g"]).cuda() if gin else None
    with torch.no_grad():
        y = m(z, g=g).cpu().numpy() if gin else m(z).cpu().numpy()
    assert np.abs(y - golden[f"hifigan_vits_g{gin}_wav"]).max() <= TOL


def test_cpu_input_raises():
    m, _ = _hifigan(vo.hifigan_v1_hp(), 80, 1234)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 80, 4))


def test_backward_through_the_generator_raises():
    """A generator in its default training mode, called with autograd on, runs (as the reference module does) -- but a backward pass
    that reaches its output fails loudly instead of leaving the generator silently untrained; under no_grad / eval / with frozen
    parameters the output carries no graph at all."""
    from types import SimpleNamespace as NS

    from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN

    hp = dict(resblock="1", upsample_rates=[2, 2], upsample_kernel_sizes=[4, 4], upsample_initial_channel=32,
              resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3, 5]])
    m = HiFiGAN(NS(preprocess=NS(n_mel=8), model=NS(hifigan=NS(**hp)))).cuda()          # training mode
    x = torch.randn(2, 8, 16, device="cuda")
    y = m(x)
    assert y.requires_grad and torch.isfinite(y).all()
    with pytest.raises(RuntimeError, match="inference-only"):
        y.square().mean().backward()
    y2 = m(x.clone().requires_grad_(True))
    with pytest.raises(RuntimeError, match="inference-only"):
        y2.sum().backward()
    with torch.no_grad():
        assert not m(x).requires_grad
    assert not m.eval()(x).requires_grad
    assert torch.equal(m(x), y.detach())


def test_jets_waveform_decoder_golden():
    """JETS (models/tts/jets/jets.py:454-458,619) decodes with the registry's HiFiGAN built from the recipe config with
    n_mel = attention_dim = 256, called on the up-sampled hidden states: golden vectors of the REAL reference class
    (tests/golden/make_golden_jets.py)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_jets.npz"))
    m, sd = _hifigan(vo.hifigan_recipe_hp(), 256, 2024)
    z = torch.from_numpy(g["z"])
    with torch.no_grad():
        y = m(z.cuda()).cpu().numpy()
    assert y.shape == g["wav"].shape
    assert np.abs(y - g["wav"]).max() <= TOL
    ref = vo.hifigan_forward(sd, vo.hifigan_recipe_hp(), z, dtype=torch.float64).numpy()
    assert np.abs(y - ref).max() <= TOL
