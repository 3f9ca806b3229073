"""GPU parity: VITS posterior encoder + flow (+ decoder) on the HIP kernels vs golden vectors of the
reference modules and vs the CPU oracle (BASELINE.json config 5).  Tolerance 1e-4 max-abs."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vocoder_oracle as vo

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_precision")]
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(HERE, "golden", "golden_vits.npz"))


def _modules(gin):
    from amphion_amd.models.tts.vits.vits import PosteriorEncoder, ResidualCouplingBlock

    enc = PosteriorEncoder(513, 192, 192, 5, 1, 16, gin_channels=gin)
    flow = ResidualCouplingBlock(192, 192, 5, 1, 4, gin_channels=gin)
    se = synth.synth_state_dict(synth.posterior_encoder_param_shapes(gin_channels=gin), 2468, g_gain=0.5)
    sf = synth.synth_state_dict(synth.coupling_block_param_shapes(gin_channels=gin), 1357, g_gain=0.5)
    enc.load_state_dict(se)
    flow.load_state_dict(sf)
    return enc.cuda().eval(), flow.cuda().eval(), se, sf


@pytest.mark.parametrize("gin", [0, 256])
def test_enc_q_and_flow_golden(gv, gin):
    enc, flow, _, _ = _modules(gin)
    t = f"vits_g{gin}_"
    y = torch.from_numpy(gv[t + "y"]).cuda()
    lens = torch.from_numpy(gv[t + "lens"])
    noise = torch.from_numpy(gv[t + "noise"]).cuda()
    g = torch.from_numpy(gv[t + "g"]).cuda() if gin else None
    with torch.no_grad():
        z, m, logs, mask = enc(y, lens, g=g, noise=noise)
        z_p = flow(z, lens, g=g)
        z_hat = flow(z_p, lens, g=g, reverse=True)
    assert tuple(mask.shape) == (2, 1, 12) and mask[1, 0, 9:].sum() == 0
    for name, val in (("z", z), ("m", m), ("logs", logs), ("z_p", z_p), ("z_hat", z_hat)):
        assert np.abs(val.cpu().numpy() - gv[t + name]).max() <= TOL, name
    # the flow is invertible: reverse(forward(z)) == z on the valid frames
    assert (z_hat - z).abs().max().item() <= 1e-4


def test_config5_end_to_end_vs_oracle():
    """enc_q -> flow -> flow(reverse) -> dec at B=3, T=37 with ragged lengths."""
    from amphion_amd.models.tts.vits.vits import SynthesizerTrnDecodePath

    hp = vo.hifigan_v1_hp()
    net = SynthesizerTrnDecodePath(513, 192, 192, "1", hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"],
                                   hp["upsample_rates"], hp["upsample_initial_channel"], hp["upsample_kernel_sizes"])
    se = synth.synth_state_dict(synth.posterior_encoder_param_shapes(), 2468, g_gain=0.5)
    sf = synth.synth_state_dict(synth.coupling_block_param_shapes(), 1357, g_gain=0.5)
    sdec = synth.synth_state_dict(synth.hifigan_param_shapes(192, hp, vits=True), 4321)
    sd = {**{"enc_q." + k: v for k, v in se.items()}, **{"flow." + k: v for k, v in sf.items()},
          **{"dec." + k: v for k, v in sdec.items()}}
    net.load_state_dict(sd)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(7)
    y = torch.rand(3, 513, 37, generator=gen)
    lens = torch.tensor([37, 20, 5])
    noise = torch.randn(3, 192, 37, generator=gen)
    with torch.no_grad():
        o, mask, (z, z_p, z_hat) = net.reconstruct(y.cuda(), lens, noise=noise.cuda())
        rz, rm, rlogs, rmask = vo.posterior_encoder_forward(se, "", y, lens, noise)
        rzp = vo.coupling_block_forward(sf, "", rz, rmask)
        rzh = vo.coupling_block_forward(sf, "", rzp, rmask, reverse=True)
        ro = vo.hifigan_forward(sdec, hp, rzh * rmask)
    assert (z.cpu() - rz).abs().max().item() <= TOL
    assert (z_p.cpu() - rzp).abs().max().item() <= TOL
    assert (z_hat.cpu() - rzh).abs().max().item() <= TOL
    assert tuple(o.shape) == (3, 1, 37 * 256)
    assert (o.cpu() - ro).abs().max().item() <= TOL


@pytest.mark.parametrize("gin,n_layers,k,rate", [(0, 4, 5, 1), (256, 16, 5, 1), (0, 3, 3, 2), (64, 2, 1, 1)])
def test_wn_fused_vs_unfused_and_oracle(gin, n_layers, k, rate, conv_precision):
    """The fused WN layer (in-conv + gate, res_skip + residual / skip update: amp_wn_forward) against the four unfused
    launches per layer on the same module, and against the oracle's WN (modules/flow/modules.py:126-151) -- ragged
    lengths, with and without the speaker condition."""
    from amphion_amd.modules.flow.modules import WN

    H, B, T = 192, 3, 83
    wn = WN(H, k, rate, n_layers, gin_channels=gin)
    from collections import OrderedDict

    sd_p = synth.synth_state_dict(synth.wn_param_shapes(OrderedDict(), "enc", H, k, n_layers, gin), 99, g_gain=0.5)
    sd = {n[len("enc."):]: v for n, v in sd_p.items()}
    wn.load_state_dict(sd)
    wn = wn.cuda().eval()
    gen = torch.Generator().manual_seed(5)
    lens = torch.tensor([83, 40, 7])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :]
    x = torch.randn(B, H, T, generator=gen) * mask
    g = torch.randn(B, gin, 1, generator=gen) if gin else None
    with torch.no_grad():
        y_f = wn(x.cuda(), lens, g=g.cuda() if gin else None)
        wn.fused = False
        y_u = wn(x.cuda(), lens, g=g.cuda() if gin else None)
        ref = vo.wn_forward(sd_p, "enc", x, mask, n_layers, H, k, rate, torch.float32, g=g)
    assert (y_u.cpu() - ref).abs().max().item() <= TOL
    assert (y_f.cpu() - ref).abs().max().item() <= TOL
    assert (y_f - y_u).abs().max().item() <= 2e-5
