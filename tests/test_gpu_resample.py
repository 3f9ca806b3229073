"""Stand-alone UpSample1d / DownSample1d / LowPassFilter1d / Snake / SnakeBeta forwards and the general (non-fused)
Activation1d on the HIP kernels, against the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_resample.npz"))
TOL = 5e-6


def _t(key):
    return torch.from_numpy(G[key])


@pytest.mark.parametrize("ratio,ks", [(2, None), (3, None), (4, 16), (2, 7)])
def test_up_down_golden(ratio, ks):
    from amphion_amd.modules.anti_aliasing import DownSample1d, UpSample1d

    tag = f"r{ratio}k{ks or 0}"
    x = _t("x").cuda()
    up, dn = UpSample1d(ratio, ks).cuda(), DownSample1d(ratio, ks).cuda()
    assert np.abs(up.filter.reshape(-1).cpu().numpy() - G[f"up_{tag}_filter"]).max() <= 1e-7
    y = up(x).cpu().numpy()
    assert y.shape == G[f"up_{tag}_y"].shape and np.abs(y - G[f"up_{tag}_y"]).max() <= TOL
    y = dn(x).cpu().numpy()
    assert y.shape == G[f"down_{tag}_y"].shape and np.abs(y - G[f"down_{tag}_y"]).max() <= TOL


def test_lowpass_variants_golden():
    from amphion_amd.modules.anti_aliasing import DownSample1d, LowPassFilter1d, UpSample1d

    x = _t("x").cuda()
    for tag, kw in {"lp_k12": dict(cutoff=0.25, half_width=0.3, kernel_size=12),
                    "lp_k9_s2_nopad": dict(cutoff=0.2, half_width=0.3, stride=2, padding=False, kernel_size=9),
                    "lp_k8_reflect": dict(cutoff=0.3, half_width=0.4, padding_mode="reflect", kernel_size=8),
                    "lp_k5_zeros": dict(cutoff=0.4, half_width=0.5, padding_mode="constant", kernel_size=5)}.items():
        y = LowPassFilter1d(**kw).cuda()(x).cpu().numpy()
        assert y.shape == G[f"{tag}_y"].shape and np.abs(y - G[f"{tag}_y"]).max() <= TOL, tag
    x1 = _t("x1").cuda()
    assert np.abs(UpSample1d(2).cuda()(x1).cpu().numpy() - G["up_T1_y"]).max() <= TOL
    assert np.abs(DownSample1d(2).cuda()(x1).cpu().numpy() - G["down_T1_y"]).max() <= TOL


@pytest.mark.parametrize("B,C,T,ratio,ks", [(1, 1, 1, 2, None), (2, 5, 1000, 2, None), (1, 3, 4097, 3, None), (3, 2, 300, 4, 16),
                                            (1, 2, 777, 5, 40), (1, 1, 50, 8, 64)])
def test_up_down_vs_oracle(B, C, T, ratio, ks):
    from amphion_amd.modules.anti_aliasing import DownSample1d, UpSample1d

    x = torch.randn(B, C, T, generator=torch.Generator().manual_seed(T + ratio)) * 1.5
    up, dn = UpSample1d(ratio, ks).cuda(), DownSample1d(ratio, ks).cuda()
    ref = vo.upsample1d(x, ratio, ks)
    y = up(x.cuda()).cpu()
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= TOL
    if T + dn.lowpass.pad_left + dn.lowpass.pad_right >= dn.kernel_size:
        ref = vo.downsample1d(x, ratio, ks)
        y = dn(x.cuda()).cpu()
        assert y.shape == ref.shape and (y - ref).abs().max().item() <= TOL
    # size-independent property: the low-pass filter sums to one, so decimating a constant returns the constant
    # (the polyphase branches of the up-sampler only sum to 1/ratio within ~1e-3, so no such identity there)
    c = torch.full((1, 1, max(T, 2 * (ks or 6 * ratio))), 0.75).cuda()
    assert (dn(c) - 0.75).abs().max().item() <= 2e-6


def test_snake_forward_golden_and_oracle():
    from amphion_amd.modules.activation_functions import Snake, SnakeBeta

    x = _t("x").cuda()
    for tag, cls, log in [("snake_lin", Snake, False), ("snake_log", Snake, True), ("snakebeta_lin", SnakeBeta, False),
                          ("snakebeta_log", SnakeBeta, True)]:
        act = cls(3, alpha_logscale=log)
        act.alpha.data = _t(f"{tag}_alpha")
        if cls is SnakeBeta:
            act.beta.data = _t(f"{tag}_beta")
        y = act.cuda()(x).cpu().numpy()
        assert np.abs(y - G[f"{tag}_y"]).max() <= TOL, tag
    g = torch.Generator().manual_seed(4)
    xb = torch.randn(2, 7, 3001, generator=g) * 2
    act = SnakeBeta(7, alpha_logscale=True)
    act.alpha.data = torch.randn(7, generator=g) * 0.3
    act.beta.data = torch.randn(7, generator=g) * 0.3
    ref = vo.snake(xb, act.alpha.data, act.beta.data, True)
    assert (act.cuda()(xb.cuda()).cpu() - ref).abs().max().item() <= TOL
    with pytest.raises(RuntimeError):
        act(xb)                                               # parameters on the GPU, input on the CPU: no fallback
    with pytest.raises(ValueError):
        act(torch.zeros(1, 6, 8).cuda())


def test_activation1d_general_ratio():
    from amphion_amd.modules.activation_functions import SnakeBeta
    from amphion_amd.modules.anti_aliasing import Activation1d

    a3 = Activation1d(SnakeBeta(3, alpha_logscale=True), up_ratio=3, down_ratio=3, up_kernel_size=18, down_kernel_size=18)
    a3.act.alpha.data = _t("act_r3_alpha")
    a3.act.beta.data = _t("act_r3_beta")
    y = a3.cuda()(_t("x").cuda()).cpu().numpy()
    assert y.shape == G["act_r3_y"].shape and np.abs(y - G["act_r3_y"]).max() <= TOL
    # the fused ratio-2 kernel and the three stand-alone ops agree
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 2500, generator=g).cuda()
    a2 = Activation1d(SnakeBeta(3, alpha_logscale=True)).cuda()
    a2.act.alpha.data = (torch.randn(3, generator=g) * 0.3).cuda()
    a2.act.beta.data = (torch.randn(3, generator=g) * 0.3).cuda()
    fused = a2(x)
    unfused = a2.downsample(a2.act(a2.upsample(x)))
    assert (fused - unfused).abs().max().item() <= TOL


def test_bad_arguments():
    from amphion_amd import _lib
    from amphion_amd.modules.anti_aliasing import LowPassFilter1d, UpSample1d

    x = torch.zeros(1, 1, 16).cuda()
    with pytest.raises(_lib.AmpError):
        UpSample1d(2, 66).cuda()(x)                           # > AMP_FIR_MAX_TAPS
    with pytest.raises(_lib.AmpError):
        LowPassFilter1d(0.25, 0.3, padding_mode="reflect", kernel_size=40).cuda()(x)   # reflect pad >= T
    with pytest.raises(NotImplementedError):
        LowPassFilter1d(0.25, 0.3, padding_mode="circular").cuda()(x)
    with pytest.raises(RuntimeError):
        UpSample1d(2)(torch.zeros(1, 1, 16))                  # CPU tensor
