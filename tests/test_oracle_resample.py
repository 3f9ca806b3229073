"""oracle/vocoder_oracle.py resampling + Snake restatements against golden vectors made by the REAL reference
classes (tests/golden/make_golden_resample.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as vo

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_resample.npz"))
RK = [(2, None), (3, None), (4, 16), (2, 7)]


def _t(key):
    return torch.from_numpy(G[key])


@pytest.mark.parametrize("ratio,ks", RK)
def test_up_down_match_reference(ratio, ks):
    tag = f"r{ratio}k{ks or 0}"
    x = _t("x")
    n = ks or int(6 * ratio // 2) * 2
    f = vo.kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, n)
    assert np.abs(f.reshape(-1).numpy() - G[f"up_{tag}_filter"]).max() <= 1e-7
    assert np.abs(f.reshape(-1).numpy() - G[f"down_{tag}_filter"]).max() <= 1e-7
    up = vo.upsample1d(x, ratio, ks, filt=_t(f"up_{tag}_filter"))
    assert up.shape == G[f"up_{tag}_y"].shape and np.abs(up.numpy() - G[f"up_{tag}_y"]).max() <= 1e-6
    dn = vo.downsample1d(x, ratio, ks, filt=_t(f"down_{tag}_filter"))
    assert dn.shape == G[f"down_{tag}_y"].shape and np.abs(dn.numpy() - G[f"down_{tag}_y"]).max() <= 1e-6


def test_lowpass_variants_and_single_sample():
    x = _t("x")
    for tag, kw in {"lp_k12": {}, "lp_k9_s2_nopad": dict(stride=2, padding=False), "lp_k8_reflect": dict(padding_mode="reflect"),
                    "lp_k5_zeros": dict(padding_mode="constant")}.items():
        y = vo.lowpass1d(x, _t(f"{tag}_filter"), **kw)
        assert y.shape == G[f"{tag}_y"].shape and np.abs(y.numpy() - G[f"{tag}_y"]).max() <= 1e-6, tag
    assert np.abs(vo.upsample1d(_t("x1")).numpy() - G["up_T1_y"]).max() <= 1e-6
    assert np.abs(vo.downsample1d(_t("x1")).numpy() - G["down_T1_y"]).max() <= 1e-6


def test_snake_and_ratio3_activation():
    x = _t("x")
    for tag, beta, log in [("snake_lin", False, False), ("snake_log", False, True), ("snakebeta_lin", True, False), ("snakebeta_log", True, True)]:
        y = vo.snake(x, _t(f"{tag}_alpha"), _t(f"{tag}_beta") if beta else None, log)
        assert np.abs(y.numpy() - G[f"{tag}_y"]).max() <= 1e-6, tag
    y = vo.downsample1d(vo.snake(vo.upsample1d(x, 3, 18), _t("act_r3_alpha"), _t("act_r3_beta"), True), 3, 18)
    assert np.abs(y.numpy() - G["act_r3_y"]).max() <= 2e-6
