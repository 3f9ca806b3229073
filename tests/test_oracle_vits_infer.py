"""oracle/vits_infer_oracle.py (full VITS inference) against golden vectors of the REAL reference
``SynthesizerTrn.infer`` (tests/golden/make_golden_vits_infer.py).  Weights: oracle/synth.py's seeded tensors over the
reference model's own key / shape list (keys_vits_infer_<tag>.json)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vits_infer_oracle as vio

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_vits_infer.npz"))
SMALL = dict(inter_channels=16, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2, kernel_size=3, resblock="1",
             resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], upsample_rates=[4, 2],
             upsample_initial_channel=32, upsample_kernel_sizes=[8, 4])
VARIANTS = {"sdp": dict(n_speakers=0, use_sdp=True), "sdp_spk": dict(n_speakers=3, use_sdp=True), "dp": dict(n_speakers=0, use_sdp=False)}


def _weights(tag):
    with open(os.path.join(HERE, "golden", f"keys_vits_infer_{tag}.json")) as f:
        shapes = {k: tuple(s) for k, s in json.load(f)}
    return synth.synth_state_dict(shapes, 77, g_gain=0.5)


def _t(tag, key):
    return torch.from_numpy(G[f"{tag}_{key}"])


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_infer_matches_reference(tag):
    sd = _weights(tag)
    hp = dict(SMALL, **VARIANTS[tag])
    with torch.no_grad():
        o = vio.vits_infer(sd, hp, _t(tag, "x"), _t(tag, "x_lengths"), _t(tag, "noise_z"),
                           noise_dp=_t(tag, "noise_dp") if hp["use_sdp"] else None,
                           sid=_t(tag, "sid") if hp["n_speakers"] else None,
                           noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8)
    assert (o["enc_x"] - _t(tag, "enc_x")).abs().max().item() <= 2e-5
    assert (o["logw"] - _t(tag, "logw")).abs().max().item() <= 2e-4           # 4 spline flows deep
    assert torch.equal(o["attn"], _t(tag, "attn"))                            # integer durations: exact
    assert torch.equal(o["mask"], _t(tag, "mask"))
    for k, tol in (("m_p", 2e-5), ("logs_p", 2e-5), ("z_p", 5e-5), ("z", 1e-4), ("y_hat", 1e-4)):
        assert o[k].shape == _t(tag, k).shape, k
        assert (o[k] - _t(tag, k)).abs().max().item() <= tol, k


def test_infer_full_size_matches_reference():
    """The oracle pinned at config/vits.json:28-75 dimensions too (hidden 192, 6 layers, filter 768, full decoder; B = 4,
    T_text up to 100): golden_vits_infer_full.npz = the REAL reference's infer (make_golden_vits_infer.py --full)."""
    F = np.load(os.path.join(HERE, "golden", "golden_vits_infer_full.npz"))
    with open(os.path.join(HERE, "golden", "keys_vits_synthesizer.json")) as f:
        shapes = {k: tuple(s) for k, s in json.load(f)}
    sd = {k: synth.synth_tensor(k, v, int(F["weight_seed"]), 1.0 if k.startswith("dec.") else 0.5) for k, v in shapes.items()}
    hp = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, resblock="1",
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[8, 8, 2, 2],
              upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], n_speakers=0, use_sdp=True)
    x, xl = torch.from_numpy(F["x"]), torch.from_numpy(F["x_lengths"])
    torch.manual_seed(int(F["noise_seed"]))
    n_dp = torch.randn(x.shape[0], 2, x.shape[1])
    n_z = torch.randn(x.shape[0], 192, int(F["y_frames"].max()), generator=torch.Generator().manual_seed(int(F["noise_seed"]) + 1))
    assert abs(float(n_z.double().sum()) - F["noise_z_check"][0]) < 1e-6 and float(n_dp[1, 1, 17]) == F["noise_dp_check"][1]
    with torch.no_grad():
        o = vio.vits_infer(sd, hp, x, xl, n_z, noise_dp=n_dp, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8)
    D = int(F["decim"])
    assert torch.equal(o["attn"].sum(2)[:, 0].to(torch.int32), torch.from_numpy(F["durations"]))
    assert (o["logw"] - torch.from_numpy(F["logw"])).abs().max().item() <= 2e-4
    for k, tol in (("m_p", 2e-5), ("logs_p", 2e-5), ("z_p", 1e-4), ("z", 1e-4)):
        assert (o[k][:, :, ::D] - torch.from_numpy(F[k])).abs().max().item() <= tol, k
    assert (o["y_hat"] - torch.from_numpy(F["y_hat"])).abs().max().item() <= 1e-4


def test_text_encoder_pieces():
    sd = _weights("sdp")
    with torch.no_grad():
        x, m, logs, mask = vio.text_encoder(sd, "enc_p", _t("sdp", "x"), _t("sdp", "x_lengths"), 32, 16, 2, 2, 3)
    assert (x - _t("sdp", "enc_x")).abs().max().item() <= 2e-5
    assert (m - _t("sdp", "enc_m")).abs().max().item() <= 2e-5
    assert (logs - _t("sdp", "enc_logs")).abs().max().item() <= 2e-5
    assert mask.sum().item() == 18 and (x[1, :, 7:] == 0).all()              # lengths 11 and 7: padding stays zero


def test_spline_is_monotone_and_invertible():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 1, 50, generator=g) * 3                                # some values beyond the tail bound 5
    uw, uh = torch.randn(4, 1, 50, 10, generator=g), torch.randn(4, 1, 50, 10, generator=g)
    ud = torch.randn(4, 1, 50, 9, generator=g)
    y = vio.rq_spline(x, uw, uh, ud, inverse=False, tail_bound=5.0)
    back = vio.rq_spline(y, uw, uh, ud, inverse=True, tail_bound=5.0)
    err = (back - x).abs()                                                     # fp32 inverse: flat bins amplify rounding
    assert err.max().item() <= 5e-4 and err.mean().item() <= 1e-5
    out = x.abs() > 5
    assert torch.equal(y[out], x[out])                                        # linear tails: identity outside
    xs, _ = torch.sort(torch.rand(200) * 10 - 5)
    ys = vio.rq_spline(xs, uw[0, 0, 0].expand(200, 10), uh[0, 0, 0].expand(200, 10), ud[0, 0, 0].expand(200, 9), False, 5.0)
    assert (ys[1:] >= ys[:-1]).all()


def test_generate_path_rows_and_columns():
    dur = torch.tensor([[[2.0, 0.0, 3.0, 1.0]], [[1.0, 1.0, 0.0, 0.0]]])
    ty = 6
    y_len = torch.tensor([6, 2])
    x_mask = torch.tensor([[[1.0, 1.0, 1.0, 1.0]], [[1.0, 1.0, 0.0, 0.0]]])
    y_mask = (torch.arange(ty).view(1, 1, ty) < y_len.view(2, 1, 1)).float()
    path = vio.generate_path(dur, x_mask.unsqueeze(2) * y_mask.unsqueeze(-1))
    assert path.shape == (2, 1, 6, 4)
    assert torch.equal(path.sum(dim=2), dur)                                  # token x owns exactly dur[x] frames
    assert torch.equal(path.sum(dim=3), y_mask)                               # every valid frame has one token
    assert path[0, 0, :, 0].tolist() == [1, 1, 0, 0, 0, 0] and path[0, 0, :, 2].tolist() == [0, 0, 1, 1, 1, 0]
