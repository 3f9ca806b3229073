"""Full VITS inference through the SynthesizerTrn drop-in against the golden vectors of the REAL
reference's ``SynthesizerTrn.infer`` and, op by op, against oracle/vits_infer_oracle.py (itself pinned on those
vectors by tests/test_oracle_vits_infer.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vits_infer_oracle as vio

pytestmark = pytest.mark.gpu        # first run on MI355X in round 2 (profiles/r2_a_first_runs.txt): 5 / 5 green
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_vits_infer.npz"))
SMALL = dict(inter_channels=16, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2, kernel_size=3, p_dropout=0.1,
             resblock="1", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], upsample_rates=[4, 2],
             upsample_initial_channel=32, upsample_kernel_sizes=[8, 4])
VARIANTS = {"sdp": dict(n_speakers=0, gin_channels=0, use_sdp=True), "sdp_spk": dict(n_speakers=3, gin_channels=8, use_sdp=True),
            "dp": dict(n_speakers=0, gin_channels=0, use_sdp=False)}


def _weights(tag):
    with open(os.path.join(HERE, "golden", f"keys_vits_infer_{tag}.json")) as f:
        shapes = {k: tuple(s) for k, s in json.load(f)}
    return synth.synth_state_dict(shapes, 77, g_gain=0.5)


def _t(tag, key):
    return torch.from_numpy(G[f"{tag}_{key}"])


def _net(tag):
    from amphion_amd.models.tts.vits.vits import SynthesizerTrn

    net = SynthesizerTrn(40, 33, 8, **SMALL, **VARIANTS[tag])
    net.load_state_dict(_weights(tag))
    return net.cuda().eval()


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_infer_matches_reference(tag):
    net = _net(tag)
    kw = VARIANTS[tag]
    with torch.no_grad():
        o = net.infer(_t(tag, "x").cuda(), _t(tag, "x_lengths"), sid=_t(tag, "sid") if kw["n_speakers"] else None,
                      noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8,
                      noise_dp=_t(tag, "noise_dp").cuda() if kw["use_sdp"] else None, noise_z=_t(tag, "noise_z").cuda())
    assert torch.equal(o["attn"].cpu(), _t(tag, "attn"))                      # integer durations -> the path is exact
    assert torch.equal(o["mask"].cpu(), _t(tag, "mask"))
    valid = _t(tag, "mask").bool()
    for k, tol in (("m_p", 5e-5), ("logs_p", 5e-5), ("z_p", 1e-4), ("z", 1e-4)):
        got, want = o[k].cpu(), _t(tag, k)
        assert got.shape == want.shape, k
        assert ((got - want) * valid).abs().max().item() <= tol, k            # padded frames: see gauss_sample / flow masks
    assert (o["y_hat"].cpu() - _t(tag, "y_hat")).abs().max().item() <= 1e-4


# ---- config/vits.json dimensions (hidden 192, 6 layers, filter 768, SDP, full decoder), B = 4, T_text up to 100 ----------
FULL = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1,
            resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[8, 8, 2, 2],
            upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], n_speakers=0, gin_channels=256, use_sdp=True)


def test_infer_full_size_matches_reference():
    """the REAL reference ``SynthesizerTrn.infer`` (vits.py:320-369) at config/vits.json:28-75 dimensions
    (tests/golden/make_golden_vits_infer.py --full): durations and therefore the alignment path exact, m_p / logs_p /
    z_p / z (every 4th frame stored) and the waveform within 1e-4."""
    from amphion_amd.models.tts.vits.vits import SynthesizerTrn

    F = np.load(os.path.join(HERE, "golden", "golden_vits_infer_full.npz"))
    with open(os.path.join(HERE, "golden", "keys_vits_synthesizer.json")) as f:
        shapes = {k: tuple(s) for k, s in json.load(f)}
    sd = {k: synth.synth_tensor(k, v, int(F["weight_seed"]), 1.0 if k.startswith("dec.") else 0.5) for k, v in shapes.items()}
    net = SynthesizerTrn(512, 513, 32, **FULL)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x, xl = torch.from_numpy(F["x"]), torch.from_numpy(F["x_lengths"])
    ty = int(F["y_frames"].max())
    torch.manual_seed(int(F["noise_seed"]))                     # the reference's two draws, replayed (checksums below)
    n_dp = torch.randn(x.shape[0], 2, x.shape[1])
    n_z = torch.randn(x.shape[0], FULL["inter_channels"], ty, generator=torch.Generator().manual_seed(int(F["noise_seed"]) + 1))
    assert abs(float(n_dp.double().sum()) - F["noise_dp_check"][0]) < 1e-6 and float(n_dp[1, 1, 17]) == F["noise_dp_check"][1]
    assert abs(float(n_z.double().sum()) - F["noise_z_check"][0]) < 1e-6 and float(n_z[2, 100, 5]) == F["noise_z_check"][1]
    with torch.no_grad():
        o = net.infer(x.cuda(), xl, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8, noise_dp=n_dp.cuda(), noise_z=n_z.cuda())
        h, m, logs, lens = net.enc_p(x.cuda(), xl)
        logw = net.dp(h, lens, g=None, reverse=True, noise_scale=0.8, noise=n_dp.cuda())
    D = int(F["decim"])
    tmask = (torch.arange(x.shape[1]).view(1, 1, -1) < xl.view(-1, 1, 1))
    e_m = ((m.cpu()[:, :, ::D] - torch.from_numpy(F["enc_m"])) * tmask[:, :, ::D]).abs().max().item()
    e_logs = ((logs.cpu()[:, :, ::D] - torch.from_numpy(F["enc_logs"])) * tmask[:, :, ::D]).abs().max().item()
    e_logw = ((logw.cpu() - torch.from_numpy(F["logw"])) * tmask).abs().max().item()
    print(f"\n[vits full] text encoder m {e_m:.2e} logs {e_logs:.2e}, logw {e_logw:.2e}")
    assert e_m <= 1e-4 and e_logs <= 1e-4 and e_logw <= 5e-4
    dur = o["attn"].sum(2)[:, 0].to(torch.int32).cpu()
    assert torch.equal(dur, torch.from_numpy(F["durations"]))                       # integer durations -> the path is exact
    assert torch.equal(o["mask"].sum(dim=(1, 2)).cpu(), torch.from_numpy(F["y_frames"]))
    valid = o["mask"].bool().cpu()[:, :, ::D]
    errs = {}
    for k in ("m_p", "logs_p", "z_p", "z"):
        got, want = o[k].cpu()[:, :, ::D], torch.from_numpy(F[k])
        assert got.shape == want.shape, k
        errs[k] = ((got - want) * valid).abs().max().item()
    errs["y_hat"] = (o["y_hat"].cpu() - torch.from_numpy(F["y_hat"])).abs().max().item()
    print("[vits full] " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()) + f"; frames {F['y_frames'].tolist()}")
    for k, v in errs.items():
        assert v <= 1e-4, (k, v)


def test_text_encoder_and_duration_predictors_vs_oracle():
    for tag in VARIANTS:
        net, sd = _net(tag), _weights(tag)
        x, xl = _t(tag, "x"), _t(tag, "x_lengths")
        with torch.no_grad():
            h, m, logs, lens = net.enc_p(x.cuda(), xl)
            rx, rm, rlogs, rmask = vio.text_encoder(sd, "enc_p", x, xl, 32, 16, 2, 2, 3)
            assert (h.cpu() - rx).abs().max().item() <= 5e-5
            assert (m.cpu() - rm).abs().max().item() <= 5e-5 and (logs.cpu() - rlogs).abs().max().item() <= 5e-5
            g = None
            if VARIANTS[tag]["n_speakers"]:
                g = net.emb_g(_t(tag, "sid").cuda().squeeze(-1)).unsqueeze(-1)
            if VARIANTS[tag]["use_sdp"]:
                logw = net.dp(h, lens, g=g, reverse=True, noise_scale=0.8, noise=_t(tag, "noise_dp").cuda())
            else:
                logw = net.dp(h, lens, g=g)
        assert (logw.cpu() - _t(tag, "logw")).abs().max().item() <= 5e-4     # 3 spline flows deep


def test_ops_vs_oracle():
    from amphion_amd.modules import hip_ops

    g = torch.Generator().manual_seed(5)
    B, C, T = 3, 24, 50
    lens = torch.tensor([50, 31, 1])
    lens_d = lens.to(torch.int32).cuda()
    mask = (torch.arange(T).view(1, 1, T) < lens.view(B, 1, 1)).float()
    x, r = torch.randn(B, C, T, generator=g), torch.randn(B, C, T, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    # LayerNorm(x + res), gelu, + post
    ref = x + torch.nn.functional.gelu(vio.layer_norm_channels(x + r, gamma, beta))
    got = hip_ops.layer_norm_c(x.cuda(), gamma.cuda(), beta.cuda(), res=r.cuda(), post=x.cuda(), gelu=True).cpu()
    assert (got - ref).abs().max().item() <= 1e-5
    # ... on shapes off the kernel's 32-column x 8-channel-group tiling: channel counts that are no multiple of 8, more than the 256
    # channels a thread keeps in registers, column counts around 32; plain (no residual, no GELU, no post)
    for Cn, Tn in ((7, 1), (192, 33), (300, 70), (513, 31)):
        xs, rs = torch.randn(2, Cn, Tn, generator=g), torch.randn(2, Cn, Tn, generator=g)
        gm, bt = 1 + 0.2 * torch.randn(Cn, generator=g), 0.1 * torch.randn(Cn, generator=g)
        got = hip_ops.layer_norm_c(xs.cuda(), gm.cuda(), bt.cuda()).cpu()
        assert (got - vio.layer_norm_channels(xs, gm, bt)).abs().max().item() <= 2e-5, (Cn, Tn)
        got = hip_ops.layer_norm_c(xs.cuda(), gm.cuda(), bt.cuda(), res=rs.cuda()).cpu()
        assert (got - vio.layer_norm_channels(xs + rs, gm, bt)).abs().max().item() <= 2e-5, (Cn, Tn, "res")
    # depthwise dilated conv of x * mask
    for K, d in ((3, 1), (3, 3), (3, 9), (5, 2)):
        w, b = torch.randn(C, 1, K, generator=g), torch.randn(C, generator=g)
        ref = torch.nn.functional.conv1d(x * mask, w, b, padding=(K * d - d) // 2, dilation=d, groups=C)
        got = hip_ops.dwconv(x.cuda(), w.cuda(), b.cuda(), lens_d, d).cpu()
        assert (got - ref).abs().max().item() <= 1e-5, (K, d)
    # relative attention (2 heads, window 4), incl. a sequence shorter than the window
    sd = {"a.conv_q.weight": torch.eye(C).unsqueeze(-1), "a.conv_q.bias": torch.zeros(C)}
    for n in ("k", "v", "o"):
        sd[f"a.conv_{n}.weight"], sd[f"a.conv_{n}.bias"] = torch.eye(C).unsqueeze(-1), torch.zeros(C)
    sd["a.emb_rel_k"], sd["a.emb_rel_v"] = torch.randn(1, 9, C // 2, generator=g) * 0.3, torch.randn(1, 9, C // 2, generator=g) * 0.3
    for tt in (T, 3):
        xs = x[:, :, :tt].contiguous()
        ls = torch.clamp(lens, max=tt)
        ms = (torch.arange(tt).view(1, 1, tt) < ls.view(B, 1, 1)).float()
        ref = vio.relative_self_attention(sd, "a", xs, ms, 2, 4)              # identity projections: q = k = v = x
        got = hip_ops.rel_attention(xs.cuda(), xs.cuda(), xs.cuda(), sd["a.emb_rel_k"][0].cuda(), sd["a.emb_rel_v"][0].cuda(),
                                    ls.to(torch.int32).cuda(), 2, 4).cpu()
        assert ((got - ref) * ms).abs().max().item() <= 2e-5, tt
    # spline, both directions, folded flips
    z = torch.randn(B, 2, T, generator=g) * 3
    h = torch.randn(B, 29, T, generator=g)
    for inverse in (True, False):
        hm = (h * mask).reshape(B, 1, 29, T).permute(0, 1, 3, 2)
        y1 = vio.rq_spline(z[:, 1:], hm[..., :10] / 8.0, hm[..., 10:20] / 8.0, hm[..., 20:], inverse, 5.0)
        ref = torch.cat([z[:, :1], y1], 1) * mask
        got = hip_ops.spline_flow(z.cuda(), h.cuda(), lens_d, 10, 64, 5.0, inverse).cpu()
        assert (got - ref).abs().max().item() <= 2e-4, inverse
        got = hip_ops.spline_flow(torch.flip(z, [1]).cuda(), h.cuda(), lens_d, 10, 64, 5.0, inverse, flip_in=True, flip_out=True).cpu()
        assert (got - torch.flip(ref, [1])).abs().max().item() <= 2e-4
    # durations -> path -> expansion
    logw = torch.randn(B, 1, T, generator=g)
    w_ceil, cum, ylen = hip_ops.durations(logw.cuda(), lens_d, 1.1)
    rw = torch.ceil(torch.exp(logw) * mask * 1.1)
    assert torch.equal(w_ceil.cpu(), rw)
    ry = torch.clamp_min(rw.sum(dim=(1, 2)), 1).long()
    assert torch.equal(ylen.cpu().long(), ry)
    ty = int(ry.max())
    ymask = (torch.arange(ty).view(1, 1, ty) < ry.view(B, 1, 1)).float()
    path = vio.generate_path(rw, mask.unsqueeze(2) * ymask.unsqueeze(-1))
    src = torch.randn(B, 7, T, generator=g)
    out, attn = hip_ops.expand_path(src.cuda(), cum, lens_d, ylen, ty, want_attn=True)
    assert torch.equal(attn.cpu(), path)
    assert torch.equal(out.cpu(), torch.matmul(path.squeeze(1), src.transpose(1, 2)).transpose(1, 2))
