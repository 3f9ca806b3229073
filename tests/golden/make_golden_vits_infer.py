#!/usr/bin/env python
"""Golden vectors for FULL VITS inference (text encoder with relative-position attention,
stochastic / standard duration predictor in reverse, generate_path, flow reverse, HiFi-GAN decoder) from the REAL
reference ``SynthesizerTrn.infer`` (models/tts/vits/vits.py:320-369) on CPU.  Build-container only:

    python tests/golden/make_golden_vits_infer.py   ->  tests/golden/golden_vits_infer.npz (+ keys_vits_synthesizer.json)

A SMALL model with oracle/synth.py's seeded weights over the reference's own key / shape list (dumped to
keys_vits_infer_<tag>.json; the tests rebuild the same tensors, nothing but inputs and outputs is stored).  That also
re-draws the layers the reference zero-initialises (ConvFlow.proj, coupling ``post``: modules/flow/modules.py:375-376,
419-420), which would otherwise make the splines and the flow the identity and the test blind.  The two Gaussian draws of ``infer`` (duration noise [B, 2, T_text],
then ``randn_like(m_p)``) are recorded by replaying the same seed."""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "shims"), "/root/reference"]
warnings.filterwarnings("ignore")

SMALL = dict(inter_channels=16, hidden_channels=32, filter_channels=64, n_heads=2, n_layers=2, kernel_size=3, p_dropout=0.1,
             resblock="1", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], upsample_rates=[4, 2],
             upsample_initial_channel=32, upsample_kernel_sizes=[8, 4])


def synth_weights(model, seed):
    """oracle/synth.py's seeded scheme over the reference model's own key / shape list: the test side rebuilds the
    same tensors from tests/golden/keys_vits_infer_<tag>.json, so no weights are stored.  (It also re-draws the layers
    the reference zero-initialises -- ConvFlow.proj, the coupling layers' post -- which would otherwise make the
    splines and the flow the identity.)"""
    from oracle import synth

    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synth.synth_state_dict(shapes, seed, g_gain=0.5))
    return model, shapes


# config/vits.json:28-75 (the model section: hidden 192, 6 layers, filter 768, SDP, gin_channels 256 with n_speakers 0)
FULL = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1,
            resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[8, 8, 2, 2],
            upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], n_speakers=0, gin_channels=256, use_sdp=True)
FULL_SEED, FULL_NOISE_SEED, FULL_DECIM = 77, 126, 4   # noise seed chosen so that no duration sits within 1.5 % of the ceil() cliff


def full_weights(shapes):
    """Seeded weights of the full-size model: oracle/synth.py's scheme with weight_g gain 0.5 everywhere but the
    waveform decoder (gain 1.0: |y_hat| up to ~0.5 instead of 0.08, so the 1e-4 bound means something)."""
    from oracle import synth

    return {k: synth.synth_tensor(k, tuple(v), FULL_SEED, 1.0 if k.startswith("dec.") else 0.5) for k, v in shapes.items()}


def make_full():
    """SynthesizerTrn.infer at config/vits.json dimensions, B = 4, T_text = 100 / 83 / 57 / 31 ->
    golden_vits_infer_full.npz.  Stored: tokens, lengths, integer durations (the alignment path follows from them), y_hat in
    full, and z / z_p / m_p / logs_p on every 4th frame (the kernels' tiling errors would not hide between frames); the two noise
    draws are NOT stored -- the test replays torch.manual_seed(noise_seed) -> randn(B, 2, T_text) and
    randn(B, 192, T_y, generator = seed + 1), and checks the checksums recorded here."""
    from models.tts.vits.vits import SynthesizerTrn

    net = SynthesizerTrn(512, 513, 32, **FULL)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "keys_vits_synthesizer.json")) as f:
        assert [[k, list(v)] for k, v in shapes.items()] == json.load(f), "keys_vits_synthesizer.json is stale"
    net.load_state_dict(full_weights(shapes))
    net.eval()
    gen = torch.Generator().manual_seed(13)
    x = torch.randint(0, 512, (4, 100), generator=gen)
    x_lengths = torch.tensor([100, 83, 57, 31])
    args = dict(noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8)
    # the duration noise is the reference's own torch.randn(B, 2, T_text) under the seed; its randn_like(m_p) fills a
    # TRANSPOSED view (vits.py:346-352) in an order a plain randn cannot replay, so that one draw is served from a seeded
    # generator of its own (same distribution; the test rebuilds it with the same call)
    zgen = torch.Generator().manual_seed(FULL_NOISE_SEED + 1)
    drawn = []
    orig_randn_like = torch.randn_like

    def seeded_randn_like(t, **kw):
        drawn.append(torch.randn(tuple(t.shape), generator=zgen))
        return drawn[-1]

    with torch.no_grad():
        torch.manual_seed(FULL_NOISE_SEED)
        torch.randn_like = seeded_randn_like
        try:
            o = net.infer(x, x_lengths, **args)
        finally:
            torch.randn_like = orig_randn_like
        assert len(drawn) == 1
        n_z = drawn[0]
        torch.manual_seed(FULL_NOISE_SEED)
        n_dp = torch.randn(4, 2, 100)
        xe, m, logs, x_mask = net.enc_p(x, x_lengths)
        torch.manual_seed(FULL_NOISE_SEED)
        logw = net.dp(xe, x_mask, g=None, reverse=True, noise_scale=args["noise_scale_w"])
    w = torch.exp(logw) * x_mask * args["length_scale"]
    margin = float((w - torch.round(w)).abs()[x_mask.bool()].min())     # distance of a duration from the ceil() cliff
    dur = torch.ceil(w)[:, 0].to(torch.int32)
    assert torch.equal(o["attn"].sum(2)[:, 0].to(torch.int32), dur)
    D = FULL_DECIM
    out = {"x": x.numpy(), "x_lengths": x_lengths.numpy(), "durations": dur.numpy(), "y_hat": o["y_hat"].numpy(),
           "logw": logw.numpy(), "enc_m": m.numpy()[:, :, ::D], "enc_logs": logs.numpy()[:, :, ::D],
           "noise_seed": np.array(FULL_NOISE_SEED), "weight_seed": np.array(FULL_SEED), "decim": np.array(D),
           "noise_dp_check": np.array([float(n_dp.double().sum()), float(n_dp[1, 1, 17])]),
           "noise_z_check": np.array([float(n_z.double().sum()), float(n_z[2, 100, 5])]), "y_frames": o["mask"].sum(dim=(1, 2)).numpy()}
    for k in ("z", "z_p", "m_p", "logs_p"):
        out[k] = o[k].numpy()[:, :, ::D]
    np.savez_compressed(os.path.join(HERE, "golden_vits_infer_full.npz"), **out)
    print("full: frames", out["y_frames"].tolist(), "y_hat", tuple(o["y_hat"].shape), "absmax", float(o["y_hat"].abs().max()), "|z| max",
          float(o["z"].abs().max()), "duration margin", margin, "max duration", int(dur.max()),
          "bytes", os.path.getsize(os.path.join(HERE, "golden_vits_infer_full.npz")))


def main():
    from models.tts.vits.vits import SynthesizerTrn

    if "--full" in sys.argv:
        return make_full()

    out = {}
    with open(os.path.join(HERE, "keys_vits_synthesizer.json"), "w") as f:   # key / shape list of the full-size model
        full = SynthesizerTrn(512, 513, 32, inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6,
                              kernel_size=3, p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11],
                              resblock_dilation_sizes=[[1, 3, 5]] * 3, upsample_rates=[8, 8, 2, 2], upsample_initial_channel=512,
                              upsample_kernel_sizes=[16, 16, 4, 4], n_speakers=0, gin_channels=256, use_sdp=True)
        json.dump([[k, list(v.shape)] for k, v in full.state_dict().items()], f)
    for tag, kw in {"sdp": dict(n_speakers=0, gin_channels=0, use_sdp=True),
                    "sdp_spk": dict(n_speakers=3, gin_channels=8, use_sdp=True),
                    "dp": dict(n_speakers=0, gin_channels=0, use_sdp=False)}.items():
        net, shapes = synth_weights(SynthesizerTrn(40, 33, 8, **SMALL, **kw), 77)
        net.eval()
        with open(os.path.join(HERE, f"keys_vits_infer_{tag}.json"), "w") as f:
            json.dump([[k, list(v)] for k, v in shapes.items()], f)
        gen = torch.Generator().manual_seed(13)
        x = torch.randint(0, 40, (2, 11), generator=gen)
        x_lengths = torch.tensor([11, 7])
        sid = torch.tensor([[2], [0]]) if kw["n_speakers"] else None
        args = dict(noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8)
        with torch.no_grad():
            torch.manual_seed(123)
            o = net.infer(x, x_lengths, sid=sid, **args)
            torch.manual_seed(123)                                        # replay the two draws of infer()
            n_dp = torch.randn(2, 2, 11) if kw["use_sdp"] else None
            n_z = torch.randn_like(o["m_p"])
            xe, m, logs, x_mask = net.enc_p(x, x_lengths)
            g = net.emb_g(sid.squeeze(-1)).unsqueeze(-1) if sid is not None else None
            if kw["use_sdp"]:
                torch.manual_seed(123)
                logw = net.dp(xe, x_mask, g=g, reverse=True, noise_scale=args["noise_scale_w"])
            else:
                logw = net.dp(xe, x_mask, g=g)
        t = tag + "_"
        out[t + "x"], out[t + "x_lengths"] = x.numpy(), x_lengths.numpy()
        if sid is not None:
            out[t + "sid"] = sid.numpy()
        if n_dp is not None:
            out[t + "noise_dp"] = n_dp.numpy()
        out[t + "noise_z"] = n_z.numpy()
        out[t + "enc_x"], out[t + "enc_m"], out[t + "enc_logs"] = xe.numpy(), m.numpy(), logs.numpy()
        out[t + "logw"] = logw.numpy()
        for k in ("y_hat", "attn", "mask", "z", "z_p", "m_p", "logs_p"):
            out[t + k] = o[k].numpy()
        print(tag, "frames", o["mask"].sum(dim=(1, 2)).tolist(), "y_hat", tuple(o["y_hat"].shape), "absmax", float(o["y_hat"].abs().max()),
              "logw range", float(logw.min()), float(logw.max()))
    np.savez_compressed(os.path.join(HERE, "golden_vits_infer.npz"), **out)
    print("wrote golden_vits_infer.npz", os.path.getsize(os.path.join(HERE, "golden_vits_infer.npz")))


if __name__ == "__main__":
    main()
