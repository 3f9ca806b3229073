#!/usr/bin/env python
"""Golden vectors for BASELINE.json config 5 (VITS posterior encoder + flow around the decoder),
produced by the REAL reference modules on CPU.  Build-container only:

    python tests/golden/make_golden_vits.py

Weights: oracle/synth.py seeded scheme; the coupling layers' zero-initialised ``post`` convs
(modules/flow/modules.py:375-376) are re-drawn (un-normed ``weight`` ~ N(0,1)/sqrt(fan_in)) so the flow
is not the identity (SURVEY.md §8d C5 caveat).  The Gaussian noise of enc_q is recorded
(seed -> torch.randn_like) so z can be compared."""
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

from oracle import synth  # noqa: E402


def main():
    from models.tts.vits.vits import PosteriorEncoder, ResidualCouplingBlock

    out = {}
    for gin in (0, 256):
        enc = PosteriorEncoder(513, 192, 192, 5, 1, 16, gin_channels=gin)
        flow = ResidualCouplingBlock(192, 192, 5, 1, 4, gin_channels=gin)
        with open(os.path.join(HERE, f"keys_vits_enc_q_g{gin}.json"), "w") as f:
            json.dump([[k, list(v.shape)] for k, v in enc.state_dict().items()], f)
        with open(os.path.join(HERE, f"keys_vits_flow_g{gin}.json"), "w") as f:
            json.dump([[k, list(v.shape)] for k, v in flow.state_dict().items()], f)
        se = synth.synth_state_dict(synth.posterior_encoder_param_shapes(gin_channels=gin), 2468, g_gain=0.5)
        sf = synth.synth_state_dict(synth.coupling_block_param_shapes(gin_channels=gin), 1357, g_gain=0.5)
        assert list(se) == list(enc.state_dict()) and list(sf) == list(flow.state_dict())
        enc.load_state_dict(se)
        flow.load_state_dict(sf)
        enc.eval()
        flow.eval()
        gen = torch.Generator().manual_seed(31 + gin)
        y = torch.rand(2, 513, 12, generator=gen)           # linear-spectrogram magnitudes >= 0
        lens = torch.tensor([12, 9])
        g = torch.randn(2, gin, 1, generator=gen) if gin else None
        with torch.no_grad():
            torch.manual_seed(99)
            z, m, logs, mask = enc(y, lens, g=g)
            torch.manual_seed(99)
            noise = torch.randn_like(m)
            z_p = flow(z, mask, g=g)
            z_hat = flow(z_p, mask, g=g, reverse=True)
        t = f"vits_g{gin}_"
        out[t + "y"], out[t + "lens"], out[t + "noise"] = y.numpy(), lens.numpy(), noise.numpy()
        if gin:
            out[t + "g"] = g.numpy()
        out[t + "z"], out[t + "m"], out[t + "logs"] = z.numpy(), m.numpy(), logs.numpy()
        out[t + "z_p"], out[t + "z_hat"] = z_p.numpy(), z_hat.numpy()
        print(gin, "z absmax", float(z.abs().max()), "z_p", float(z_p.abs().max()), "roundtrip", float((z_hat - z).abs().max()))
    np.savez_compressed(os.path.join(HERE, "golden_vits.npz"), **out)
    print("wrote golden_vits.npz", os.path.getsize(os.path.join(HERE, "golden_vits.npz")))


if __name__ == "__main__":
    main()
