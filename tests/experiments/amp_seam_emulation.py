#!/usr/bin/env python
"""CPU emulation of the seam of the EXPERIMENTAL fused AMPBlock1 pair (amphion_amd/csrc/amp_pair_f16x3.hip): the tile
geometry (qa = q0 - H2 - 5, N1 conv1 columns, NV = N1 - 10 activated columns), the replicate clamps at the utterance's
own ends, the sliding 12-value ring under the down filter and the zeroing outside the utterance, restated in numpy and
compared with the oracle's Activation1d on whole rows -- random lengths, ragged valid lengths, every tile of the row,
k = 3 / 7 / 11.  Prints the worst difference (fp32 noise, ~1e-6, when the indexing is right)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vocoder_oracle as vo  # noqa: E402

AH = 5


def seam_row(xt, Tv, q0, H2, N1, fu2, fd, a, invb, cpl=11):
    qa = q0 - H2 - AH
    NV = N1 - 2 * AH
    scr = np.full(N1, np.nan, dtype=np.float32)          # scratch row: tile column c holds global column qa + c
    for c in range(N1):
        if 0 <= qa + c < len(xt):
            scr[c] = xt[qa + c]
    twoT = 2 * Tv

    def sval(n):
        n = min(max(n, 0), twoT - 1)
        mmax, odd = (n + 15) >> 1, (n + 15) & 1
        u = np.float32(0)
        for k in range(6):
            xi = min(max(mmax - k - 5, 0), Tv - 1)
            ci = min(max(xi - qa, 0), N1 - 1)
            u = np.float32(np.float32(scr[ci]) * np.float32(fu2[2 * k + odd]) + u)
        return np.float32(u + invb * np.sin(np.float64(u) * a) ** 2)

    out = np.zeros(NV, dtype=np.float32)
    for oc0 in range(0, NV, cpl):                          # one lane's stretch
        qf = qa + AH + oc0
        ring, n = [0.0] * 12, 2 * qf - 5
        for _ in range(10):
            ring = ring[1:] + [sval(n)]
            n += 1
        for o in range(cpl):
            ring = ring[1:] + [sval(n)]
            ring = ring[1:] + [sval(n + 1)]
            n += 2
            ae = sum(np.float32(fd[2 * m]) * ring[2 * m] for m in range(6))
            ao = sum(np.float32(fd[2 * m + 1]) * ring[2 * m + 1] for m in range(6))
            if oc0 + o < NV:
                out[oc0 + o] = (ae + ao) if 0 <= qf + o < Tv else 0.0
    return out


def main():
    rng = np.random.default_rng(0)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12).reshape(-1).numpy()
    worst = 0.0
    for trial in range(40):
        T = int(rng.integers(30, 400))
        Tv = int(rng.integers(1, T + 1)) if trial % 2 else T
        x = rng.standard_normal(T).astype(np.float32) * 1.5
        al, be = float(rng.standard_normal() * 0.3), float(rng.standard_normal() * 0.3)
        ref = vo.activation1d(torch.from_numpy(x[:Tv]).reshape(1, 1, -1), torch.tensor([al]), torch.tensor([be]), True).reshape(-1).numpy()
        KT = int(rng.choice([3, 7, 11]))
        H2, N1 = (KT - 1) // 2, 96
        NV = N1 - 2 * AH
        NT = NV - 2 * H2
        for tile in range((T + NT - 1) // NT):
            q0 = tile * NT
            y = seam_row(x, Tv, q0, H2, N1, 2 * f, f, np.exp(al), 1.0 / (np.exp(be) + 1e-9))
            for oc in range(NV):
                q = q0 - H2 + oc
                worst = max(worst, abs(float(y[oc]) - (float(ref[q]) if 0 <= q < Tv else 0.0)))
    print("max |seam emulation - oracle activation1d| =", worst)
    assert worst < 5e-6


if __name__ == "__main__":
    main()
