"""CPU emulation of the index scheme of conv_f16x3.hip's fused Activation1d epilogue (ACT variant): tiles advancing by
NT - 16 columns from q0 = tile * AT - 8, a 12-value window per lane, Snake values exchanged through a per-row buffer, the
replicate padding of both stages at the utterance's ends -- against the definition (oracle formulas of
modules/anti_aliasing: replicate-pad 5, up-sample x2 with the 12-tap filter, Snake, replicate-pad (5, 6), 12-tap
down-sample).  Arithmetic in float64; this checks INDICES, the GPU tests check bits."""
import numpy as np


def reference(y, fu, fd, a, invb):
    T = len(y)
    n = np.arange(2 * T)
    u = np.zeros(2 * T)
    for k in range(6):                      # u[n] = sum_k xp[(n + 5 >> 1) - k] * 2 f[par + 2k],  xp = replicate
        idx = np.clip(((n + 5) >> 1) - k, 0, T - 1)
        u += y[idx] * 2 * fu[((n + 1) & 1) + 2 * k]
    s = u + invb * np.sin(a * u) ** 2
    out = np.zeros(T)
    for j in range(12):
        out += fd[j] * s[np.clip(2 * np.arange(T) + j - 5, 0, 2 * T - 1)]
    return out


def tiled(y, fu, fd, a, invb, NT):
    Tv = len(y)
    AT = NT - 16
    out = np.full(Tv, np.nan)
    for tile in range((Tv + AT - 1) // AT):
        q0 = tile * AT - 8
        if q0 + 8 >= Tv:
            continue
        yl = np.array([y[q0 + c] if 0 <= q0 + c < Tv else np.nan for c in range(NT)])
        edge = q0 < 0 or q0 + NT > Tv
        if edge:
            clo = -q0 if q0 < 0 else 0
            chi = min(Tv - 1 - q0, NT - 1)
            yl[:clo] = yl[clo]
            yl[chi + 1:] = yl[chi]
        srow = np.full(2 * NT + 32, np.nan)
        for cg in range(NT // 4):
            col = 4 * cg
            base = col - 8 if col >= 8 else 0
            xw = yl[base:base + 12]
            for e in range(8):
                top, par = ((e + 10) >> 1) + 3, e & 1
                u = sum(xw[top - k] * 2 * fu[par + 2 * k] for k in range(6))
                srow[8 * cg + e] = u + invb * np.sin(a * u) ** 2
        if edge:
            ilo, ihi = 5 - 2 * q0, 2 * Tv + 4 - 2 * q0
            slo = srow[min(max(ilo, 0), 2 * NT - 1)]
            shi = srow[max(min(ihi, 2 * NT - 1), 0)]
            for i in range(2 * NT):
                if i < ilo:
                    srow[i] = slo
                elif i > ihi:
                    srow[i] = shi
        for cg in range(2, NT // 4 - 2):
            for e in range(4):
                t = q0 + 4 * cg + e
                if t < Tv:
                    out[t] = sum(fd[j] * srow[8 * cg + 2 * e + j] for j in range(12))
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    fu, fd = rng.standard_normal(12), rng.standard_normal(12)
    for NT in (128, 256, 512):
        for Tv in (NT - 16, NT - 15, 3 * NT + 7, 2 * (NT - 16), 2 * (NT - 16) - 9, 2 * (NT - 16) + 1, 1000, 1024):
            y = rng.standard_normal(Tv)
            r = reference(y, fu, fd, 0.7, 1.3)
            t = tiled(y, fu, fd, 0.7, 1.3, NT)
            assert not np.isnan(t).any(), (NT, Tv, np.where(np.isnan(t))[0][:5])
            assert np.abs(r - t).max() < 1e-12, (NT, Tv, np.abs(r - t).max(), np.argmax(np.abs(r - t)))
    print("tile scheme == definition for every case")
