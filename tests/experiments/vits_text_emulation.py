#!/usr/bin/env python
"""Line-by-line numpy transliteration of three EXPERIMENTAL kernels of amphion_amd/csrc/vits_text.hip (which have not
run on hardware yet) checked against the oracle (oracle/vits_infer_oracle.py, pinned on the reference):
  spline_flow_kernel   (knot construction, bin search, forward / inverse formula, folded flips, masking)
  rel_attention_kernel (scaled scores + windowed relative logits, -1e4 masking, softmax, relative values)
  durations_kernel + expand_path_kernel  (ceil / running sum / frame -> token search vs generate_path @ src)
Catches algorithm / indexing slips before GPU time is spent on them; says nothing about HIP-specific mistakes."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vits_infer_oracle as vio  # noqa: E402

f32 = np.float32


def spline_kernel(z, h, lens, K, fc, tail, inverse, flip_in, flip_out):
    B, _, T = z.shape
    zo = np.zeros_like(z)
    isf = f32(1.0 / math.sqrt(fc))
    for b in range(B):
        for t in range(T):
            m = f32(1.0 if t < lens[b] else 0.0)
            x0 = z[b, 1 if flip_in else 0, t]
            x1 = z[b, 0 if flip_in else 1, t]
            out = x1
            if -tail <= x1 <= tail:
                kn = []
                for p, lo in ((0, 1e-3), (1, 1e-3)):
                    u = np.array([h[b, p * K + i, t] * isf * m for i in range(K)], dtype=f32)
                    e = np.exp(u - u.max())
                    frac = f32(lo) + f32(1 - lo * K) * (e / e.sum())
                    k = np.empty(K + 1, dtype=f32)
                    k[0] = -tail
                    cum = f32(0)
                    for i in range(K):
                        cum += frac[i]
                        k[i + 1] = f32(2 * tail) * cum - f32(tail)
                    k[K] = tail
                    kn.append(k)
                xk, yk = kn
                dd = np.ones(K + 1, dtype=f32)
                for i in range(1, K):
                    u = h[b, 2 * K + i - 1, t] * m
                    dd[i] = f32(1e-3) + (u if u > 20 else f32(np.log1p(np.exp(u))))
                src = yk if inverse else xk
                bn = sum(1 for i in range(1, K) if x1 >= src[i])
                xa, wb, ya, hb = xk[bn], xk[bn + 1] - xk[bn], yk[bn], yk[bn + 1] - yk[bn]
                d0, d1 = dd[bn], dd[bn + 1]
                s = hb / wb
                if inverse:
                    dy = x1 - ya
                    e = d0 + d1 - 2 * s
                    qa, qb, qc = dy * e + hb * (s - d0), hb * d0 - dy * e, -s * dy
                    root = (2 * qc) / (-qb - np.sqrt(qb * qb - 4 * qa * qc))
                    out = root * wb + xa
                else:
                    th = (x1 - xa) / wb
                    tt = th * (1 - th)
                    out = ya + hb * (s * th * th + d0 * tt) / (s + (d0 + d1 - 2 * s) * tt)
            zo[b, 1 if flip_out else 0, t] = x0 * m
            zo[b, 0 if flip_out else 1, t] = out * m
    return zo


def attention_kernel(q, k, v, ek, ev, lens, H, window):
    B, C, T = q.shape
    dk = C // H
    out = np.zeros_like(q)
    scale = f32(1.0 / math.sqrt(dk))
    for b in range(B):
        for h in range(H):
            base = slice(h * dk, (h + 1) * dk)
            for i in range(T):
                qs = q[b, base, i] * scale
                p = np.empty(T, dtype=f32)
                for j in range(T):
                    s = f32(np.dot(qs, k[b, base, j]))
                    r = j - i + window
                    if 0 <= r <= 2 * window:
                        s += f32(np.dot(qs, ek[r]))
                    if i >= lens[b] or j >= lens[b]:
                        s = f32(-1e4)
                    p[j] = s
                p = np.exp(p - p.max())
                inv = f32(1.0) / p.sum()
                for d in range(dk):
                    acc = f32(np.dot(p, v[b, h * dk + d, :]))
                    jlo, jhi = max(i - window, 0), min(i + window, T - 1)
                    ar = sum(p[j] * ev[j - i + window, d] for j in range(jlo, jhi + 1))
                    out[b, h * dk + d, i] = (acc + ar) * inv
    return out


def durations_expand(logw, lens, ls, src):
    B, _, T = logw.shape
    w_ceil = np.zeros((B, 1, T), dtype=f32)
    cum = np.zeros((B, T), dtype=np.int32)
    ylen = np.zeros(B, dtype=np.int32)
    for b in range(B):
        run = f32(0)
        for t in range(T):
            wv = f32(np.ceil(np.exp(logw[b, 0, t]) * f32(ls))) if t < lens[b] else f32(0)
            w_ceil[b, 0, t] = wv
            run += wv
            cum[b, t] = int(run)
        ylen[b] = 1 if run < 1 else int(run)
    Ty = int(ylen.max())
    D = src.shape[1]
    out = np.zeros((B, D, Ty), dtype=f32)
    attn = np.zeros((B, 1, Ty, T), dtype=f32)
    for b in range(B):
        for y in range(Ty):
            tok = -1
            if y < ylen[b]:
                for x in range(lens[b]):
                    lo = 0 if x == 0 else cum[b, x - 1]
                    if lo <= y < cum[b, x]:
                        tok = x
                        break
            if tok >= 0:
                out[b, :, y] = src[b, :, tok]
                attn[b, 0, y, tok] = 1
    return w_ceil, ylen, out, attn


def main():
    g = torch.Generator().manual_seed(9)
    B, T = 3, 40
    lens = [40, 23, 1]
    mask = (torch.arange(T).view(1, 1, T) < torch.tensor(lens).view(B, 1, 1)).float()
    # spline
    z = torch.randn(B, 2, T, generator=g) * 3
    h = torch.randn(B, 29, T, generator=g) * 2
    worst = 0.0
    for inverse in (True, False):
        hm = (h * mask).reshape(B, 1, 29, T).permute(0, 1, 3, 2)
        y1 = vio.rq_spline(z[:, 1:], hm[..., :10] / 8.0, hm[..., 10:20] / 8.0, hm[..., 20:], inverse, 5.0)
        ref = (torch.cat([z[:, :1], y1], 1) * mask).numpy()
        got = spline_kernel(z.numpy(), h.numpy(), lens, 10, 64, f32(5.0), inverse, False, False)
        worst = max(worst, float(np.abs(got - ref).max()))
        got = spline_kernel(torch.flip(z, [1]).numpy(), h.numpy(), lens, 10, 64, f32(5.0), inverse, True, True)
        worst = max(worst, float(np.abs(got - np.flip(ref, 1)).max()))
    print("spline kernel transliteration vs oracle:", worst)
    assert worst < 5e-4
    # attention
    C, H, w = 16, 2, 4
    x = torch.randn(B, C, T, generator=g)
    sd = {"a.emb_rel_k": torch.randn(1, 9, C // H, generator=g) * 0.3, "a.emb_rel_v": torch.randn(1, 9, C // H, generator=g) * 0.3}
    for n in "qkvo":
        sd[f"a.conv_{n}.weight"], sd[f"a.conv_{n}.bias"] = torch.eye(C).unsqueeze(-1), torch.zeros(C)
    worst = 0.0
    for tt in (T, 3):
        xs = x[:, :, :tt].contiguous()
        ls = [min(n, tt) for n in lens]
        ms = (torch.arange(tt).view(1, 1, tt) < torch.tensor(ls).view(B, 1, 1)).float()
        ref = (vio.relative_self_attention(sd, "a", xs, ms, H, w) * ms).numpy()
        got = attention_kernel(xs.numpy(), xs.numpy(), xs.numpy(), sd["a.emb_rel_k"][0].numpy(), sd["a.emb_rel_v"][0].numpy(), ls, H, w)
        worst = max(worst, float(np.abs(got * ms.numpy() - ref).max()))
    print("attention kernel transliteration vs oracle:", worst)
    assert worst < 2e-5
    # durations -> path -> expansion
    logw = torch.randn(B, 1, T, generator=g)
    src = torch.randn(B, 5, T, generator=g)
    w_ceil, ylen, out, attn = durations_expand(logw.numpy(), lens, 1.1, src.numpy())
    rw = torch.ceil(torch.exp(logw) * mask * 1.1)
    ry = torch.clamp_min(rw.sum(dim=(1, 2)), 1).long()
    ty = int(ry.max())
    ymask = (torch.arange(ty).view(1, 1, ty) < ry.view(B, 1, 1)).float()
    path = vio.generate_path(rw, mask.unsqueeze(2) * ymask.unsqueeze(-1))
    assert np.array_equal(w_ceil, rw.numpy()) and np.array_equal(ylen, ry.numpy())
    assert np.array_equal(attn, path.numpy())
    assert np.array_equal(out, torch.matmul(path.squeeze(1), src.transpose(1, 2)).transpose(1, 2).numpy())
    print("durations / path / expansion transliteration: exact")


if __name__ == "__main__":
    main()
