#!/usr/bin/env python
"""Diagnostics for the whole-AMPBlock kernel: fused / unfused against an fp64 reference, by region and by column."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from amphion_amd import _lib
from oracle import vocoder_oracle as vo
from hip_helpers import ampblock_forward
from test_gpu_ampblock import _params, _ref64, _rand


def ranges(idx):
    out, s, p = [], None, None
    for i in idx:
        if s is None: s = p = i
        elif i == p + 1: p = i
        else: out.append((s, p)); s = p = i
    if s is not None: out.append((s, p))
    return out


def run(C, k, dils, B, T, al_shift=0.0, mode=2, tag=""):
    _lib.set_precision("f16x3")
    _lib.check(_lib.lib().amp_set_ampblock_fusion(mode))
    n = len(dils)
    ws1, bs1, ws2, bs2, al, be = _params(C, k, n)
    al = al + al_shift
    x = _rand(B, C, T, seed=B + T, scale=1.5)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    ref = _ref64(ws1, bs1, ws2, bs2, al, be, x, dils)
    yu = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=dils, fused=False).double()
    yf = ampblock_forward(ws1, bs1, ws2, bs2, al, be, True, f, f, x, dilations=dils, fused=True).double()
    eu, ef = (yu - ref).abs(), (yf - ref).abs()
    print(f"== {tag} C={C} k={k} dils={dils} B={B} T={T} al_shift={al_shift} mode={mode}: |unfused-ref| {eu.max():.2e}  |fused-ref| {ef.max():.2e}  "
          f"fused==unfused {torch.equal(yu, yf)}  ndiff {(yu != yf).sum().item()} / {yu.numel()}")
    if not torch.equal(yu, yf):
        d = (yf - yu).abs()
        colerr = d.amax(dim=(0, 1))
        bad = (colerr > 0).nonzero().flatten().tolist()
        print("   columns that differ:", ranges(bad)[:40])
        big = (colerr > 1e-3).nonzero().flatten().tolist()
        print("   columns with |d| > 1e-3:", ranges(big)[:40])
        cherr = d.amax(dim=(0, 2))
        print("   channels with |d| > 1e-3:", ranges((cherr > 1e-3).nonzero().flatten().tolist()))
        print("   sample  fused", yf[0, 0, 500:504].tolist(), " unfused", yu[0, 0, 500:504].tolist(), " ref", ref[0, 0, 500:504].tolist())


if __name__ == "__main__" and len(sys.argv) == 1:
    run(32, 3, (1,), 1, 2000, tag="one pair")
    run(32, 3, (1,), 1, 2000, al_shift=-40.0, tag="one pair, alpha -> 0")
    run(32, 3, (1, 3, 5), 1, 2000, tag="three pairs")
    run(32, 3, (1,), 1, 2000, mode=3, tag="one pair, 4-wave tiles")
    run(64, 3, (1,), 1, 1200, tag="C = 64")


def probe(tag, C, k, T, w1, b1, w2, b2, al, be, mode=2, d=1):
    _lib.set_precision("f16x3")
    _lib.check(_lib.lib().amp_set_ampblock_fusion(mode))
    x = _rand(1, C, T, seed=7, scale=1.5)
    f = vo.kaiser_sinc_filter1d(0.25, 0.3, 12)
    yu = ampblock_forward([w1], [b1], [w2], [b2], al, be, True, f, f, x, dilations=(d,), fused=False)
    yf = ampblock_forward([w1], [b1], [w2], [b2], al, be, True, f, f, x, dilations=(d,), fused=True)
    dd = (yf - yu).abs()
    colerr = dd.amax(dim=(0, 1))
    cherr = dd.amax(dim=(0, 2))
    bad = (colerr > 1e-5).nonzero().flatten().tolist()
    print(f"-- {tag}: max|d| {dd.max():.3e}  bad columns {len(bad)}/{T} {ranges(bad)[:16]}  bad channels {ranges((cherr > 1e-5).nonzero().flatten().tolist())}")
    if bad:
        c0 = bad[len(bad) // 2]
        print(f"   at column {c0}: fused {yf[0, :6, c0].tolist()}\n                 unfused {yu[0, :6, c0].tolist()}")
        print(f"   channel 3, columns {c0}..{c0 + 5}: fused {yf[0, 3, c0:c0 + 6].tolist()}\n                 unfused {yu[0, 3, c0:c0 + 6].tolist()}")


def probes():
    C, k, T = 32, 3, 2000
    z = torch.zeros(C, C, k)
    ident = torch.zeros(C, C, k)
    for c in range(C):
        ident[c, c, k // 2] = 1.0
    shift = torch.zeros(C, C, k)           # out channel c <- in channel (c + 1) % C, tap 0 (column t - 1)
    for c in range(C):
        shift[c, (c + 1) % C, 0] = 1.0
    zb = torch.zeros(C)
    rb_ = torch.arange(C).float() * 0.1 - 1.0
    al0 = torch.full((2, C), -40.0)
    be0 = torch.zeros(2, C)
    al1 = _rand(2, C, seed=50, scale=0.3)
    be1 = _rand(2, C, seed=51, scale=0.3)
    probe("A  w1=0 w2=0 b2=ramp           (load / init / swap / store)", C, k, T, z, zb, z, rb_, al0, be0)
    probe("B  w1=0 b1=ramp w2=ident alpha->0 (a2 of constants, conv ident)", C, k, T, z, rb_, ident, zb, al0, be0)
    probe("C  w1=ident w2=0 (dead c1)", C, k, T, ident, zb, z, rb_, al0, be0)
    probe("D  w1=ident w2=ident alpha->0    (two low-pass activations + identity convs)", C, k, T, ident, zb, ident, zb, al0, be0)
    probe("E  w1=ident w2=ident snake       (full activations, identity convs)", C, k, T, ident, zb, ident, zb, al1, be1)
    probe("F  w1=shift w2=ident alpha->0    (channel / column mapping of the conv)", C, k, T, shift, zb, ident, zb, al0, be0)
    probe("G  random weights alpha->0", C, k, T, _rand(C, C, k, seed=1, scale=0.1), rb_, _rand(C, C, k, seed=2, scale=0.1), zb, al0, be0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "probes":
    probes()
