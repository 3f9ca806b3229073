#!/usr/bin/env python
"""Phase timeline of pair_strip_kernel (the dominant kernel of the headline) from in-kernel clock stamps of EVERY wave of EVERY workgroup.
Experiment build: apply profiles/negative_kernels/r5_pair_strip_stamps.patch, then
    AMP_BUILD_TAG=ps AMP_BUILD_FLAGS=-DPS_TIMING python -m amphion_amd.build
    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_ps.so python tools/pair_stamps.py [k] [dilation]
Stamps (per wave): 0 entry, 1 first loads issued, 2 accumulators initialised, 3 chunk 0 staged (waits for the loads), 4 barrier,
5+3c / 6+3c / 7+3c conv1 chunk c: MFMAs issued / every wave has read the tile / next chunk staged + barrier, 29 seam written,
30 barrier, 31+c conv2 chunk c issued, 39 next loads issued, 40 epilogue stores issued, 41 stores have left.
Per workgroup: start, end, HW_ID, XCC_ID -> gap between consecutive workgroups of one CU.  Tuning aid; not part of the product."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd import _lib  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, C, T = 64, 128, 16384
L = _lib.lib()
_lib.set_precision("f16x3")
fn = getattr(L, f"amp_debug_ps_stamps_kt{k}", None)
if fn is None:
    sys.exit("this library has no stamps (build with -DPS_TIMING, see the docstring)")
g = torch.Generator().manual_seed(1)


def conv(dil, seed):
    gg = torch.Generator().manual_seed(seed)
    w = (torch.randn(C, C, k, generator=gg) * (C * k) ** -0.5).contiguous()
    b = torch.randn(C, generator=gg) * 0.1
    h = ctypes.c_void_p()
    _lib.check(L.amp_conv_create(0, C, C, k, 1, dil, (k * dil - dil) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
    return h


h1, h2 = conv(d, 1), conv(1, 2)
x = torch.randn(B, C, T, device="cuda") * 0.5
y = torch.empty_like(x)
st = _lib.current_stream_ptr(x.device)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.check(L.amp_pair_forward(h1, h2, p(x), B, T, 0.1, p(y), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(L.amp_pair_forward(h1, h2, p(x), B, T, 0.1, p(y), st))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
NWG = 8192
allbuf = np.zeros(NWG * 4 * 64, dtype=np.int64)
wgbuf = np.zeros(NWG * 4, dtype=np.int64)
rc = fn(ctypes.c_void_p(allbuf.ctypes.data), ctypes.c_void_p(wgbuf.ctypes.data))
assert rc == 0, rc
wg = wgbuf.reshape(NWG, 4)
n = int((wg[:, 1] > 0).sum())
wg = wg[:n]
S = allbuf.reshape(NWG, 4, 64)[:n]
hw, xcc = wg[:, 2], wg[:, 3] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)     # cu_id, se_id, sh_id, xcc
cus = sorted(set(cu.tolist()))
# the counters of different XCDs / shader engines are not synchronised: calibrate on the span of ONE CU's workgroups (median over the CUs)
spans = np.array([wg[cu == c, 1].max() - wg[cu == c, 0].min() for c in cus], dtype=np.float64)
tick_us = float(np.median(spans)) / us
print(f"# pair_strip_kernel k={k} d={d} C={C} B={B} T={T}: launch {us:.0f} us, {n} workgroups on {len(cus)} CUs; a CU's first start -> last end: median {np.median(spans):.0f} ticks "
      f"=> {tick_us:.0f} ticks / us = the shader clock in MHz (the launch includes the ramp; p10 {np.percentile(spans, 10):.0f}, p90 {np.percentile(spans, 90):.0f})")
life = wg[:, 1] - wg[:, 0]
print(f"workgroup life: median {np.median(life):.0f} ticks = {np.median(life) / tick_us:.1f} us (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}); "
      f"rounds per CU: {n / len(set(cu.tolist())):.2f}")
gaps = []
for c in set(cu.tolist()):
    m = np.where(cu == c)[0]
    o = m[np.argsort(wg[m, 0])]
    gaps += (wg[o[1:], 0] - wg[o[:-1], 1]).tolist()
gaps = np.array(gaps)
print(f"gap between consecutive workgroups of one CU (end of stores -> first instruction): median {np.median(gaps):.0f} ticks = {np.median(gaps) / tick_us:.2f} us "
      f"(p10 {np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f}, negative = overlapped: {int((gaps < 0).sum())} of {len(gaps)})")


def phase(a, b):
    dlt = (S[:, :, b] - S[:, :, a]).astype(np.float64)
    return dlt


names = [("entry -> first loads issued", 0, 1), ("-> accumulators initialised", 1, 2), ("-> chunk 0 staged (waits for HBM)", 2, 3), ("-> barrier", 3, 4)]
rows = []
for nm, a_, b_ in names:
    rows.append((nm, phase(a_, b_)))
c1_issue = sum(phase(4 if c == 0 else 7 + 3 * (c - 1), 5 + 3 * c) for c in range(8))
c1_bar1 = sum(phase(5 + 3 * c, 6 + 3 * c) for c in range(8))
c1_stage = sum(phase(6 + 3 * c, 7 + 3 * c) for c in range(8))
rows.append((f"conv1: 8 x MFMA issue ({24 * k} MFMAs each)", c1_issue))
rows.append(("conv1: 8 x barrier 'tile read'", c1_bar1))
rows.append(("conv1: 8 x stage next chunk + barrier", c1_stage))
rows.append(("seam (bias, lrelu, split, xt tile write)", phase(28, 29)))
rows.append(("-> barrier", phase(29, 30)))
c2 = sum(phase(30 if c == 0 else 31 + (c - 1), 31 + c) for c in range(8))
rows.append(("conv2: 8 x MFMA issue (no barriers)", c2))
rows.append(("epilogue: next loads issued", phase(38, 39)))
rows.append(("epilogue: residual, sum, stores issued", phase(39, 40)))
rows.append(("stores have left (vmcnt 0)", phase(40, 41)))
tot = phase(0, 41)
print(f"\n{'phase (median over all waves of all workgroups)':52s} {'ticks':>8s} {'us':>7s} {'share':>6s}   p10 / p90")
for nm, v in rows:
    print(f"{nm:52s} {np.median(v):8.0f} {np.median(v) / tick_us:7.2f} {np.median(v) / np.median(tot):6.1%}   {np.percentile(v, 10):.0f} / {np.percentile(v, 90):.0f}")
print(f"{'wave life (stamp 0 -> 41)':52s} {np.median(tot):8.0f} {np.median(tot) / tick_us:7.2f}")
mf = 2 * 8 * 24 * k
print(f"\nMFMAs per wave: {mf} x 32 matrix-pipe cycles = {mf * 32} cycles; at the launch's average clock that is the floor of a wave's life.")
per = [np.median(phase(4 if c == 0 else 7 + 3 * (c - 1), 5 + 3 * c)) for c in range(8)]
print("conv1 MFMA-issue ticks per chunk:", " ".join(f"{v:.0f}" for v in per))
per2 = [np.median(phase(30 if c == 0 else 31 + (c - 1), 31 + c)) for c in range(8)]
print("conv2 MFMA-issue ticks per chunk:", " ".join(f"{v:.0f}" for v in per2))
# first round against later rounds (cold instruction cache / weights not yet in L2)
order = np.argsort(wg[:, 0])
first, later = order[: len(set(cu.tolist()))], order[len(set(cu.tolist())):]
print(f"workgroup life, first round {np.median(life[first]) / tick_us:.1f} us, later rounds {np.median(life[later]) / tick_us:.1f} us")
L.amp_conv_destroy(h1)
L.amp_conv_destroy(h2)
