#!/usr/bin/env python
"""Where a wave of conv_strip_kernel (csrc/conv_strip_f16x3.hip) spends its cycles, from in-kernel clock stamps of EVERY wave of EVERY workgroup.
Experiment build:
    AMP_BUILD_TAG=ss AMP_BUILD_FLAGS=-DAMP_STRIP_STAMPS python -m amphion_amd.build
    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_ss.so python tools/strip_stamps.py [C] [k] [dilation] [steps] [res]
Stamps per wave (shader clock): 0 entry, 1 prologue done; for step 1: 2+3c round c starts, 3+3c its MFMAs are issued, 4+3c barrier passed; 64 + 32 rr + h
behind half-tap h of round rr = 0 | 1 | second-to-last | last; 56 / 57 around the X <-> Y exchange; 58 step loop left, 59 last stores have left.  Tuning aid; not part of the product."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amphion_amd import _lib  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
with_res = int(sys.argv[5]) if len(sys.argv) > 5 else 1
B = 32
T = {128: 16384, 64: 32768, 256: 2048}[C]
L = _lib.lib()
_lib.set_precision("f16x3")
fn = getattr(L, "amp_debug_strip_stamps", None)
if fn is None:
    sys.exit("this library has no stamps (build with -DAMP_STRIP_STAMPS, see the docstring)")
fn.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous()
b = torch.randn(C, generator=g) * 0.1
h = ctypes.c_void_p()
_lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
x = torch.randn(B, C, T, device="cuda")
y = torch.empty_like(x)
res = torch.randn(B, C, T, device="cuda") if with_res else None
st = _lib.current_stream_ptr(x.device)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
_lib.check(L.amp_set_conv_strip(2)); _lib.check(L.amp_set_conv_strip_steps(steps))
NWG = 16384
buf = torch.zeros(NWG * 4 * 256, dtype=torch.int64, device="cuda")


def go():
    _lib.check(L.amp_conv_forward(h, p(x), B, T, 1.0, p(res), 1.0, p(y), st))


for _ in range(3):
    go()
torch.cuda.synchronize()
fn(ctypes.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
fn(None)
S = buf.cpu().numpy().reshape(NWG, 4, 256)
n = int((S[:, 0, 0] > 0).sum())
S = S[:n]
flop = 2.0 * C * C * k * B * T
print(f"# conv_strip C={C} k={k} d={d} steps={steps} res={with_res} B={B} T={T}: launch {us:.0f} us = {flop / us / 1e6 / (2516.6 / 3):.3f} of the f16x3 peak; {n} workgroups")
life = S[:, :, 59] - S[:, :, 0]
print(f"wave life (0 -> 59): median {np.median(life):.0f} ticks (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
full = S[:, 0, 2] > 0     # workgroups that ran a step 1
F = S[full]
print(f"workgroups with a step 1: {F.shape[0]}")


def med(a):
    return float(np.median(a))


print(f"prologue (0 -> 1): {med(S[:, :, 1] - S[:, :, 0]):.0f}")
nr = 0
while nr < 17 and (F[:, 0, 2 + 3 * nr] > 0).all():
    nr += 1
tot = 0.0
for c in range(nr):
    a0, a1, a2 = F[:, :, 2 + 3 * c], F[:, :, 3 + 3 * c], F[:, :, 4 + 3 * c]
    print(f"step 1 round {c}: MFMA phase {med(a1 - a0):.0f}   barrier wait {med(a2 - a1):.0f}" + (f"   gap to next round {med(F[:, :, 2 + 3 * (c + 1)] - a2):.0f}" if c + 1 < nr else ""))
    tot += med(a2 - a0)
print(f"boundary: last barrier -> exchange {med(F[:, :, 56] - F[:, :, 4 + 3 * (nr - 1)]):.0f}, exchange {med(F[:, :, 57] - F[:, :, 56]):.0f}")
print(f"step 1 total (round 0 start -> exchange done): {med(F[:, :, 57] - F[:, :, 2]):.0f}")
for rr, cname in enumerate(("round 0 (stores)", "round 1", "second-to-last round (residual loads)", "last round (conversions)")):
    c = rr if rr <= 1 else nr - 4 + rr
    if c < 0 or c >= nr or (rr >= 2 and c <= 1):
        continue
    base = 64 + 32 * rr
    hs = []
    hh = 0
    while hh < 32 and (F[:, 0, base + hh] > 0).all():
        hs.append(hh); hh += 1
    if hs:
        prev = F[:, :, 2 + 3 * c]
        out = []
        for hh in hs:
            cur = F[:, :, base + hh]
            out.append(med(cur - prev)); prev = cur
        print(f"{cname} half-taps (ticks each):", " ".join(f"{v:.0f}" for v in out))
print(f"flush (58 -> 59): {med(S[:, :, 59] - S[:, :, 58]):.0f}")
