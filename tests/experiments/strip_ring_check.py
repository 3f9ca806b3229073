"""First hardware run of the -DAMP_STRIP_RING experiment library (pair_strip_f16x3.hip: A-fragment ring, 64 x 128-column wave tiles):
bit-for-bit against the per-tile kernel at op level, then tile vs strip timing at the config-2 stage-1 shape in ONE process.
    AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_ring.so python tests/experiments/strip_ring_check.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from amphion_amd import _lib
from hip_helpers import pair_forward

_lib.set_precision("f16x3")
L = _lib.lib()
print("library:", os.environ.get("AMP_LIB_PATH", "default"), flush=True)


def rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


ok = True
for C, k, d, B, T, steps in ((128, 11, 5, 64, 2100, "1"), (128, 7, 3, 64, 4100, "2"), (128, 11, 1, 48, 6000, "3")):
    w1, b1 = rnd(C, C, k, seed=1, scale=(C * k) ** -0.5), rnd(C, seed=2, scale=0.1)
    w2, b2 = rnd(C, C, k, seed=3, scale=(C * k) ** -0.5), rnd(C, seed=4, scale=0.1)
    x = rnd(B, C, T, seed=5)
    L.amp_set_pair_strips(0)
    y_tile = pair_forward(w1, b1, w2, b2, x, dilation=d)
    L.amp_set_pair_strips(-1)
    os.environ["AMP_STRIP_STEPS"] = steps
    y_pol = pair_forward(w1, b1, w2, b2, x, dilation=d)
    same = torch.equal(y_tile, y_pol) and bool(torch.isfinite(y_pol).all())
    ok &= same
    print(f"parity C={C} k={k} d={d} B={B} T={T} steps={steps}: {'bitwise equal' if same else 'MISMATCH max ' + str((y_tile - y_pol).abs().max().item())}", flush=True)

st = _lib.current_stream_ptr(torch.device("cuda", 0))
B, C, T = 64, 128, 16384
x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x)
for k, d in ((11, 5), (7, 3)):
    hs = []
    g = torch.Generator().manual_seed(1)
    for dd in (d, 1):
        w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous(); b = torch.randn(C, generator=g) * 0.1
        h = ctypes.c_void_p()
        _lib.check(L.amp_conv_create(0, C, C, k, 1, dd, (k * dd - dd) // 2, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
        hs.append(h)
    for mode, steps in ((0, ""), (-1, "1"), (-1, "2"), (-1, "3")):
        L.amp_set_pair_strips(mode)
        if steps: os.environ["AMP_STRIP_STEPS"] = steps
        go = lambda: L.amp_pair_forward(hs[0], hs[1], ctypes.c_void_p(x.data_ptr()), B, T, 0.1, ctypes.c_void_p(y.data_ptr()), st)
        go(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        print(f"time k={k} d={d} {'tile kernel' if mode == 0 else 'policy strips, steps=' + steps}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    for h in hs: L.amp_conv_destroy(h)
print("PARITY", "OK" if ok else "FAILED")
