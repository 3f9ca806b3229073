#!/usr/bin/env python
"""A whole ResBlock1 in one rb_f16x3 launch against the same resblock as 2 + 1, 1 + 2 and 1 + 1 + 1 pairs per launch (op-level C ABI,
amp_resblock_forward on sub-lists of the conv handles; each variant looped for ~2 s so that the package sits at its power cap as it does in
the forward).  Behind generator.hip's rb_split() policy: profiles/r5_l_rb_split.txt.  Tuning aid; not part of the product."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amphion_amd import _lib
L = _lib.lib(); _lib.set_precision("f16x3")
_lib.check(L.amp_set_resblock_fusion(2))          # the kernel for every shape it is built for, whatever the grid
st = _lib.current_stream_ptr(torch.device("cuda", 0))
p = lambda t: ctypes.c_void_p(t.data_ptr())


def conv(C, k, dil, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).contiguous(); b = torch.randn(C, generator=g) * 0.1
    h = ctypes.c_void_p()
    _lib.check(L.amp_conv_create(0, C, C, k, 1, dil, (k * dil - dil) // 2, p(w), p(b), ctypes.byref(h)))
    return h


def loop_ms(fn, seconds=2.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(5, int(seconds * 1e3 / e0.elapsed_time(e1)))
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, C, k, T in ((8, 64, 11, 32768), (16, 64, 11, 32768), (32, 64, 11, 32768), (64, 64, 11, 32768), (64, 64, 11, 8192), (64, 64, 7, 32768),
                   (64, 64, 5, 32768), (64, 32, 11, 65536), (64, 32, 7, 65536), (64, 128, 3, 16384)):
    h1 = [conv(C, k, d, 10 + d) for d in (1, 3, 5)]; h2 = [conv(C, k, 1, 20 + d) for d in (1, 3, 5)]
    arr = lambda hs: (ctypes.c_void_p * len(hs))(*[h.value for h in hs])
    x = torch.randn(B, C, T, device="cuda") * 0.5; y = torch.empty_like(x); z = torch.empty_like(x); w = torch.empty_like(x)
    a3, b3 = arr(h1), arr(h2)
    def whole(): _lib.check(L.amp_resblock_forward(a3, b3, 3, p(x), B, T, 0.1, p(y), st))
    def make(parts):
        segs = []; i = 0
        for n in parts:
            segs.append((arr(h1[i:i + n]), arr(h2[i:i + n]), n)); i += n
        def run():
            src = x
            for j, (aa, bb, n) in enumerate(segs):
                dst = y if j == len(segs) - 1 else (z if j % 2 == 0 else w)
                _lib.check(L.amp_resblock_forward(aa, bb, n, p(src), B, T, 0.1, p(dst), st))
                src = dst
        return run
    whole(); torch.cuda.synchronize(); yw = y.clone()
    out = [f"whole {loop_ms(whole):.3f}"]
    for parts in ((2, 1), (1, 2), (1, 1, 1)):
        f = make(parts); f(); torch.cuda.synchronize()
        out.append(f"{'+'.join(map(str, parts))} {loop_ms(f):.3f}{'' if torch.equal(yw, y) else ' (BITS DIFFER)'}")
    out.append(f"whole {loop_ms(whole):.3f}")
    print(f"B={B} T={T} C={C} k={k}: " + ", ".join(out) + " ms", flush=True)
    for h in h1 + h2: L.amp_conv_destroy(h)
