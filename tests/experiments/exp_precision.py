"""GPU experiment: where does the f16x3 conv's error come from? (single conv + fused pair, vs fp64)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from amphion_amd import _lib
from hip_helpers import conv_forward, pair_forward

def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale

def stats(name, y, ref, ref32=None):
    e = (y.double() - ref)
    i = e.abs().argmax().item()
    idx = torch.unravel_index(torch.tensor(i), e.shape)
    msg = f"{name}: max {e.abs().max().item():.2e} mean|e| {e.abs().mean().item():.2e} mean(e) {e.mean().item():+.2e} at {[int(v) for v in idx]} ref there {ref.flatten()[i].item():+.3f}"
    if ref32 is not None:
        e2 = ref32.double() - ref
        msg += f" | torch32: max {e2.abs().max().item():.2e} mean|e| {e2.abs().mean().item():.2e}"
    print(msg)
    # error by column, top 5 columns
    ec = e.abs().amax(dim=(0, 1))
    top = torch.topk(ec, min(8, ec.numel()))
    print("   worst columns:", [(int(c), f"{v:.1e}") for v, c in zip(top.values, top.indices)])

for prec in ("f16x3", "f32"):
    _lib.set_precision(prec)
    for (C, k, d, B, T) in [(64, 3, 5, 2, 1000), (64, 7, 1, 1, 129), (128, 11, 5, 1, 300), (32, 3, 3, 1, 2050)]:
        w = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5); b = _rand(C, seed=2, scale=0.1); x = _rand(B, C, T, seed=5)
        pad = (k * d - d) // 2
        ref = F.conv1d(x.double(), w.double(), b.double(), dilation=d, padding=pad)
        ref32 = F.conv1d(x, w, b, dilation=d, padding=pad)
        stats(f"[{prec}] conv C={C} k={k} d={d}", conv_forward(w, b, x, dilation=d, padding=pad), ref, ref32)
        # no bias, x scaled
        for sc in (1.0, 2.0 ** -6, 2.0 ** 6):
            ref = F.conv1d(x.double() * sc, w.double(), None, dilation=d, padding=pad)
            y = conv_forward(w, None, x * sc, dilation=d, padding=pad)
            print(f"     xscale {sc:g}: rel err {(y.double() - ref).abs().max().item() / ref.abs().max().item():.2e}")
_lib.set_precision("f16x3")
for (C, k, d, B, T) in [(64, 3, 5, 2, 1000), (32, 3, 3, 1, 2050), (128, 3, 1, 2, 300)]:
    w1 = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5); b1 = _rand(C, seed=2, scale=0.1)
    w2 = _rand(C, C, k, seed=3, scale=(C * k) ** -0.5); b2 = _rand(C, seed=4, scale=0.1); x = _rand(B, C, T, seed=5)
    def ref(dt):
        xt = F.conv1d(F.leaky_relu(x.to(dt), 0.1), w1.to(dt), b1.to(dt), dilation=d, padding=(k * d - d) // 2)
        xt = F.conv1d(F.leaky_relu(xt, 0.1), w2.to(dt), b2.to(dt), padding=(k - 1) // 2); return xt + x.to(dt)
    stats(f"pair C={C} k={k} d={d}", pair_forward(w1, b1, w2, b2, x, dilation=d), ref(torch.float64), ref(torch.float32))
