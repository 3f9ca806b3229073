import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from amphion_amd import _lib
from hip_helpers import conv_forward, pair_forward
def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
_lib.set_precision("f16x3")
for (C, k, d, B, T) in [(64, 3, 5, 2, 1000), (128, 3, 1, 2, 300)]:
    w1 = _rand(C, C, k, seed=1, scale=(C * k) ** -0.5); b1 = _rand(C, seed=2, scale=0.1)
    w2 = _rand(C, C, k, seed=3, scale=(C * k) ** -0.5); b2 = _rand(C, seed=4, scale=0.1); x = _rand(B, C, T, seed=5)
    dt = torch.float64
    xt = F.conv1d(F.leaky_relu(x.to(dt), 0.1), w1.to(dt), b1.to(dt), dilation=d, padding=(k * d - d) // 2)
    y64 = F.conv1d(F.leaky_relu(xt, 0.1), w2.to(dt), b2.to(dt), padding=(k - 1) // 2) + x.to(dt)
    y = pair_forward(w1, b1, w2, b2, x, dilation=d)
    e = (y.double() - y64)
    i = e.abs().argmax().item(); b, c, t = [int(v) for v in torch.unravel_index(torch.tensor(i), e.shape)]
    print(f"C={C} k={k} d={d}: worst at b={b} c={c} t={t}")
    for tt in range(max(0, t - 8), min(T, t + 9)):
        col = e[b, :, tt].abs()
        print(f"   t={tt}: max|e| over ch {col.max().item():.2e}  #ch>1e-6: {(col > 1e-6).sum().item()}  xt64 absmax {xt[b,:,tt].abs().max().item():.3f}  x absmin {x[b,:,tt].abs().min().item():.2e} xt absmin {xt[b,:,tt].abs().min().item():.2e}")
    # repeat to check determinism
    y2 = pair_forward(w1, b1, w2, b2, x, dilation=d)
    print("   deterministic:", torch.equal(y, y2))
    # unfused path via two convs
    xt_h = conv_forward(w1, b1, x, dilation=d, padding=(k*d-d)//2, slope_in=0.1, slope_out=0.1)
    y_h = conv_forward(w2, b2, xt_h, padding=(k-1)//2, res=x)
    print("   unfused f16x3 max err:", (y_h.double() - y64).abs().max().item(), " xt err:", (xt_h.double() - F.leaky_relu(xt, 0.1)).abs().max().item())
