"""CPU emulation of the split-f16 (3 x f16 MFMA, f32 accumulate) operand rounding, end to end through
HiFi-GAN V1, against an fp64 run of the oracle.  Design-time experiment for conv_f16x3.hip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import synth, vocoder_oracle as vo

XS = 16.0

def split(x, scale, rtz=False):
    xs = (x * scale).float()
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi.double(), lo.double()

def wscale(w):
    m = w.abs().max().item()
    import math
    return 2.0 ** (13 - math.ceil(math.log2(m)))

mode = {"on": False, "terms": 3}
_c1, _ct = F.conv1d, F.conv_transpose1d

def conv1d(x, w, b=None, **kw):
    if not mode["on"]:
        return _c1(x, w, b, **kw)
    S = wscale(w)
    xh, xl = split(x, XS); wh, wl = split(w, S)
    y = _c1(xh, wh, None, **kw)
    if mode["terms"] in (3, "w11"):
        y = y + _c1(xl, wh, None, **kw)      # Whi * Xlo
    if mode["terms"] in (3, "x11"):
        y = y + _c1(xh, wl, None, **kw)      # Wlo * Xhi
    y = (y / (S * XS)).float()
    if b is not None: y = y + b.view(1, -1, 1)
    return y

def convt(x, w, b=None, **kw):
    if not mode["on"]:
        return _ct(x, w, b, **kw)
    S = wscale(w)
    xh, xl = split(x, XS); wh, wl = split(w, S)
    y = _ct(xh, wh, None, **kw)
    if mode["terms"] in (3, "w11"):
        y = y + _ct(xl, wh, None, **kw)
    if mode["terms"] in (3, "x11"):
        y = y + _ct(xh, wl, None, **kw)
    y = (y / (S * XS)).float()
    if b is not None: y = y + b.view(1, -1, 1)
    return y

F.conv1d, F.conv_transpose1d = conv1d, convt
hp = vo.hifigan_v1_hp()
sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
mel = synth.synth_mel(2, 80, 48, seed=3)
with torch.no_grad():
    ref64 = vo.hifigan_forward(sd, hp, mel, dtype=torch.float64)
    ref32 = vo.hifigan_forward(sd, hp, mel)
    mode["on"] = True
    y3 = vo.hifigan_forward(sd, hp, mel)
    mode["terms"] = "w11"      # weights rounded to f16 (11 bits), activations split: 2 MFMAs per term
    yw = vo.hifigan_forward(sd, hp, mel)
    mode["terms"] = "x11"      # activations rounded to f16, weights split: 2 MFMAs per term
    yx = vo.hifigan_forward(sd, hp, mel)
    mode["terms"] = 1
    y1 = vo.hifigan_forward(sd, hp, mel)
print("out absmax", ref64.abs().max().item())
print("fp32 oracle vs fp64:", (ref32.double() - ref64).abs().max().item())
print("f16x3       vs fp64:", (y3.double() - ref64).abs().max().item())
print("f16x2 (W 11b) vs fp64:", (yw.double() - ref64).abs().max().item())
print("f16x2 (X 11b) vs fp64:", (yx.double() - ref64).abs().max().item())
print("f16x1       vs fp64:", (y1.double() - ref64).abs().max().item())
