"""Design-time model for VERDICT r4 item 1: Activation1d's two 12-tap FIRs as split-f16 Toeplitz products on the matrix pipe
inside ampb_f16x3 (modules/anti_aliasing/resample.py:36-65, filter.py:92-99).

  (1) numerics: BigVGAN-base end to end (oracle, CPU) with the convs AND the FIRs in the f16x3 form
      (x = hi + lo f16 after an exact x16, taps = hi + lo f16, products hi*hi + lo*hi + hi*lo, fp32 accumulate) against an fp64 run;
  (2) VALU issue slots per step and wave of the kernel, now vs with the FIRs as MFMAs (counts from ampb_f16x3.hip's act_run and the
      packed-op issue cost measured in round 4: v_pk_*_f32 = 2 slots, v_sin_f32 = 4).

    python tests/experiments/ampb_fir_mfma_model.py
Kill criterion (VERDICT): modelled VALU slots fall by < 25 %, or end-to-end error vs fp64 > 1e-5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import synth, vocoder_oracle as vo

XS = 16.0


def split(x):
    hi = x.float().half()
    lo = (x.float() - hi.float()).half()
    return hi.double(), lo.double()


def wscale(w):
    import math
    return 2.0 ** (13 - math.ceil(math.log2(w.abs().max().item())))


mode = {"conv": False, "fir": False}
_c1, _ct = F.conv1d, F.conv_transpose1d


def conv1d(x, w, b=None, **kw):
    if not mode["conv"] or kw.get("groups", 1) != 1:
        return _c1(x, w, b, **kw)
    S = wscale(w)
    xh, xl = split(x * XS); wh, wl = split(w * S)
    y = _c1(xh, wh, None, **kw) + _c1(xl, wh, None, **kw) + _c1(xh, wl, None, **kw)
    y = (y / (S * XS)).float()
    return y if b is None else y + b.view(1, -1, 1)


def convt(x, w, b=None, **kw):
    if not mode["conv"] or kw.get("groups", 1) != 1:
        return _ct(x, w, b, **kw)
    S = wscale(w)
    xh, xl = split(x * XS); wh, wl = split(w * S)
    y = _ct(xh, wh, None, **kw) + _ct(xl, wh, None, **kw) + _ct(xh, wl, None, **kw)
    y = (y / (S * XS)).float()
    return y if b is None else y + b.view(1, -1, 1)


_act = vo.activation1d


def activation1d(x, alpha, beta=None, logscale=False, filt_up=None, filt_down=None):
    if not mode["fir"] or x.dtype != torch.float32:
        return _act(x, alpha, beta, logscale, filt_up, filt_down)
    C = x.shape[1]
    fu2 = (2.0 * filt_up.float()).reshape(1, 1, 12)          # the kernel's doubled taps (exact)
    fd = filt_down.float().reshape(1, 1, 12)
    fuh, ful = split(fu2); fdh, fdl = split(fd)
    e = lambda f: f.expand(C, -1, -1)
    xp = F.pad(x, (5, 5), mode="replicate") * XS             # everything inside the block is 16 x true (exact)
    xh, xl = split(xp)
    up = lambda a, f: _ct(a, e(f), stride=2, groups=C)
    u = (up(xh, fuh) + up(xl, fuh) + up(xh, ful))[..., 15:-15].float()          # fp32 accumulator of the MFMA
    a = alpha.reshape(1, -1, 1).float(); b = a if beta is None else beta.reshape(1, -1, 1).float()
    if logscale:
        a = torch.exp(a); b = torch.exp(b) if beta is not None else a
    invb16 = (1.0 / (b + 1e-9)) * XS
    s = u + invb16 * torch.sin(u * (a / XS)) ** 2                               # 16 x snake, fp32
    sp = F.pad(s, (5, 6), mode="replicate")
    sh, sl = split(sp)
    dn = lambda a_, f: _c1(a_, e(f), stride=2, groups=C)
    y = (dn(sh, fdh) + dn(sl, fdh) + dn(sh, fdl)).float()
    return y / XS


F.conv1d, F.conv_transpose1d = conv1d, convt
vo.activation1d = activation1d


def numerics():
    hp = vo.bigvgan_base_hp()
    sd = synth.synth_state_dict(synth.bigvgan_param_shapes(100, hp), 1234, g_gain=0.75) if hasattr(synth, "bigvgan_param_shapes") else None
    return hp, sd


def slots():
    # per step and wave (64 columns per lane); packed fp32 = 2 issue slots, v_sin_f32 = 4 (quarter rate), everything else 1
    fir_now = 72 * 6 * 2 + 64 * 6 * 2 + 64                     # up: 72 pairs x 6 pk_fma; down: 64 x 6 pk_fma + the final add
    snake = 72 * (5 * 2 + 2 + 2 * 4)                           # per pair: 5 packed + 2 rint + 2 sines
    now_total = 1150 * 2 + 1350 + 144 * 3                      # ~1150 packed + ~1350 other per step (+3 extra slots per sine)
    split_z = 80 * 1.5                                         # 10 K blocks of x: cvt_pk + 2 mix per pair
    split_s = 144 * 1.5
    guards = 32 + 72                                           # v_max3 on |x16| and |s16| (the f16 range flag)
    saved_scale = 64                                           # the scatter's x16 multiplies disappear (values stay 16 x inside the block)
    new_total = now_total - fir_now + split_z + split_s + guards - saved_scale
    mfma_fir = 9 * 2 * 3 + 4 * 6 * 3
    print(f"VALU issue slots per step and wave: now ~{now_total} (FIRs {fir_now}, Snake {snake}); with the FIRs on MFMA ~{new_total:.0f} "
          f"({100 * (1 - new_total / now_total):.0f} % fewer); +{mfma_fir} MFMAs (32x32x16, 8 passes) = {mfma_fir * 32} matrix-pipe cycles beside "
          f"72 / 168 / 264 conv MFMAs at C = 32, k = 3 / 7 / 11")


if __name__ == "__main__":
    slots()
    torch.manual_seed(0)
    hp = vo.bigvgan_base_hp()
    shapes = synth.bigvgan_param_shapes(100, hp)
    sd = synth.synth_state_dict(shapes, 1234, g_gain=0.75)
    mel = torch.randn(2, 100, 24, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref64 = vo.bigvgan_forward(sd, hp, mel, dtype=torch.float64)
        ref32 = vo.bigvgan_forward(sd, hp, mel)
        mode["conv"] = True
        yc = vo.bigvgan_forward(sd, hp, mel)
        mode["fir"] = True
        ycf = vo.bigvgan_forward(sd, hp, mel)
        mode["conv"] = False
        yf = vo.bigvgan_forward(sd, hp, mel)
    err = lambda y: (y.double() - ref64).abs().max().item()
    print("out absmax %.3f" % ref64.abs().max().item())
    print("fp32 oracle            vs fp64: %.2e" % err(ref32))
    print("convs f16x3            vs fp64: %.2e" % err(yc))
    print("convs + FIRs f16x3     vs fp64: %.2e" % err(ycf))
    print("FIRs f16x3 only        vs fp64: %.2e" % err(yf))
