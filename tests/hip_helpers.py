"""ctypes-level helpers for the GPU parity tests (call the C ABI with raw device pointers)."""
import ctypes

import torch

from amphion_amd import _lib


def conv_forward(w, b, x, *, transposed=False, stride=1, dilation=1, padding=0, slope_in=1.0, res=None,
                 slope_out=1.0):
    """Run amp_conv_* on cuda:0.  w/b are CPU tensors (folded weight), x a CPU tensor [B, Cin, T]."""
    L = _lib.lib()
    h = ctypes.c_void_p()
    w = w.contiguous().float()
    cin = w.shape[0] if transposed else w.shape[1]
    cout = w.shape[1] if transposed else w.shape[0]
    bptr = ctypes.c_void_p(b.contiguous().float().data_ptr()) if b is not None else None
    _lib.check(L.amp_conv_create(int(transposed), cin, cout, w.shape[2], stride, dilation, padding,
                                 ctypes.c_void_p(w.data_ptr()), bptr, ctypes.byref(h)))
    try:
        xd = x.contiguous().float().cuda()
        B, _, T = xd.shape
        Tout = L.amp_conv_out_len(h, T)
        y = torch.full((B, cout, Tout), float("nan"), device="cuda")
        rd = res.contiguous().float().cuda() if res is not None else None
        _lib.check(L.amp_conv_forward(h, ctypes.c_void_p(xd.data_ptr()), B, T, slope_in,
                                      ctypes.c_void_p(rd.data_ptr()) if rd is not None else None, slope_out,
                                      ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(xd.device)))
        torch.cuda.synchronize()
        return y.cpu()
    finally:
        L.amp_conv_destroy(h)


def pair_forward(w1, b1, w2, b2, x, *, dilation, slope=0.1):
    """amp_pair_forward on cuda:0: y = x + c2(lrelu(c1(lrelu(x))))  (fused ResBlock1 pair)."""
    L = _lib.lib()
    C, _, k = w1.shape
    hs = []
    try:
        for w, b, d in ((w1, b1, dilation), (w2, b2, 1)):
            h = ctypes.c_void_p()
            w = w.contiguous().float()
            b = b.contiguous().float()
            _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()),
                                         ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
            hs.append(h)
        xd = x.contiguous().float().cuda()
        B, _, T = xd.shape
        y = torch.full_like(xd, float("nan"))
        _lib.check(L.amp_pair_forward(hs[0], hs[1], ctypes.c_void_p(xd.data_ptr()), B, T, slope,
                                      ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(xd.device)))
        torch.cuda.synchronize()
        return y.cpu()
    finally:
        for h in hs:
            L.amp_conv_destroy(h)


def resblock_forward(ws1, bs1, ws2, bs2, x, *, dilations, slope=0.1, fused=True):
    """ResBlock1 on cuda:0 from per-pair weights: fused=True -> amp_resblock_forward (ONE launch, csrc/rb_f16x3.hip),
    fused=False -> len(dilations) x amp_pair_forward (the fused pairs)."""
    L = _lib.lib()
    C, _, k = ws1[0].shape
    n = len(dilations)
    h1, h2 = [], []
    try:
        for ws, bs, hs, ds in ((ws1, bs1, h1, dilations), (ws2, bs2, h2, [1] * n)):
            for w, b, d in zip(ws, bs, ds):
                h = ctypes.c_void_p()
                w, b = w.contiguous().float(), b.contiguous().float()
                _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, ctypes.c_void_p(w.data_ptr()),
                                             ctypes.c_void_p(b.data_ptr()), ctypes.byref(h)))
                hs.append(h)
        xd = x.contiguous().float().cuda()
        B, _, T = xd.shape
        st = _lib.current_stream_ptr(xd.device)
        if fused:
            y = torch.full_like(xd, float("nan"))
            a1, a2 = (ctypes.c_void_p * n)(*[h.value for h in h1]), (ctypes.c_void_p * n)(*[h.value for h in h2])
            _lib.check(L.amp_resblock_forward(a1, a2, n, ctypes.c_void_p(xd.data_ptr()), B, T, slope, ctypes.c_void_p(y.data_ptr()), st))
        else:
            cur = xd
            for p in range(n):
                y = torch.full_like(xd, float("nan"))
                _lib.check(L.amp_pair_forward(h1[p], h2[p], ctypes.c_void_p(cur.data_ptr()), B, T, slope, ctypes.c_void_p(y.data_ptr()), st))
                cur = y
        torch.cuda.synchronize()
        return y.cpu()
    finally:
        for h in h1 + h2:
            L.amp_conv_destroy(h)


def act1d_forward(x, alpha, beta, logscale, fu, fd):
    L = _lib.lib()
    xd = x.contiguous().float().cuda()
    B, C, T = xd.shape
    y = torch.full_like(xd, float("nan"))
    ad = alpha.contiguous().float().cuda()
    bd = beta.contiguous().float().cuda() if beta is not None else None
    fu = fu.contiguous().float()
    fd = fd.contiguous().float()
    _lib.check(L.amp_antialias_snake(ctypes.c_void_p(xd.data_ptr()), B, C, T, ctypes.c_void_p(ad.data_ptr()),
                                     ctypes.c_void_p(bd.data_ptr()) if bd is not None else None, int(logscale),
                                     ctypes.c_void_p(fu.data_ptr()), ctypes.c_void_p(fd.data_ptr()),
                                     ctypes.c_void_p(y.data_ptr()), _lib.current_stream_ptr(xd.device)))
    torch.cuda.synchronize()
    return y.cpu()


def ampblock_forward(ws1, bs1, ws2, bs2, alphas, betas, logscale, fu, fd, x, *, dilations, fused=True, mode=0, div=1.0, y0=None):
    """AMPBlock1 (bigvgan.py:137-146) on cuda:0 from per-pair weights and per-activation parameters (``alphas`` / ``betas``
    [2 * n, C]; ``betas`` None -> Snake).  fused=True -> ``amp_ampblock_forward`` (ONE launch, csrc/ampb_f16x3.hip);
    fused=False -> the launches it replaces: ``amp_antialias_snake`` -> ``amp_conv_forward`` -> ``amp_antialias_snake`` ->
    ``amp_conv_forward`` (with the residual; the last conv through ``amp_conv_forward_mrf`` with the MRF ``mode``, bigvgan.py:320-327).
    ``y0``: the running MRF sum for modes 1 / 2.  CPU tensors in and out."""
    L = _lib.lib()
    C, _, k = ws1[0].shape
    n = len(dilations)
    h1, h2 = [], []
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    try:
        for ws, bs, hs, ds in ((ws1, bs1, h1, dilations), (ws2, bs2, h2, [1] * n)):
            for w, b, d in zip(ws, bs, ds):
                h = ctypes.c_void_p()
                w, b = w.contiguous().float(), b.contiguous().float()
                _lib.check(L.amp_conv_create(0, C, C, k, 1, d, (k * d - d) // 2, p(w), p(b), ctypes.byref(h)))
                hs.append(h)
        xd = x.contiguous().float().cuda()
        B, _, T = xd.shape
        st = _lib.current_stream_ptr(xd.device)
        ad = alphas.contiguous().float().cuda()
        bd = betas.contiguous().float().cuda() if betas is not None else None
        fu, fd = fu.contiguous().float(), fd.contiguous().float()
        y = y0.contiguous().float().cuda().clone() if y0 is not None else torch.full_like(xd, float("nan"))
        if fused:
            a1, a2 = (ctypes.c_void_p * n)(*[h.value for h in h1]), (ctypes.c_void_p * n)(*[h.value for h in h2])
            _lib.check(L.amp_ampblock_forward(a1, a2, n, p(ad), p(bd), int(logscale), p(fu), p(fd), p(xd), B, T, p(y), mode, div, st))
        else:
            act = torch.empty_like(xd)
            xt = torch.empty_like(xd)
            cur = xd
            for i in range(n):
                for j, h in ((0, h1[i]), (1, h2[i])):
                    s = 2 * i + j
                    src = cur if j == 0 else xt
                    _lib.check(L.amp_antialias_snake(p(src), B, C, T, p(ad[s]), p(bd[s]) if bd is not None else None, int(logscale),
                                                     p(fu), p(fd), p(act), st))
                    if j == 0:
                        _lib.check(L.amp_conv_forward(h, p(act), B, T, 1.0, None, 1.0, p(xt), st))
                    elif i + 1 < n:
                        nxt = torch.empty_like(xd)
                        _lib.check(L.amp_conv_forward(h, p(act), B, T, 1.0, p(cur), 1.0, p(nxt), st))
                        cur = nxt
                    else:
                        _lib.check(L.amp_conv_forward_mrf(h, p(act), B, T, 1.0, p(cur), p(y), mode, div, st))
        torch.cuda.synchronize()
        return y.cpu()
    finally:
        for h in h1 + h2:
            L.amp_conv_destroy(h)
