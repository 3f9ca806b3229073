"""Thin torch-tensor wrappers over the op-level C ABI (conv + VITS element-wise kernels)."""
from __future__ import annotations

import ctypes
import weakref

import torch

from amphion_amd import _lib
from amphion_amd.models.vocoders.gan.generator._engine import ConvParams


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _destroy_conv(ptr):
    try:
        _lib.lib().amp_conv_destroy(ctypes.c_void_p(ptr))
    except Exception:
        pass


class HipConv1d(ConvParams):
    """A Conv1d / ConvTranspose1d whose forward is the fused gfx950 implicit-GEMM kernel:
    ``y = lrelu_out(conv(lrelu_in(x)) + bias + res)``.  Parameters keep torch's key names
    (``weight``/``bias`` or ``weight_g``/``weight_v`` under weight-norm); the packed device copy is
    rebuilt lazily when they change."""

    def __init__(self, *a, pad_mode="zeros", tanh=False, **k):
        super().__init__(*a, **k)
        if pad_mode not in ("zeros", "reflect"):
            raise ValueError(f"pad_mode must be 'zeros' or 'reflect', got {pad_mode!r}")
        self.pad_mode, self.tanh = pad_mode, bool(tanh)
        self._h = None
        self._fin = None
        self._sig = None

    def _param_sig(self):
        """identity + version of every parameter (a leaf module: its own ``_parameters`` only -- ``named_parameters()`` walks the
        module tree with prefixes and costs several microseconds per launch)"""
        return tuple((n, p.data_ptr(), p._version) for n, p in self._parameters.items() if p is not None)

    def _ensure(self, device):
        sig = self._param_sig() + (device,)
        if self._h is not None and sig == self._sig:
            return self._h
        if self._fin is not None:
            self._fin()
        w = self.folded_weight().detach().to("cpu", torch.float32).contiguous()
        b = self.bias.detach().to("cpu", torch.float32).contiguous() if self.bias is not None else None
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().amp_conv_create(int(self.transposed), self.cin, self.cout, self.k, self.stride,
                                                  self.dilation, self.padding, _ptr(w), _ptr(b), ctypes.byref(h)))
            if self.pad_mode == "reflect":
                _lib.check(_lib.lib().amp_conv_set_option(h, _lib.AMP_CONV_OPT_PAD_REFLECT, 1))
            if self.tanh:
                _lib.check(_lib.lib().amp_conv_set_option(h, _lib.AMP_CONV_OPT_TANH, 1))
        self._h, self._fin, self._sig = h, weakref.finalize(self, _destroy_conv, h.value), sig
        self._prec = _lib.get_precision()      # a handle keeps the arithmetic it was built with (amphion_hip.h)
        return h

    def _ensure_gated(self, device):
        """A second handle with the 2H rows packed for the gate epilogue of the fused WN layer
        (``amp_conv_create_gated``); None when this conv / arithmetic is outside what that kernel covers."""
        sig = self._param_sig() + (device, _lib.get_precision())
        if getattr(self, "_sig_g", None) == sig:
            return self._hg
        if getattr(self, "_fin_g", None) is not None:
            self._fin_g()
        self._hg, self._fin_g, self._sig_g = None, None, sig
        if (_lib.get_precision() != "f16x3" or self.transposed or self.cout != 2 * self.cin or self.cin % 32 or self.cin > 256 or self.cout <= 64
                or self.k not in (1, 3, 5) or self.stride != 1 or 2 * self.padding != self.dilation * (self.k - 1)
                or (self.k - 1) * self.dilation > 64 or self.pad_mode != "zeros" or self.tanh):
            return None
        w = self.folded_weight().detach().to("cpu", torch.float32).contiguous()
        b = self.bias.detach().to("cpu", torch.float32).contiguous() if self.bias is not None else None
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().amp_conv_create_gated(self.cin, self.k, self.dilation, self.padding, _ptr(w), _ptr(b), ctypes.byref(h)))
        self._hg, self._fin_g = h, weakref.finalize(self, _destroy_conv, h.value)
        return h

    def forward(self, x, *, slope_in=1.0, res=None, slope_out=1.0, out=None, x_batch_stride=None, T=None, lens=None):
        """x [B, cin, T] (or a channel slice of a wider tensor when ``x_batch_stride`` is given).  ``lens`` (int32 [B] on the
        device): conv(x * mask) with the mask taken by the kernel -- output columns beyond an item's length are UNSPECIFIED
        (``amp_conv_forward_ragged``)."""
        x = _lib.require_device_tensor(x, "conv input") if x_batch_stride is None else x
        dev = x.device
        B = x.shape[0]
        T = x.shape[-1] if T is None else T
        L = _lib.lib()
        h = self._ensure(dev)
        # amp_conv_out_len, without the call
        Tout = (T - 1) * self.stride - 2 * self.padding + self.k if self.transposed else T + 2 * self.padding - self.dilation * (self.k - 1)
        if out is None:
            out = torch.empty((B, self.cout, Tout), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            if lens is not None:
                _lib.check(L.amp_conv_forward_ragged(h, _ptr(x), int(x_batch_stride or 0), B, T, _ptr(lens), slope_in, _ptr(res),
                                                     slope_out, _ptr(out), _lib.current_stream_ptr(dev)))
            elif x_batch_stride is None:
                _lib.check(L.amp_conv_forward(h, _ptr(x), B, T, slope_in, _ptr(res), slope_out, _ptr(out),
                                              _lib.current_stream_ptr(dev)))
            else:
                _lib.check(L.amp_conv_forward_strided(h, _ptr(x), int(x_batch_stride), B, T, slope_in, _ptr(res),
                                                      slope_out, _ptr(out), _lib.current_stream_ptr(dev)))
        return out


class MergedConv1d:
    """Several ``HipConv1d`` that read the same input (the q / k / v projections of an attention layer), run as ONE launch: their
    folded weights stacked along the output rows in one handle -> [B, sum(cout), T].  Not a module and it owns no parameters: the
    convs are passed at every call, this object only caches the packed device copy (rebuilt when a parameter changes; a copy /
    pickle of the owning module starts with an empty cache)."""

    def __init__(self):
        self._h = self._fin = self._sig = None

    def __deepcopy__(self, memo):
        return MergedConv1d()

    def __reduce__(self):
        return (MergedConv1d, ())

    def _ensure(self, convs, device):
        sig = tuple(c._param_sig() for c in convs) + (device,)
        if self._h is not None and sig == self._sig:
            return self._h
        c0 = convs[0]
        if any(c.transposed or (c.cin, c.k, c.stride, c.dilation, c.padding) != (c0.cin, c0.k, c0.stride, c0.dilation, c0.padding)
               or (c.bias is None) != (c0.bias is None) for c in convs):
            raise ValueError("MergedConv1d: the convs must be Conv1d of one geometry")
        if self._fin is not None:
            self._fin()
        w = torch.cat([c.folded_weight().detach().to("cpu", torch.float32) for c in convs], dim=0).contiguous()
        b = torch.cat([c.bias.detach().to("cpu", torch.float32) for c in convs]).contiguous() if c0.bias is not None else None
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().amp_conv_create(0, c0.cin, w.shape[0], c0.k, c0.stride, c0.dilation, c0.padding, _ptr(w), _ptr(b),
                                                  ctypes.byref(h)))
        self._h, self._fin, self._sig = h, weakref.finalize(self, _destroy_conv, h.value), sig
        return h

    def __call__(self, convs, x, lens=None):
        x = _lib.require_device_tensor(x, "conv input")
        B, _, T = x.shape
        h = self._ensure(convs, x.device)
        out = torch.empty((B, sum(c.cout for c in convs), T), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().amp_conv_forward_ragged(h, _ptr(x), 0, B, T, _ptr(lens), 1.0, None, 1.0, _ptr(out),
                                                          _lib.current_stream_ptr(x.device)))
        return out


def lens_tensor(lengths, device):
    """int32 device tensor of valid lengths (sequence_mask argument), or None."""
    if lengths is None:
        return None
    return torch.as_tensor(lengths).to(device=device, dtype=torch.int32).contiguous()


def wn_fused(in_layers, res_skip_layers, x, cond, lens, out, acts):
    """The whole WN stack on the fused kernels (``amp_wn_forward``): two launches per layer.  ``x`` is modified.
    Returns False (nothing launched) when a layer is outside what the fused kernels cover -- the caller then runs the
    unfused ops."""
    dev = x.device
    n = len(in_layers)
    H = x.shape[1]
    hi, hr = [], []
    for i in range(n):
        rs = res_skip_layers[i]
        if (rs.k != 1 or rs.cin != H or rs.cout != (2 * H if i < n - 1 else H) or rs.cout <= 64 or rs.bias is None or rs.transposed or rs.tanh
                or rs.pad_mode != "zeros"):      # cout <= 64: fewer than the 128 rows the whole-K kernel's workgroups cover (H <= 64)
            return False
        g = in_layers[i]._ensure_gated(dev)
        if g is None:
            return False
        hr_i = rs._ensure(dev)
        if getattr(rs, "_prec", None) != "f16x3":       # a res_skip handle built while the exact-fp32 mode was selected
            return False
        hi.append(g.value)
        hr.append(hr_i.value)
    B, _, T = x.shape
    arr_i = (ctypes.c_void_p * n)(*hi)
    arr_r = (ctypes.c_void_p * n)(*hr)
    bs = cond.stride(0) if cond is not None else 0
    with _lib.on_device(dev):
        _lib.check(_lib.lib().amp_wn_forward(arr_i, arr_r, n, _ptr(x), _ptr(cond), bs, _ptr(lens), B, T, _ptr(acts), _ptr(out),
                                             _lib.current_stream_ptr(dev)))
    return True


def wn_gate(a, cond, out):
    B, H2, T = a.shape
    bs = cond.stride(0) if cond is not None else 0
    _lib.check(_lib.lib().amp_wn_gate(_ptr(a), _ptr(cond), bs, _ptr(out), B, H2 // 2, T, _lib.current_stream_ptr(a.device)))
    return out


def wn_accumulate(x, out, rs, lens, first, last):
    B, H, T = x.shape
    _lib.check(_lib.lib().amp_wn_accumulate(_ptr(x), _ptr(out), _ptr(rs), _ptr(lens), B, H, T, int(first), int(last),
                                            _lib.current_stream_ptr(x.device)))


def sequence_mask_(x, lens):
    B, C, T = x.shape
    _lib.check(_lib.lib().amp_sequence_mask(_ptr(x), _ptr(lens), B, C, T, _lib.current_stream_ptr(x.device)))
    return x


def coupling_apply_(x, m, lens, reverse):
    B, C, T = x.shape
    _lib.check(_lib.lib().amp_coupling_apply(_ptr(x), _ptr(m), _ptr(lens), B, C // 2, T, int(reverse),
                                             _lib.current_stream_ptr(x.device)))
    return x


def flip_channels(x):
    B, C, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().amp_flip_channels(_ptr(x), _ptr(y), B, C, T, _lib.current_stream_ptr(x.device)))
    return y


def posterior_sample(stats, eps, lens):
    B, C2, T = stats.shape
    z = torch.empty((B, C2 // 2, T), dtype=torch.float32, device=stats.device)
    _lib.check(_lib.lib().amp_posterior_sample(_ptr(stats), _ptr(eps), _ptr(lens), _ptr(z), B, C2 // 2, T,
                                               _lib.current_stream_ptr(stats.device)))
    return z


# ---- frame-rate ops of VITS' text -> duration -> alignment front ----------------
def _stream(t):
    return _lib.current_stream_ptr(t.device)


def layer_norm_c(x, gamma, beta, res=None, post=None, eps=1e-5, gelu=False, lens=None):
    """post + act(LN(x + res)) over the channel axis; with ``lens`` the columns beyond an item's length are written as zero
    (whatever x / res hold there)"""
    B, C, T = x.shape
    y = torch.empty_like(x)
    if lens is None:
        _lib.check(_lib.lib().amp_layer_norm_c(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(post), B, C, T, float(eps), int(gelu),
                                               _ptr(y), _stream(x)))
    else:
        _lib.check(_lib.lib().amp_layer_norm_c_ragged(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(post), _ptr(lens), B, C, T,
                                                      float(eps), int(gelu), _ptr(y), _stream(x)))
    return y


def dwconv_layer_norm_c(x, weight, bias, dilation, gamma, beta, lens=None, eps=1e-5, gelu=False):
    """act(LN(dwconv(x * mask))) in one launch (DDSConv's convs_sep + norms_1, modules/flow/modules.py:63-65); K = 3"""
    B, C, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().amp_dwconv_layer_norm_c(_ptr(x), _ptr(weight), _ptr(bias), weight.shape[-1], int(dilation), _ptr(gamma),
                                                  _ptr(beta), _ptr(lens), B, C, T, float(eps), int(gelu), _ptr(y), _stream(x)))
    return y


def dds_seam(y, x, n2, sep, n1, lens=None):
    """The seam between DDSConv layers i and i + 1 in one launch: ``x_new = x + gelu(n2(y))``; ``z = gelu(n1(sep(x_new * mask)))``
    -> (x_new, z), or None when the shape is outside the fused kernel (depthwise K = 3, C <= 192)."""
    B, C, T = y.shape
    w = sep.weight.detach()
    if w.shape[-1] != 3 or C > 192:
        return None
    xo, zo = torch.empty_like(y), torch.empty_like(y)
    _lib.check(_lib.lib().amp_dds_seam(_ptr(y), _ptr(x), _ptr(n2.gamma.detach()), _ptr(n2.beta.detach()), float(n2.eps), _ptr(w.contiguous()),
                                       _ptr(sep.bias.detach()), 3, int(sep.dilation), _ptr(n1.gamma.detach()), _ptr(n1.beta.detach()),
                                       float(n1.eps), _ptr(lens), B, C, T, _ptr(xo), _ptr(zo), _stream(y)))
    return xo, zo


def add_channel_bias_(x, cb):
    """x [B, C, T] += cb [B, C, 1] in place (x + cond(g) for a length-1 condition)"""
    B, C, T = x.shape
    _lib.check(_lib.lib().amp_add_channel_bias(_ptr(x), _ptr(cb.reshape(B, C).contiguous()), B, C, T, _stream(x)))
    return x


def rel_attention(q, k, v, emb_k, emb_v, lens, n_heads, window):
    B, C, T = q.shape
    out = torch.empty_like(q)
    _lib.check(_lib.lib().amp_rel_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(emb_k), _ptr(emb_v), _ptr(lens), B, n_heads, C // n_heads, T,
                                            int(window), _ptr(out), _stream(q)))
    return out


def rel_attention_qkv(qkv, emb_k, emb_v, lens, n_heads, window):
    """the same on the output [B, 3C, T] of a merged q | k | v projection (no copies: the three slices share a batch stride)"""
    B, C3, T = qkv.shape
    C = C3 // 3
    out = torch.empty((B, C, T), dtype=torch.float32, device=qkv.device)
    p = qkv.data_ptr()
    step = C * T * 4
    _lib.check(_lib.lib().amp_rel_attention_strided(ctypes.c_void_p(p), ctypes.c_void_p(p + step), ctypes.c_void_p(p + 2 * step), C3 * T,
                                                    _ptr(emb_k), _ptr(emb_v), _ptr(lens), B, n_heads, C // n_heads, T, int(window),
                                                    _ptr(out), _stream(qkv)))
    return out


def dwconv(x, weight, bias, lens, dilation):
    B, C, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().amp_dwconv(_ptr(x), _ptr(weight), _ptr(bias), _ptr(lens), B, C, T, weight.shape[-1], int(dilation), _ptr(y), _stream(x)))
    return y


def spline_flow(z, h, lens, num_bins, filter_channels, tail_bound, inverse, flip_in=False, flip_out=False):
    B, _, T = z.shape
    out = torch.empty_like(z)
    _lib.check(_lib.lib().amp_spline_flow(_ptr(z), _ptr(h), _ptr(lens), B, T, int(num_bins), int(filter_channels), float(tail_bound),
                                          int(inverse), int(flip_in), int(flip_out), _ptr(out), _stream(z)))
    return out


def spline_flow_proj(z, hc, proj_w, proj_b, lens, num_bins, filter_channels, tail_bound, inverse, flip_in=False, flip_out=False):
    """ConvFlow's ``proj`` (1 x 1 conv to the 3K - 1 spline parameters) and the spline step in one launch; hc [B, C, T] is the DDSConv
    output (its columns beyond the lengths are never used)"""
    B, _, T = z.shape
    C = hc.shape[1]
    out = torch.empty_like(z)
    _lib.check(_lib.lib().amp_spline_flow_proj(_ptr(z), _ptr(hc), _ptr(proj_w), _ptr(proj_b), _ptr(lens), B, C, T, int(num_bins),
                                               int(filter_channels), float(tail_bound), int(inverse), int(flip_in), int(flip_out), _ptr(out),
                                               _stream(z)))
    return out


def affine_reverse(x, m, logs, lens):
    B, C, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().amp_affine_reverse(_ptr(x), _ptr(m), _ptr(logs), _ptr(lens), B, C, T, _ptr(y), _stream(x)))
    return y


def embed_tokens(tokens, weight, lens, scale):
    B, T = tokens.shape
    n_vocab, hidden = weight.shape
    y = torch.empty((B, hidden, T), dtype=torch.float32, device=weight.device)
    _lib.check(_lib.lib().amp_embed_tokens(_ptr(tokens), _ptr(weight), _ptr(lens), B, T, hidden, n_vocab, float(scale), _ptr(y), _stream(weight)))
    return y


def durations(logw, lens, length_scale):
    """-> (w_ceil [B, 1, T] float, cum [B, T] int32, y_lengths [B] int32)"""
    B, _, T = logw.shape
    w_ceil = torch.empty_like(logw)
    cum = torch.empty((B, T), dtype=torch.int32, device=logw.device)
    ylen = torch.empty((B,), dtype=torch.int32, device=logw.device)
    _lib.check(_lib.lib().amp_durations(_ptr(logw), _ptr(lens), B, T, float(length_scale), _ptr(w_ceil), _ptr(cum), _ptr(ylen), _stream(logw)))
    return w_ceil, cum, ylen


def expand_path(src, cum, xlens, ylens, t_y, want_attn=False):
    """src [B, D, Tx]: contiguous, or a channel slice of a wider [B, D', Tx] tensor (the halves of the text encoder's stats) --
    read in place through its batch stride"""
    B, D, Tx = src.shape
    if src.stride(2) != 1 or src.stride(1) != Tx or (B > 1 and src.stride(0) < D * Tx):
        src = src.contiguous()
    out = torch.empty((B, D, t_y), dtype=torch.float32, device=src.device)
    attn = torch.empty((B, 1, t_y, Tx), dtype=torch.float32, device=src.device) if want_attn else None
    _lib.check(_lib.lib().amp_expand_path_strided(_ptr(src), int(src.stride(0)) if B > 1 else D * Tx, _ptr(cum), _ptr(xlens), _ptr(ylens), B, D,
                                                  Tx, int(t_y), _ptr(out), _ptr(attn), _stream(src)))
    return out, attn


def gauss_sample(m, logs, noise, noise_scale):
    out = torch.empty_like(m)
    _lib.check(_lib.lib().amp_gauss_sample(_ptr(m), _ptr(logs), _ptr(noise), m.numel(), float(noise_scale), _ptr(out), _stream(m)))
    return out
