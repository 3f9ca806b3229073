"""Seeded synthetic weights / inputs for parity tests and the bench (TEST INFRASTRUCTURE).

There are no checkpoints offline, so every test/bench model uses random-init
weights.  The reference's own init (``init_weights`` gan_utils.py:25-28, sigma
0.01 on weight_v, g=||v||) gives |out| <= 0.06 and bias-dominated signals,
which makes parity checks insensitive.  We use the
variance-preserving scheme of SURVEY.md §8(d) instead:

  weight_v ~ N(0,1), weight_g ~ gain*U(0.7,1.3), bias ~ N(0,0.05),
  Snake alpha/beta ~ N(0,0.3) (log-scale), un-normed ``weight`` ~ N(0,1)/sqrt(fan_in).

Each tensor is drawn from its own generator seeded by crc32(key) ^ seed, so the
values do not depend on key order or on which other tensors exist; they are
reproducible wherever the same torch CPU RNG runs (same image on the GPU box).

``*_param_shapes`` restate the reference constructors' parameter lists
(hifigan.py:151-199, :376-422; bigvgan.py:232-306) -- pinned against the real
reference ``state_dict`` key/shape lists committed in tests/golden/keys_*.json.
"""

from __future__ import annotations

import zlib
from collections import OrderedDict

import torch

from .vocoder_oracle import kaiser_sinc_filter1d


def _wn_conv(shapes, prefix, cout, cin, k, transposed=False, wn=True, bias=True):
    d0 = cin if transposed else cout
    d1 = cout if transposed else cin
    if wn:  # weight_norm re-registers g/v AFTER bias
        if bias:
            shapes[prefix + ".bias"] = (cout,)
        shapes[prefix + ".weight_g"] = (d0, 1, 1)
        shapes[prefix + ".weight_v"] = (d0, d1, k)
    else:
        shapes[prefix + ".weight"] = (d0, d1, k)
        if bias:
            shapes[prefix + ".bias"] = (cout,)


def _generator_param_shapes(n_in, hp, vits=False, gin_channels=0, ups_fmt="ups.{i}", act=None):
    s = OrderedDict()
    c0 = hp["upsample_initial_channel"]
    _wn_conv(s, "conv_pre", c0, n_in, 7, wn=not vits)
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        _wn_conv(s, ups_fmt.format(i=i), c0 // 2 ** (i + 1), c0 // 2**i, k, transposed=True)
    nk = len(hp["resblock_kernel_sizes"])
    ch = c0
    for i in range(len(hp["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, (k, d) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            pre = f"resblocks.{i * nk + j}"
            if str(hp["resblock"]) == "1":
                for p in range(len(d)):
                    _wn_conv(s, f"{pre}.convs1.{p}", ch, ch, k)
                for p in range(len(d)):
                    _wn_conv(s, f"{pre}.convs2.{p}", ch, ch, k)
                n_act = 2 * len(d)
            else:
                for p in range(len(d)):
                    _wn_conv(s, f"{pre}.convs.{p}", ch, ch, k)
                n_act = len(d)
            if act is not None:
                for m in range(n_act):
                    act(s, f"{pre}.activations.{m}", ch)
    if act is not None:
        act(s, "activation_post", ch)
    _wn_conv(s, "conv_post", 1, ch, 7, wn=not vits, bias=not vits)
    if vits and gin_channels:
        _wn_conv(s, "cond", c0, gin_channels, 1, wn=False)
    return s


def hifigan_param_shapes(n_in, hp, vits=False, gin_channels=0):
    """Ordered {key: shape} of HiFiGAN (hifigan.py:151-199) or HiFiGAN_vits (:376-422)."""
    return _generator_param_shapes(n_in, hp, vits=vits, gin_channels=gin_channels)


def bigvgan_param_shapes(n_in, hp):
    """Ordered {key: shape} of BigVGAN (bigvgan.py:232-306) incl. Snake params and the
    persistent anti-aliasing filter buffers (resample.py:31-34, filter.py:89-90)."""
    beta = hp["activation"] == "snakebeta"

    def act(s, prefix, c):
        s[prefix + ".act.alpha"] = (c,)
        if beta:
            s[prefix + ".act.beta"] = (c,)
        s[prefix + ".upsample.filter"] = (1, 1, 12)
        s[prefix + ".downsample.lowpass.filter"] = (1, 1, 12)

    return _generator_param_shapes(n_in, hp, ups_fmt="ups.{i}.0", act=act)


def apnet_param_shapes(n_mel, n_fft, hp):
    """Parameter list of the reference APNet (apnet.py:280-352) in state_dict order."""
    s = OrderedDict()
    bins = n_fft // 2 + 1
    _wn_conv(s, "ASP_input_conv", hp["ASP_channel"], n_mel, hp["ASP_input_conv_kernel_size"])
    _wn_conv(s, "PSP_input_conv", hp["PSP_channel"], n_mel, hp["PSP_input_conv_kernel_size"])
    for pre in ("ASP", "PSP"):
        ch = hp[f"{pre}_channel"]
        for j, (k, d) in enumerate(zip(hp[f"{pre}_resblock_kernel_sizes"], hp[f"{pre}_resblock_dilation_sizes"])):
            for p in range(len(d)):
                _wn_conv(s, f"{pre}_ResNet.{j}.convs1.{p}", ch, ch, k)
            for p in range(len(d)):
                _wn_conv(s, f"{pre}_ResNet.{j}.convs2.{p}", ch, ch, k)
    _wn_conv(s, "ASP_output_conv", bins, hp["ASP_channel"], hp["ASP_output_conv_kernel_size"])
    _wn_conv(s, "PSP_output_R_conv", bins, hp["PSP_channel"], hp["PSP_output_R_conv_kernel_size"])
    _wn_conv(s, "PSP_output_I_conv", bins, hp["PSP_channel"], hp["PSP_output_I_conv_kernel_size"])
    return s


def nsfhifigan_param_shapes(n_mel, hp):
    """Parameter list of the reference NSFHiFiGAN (nsfhifigan.py:181-256) in state_dict order:
    m_source.l_linear, noise_convs (plain Conv1d: weight then bias), conv_pre, ups, resblocks, conv_post."""
    s = OrderedDict()
    c0 = hp["upsample_initial_channel"]
    rates = list(hp["upsample_rates"])
    s["m_source.l_linear.weight"] = (1, hp["harmonic_num"] + 1)
    s["m_source.l_linear.bias"] = (1,)
    for i in range(len(rates)):
        c_cur = c0 // 2 ** (i + 1)
        if i + 1 < len(rates):
            st = 1
            for r in rates[i + 1:]:
                st *= r
            s[f"noise_convs.{i}.weight"] = (c_cur, 1, 2 * st)
        else:
            s[f"noise_convs.{i}.weight"] = (c_cur, 1, 1)
        s[f"noise_convs.{i}.bias"] = (c_cur,)
    g = _generator_param_shapes(n_mel, hp)
    s.update(g)
    return s


def melgan_param_shapes(n_mel, hp):
    """Parameter list of the reference MelGAN Sequential (melgan.py:51-97), in state_dict order."""
    s = OrderedDict()
    ratios, ngf, nres = list(hp["ratios"]), hp["ngf"], hp["n_residual_layers"]
    mult = 2 ** len(ratios)
    idx = 1
    _wn_conv(s, f"model.{idx}", mult * ngf, n_mel, 7)
    idx += 1
    for r in ratios:
        idx += 1
        _wn_conv(s, f"model.{idx}", mult * ngf // 2, mult * ngf, 2 * r, transposed=True)
        idx += 1
        for _ in range(nres):
            dim = mult * ngf // 2
            _wn_conv(s, f"model.{idx}.block.2", dim, dim, 3)
            _wn_conv(s, f"model.{idx}.block.4", dim, dim, 1)
            _wn_conv(s, f"model.{idx}.shortcut", dim, dim, 1)
            idx += 1
        mult //= 2
    idx += 2
    _wn_conv(s, f"model.{idx}", 1, ngf, 7)
    return s


def synth_tensor(key, shape, seed=1234, g_gain=1.0):
    gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "filter":
        return kaiser_sinc_filter1d(0.25, 0.3, 12).reshape(shape)
    if leaf == "weight_v":
        return torch.randn(shape, generator=gen)
    if leaf == "weight_g":
        return g_gain * (0.7 + 0.6 * torch.rand(shape, generator=gen))
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=gen)
    if leaf in ("alpha", "beta"):
        return 0.3 * torch.randn(shape, generator=gen)
    if leaf == "gamma":  # LayerNorm scale (modules/base/base_module.py:17)
        return 1.0 + 0.2 * torch.randn(shape, generator=gen)
    if leaf in ("m", "logs"):  # ElementwiseAffine (modules/flow/modules.py:328-329)
        return 0.1 * torch.randn(shape, generator=gen)
    if leaf in ("emb_rel_k", "emb_rel_v"):  # relative-position embeddings (modules/transformer/attentions.py:200-209)
        return torch.randn(shape, generator=gen) * shape[-1] ** -0.5
    if leaf == "weight":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return torch.randn(shape, generator=gen) / fan_in**0.5
    raise KeyError(key)


def synth_state_dict(shapes, seed=1234, g_gain=1.0):
    return OrderedDict((k, synth_tensor(k, tuple(v), seed, g_gain)) for k, v in shapes.items())


def synth_mel(B, n_mel, T, seed=0):
    """Log-mel-like input: randn*2-5."""
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(B, n_mel, T, generator=gen) * 2 - 5


def _plain_conv(s, prefix, cout, cin, k):
    s[prefix + ".weight"] = (cout, cin, k)
    s[prefix + ".bias"] = (cout,)


def wn_param_shapes(s, prefix, hidden, kernel_size, n_layers, gin_channels=0):
    """WN.__init__ modules/flow/modules.py:74-124 (module registration order: in_layers,
    res_skip_layers, then cond_layer)."""
    for i in range(n_layers):
        _wn_conv(s, f"{prefix}.in_layers.{i}", 2 * hidden, hidden, kernel_size)
    for i in range(n_layers):
        _wn_conv(s, f"{prefix}.res_skip_layers.{i}", 2 * hidden if i < n_layers - 1 else hidden, hidden, 1)
    if gin_channels:
        _wn_conv(s, f"{prefix}.cond_layer", 2 * hidden * n_layers, gin_channels, 1)
    return s


def posterior_encoder_param_shapes(in_ch=513, out_ch=192, hidden=192, n_layers=16, gin_channels=0):
    """PosteriorEncoder.__init__ vits.py:116-143"""
    s = OrderedDict()
    _plain_conv(s, "pre", hidden, in_ch, 1)
    wn_param_shapes(s, "enc", hidden, 5, n_layers, gin_channels)
    _plain_conv(s, "proj", 2 * out_ch, hidden, 1)
    return s


def coupling_block_param_shapes(channels=192, hidden=192, n_layers=4, n_flows=4, gin_channels=0):
    """ResidualCouplingBlock.__init__ vits.py:71-103 (Flip has no parameters; indices 0,2,4,6)."""
    s = OrderedDict()
    for f in range(n_flows):
        p = f"flows.{2 * f}"
        _plain_conv(s, f"{p}.pre", hidden, channels // 2, 1)
        wn_param_shapes(s, f"{p}.enc", hidden, 5, n_layers, gin_channels)
        _plain_conv(s, f"{p}.post", channels // 2, hidden, 1)
    return s
