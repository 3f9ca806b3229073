"""CPU restatement of FULL VITS inference (TEST INFRASTRUCTURE: the checker for SURVEY.md §8 row f.4, never the product).

Reference path: ``SynthesizerTrn.infer`` models/tts/vits/vits.py:320-369 =
  TextEncoder (vits.py:28-67; Encoder / MultiHeadAttention with windowed relative-position embeddings / FFN,
  modules/transformer/attentions.py:16-77,165-358,361-417; LayerNorm modules/base/base_module.py:11-24)
  -> StochasticDurationPredictor in reverse (modules/duration_predictor/stochastic_duration_predictor.py:14-130:
     DDSConv, ElementwiseAffine, ConvFlow with the piecewise rational-quadratic spline of
     modules/transformer/transforms.py, Flip; modules/flow/modules.py:25-72,304-340,400-458)
     or DurationPredictor (modules/duration_predictor/standard_duration_predictor.py:13-61)
  -> ceil(exp(logw) * mask * length_scale), generate_path (utils/util.py:625-640)
  -> z_p = m_p + noise * exp(logs_p) * noise_scale -> flow reverse -> HiFiGAN_vits decoder
     (both already restated in oracle/vocoder_oracle.py).

Everything is written from the formulas (explicit relative-offset indexing instead of the reference's pad / reshape
skewing, closed-form spline bins), in torch on the CPU, and pinned against golden vectors produced by the real
reference classes: tests/golden/make_golden_vits_infer.py -> tests/test_oracle_vits_infer.py.  The two Gaussian draws
of ``infer`` are inputs here (``noise_dp`` [B, 2, T_text], ``noise_z`` like m_p)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import vocoder_oracle as vo


def _sub(sd, prefix):
    """state-dict view of one sub-module: keys under ``prefix`` with the prefix removed"""
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _conv(sd, name, x, padding=0, dilation=1, groups=1):
    return F.conv1d(x, sd[name + ".weight"], sd.get(name + ".bias"), padding=padding, dilation=dilation, groups=groups)


def layer_norm_channels(x, gamma, beta, eps=1e-5):
    """LayerNorm over the CHANNEL axis of [B, C, T] (base_module.py:20-23: transpose, F.layer_norm, transpose)."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


# ----------------------------------------------------------------------------------------------------------------
# text encoder
# ----------------------------------------------------------------------------------------------------------------
def relative_self_attention(sd, p, x, x_mask, n_heads, window):
    """MultiHeadAttention.forward(x, x, attn_mask) attentions.py:222-272 for self-attention with heads_share=True.

    With r = j - i the offset of key j from query i and E_k / E_v the [2*window+1, d_k] embeddings:
      score[i, j] = q_i . k_j / sqrt(d_k) + [|r| <= window] q_i . E_k[r + window] / sqrt(d_k)
      out[i]      = sum_j p[i, j] v_j     + sum_{|r| <= window} p[i, j] E_v[r + window]
    (what _get_relative_embeddings / _relative_position_to_absolute_position / _absolute_position_to_relative_position,
    :291-345, compute through padding and reshaping)."""
    B, C, T = x.shape
    dk = C // n_heads
    q = _conv(sd, p + ".conv_q", x).view(B, n_heads, dk, T)
    k = _conv(sd, p + ".conv_k", x).view(B, n_heads, dk, T)
    v = _conv(sd, p + ".conv_v", x).view(B, n_heads, dk, T)
    qs = q / math.sqrt(dk)
    scores = torch.einsum("bhdi,bhdj->bhij", qs, k)
    ar = torch.arange(T)
    off = ar.view(1, T) - ar.view(T, 1) + window                  # [i, j] -> r + window
    near = ((off >= 0) & (off <= 2 * window)).to(x.dtype)
    off = off.clamp(0, 2 * window)
    ek = sd[p + ".emb_rel_k"][0][off] * near.unsqueeze(-1)         # [T, T, dk]
    ev = sd[p + ".emb_rel_v"][0][off] * near.unsqueeze(-1)
    scores = scores + torch.einsum("bhdi,ijd->bhij", qs, ek)
    mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)              # [B, 1, T, T]   (Encoder.forward :65)
    scores = scores.masked_fill(mask == 0, -1e4)
    prob = torch.softmax(scores, dim=-1)
    out = torch.einsum("bhij,bhdj->bhdi", prob, v) + torch.einsum("bhij,ijd->bhdi", prob, ev)
    return _conv(sd, p + ".conv_o", out.reshape(B, C, T))


def ffn(sd, p, x, x_mask, kernel_size):
    """FFN.forward attentions.py:392-400 (activation None -> relu, 'same' padding (k-1)//2, k//2)."""
    pad = ((kernel_size - 1) // 2, kernel_size // 2)
    h = torch.relu(_conv(sd, p + ".conv_1", F.pad(x * x_mask, pad)))
    return _conv(sd, p + ".conv_2", F.pad(h * x_mask, pad)) * x_mask


def encoder(sd, p, x, x_mask, n_layers, n_heads, kernel_size, window=4):
    """Encoder.forward attentions.py:64-76 (dropout is the identity in eval mode)."""
    x = x * x_mask
    for i in range(n_layers):
        y = relative_self_attention(sd, f"{p}.attn_layers.{i}", x, x_mask, n_heads, window)
        x = layer_norm_channels(x + y, sd[f"{p}.norm_layers_1.{i}.gamma"], sd[f"{p}.norm_layers_1.{i}.beta"])
        y = ffn(sd, f"{p}.ffn_layers.{i}", x, x_mask, kernel_size)
        x = layer_norm_channels(x + y, sd[f"{p}.norm_layers_2.{i}.gamma"], sd[f"{p}.norm_layers_2.{i}.beta"])
    return x * x_mask


def text_encoder(sd, p, tokens, lengths, hidden, out_channels, n_layers, n_heads, kernel_size):
    """TextEncoder.forward vits.py:57-67 -> (x, m, logs, x_mask)."""
    x = sd[p + ".emb.weight"][tokens] * math.sqrt(hidden)          # [B, T, H]
    x = x.transpose(1, 2)
    x_mask = vo.sequence_mask(lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    x = encoder(sd, p + ".encoder", x * x_mask, x_mask, n_layers, n_heads, kernel_size)
    stats = _conv(sd, p + ".proj", x) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    return x, m, logs, x_mask


# ----------------------------------------------------------------------------------------------------------------
# duration predictors
# ----------------------------------------------------------------------------------------------------------------
def dds_conv(sd, p, x, x_mask, kernel_size, n_layers, g=None):
    """DDSConv.forward modules/flow/modules.py:60-72: depthwise conv at dilation k**i, LayerNorm, GELU, 1x1, LayerNorm,
    GELU, residual."""
    C = x.shape[1]
    if g is not None:
        x = x + g
    for i in range(n_layers):
        d = kernel_size**i
        y = _conv(sd, f"{p}.convs_sep.{i}", x * x_mask, padding=(kernel_size * d - d) // 2, dilation=d, groups=C)
        y = F.gelu(layer_norm_channels(y, sd[f"{p}.norms_1.{i}.gamma"], sd[f"{p}.norms_1.{i}.beta"]))
        y = _conv(sd, f"{p}.convs_1x1.{i}", y)
        y = F.gelu(layer_norm_channels(y, sd[f"{p}.norms_2.{i}.gamma"], sd[f"{p}.norms_2.{i}.beta"]))
        x = x + y
    return x * x_mask


def rq_spline(x, uw, uh, ud, inverse, tail_bound, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """Monotone piecewise rational-quadratic spline with linear tails (Durkan et al. 2019), element-wise over ``x`` with
    K bins parametrised by ``uw`` / ``uh`` [..., K] and ``ud`` [..., K-1]   (modules/transformer/transforms.py:56-215).

    Inside [-B, B]: bin widths w = min_w + (1 - K min_w) softmax(uw), likewise heights; knot derivatives
    d = min_d + softplus(ud) with d = 1 at both ends.  In bin k with slope s = h_k / w_k and t the position inside it,
      forward  y = y_k + h_k (s t^2 + d_k t (1 - t)) / (s + (d_k + d_{k+1} - 2 s) t (1 - t))
      inverse  t = 2c / (-b - sqrt(b^2 - 4ac)) for the quadratic the forward map gives in t.
    Outside: identity.  Returns the transformed values (the log-determinant is not needed for inference)."""
    K = uw.shape[-1]
    B = tail_bound
    inside = (x >= -B) & (x <= B)
    const = math.log(math.exp(1 - min_d) - 1)                      # softplus(const) = 1 - min_d  ->  d = 1 at the ends
    ud = F.pad(ud, (1, 1), value=const)

    def knots(u, lo):
        frac = lo + (1 - lo * K) * torch.softmax(u, dim=-1)
        edges = F.pad(torch.cumsum(frac, dim=-1), (1, 0)) * (2 * B) - B
        edges = torch.cat([torch.full_like(edges[..., :1], -B), edges[..., 1:-1], torch.full_like(edges[..., :1], B)], dim=-1)
        return edges, edges[..., 1:] - edges[..., :-1]

    xk, w = knots(uw, min_w)
    yk, h = knots(uh, min_h)
    d = min_d + F.softplus(ud)
    xin = x.clamp(-B, B)                                           # values outside keep the identity below
    search = (yk if inverse else xk).clone()
    search[..., -1] += 1e-6
    k = ((xin.unsqueeze(-1) >= search).sum(dim=-1) - 1).clamp(0, K - 1).unsqueeze(-1)
    pick = lambda t: t.gather(-1, k).squeeze(-1)  # noqa: E731
    x0, w0, y0, h0 = pick(xk), pick(w), pick(yk), pick(h)
    d0, d1 = pick(d), pick(d[..., 1:])
    s = h0 / w0
    if inverse:
        dy = xin - y0
        e = d0 + d1 - 2 * s
        a = dy * e + h0 * (s - d0)
        b = h0 * d0 - dy * e
        c = -s * dy
        t = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
        out = t * w0 + x0
    else:
        t = (xin - x0) / w0
        tt = t * (1 - t)
        out = y0 + h0 * (s * t * t + d0 * tt) / (s + (d0 + d1 - 2 * s) * tt)
    return torch.where(inside, out, x)


def conv_flow(sd, p, x, x_mask, g, reverse, filter_channels, kernel_size, n_layers=3, num_bins=10, tail_bound=5.0):
    """ConvFlow.forward modules/flow/modules.py:424-458 (half_channels = 1: x0 conditions the spline applied to x1)."""
    x0, x1 = x[:, :1], x[:, 1:]
    h = _conv(sd, p + ".pre", x0)
    h = dds_conv(sd, p + ".convs", h, x_mask, kernel_size, n_layers, g=g)
    h = _conv(sd, p + ".proj", h) * x_mask                          # [B, 3K-1, T]
    B, _, T = x0.shape
    h = h.reshape(B, 1, -1, T).permute(0, 1, 3, 2)                   # [B, 1, T, 3K-1]
    uw = h[..., :num_bins] / math.sqrt(filter_channels)
    uh = h[..., num_bins:2 * num_bins] / math.sqrt(filter_channels)
    ud = h[..., 2 * num_bins:]
    x1 = rq_spline(x1, uw, uh, ud, inverse=reverse, tail_bound=tail_bound)
    return torch.cat([x0, x1], 1) * x_mask


def stochastic_duration_predictor_reverse(sd, p, x, x_mask, noise, noise_scale, filter_channels, kernel_size, n_flows=4, g=None):
    """StochasticDurationPredictor.forward(reverse=True) stochastic_duration_predictor.py:61-70,117-130 -> logw [B, 1, T].

    flows = [ElementwiseAffine, ConvFlow_1, Flip, ..., ConvFlow_n, Flip]; the reverse pass runs them backwards and
    SKIPS ConvFlow_1 ("flows[:-2] + [flows[-1]]", :119-120)."""
    h = _conv(sd, p + ".pre", x)
    if g is not None:
        h = h + _conv(sd, p + ".cond", g)
    h = dds_conv(sd, p + ".convs", h, x_mask, kernel_size, 3)
    h = _conv(sd, p + ".proj", h) * x_mask
    z = noise * noise_scale
    for idx in range(2 * n_flows, 1, -1):                            # Flip_n, ConvFlow_n, ..., ConvFlow_2, Flip_1
        if idx % 2 == 0:
            z = torch.flip(z, [1])
        else:
            z = conv_flow(sd, f"{p}.flows.{idx}", z, x_mask, h, True, filter_channels, kernel_size)
    z = (z - sd[p + ".flows.0.m"]) * torch.exp(-sd[p + ".flows.0.logs"]) * x_mask     # ElementwiseAffine reverse :338-340
    return z[:, :1]


def duration_predictor(sd, p, x, x_mask, kernel_size, g=None):
    """DurationPredictor.forward standard_duration_predictor.py:40-61 -> logw [B, 1, T]."""
    if g is not None:
        x = x + _conv(sd, p + ".cond", g)
    h = torch.relu(_conv(sd, p + ".conv_1", x * x_mask, padding=kernel_size // 2))
    h = layer_norm_channels(h, sd[p + ".norm_1.gamma"], sd[p + ".norm_1.beta"])
    h = torch.relu(_conv(sd, p + ".conv_2", h * x_mask, padding=kernel_size // 2))
    h = layer_norm_channels(h, sd[p + ".norm_2.gamma"], sd[p + ".norm_2.beta"])
    return _conv(sd, p + ".proj", h * x_mask) * x_mask


def generate_path(duration, mask):
    """utils/util.py:625-640: duration [B, 1, Tx] (integers), mask [B, 1, Ty, Tx] -> path [B, 1, Ty, Tx] with
    path[y, x] = 1 where cum[x-1] <= y < cum[x]."""
    cum = torch.cumsum(duration, -1)                                 # [B, 1, Tx]
    ty = mask.shape[2]
    frame = torch.arange(ty, dtype=cum.dtype).view(1, 1, ty, 1)
    below = (frame < cum.unsqueeze(2)).to(mask.dtype)                # [B, 1, Ty, Tx]
    prev = F.pad(below, (1, 0))[..., :-1]
    return (below - prev) * mask


# ----------------------------------------------------------------------------------------------------------------
# SynthesizerTrn.infer
# ----------------------------------------------------------------------------------------------------------------
def vits_infer(sd, hp, tokens, lengths, noise_z, noise_dp=None, sid=None, noise_scale=1.0, length_scale=1.0,
               noise_scale_w=1.0, max_len=None):
    """SynthesizerTrn.infer vits.py:320-369.  ``hp``: inter_channels, hidden_channels, n_heads, n_layers, kernel_size,
    use_sdp, n_speakers + the decoder's hifigan hyper-parameters.  Returns the reference's output dict."""
    inter, hidden = hp["inter_channels"], hp["hidden_channels"]
    x, m_p, logs_p, x_mask = text_encoder(sd, "enc_p", tokens, lengths, hidden, inter, hp["n_layers"], hp["n_heads"], hp["kernel_size"])
    g = sd["emb_g.weight"][sid.squeeze(-1)].unsqueeze(-1) if hp.get("n_speakers", 0) > 0 else None
    if hp.get("use_sdp", True):
        logw = stochastic_duration_predictor_reverse(sd, "dp", x, x_mask, noise_dp, noise_scale_w, hidden, 3, 4, g=g)
    else:
        logw = duration_predictor(sd, "dp", x, x_mask, 3, g=g)
    w_ceil = torch.ceil(torch.exp(logw) * x_mask * length_scale)
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_mask = vo.sequence_mask(y_lengths, int(y_lengths.max())).unsqueeze(1).to(x_mask.dtype)
    attn = generate_path(w_ceil, x_mask.unsqueeze(2) * y_mask.unsqueeze(-1))
    m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)          # [B, Ty, Tx] x [B, Tx, D]
    logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
    z_p = m_p + noise_z * torch.exp(logs_p) * noise_scale
    z = vo.coupling_block_forward(_sub(sd, "flow."), "", z_p, y_mask, reverse=True, channels=inter, hidden=hidden, g=g)
    y_hat = vo.hifigan_forward(_sub(sd, "dec."), hp, (z * y_mask)[:, :, :max_len], g=g)
    return {"y_hat": y_hat, "attn": attn, "mask": y_mask, "z": z, "z_p": z_p, "m_p": m_p, "logs_p": logs_p,
            "logw": logw, "enc_x": x}
