"""MI355X-native pieces of VITS that BASELINE.json config 5 runs around the HiFi-GAN decoder
(models/tts/vits/vits.py): ``PosteriorEncoder`` (:115-152), ``ResidualCouplingBlock`` (:70-112) and the
``enc_q -> flow -> flow(reverse) -> dec`` composition of ``SynthesizerTrn.voice_conversion`` (:371-379).
Same constructor arguments and state_dict keys as the reference sub-modules.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.models.vocoders.gan.generator.hifigan import HiFiGAN_vits
from amphion_amd.modules import hip_ops
from amphion_amd.modules.flow.modules import WN, Flip, ResidualCouplingLayer
from amphion_amd.modules.hip_ops import HipConv1d


class ResidualCouplingBlock(nn.Module):
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.n_flows = n_flows
        self.gin_channels = gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                    gin_channels=gin_channels, mean_only=True))
            self.flows.append(Flip())

    def forward(self, x, x_lengths=None, g=None, reverse=False):
        """vits.py:105-112.  ``x_lengths`` replaces the dense ``x_mask``."""
        if x_lengths is not None:
            x_lengths = hip_ops.lens_tensor(x_lengths, x.device)   # once: every flow reuses the device tensor
        # a coupling layer that follows a Flip works in place on the Flip's output (a fresh tensor): no clone
        fresh = False
        if not reverse:
            for flow in self.flows:
                x, _ = flow(x, x_lengths, g=g, reverse=reverse, **({"owns_input": fresh} if isinstance(flow, ResidualCouplingLayer) else {}))
                fresh = True
        else:
            for flow in reversed(self.flows):
                x = flow(x, x_lengths, g=g, reverse=reverse, **({"owns_input": fresh} if isinstance(flow, ResidualCouplingLayer) else {}))
                fresh = True
        return x


class PosteriorEncoder(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.pre = HipConv1d(in_channels, hidden_channels, 1, weight_norm=False)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = HipConv1d(hidden_channels, out_channels * 2, 1, weight_norm=False)

    def forward(self, x, x_lengths, g=None, noise=None):
        """vits.py:145-152 -> (z, m, logs, x_mask).  ``noise`` (same shape as m) replaces the
        reference's ``torch.randn_like(m)`` when given, so results can be pinned."""
        x = _lib.require_device_tensor(x, "PosteriorEncoder input")
        B, _, T = x.shape
        lens = hip_ops.lens_tensor(x_lengths, x.device)
        h = self.pre(x)
        hip_ops.sequence_mask_(h, lens)
        h = self.enc(h, lens, g=g)
        stats = self.proj(h)
        hip_ops.sequence_mask_(stats, lens)
        m, logs = torch.split(stats, self.out_channels, dim=1)
        if noise is None:
            noise = torch.randn_like(m)
        z = hip_ops.posterior_sample(stats, _lib.require_device_tensor(noise, "noise"), lens)
        x_mask = (torch.arange(T, device=x.device).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1).to(x.dtype)
        return z, m, logs, x_mask


def _dec_exact(self, z, g):
    """``self.dec(z, g=g)`` with the fp32 reference's operand range: the decoder's own range flag is checked for THIS call and
    the exact-fp32 kernels take over if an activation did not fit (``HipGenerator.forward_exact_range``) -- the waveform
    that leaves ``infer`` / ``reconstruct`` is never silently inf / NaN.  The check synchronises once, after the last
    launch of the call; the op-level check that follows it finds the stream already drained."""
    if g is not None and not hasattr(self.dec, "cond"):
        raise AttributeError("'HiFiGAN_vits' object has no attribute 'cond'")
    return self.dec.forward_exact_range(z, g)


class SynthesizerTrnDecodePath(nn.Module):
    """The sub-modules of ``SynthesizerTrn`` (vits.py:155-379) that config 5 exercises -- ``enc_q``,
    ``flow``, ``dec`` -- under the reference's attribute names, so a full VITS checkpoint loads with
    ``load_state_dict(sd, strict=False)``.  Text encoder / duration predictor are out of scope."""

    def __init__(self, spec_channels, inter_channels, hidden_channels, resblock, resblock_kernel_sizes,
                 resblock_dilation_sizes, upsample_rates, upsample_initial_channel, upsample_kernel_sizes,
                 gin_channels=0, **unused):
        super().__init__()
        self.dec = HiFiGAN_vits(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                                upsample_rates, upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)

    _dec_exact = _dec_exact

    def reconstruct(self, y, y_lengths, g_src=None, g_tgt=None, noise=None):
        """voice_conversion topology (vits.py:371-379): enc_q -> flow -> flow(reverse) -> dec."""
        lens = hip_ops.lens_tensor(y_lengths, _lib.require_device_tensor(y, "y").device)
        z, m_q, logs_q, y_mask = self.enc_q(y, lens, g=g_src, noise=noise)
        z_p = self.flow(z, lens, g=g_src)
        z_hat = self.flow(z_p, lens, g=g_tgt, reverse=True)
        o_hat = self._dec_exact(hip_ops.sequence_mask_(z_hat.clone(), lens), g_tgt)
        _lib.range_check(z_hat.device)       # enc_q / flow are op-level f16x3 launches: an operand beyond |x| = 4094 raises HERE
        return o_hat, y_mask, (z, z_p, z_hat)


# ---- the text -> duration -> alignment front and SynthesizerTrn.infer ----------
class TextEncoder(nn.Module):
    """vits.py:28-67, same arguments and state_dict keys; ``forward(tokens, lengths) -> (x, m, logs, lens)``."""

    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        from amphion_amd.modules.transformer import Encoder

        self.n_vocab, self.out_channels, self.hidden_channels = n_vocab, out_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers = filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels**-0.5)
        self.encoder = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = HipConv1d(hidden_channels, out_channels * 2, 1, weight_norm=False)

    def forward(self, x, x_lengths):
        w = self.emb.weight.detach()
        if not w.is_cuda:
            raise RuntimeError("TextEncoder parameters are on the CPU: the HIP path has no CPU fallback (move the model to 'cuda')")
        tokens = torch.as_tensor(x).to(device=w.device, dtype=torch.int64).contiguous()
        lens = hip_ops.lens_tensor(x_lengths, w.device)
        h = hip_ops.embed_tokens(tokens, w.contiguous(), lens, self.hidden_channels**0.5)   # emb * sqrt(H), [B, H, T], masked
        h = self.encoder(h, lens)
        stats = hip_ops.sequence_mask_(self.proj(h), lens)
        m, logs = torch.split(stats, self.out_channels, dim=1)   # views, as in the reference (:65); expand_path reads them in place
        return h, m, logs, lens


class SynthesizerTrn(nn.Module):
    """Inference side of vits.py:155-379 under the reference's constructor signature and state_dict keys (enc_p, dec,
    enc_q, flow, dp, emb_g): ``infer`` (:320-369) and ``voice_conversion`` (:371-379).  Training (``forward``) is out of
    scope -- the kernels have no backward."""

    def __init__(self, n_vocab, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0, use_sdp=True, **kwargs):
        super().__init__()
        from amphion_amd.modules.duration_predictor.standard_duration_predictor import DurationPredictor
        from amphion_amd.modules.duration_predictor.stochastic_duration_predictor import StochasticDurationPredictor

        self.n_vocab, self.spec_channels, self.inter_channels, self.hidden_channels = n_vocab, spec_channels, inter_channels, hidden_channels
        self.segment_size, self.n_speakers, self.gin_channels, self.use_sdp = segment_size, n_speakers, gin_channels, use_sdp
        self.enc_p = TextEncoder(n_vocab, inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.dec = HiFiGAN_vits(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                                upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)
        if use_sdp:
            self.dp = StochasticDurationPredictor(hidden_channels, 192, 3, 0.5, 4, gin_channels=gin_channels)
        else:
            self.dp = DurationPredictor(hidden_channels, 256, 3, 0.5, gin_channels=gin_channels)
        if n_speakers >= 1:
            self.emb_g = nn.Embedding(n_speakers, gin_channels)

    _dec_exact = _dec_exact

    def forward(self, data):
        raise NotImplementedError("SynthesizerTrn.forward is the training step; this package is inference-only")

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.0, max_len=None, noise_dp=None, noise_z=None):
        """vits.py:320-369.  ``noise_dp`` [B, 2, T_text] / ``noise_z`` [B, inter, T_frames] replace the reference's two
        ``torch.randn`` draws when given (pinned tests).  One host sync: the frame count max(y_lengths) sizes the tensors,
        as ``sequence_mask(y_lengths, None)`` does in the reference (:344)."""
        xe, m_p, logs_p, lens = self.enc_p(x, x_lengths)
        dev = xe.device
        g = None
        if self.n_speakers > 0:
            g = self.emb_g(torch.as_tensor(sid).to(dev).squeeze(-1)).unsqueeze(-1).detach().contiguous()   # [B, gin, 1]: a row lookup
        if self.use_sdp:
            logw = self.dp(xe, lens, g=g, reverse=True, noise_scale=noise_scale_w, noise=noise_dp)
        else:
            logw = self.dp(xe, lens, g=g)
        w_ceil, cum, y_lengths = hip_ops.durations(logw, lens, length_scale)
        t_y = int(y_lengths.max().item())
        m_e, attn = hip_ops.expand_path(m_p, cum, lens, y_lengths, t_y, want_attn=True)
        logs_e, _ = hip_ops.expand_path(logs_p, cum, lens, y_lengths, t_y)
        if noise_z is None:
            noise_z = torch.randn_like(m_e)
        z_p = hip_ops.gauss_sample(m_e, logs_e, _lib.require_device_tensor(noise_z, "noise_z"), noise_scale)
        z = self.flow(z_p, y_lengths, g=g, reverse=True)
        zm = hip_ops.sequence_mask_(z.clone(), y_lengths)
        o = self._dec_exact(zm[:, :, :max_len].contiguous() if max_len is not None else zm, g)
        _lib.range_check(dev)                # text encoder / duration predictor / flow: op-level f16x3 launches (as in reconstruct)
        y_mask = (torch.arange(t_y, device=dev).unsqueeze(0) < y_lengths.unsqueeze(1)).unsqueeze(1).to(z.dtype)
        return {"y_hat": o, "attn": attn, "mask": y_mask, "z": z, "z_p": z_p, "m_p": m_e, "logs_p": logs_e}

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt):
        assert self.n_speakers > 0, "n_speakers have to be larger than 0."
        dev = next(self.parameters()).device
        g_src = self.emb_g(torch.as_tensor(sid_src).to(dev)).unsqueeze(-1).detach().contiguous()
        g_tgt = self.emb_g(torch.as_tensor(sid_tgt).to(dev)).unsqueeze(-1).detach().contiguous()
        lens = hip_ops.lens_tensor(y_lengths, dev)
        z, m_q, logs_q, y_mask = self.enc_q(y, lens, g=g_src)
        z_p = self.flow(z, lens, g=g_src)
        z_hat = self.flow(z_p, lens, g=g_tgt, reverse=True)
        o_hat = self._dec_exact(hip_ops.sequence_mask_(z_hat.clone(), lens), g_tgt)
        _lib.range_check(z_hat.device)       # enc_q / flow are op-level f16x3 launches: an operand beyond |x| = 4094 raises HERE
        return o_hat, y_mask, (z, z_p, z_hat)
