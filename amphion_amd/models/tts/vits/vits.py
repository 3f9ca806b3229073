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
        if not reverse:
            for flow in self.flows:
                x, _ = flow(x, x_lengths, g=g, reverse=reverse)
        else:
            for flow in reversed(self.flows):
                x = flow(x, x_lengths, g=g, reverse=reverse)
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
        h = self.enc(h, x_lengths, g=g)
        stats = self.proj(h)
        hip_ops.sequence_mask_(stats, lens)
        m, logs = torch.split(stats, self.out_channels, dim=1)
        if noise is None:
            noise = torch.randn_like(m)
        z = hip_ops.posterior_sample(stats, _lib.require_device_tensor(noise, "noise"), lens)
        x_mask = (torch.arange(T, device=x.device).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1).to(x.dtype)
        return z, m, logs, x_mask


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

    def reconstruct(self, y, y_lengths, g_src=None, g_tgt=None, noise=None):
        """voice_conversion topology (vits.py:371-379): enc_q -> flow -> flow(reverse) -> dec."""
        z, m_q, logs_q, y_mask = self.enc_q(y, y_lengths, g=g_src, noise=noise)
        z_p = self.flow(z, y_lengths, g=g_src)
        z_hat = self.flow(z_p, y_lengths, g=g_tgt, reverse=True)
        lens = hip_ops.lens_tensor(y_lengths, z_hat.device)
        o_hat = self.dec(hip_ops.sequence_mask_(z_hat.clone(), lens), g=g_tgt)
        return o_hat, y_mask, (z, z_p, z_hat)
