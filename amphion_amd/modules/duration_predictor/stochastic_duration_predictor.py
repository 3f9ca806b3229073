"""StochasticDurationPredictor, inference direction (modules/duration_predictor/stochastic_duration_predictor.py:14-130),
same arguments and state_dict keys (the posterior branch used only in training keeps its parameters so checkpoints load)."""
import torch
import torch.nn as nn

from amphion_amd import _lib
from amphion_amd.modules import hip_ops
from amphion_amd.modules.flow.modules import ConvFlow, DDSConv, ElementwiseAffine, Flip, Log
from amphion_amd.modules.hip_ops import HipConv1d


class StochasticDurationPredictor(nn.Module):
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, n_flows=4, gin_channels=0):
        super().__init__()
        filter_channels = in_channels                       # :24 (the argument is overridden in the reference)
        self.in_channels, self.filter_channels, self.kernel_size = in_channels, filter_channels, kernel_size
        self.p_dropout, self.n_flows, self.gin_channels = p_dropout, n_flows, gin_channels
        self.log_flow = Log()
        self.flows = nn.ModuleList([ElementwiseAffine(2)])
        for _ in range(n_flows):
            self.flows.append(ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.flows.append(Flip())
        self.post_pre = HipConv1d(1, filter_channels, 1, weight_norm=False)
        self.post_proj = HipConv1d(filter_channels, filter_channels, 1, weight_norm=False)
        self.post_convs = DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        self.post_flows = nn.ModuleList([ElementwiseAffine(2)])
        for _ in range(4):
            self.post_flows.append(ConvFlow(2, filter_channels, kernel_size, n_layers=3))
            self.post_flows.append(Flip())
        self.pre = HipConv1d(in_channels, filter_channels, 1, weight_norm=False)
        self.proj = HipConv1d(filter_channels, filter_channels, 1, weight_norm=False)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers=3, p_dropout=p_dropout)
        if gin_channels != 0:
            self.cond = HipConv1d(gin_channels, filter_channels, 1, weight_norm=False)

    def forward(self, x, lens=None, w=None, g=None, reverse=False, noise_scale=1.0, noise=None):
        """reverse=True only (:117-130) -> logw [B, 1, T].  ``noise`` [B, 2, T] replaces torch.randn when given."""
        if not reverse:
            raise NotImplementedError("StochasticDurationPredictor forward (NLL): training direction only")
        x = _lib.require_device_tensor(x, "duration predictor input")
        B, _, T = x.shape
        h = self.pre(x)
        if g is not None:
            hip_ops.add_channel_bias_(h, self.cond(_lib.require_device_tensor(g, "g")))
        h = self.convs(h, lens)
        h = self.proj(h)
        if lens is not None:
            hip_ops.sequence_mask_(h, lens)
        if noise is None:
            noise = torch.randn(B, 2, T, device=x.device, dtype=x.dtype)
        z = _lib.require_device_tensor(noise, "noise") * noise_scale
        # reversed flows without the first ConvFlow (:118-120): Flip, CF_n, Flip, ..., CF_2, Flip, ElementwiseAffine.
        # Every Flip is folded into the spline kernel of a neighbouring ConvFlow.
        n = self.n_flows
        for i in range(n, 1, -1):                               # CF_n ... CF_2 = self.flows[2i - 1]
            z = self.flows[2 * i - 1](z, lens, g=h, reverse=True, flip_in=(i == n), flip_out=True)
        z = self.flows[0](z, lens, reverse=True)
        return z[:, :1].contiguous()
