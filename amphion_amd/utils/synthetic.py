"""Seeded random initialisation for benchmarking without checkpoints (there is no network for weights).

The reference's own init (``init_weights`` modules/vocoder_blocks/gan_utils.py:25-28: sigma 0.01 on weight_v,
g = ||v||) gives |out| <= 0.06 and bias-dominated signals; this is the variance-preserving scheme of
SURVEY.md §8(d) instead, so that activations stay O(1) through the whole generator:

    weight_v ~ N(0,1), weight_g ~ gain * U(0.7, 1.3), bias ~ N(0, 0.05), Snake alpha/beta ~ N(0, 0.3),
    un-normed weight ~ N(0,1)/sqrt(fan_in); LayerNorm gamma ~ 1 + N(0, 0.2), ElementwiseAffine m/logs ~ N(0, 0.1),
    emb_rel_k/v ~ N(0,1)/sqrt(dk);   anti-aliasing ``filter`` buffers are left as constructed.

Each tensor comes from its own generator seeded by crc32(key) ^ seed, so values depend neither on key
order nor on which other tensors exist (same values as the test suite's fixtures for the same keys).
"""
from __future__ import annotations

import zlib

import torch


def synthetic_tensor(key: str, shape, seed: int = 1234, g_gain: float = 1.0):
    gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf == "weight_v":
        return torch.randn(shape, generator=gen)
    if leaf == "weight_g":
        return g_gain * (0.7 + 0.6 * torch.rand(shape, generator=gen))
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=gen)
    if leaf in ("alpha", "beta"):
        return 0.3 * torch.randn(shape, generator=gen)
    # VITS text side (round 4: these three used to keep the constructor's UNSEEDED draws, so the synthetic VITS workload -- and its
    # frame count, 2389..2404 -- changed from process to process)
    if leaf == "gamma":  # LayerNorm scale
        return 1.0 + 0.2 * torch.randn(shape, generator=gen)
    if leaf in ("m", "logs"):  # ElementwiseAffine
        return 0.1 * torch.randn(shape, generator=gen)
    if leaf in ("emb_rel_k", "emb_rel_v"):  # relative-position embeddings
        return torch.randn(shape, generator=gen) * shape[-1] ** -0.5
    if leaf == "weight":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return torch.randn(shape, generator=gen) / fan_in**0.5
    return None  # buffers such as the Kaiser-sinc filters keep their constructed value


@torch.no_grad()
def randomize_(module: torch.nn.Module, seed: int = 1234, g_gain: float = 1.0):
    """In-place synthetic init of every parameter of ``module`` (by state_dict key); returns the module."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        t = synthetic_tensor(k, v.shape, seed, g_gain)
        new[k] = v if t is None else t.to(v.dtype)
    module.load_state_dict(new)
    return module


def synthetic_mel(B: int, n_mel: int, T: int, seed: int = 0):
    """Log-mel-like input: randn * 2 - 5."""
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(B, n_mel, T, generator=gen) * 2 - 5
