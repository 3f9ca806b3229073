"""Multi-GPU sharding of vocoder inference: one process per GPU, utterances sharded along the batch
dimension, no data-path collective -- only the final result gather over RCCL/xGMI.

The reference shards its DataLoader with ``accelerator.prepare`` and never gathers
(vocoder_inference.py:151-154); each rank writes its own wavs.  Here the root optionally collects
the audio so a caller sees the same [B, L] tensor a single-GPU run returns.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous block partition of ``n_items`` utterances; ragged tail allowed (first ranks get +1)."""
    q, r = divmod(n_items, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(mels: torch.Tensor, world_size: int = None, rank: int = None) -> torch.Tensor:
    """This rank's contiguous slice of a [B, n_mel, T] batch."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    s, e = shard_bounds(mels.shape[0], world_size, rank)
    return mels[s:e]


# element types RCCL / gloo move natively; anything else (16-bit PCM) travels as raw bytes -- a gather only copies
_WIRE_DTYPES = {torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64, torch.int8, torch.uint8}


def gather_audio(local: torch.Tensor, total_items: int, dst: int = 0, group=None, async_op: bool = False):
    """Gather per-rank [b_r, L] audio (fp32, or int16 PCM from ``utils.io.wav_to_pcm16``: 2 B per sample on the
    links) on ``dst`` in rank order -> [total_items, L] (None elsewhere).

    Shards may be ragged by one item: every rank pads to the largest shard so the collective is a
    fixed-size gather (each peer sends over its own xGMI link to the root; no ring, no reduction).

    ``async_op=True`` (equal shards only) returns ``(result, work)`` without making the compute stream wait:
    the transfer then overlaps the vocoding of the NEXT batch; call ``work.wait()`` before reading ``result``.
    """
    if local.dtype not in _WIRE_DTYPES:
        if local.dim() < 2:
            raise ValueError("gather_audio expects [items, samples] rows")
        got = gather_audio(local.contiguous().view(torch.uint8), total_items, dst=dst, group=group, async_op=async_op)
        res, work = got if async_op else (got, None)
        res = None if res is None else res.view(local.dtype)
        return (res, work) if async_op else res
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_bounds(total_items, world, r)[1] - shard_bounds(total_items, world, r)[0] for r in range(world)]
    mx = max(counts)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} items, expected {counts[rank]}")
    send = local
    if local.shape[0] < mx:
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == dst:
        if min(counts) == mx:
            # equal shards: receive straight into the row blocks of the result (no staging copies)
            out = torch.empty((total_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            work = dist.gather(send, list(out.split(mx, dim=0)), dst=dst, group=group, async_op=async_op)
            return (out, work) if async_op else out
        if async_op:
            raise ValueError("async gather needs equal shards")
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    work = dist.gather(send, None, dst=dst, group=group, async_op=async_op)
    return (None, work) if async_op else None


def sharded_vocoder_forward(model, mels: torch.Tensor, gather: bool = True, dst: int = 0):
    """Run ``model`` on this rank's shard of ``mels`` ([B, n_mel, T], identical on every rank) and
    gather the [B, L] audio on ``dst``.  ``model`` and the shard live on this rank's device."""
    total = mels.shape[0]
    local_in = shard_batch(mels)
    dev = next(model.parameters()).device
    with torch.no_grad():
        if local_in.shape[0] > 0:
            out = model(local_in.to(dev)).squeeze(1)
        else:
            out = torch.zeros((0, mels.shape[-1] * model.hop_factor), device=dev)
    if not gather:
        return out
    return gather_audio(out, total, dst=dst)
