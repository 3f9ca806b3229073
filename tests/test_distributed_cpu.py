"""N>1 path on CPU: world_size-2 gloo processes shard a batch, run a stand-in generator and gather the
audio on rank 0; the gathered tensor must equal the single-process result bit-for-bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from amphion_amd.distributed import gather_audio, shard_bounds, sharded_vocoder_forward


class _StubGenerator(torch.nn.Module):
    """Deterministic stand-in with the generator contract ([B, C, T] -> [B, 1, T*hop]); the real
    kernels need a GPU and are covered by the -m gpu tests."""

    hop_factor = 4

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.arange(1.0, 4.0))

    def forward(self, x):
        y = (x * self.w.view(1, -1, 1)).sum(1, keepdim=True)
        return torch.tanh(y.repeat_interleave(self.hop_factor, dim=-1))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 8, 64, 511, 512):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        mels = torch.randn(n_items, 3, 6)
        model = _StubGenerator()
        out = sharded_vocoder_forward(model, mels)
        if rank == 0:
            ref = model(mels).squeeze(1)
            q.put((out.shape == ref.shape) and bool(torch.equal(out, ref)))
        else:
            assert out is None
        # plain gather with explicit ragged shards
        s, e = shard_bounds(n_items, world, rank)
        g = gather_audio(torch.full((e - s, 2), float(rank)), n_items)
        if rank == 0:
            q.put(g[:, 0].tolist())
        # asynchronous gather of equal shards (what bench.py does for N > 1): several in flight, waited later
        if n_items % world == 0:
            per = n_items // world
            inflight = [gather_audio(torch.full((per, 3), float(10 * k + rank)), n_items, async_op=True) for k in range(3)]
            for k, (res, work) in enumerate(inflight):
                work.wait()
                if rank == 0:
                    want = torch.cat([torch.full((per, 3), float(10 * k + r)) for r in range(world)])
                    assert torch.equal(res, want)
                else:
                    assert res is None
        # 16-bit PCM rows (amphion_amd.utils.io.wav_to_pcm16 output) travel through the same gather: 2 B / sample
        pcm = torch.arange((e - s) * 5, dtype=torch.int16).reshape(e - s, 5) + 1000 * rank
        gp = gather_audio(pcm, n_items)
        if rank == 0:
            want = torch.cat([torch.arange((shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0]) * 5,
                                           dtype=torch.int16).reshape(-1, 5) + 1000 * r for r in range(world)])
            assert gp.dtype == torch.int16 and torch.equal(gp, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 5, 1])
def test_two_rank_shard_and_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
    ranks = q.get(timeout=10)
    s0 = shard_bounds(n_items, 2, 0)
    assert ranks == [0.0] * (s0[1] - s0[0]) + [1.0] * (n_items - (s0[1] - s0[0]))


def test_eight_rank_ragged_tail_b509():
    """BASELINE configs[3]'s world size with a batch that does not divide: B = 509 over 8 ranks (five shards of 64, three of 63) --
    sharded forward + gather equals the single-process result bit for bit, and the PCM rows arrive in rank order."""
    n_items, world = 509, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=20) is True
    ranks = q.get(timeout=20)
    want = []
    for r in range(world):
        s, e = shard_bounds(n_items, world, r)
        want += [float(r)] * (e - s)
    assert ranks == want
    sizes = [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0] for r in range(world)]
    assert sizes == [64] * 5 + [63] * 3
