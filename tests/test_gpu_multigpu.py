"""N > 1 on real hardware: `python bench.py --gpus 2` launches its own two ranks (one process per
GPU, RCCL over xGMI), rank 0 checks that the gathered [128, L] audio equals two single-GPU runs bit for bit, and one
JSON line with n_gpus = 2 comes back.  Needs two visible GPUs: skipped (and says so) on the 1-GPU test boxes; the
sharding / gather logic itself is covered on CPU by tests/test_distributed_cpu.py (gloo, world size 2)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs on the node (this box exposes %d)" % torch.cuda.device_count())
def test_bench_two_ranks_rccl_gather_is_bitwise():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert "bitwise" in d["gather_check"]


def test_multi_gpu_diagnostics_on_a_one_rank_group():
    """The N > 1 diagnostics of the bench line (per-rank ms, generator alone, fp32 / PCM16 gather alone) cannot meet a second GPU on
    these boxes; a one-rank RCCL group at least runs every collective and every key of the code path on the hardware, so that a first
    8-GPU run does not die of a typo."""
    import importlib.util
    import socket

    import torch.distributed as dist

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    try:
        from amphion_amd.utils.synthetic import synthetic_mel

        model, _, _ = bench.build_model(device)
        mel = synthetic_mel(4, bench.N_MEL, 32, seed=0).to(device)
        d = bench.multi_gpu_diagnostics(model, mel, 4, device, [1.5], reps=2)
    finally:
        dist.destroy_process_group()
    for k in ("per_rank_ms", "compute_only_ms", "gather_ms", "gather_ms_overlapped", "gather_ms_pcm16", "pcm16_convert_ms", "gather_GBps_per_link"):
        assert k in d, k
    assert d["per_rank_ms"] == [1.5] and d["compute_only_ms"] > 0 and d["gather_ms"] > 0 and d["gather_ms_pcm16"] > 0
    json.dumps(d)


def test_bench_under_the_drivers_own_launcher_line():
    """The driver starts N > 1 runs as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`; with the one GPU of these boxes the same line at N = 1 must come back with ONE well-formed JSON line (the head-of-line
    summary, the contract's keys, the roofline object)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert list(d)[0] == "summary_ms" and d["summary_ms"]["c2_f16x3_ms"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None and d["config"]["workload"].startswith("HiFi-GAN V1 22.05 kHz")
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] == "mfma"
