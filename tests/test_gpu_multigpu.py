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
