import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is torch on the host cores.  On a 256-thread GPU host torch's default (all hardware threads)
    # is ~40x slower than 16 threads for these conv shapes (tests/experiments/cpu_threads_sweep.py), and xdist
    # workers would oversubscribe further: cap it.
    import torch

    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(params=["f16x3", "f32"])
def conv_precision(request):
    """Runs a GPU test once per arithmetic of the conv contractions (include/amphion_hip.h: amp_precision);
    both must meet the same parity bounds.  Handles pick the mode up when they are created."""
    from amphion_amd import _lib

    _lib.set_precision(request.param)
    yield request.param
    _lib.set_precision("f16x3")
