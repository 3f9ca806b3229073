"""Host helpers mirrored from utils/util.py that the vocoder inference path calls."""
import os

import torch


def _cpu_budget():
    """CPUs this process may actually burn: the cgroup quota (cpu.max, cgroup v2; cfs_quota_us, v1) when there is one, else
    its affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


class few_host_threads:
    """Scope for the host-side tensor work of the list / batch entry points (padding, crops): at most ``n`` intra-op threads,
    and never more than a quarter of the CPU budget.  torch sizes its OpenMP pool by the VISIBLE CPUs (128 threads on the 256-CPU
    MI355X hosts) although a container may be allowed 16: every padded copy and crop then wakes 128 spinning threads, the
    cgroup's quota for the 100-ms period is gone and the whole process is parked -- the list API alternated 37 / 60 ms per call with
    the GPU idle (profiles/r3_o_list_api_cgroup_throttle.txt).  These copies are a few MB: four threads are plenty."""

    def __init__(self, n=4):
        self.n = max(1, min(n, _cpu_budget() // 4 or 1))

    # torch.set_num_threads is omp_set_num_threads: the nthreads ICV of the CALLING thread, not process-wide state -- another thread's
    # setting is neither changed nor observed by this scope (tests/test_host_logic.py checks that), so enter / exit only have to save
    # and restore the caller's own value, nesting included.
    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


def pad_mels_to_tensors(mels, batched=None):
    """utils/util.py:114-182 -- zero-pad a list of [n_mel, T_i] mels into batches of ``batched``
    (one batch when None); returns (tensors, mel_frames) exactly as the reference does."""
    tensors, mel_frames = [], []
    n = len(mels)
    step = n if batched is None else batched
    start = 0
    while start < n:
        group = mels[start:start + step]
        size = max(m.shape[-1] for m in group)
        tensor = torch.zeros(len(group), mels[0].shape[0], size)
        frame = torch.zeros(len(group), dtype=torch.int32)
        for i, m in enumerate(group):
            tensor[i, :, : m.shape[-1]] = m[:]
            frame[i] = m.shape[-1]
        tensors.append(tensor)
        mel_frames.append(frame)
        start += step
    return tensors, mel_frames
