export CMD='
python -c "import torch, os; print(\"threads\", torch.get_num_threads(), \"cpus\", os.cpu_count(), \"affinity\", len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc
echo "## OMP_NUM_THREADS=1"; OMP_NUM_THREADS=1 timeout 300 python tools/list_api_probe.py --elim | head -4
echo "## OMP_NUM_THREADS=8"; OMP_NUM_THREADS=8 timeout 300 python tools/list_api_probe.py --elim | head -3
echo "## default"; timeout 300 python tools/list_api_probe.py --elim | head -3
'
bash tools/gpu_round.sh r3_o cmd
