#!/bin/bash
# Round 2, visit AC: act1d edge tiles through float4 windows (stage-0 rows): parity + C3 kernel stats
OUT=gpurun_out/r2_ac
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py tests/test_gpu_resample.py tests/test_gpu_f16x3_kernels.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c3 -o kt -- python $REPO/tools/bench_configs.py --only c3 --reps 5 > $REPO/$OUT/c3.json 2> $REPO/$OUT/c3.err )
cat $OUT/c3.json
grep act1d $OUT/c3/kt_kernel_stats.csv | cut -c1-45,150-300
python - <<'PY'
import csv, collections
rows = [r for r in csv.DictReader(open("gpurun_out/r2_ac/c3/kt_kernel_trace.csv")) if "act1d" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows: by[(r["Kernel_Name"][10:30], r["Grid_Size"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()): print(k, len(v), "avg us", round(sum(v) / len(v), 1))
PY
rm -f $OUT/c3/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
python tools/bench_configs.py --only c3 --reps 10
