#!/bin/bash
# Round 2, visit Y: conv + Activation1d in one launch (conv_f16x3.hip ACT variant): parity, then C3 A/B with kernel stats
OUT=gpurun_out/r2_y
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_bigvgan.py -m gpu -q -x --timeout 600 -k "fused" 2>&1 | tail -15 ) > $OUT/pytest_fused.txt; cat $OUT/pytest_fused.txt
( timeout 900 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py tests/test_gpu_range_guard.py tests/test_gpu_inference_api.py -m gpu -q --timeout 600 2>&1 | tail -6 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for v in 1 0; do
  ( cd /tmp && AMP_FUSE_ACT=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c3_$v -o kt -- python $REPO/tools/bench_configs.py --only c3 --reps 5 > $REPO/$OUT/c3_$v.json 2> $REPO/$OUT/c3_$v.err )
  echo "== AMP_FUSE_ACT=$v $(cat $OUT/c3_$v.json)"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/c3_$v/kt_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per forward ms", round(tot/6e6, 2))
for r in rows[:14]:
    print(r["Name"].replace("void amp::","")[:70], r["Calls"], round(float(r["TotalDurationNs"])/6e6,3), round(float(r["AverageNs"])/1e3,1))
PY
  rm -f $OUT/c3_$v/kt_kernel_trace.csv
done
AMP_FUSE_ACT=1 python tools/bench_configs.py --only c3 --reps 10
AMP_FUSE_ACT=0 python tools/bench_configs.py --only c3 --reps 10
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
