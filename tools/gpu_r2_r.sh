#!/bin/bash
# Round 2, visit R: whole-K frame-rate conv kernel (conv_small_f16x3.hip) + fused WN layer: parity, then C5 A/B with per-kernel stats
OUT=gpurun_out/r2_r
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_vits.py tests/test_gpu_vits_infer.py tests/test_gpu_apnet.py -m gpu -q -x --timeout 300 2>&1 | tail -15 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for v in 1 0; do
  ( cd /tmp && AMP_SMALL_CONV=$v AMP_WN_FUSED=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c5_$v -o kt -- python $REPO/tools/bench_configs.py --only c5 --reps 10 > $REPO/$OUT/c5_$v.json 2> $REPO/$OUT/c5_$v.err )
  echo "== AMP_SMALL_CONV=$v AMP_WN_FUSED=$v"; cat $OUT/c5_$v.json; tail -3 $OUT/c5_$v.err
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/c5_$v/kt_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total ms", round(tot/1e6,2), "calls", sum(int(r["Calls"]) for r in rows))
for r in rows:
    n = r["Name"]
    if "pair_" in n: continue
    print(n.replace("void amp::","")[:80], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), round(float(r["AverageNs"])/1e3,1))
PY
  rm -f $OUT/c5_$v/kt_kernel_trace.csv
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
