#!/bin/bash
# Round 2, visit S: whole-K kernel tile width A/B on C5 (AMP_SMALL_NI=1: 128 x 32 tiles, 2: 128 x 64)
OUT=gpurun_out/r2_s
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for v in 1 2; do
  ( cd /tmp && AMP_SMALL_NI=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c5_$v -o kt -- python $REPO/tools/bench_configs.py --only c5 --reps 10 > $REPO/$OUT/c5_$v.json 2> $REPO/$OUT/c5_$v.err )
  echo "== AMP_SMALL_NI=$v"; cat $OUT/c5_$v.json
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/c5_$v/kt_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total ms", round(tot/1e6,2), "calls", sum(int(r["Calls"]) for r in rows))
for r in rows:
    n = r["Name"]
    if "conv_small" in n or "copyBuffer" in n or "mask" in n:
        print(n.replace("void amp::","")[:80], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), round(float(r["AverageNs"])/1e3,1))
PY
  rm -f $OUT/c5_$v/kt_kernel_trace.csv
done
python tools/bench_configs.py --only c5 --reps 20
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
