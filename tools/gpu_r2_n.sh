#!/bin/bash
# Round 2, visit N: where the VITS decode path (C5) spends its time -- kernel sum vs wall
OUT=gpurun_out/r2_n
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c5 -o kt -- python $REPO/tools/bench_configs.py --only c5 --reps 10 > $REPO/$OUT/c5.json 2> $REPO/$OUT/c5.err )
cat $OUT/c5.json
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r2_n/c5/kt_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total ms", tot/1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:14]: print(r["Name"][:90], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), round(float(r["AverageNs"])/1e3,1))
PY
rm -f $OUT/c5/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
