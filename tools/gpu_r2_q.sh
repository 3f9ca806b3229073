#!/bin/bash
# Round 2, visit Q: the 2 x 2-blocked variant for the C = 64 pairs (4 waves x (64 rows x 64 columns), one workgroup per CU)
OUT=gpurun_out/r2_q
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for v in 5 6 7; do
  ( AMP_STRIP_C64=$v timeout 600 python -m pytest tests/test_gpu_pair.py -m gpu -q -k "policy" --timeout 300 2>&1 | tail -1 ) > $OUT/pytest_c64_$v.txt; echo "variant $v: $(cat $OUT/pytest_c64_$v.txt)"
done
for v in 0 6 7; do
  ( cd /tmp && AMP_STRIP_C64=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/p$v -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/bench_$v.json 2> $REPO/$OUT/p$v.err )
  rm -f $OUT/p$v/kt_kernel_trace.csv
  python - <<PY
import csv, json
print("== AMP_STRIP_C64=$v", json.load(open("$OUT/bench_$v.json"))["ms_per_step"])
for r in csv.DictReader(open("$OUT/p$v/kt_kernel_stats.csv")):
    n = r["Name"]
    if ("pair_f16x3_kernel<" in n and ", 2, 2, 2" in n) or ("pair_strip_kernel" in n and "1, 4, 2, 320, 2" in n):
        print(n.replace("void amp::","")[:52], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
