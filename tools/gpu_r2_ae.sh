#!/bin/bash
# Round 2, visit AE: C = 32 fused pairs with both chunks' loads in flight from the start vs the build before (same box)
OUT=gpurun_out/r2_ae
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_generator.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 600 2>&1 | tail -3 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for v in new base; do
  if [ $v = base ]; then export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_base.so; else unset AMP_LIB_PATH; fi
  ( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof_$v -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/bench_$v.json 2> /dev/null )
  python - <<PY
import csv, json
print("== $v", round(json.load(open("$OUT/bench_$v.json"))["ms_per_step"], 3))
for r in csv.DictReader(open("$OUT/prof_$v/kt_kernel_stats.csv")):
    if "1, 4, 2, 320" in r["Name"]: print("  ", r["Name"].replace("void amp::", "")[:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
  rm -f $OUT/prof_$v/kt_kernel_trace.csv
done
unset AMP_LIB_PATH
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
