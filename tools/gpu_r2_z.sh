#!/bin/bash
# Round 2, visit Z: packed staging / seam conversions (stage4_f16, seam4_f16) vs the scalar form, same box:
# AMP_LIB_PATH=amphion_amd/lib/libamphion_hip_base.so is the library built from the commit before
OUT=gpurun_out/r2_z
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_conv.py tests/test_gpu_generator.py tests/test_gpu_fuzz.py tests/test_gpu_range_guard.py tests/test_gpu_bigvgan.py -m gpu -q -x --timeout 600 2>&1 | tail -5 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for rep in 1 2; do
for v in new base; do
  if [ $v = base ]; then export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_base.so; else unset AMP_LIB_PATH; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err || tail -3 $OUT/bench_${v}_$rep.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_${v}_$rep.json"))
r = d["roofline"]
print("$v $rep: ms/step", round(d["ms_per_step"], 3), "dominant launch us", round(r["launch_us"], 1), "frac", round(r["frac"], 4), "stages", [round(x, 2) for x in r["mrf_stack"]["ms_per_stage"]])
PY
done
done
unset AMP_LIB_PATH
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof_new -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
( cd /tmp && AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_base.so timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof_base -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python - <<'PY'
import csv
for v in ("new", "base"):
    print("==", v)
    for r in list(csv.DictReader(open(f"gpurun_out/r2_z/prof_{v}/kt_kernel_stats.csv")))[:14]:
        print("  ", r["Name"].replace("void amp::", "")[:58], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -f $OUT/prof_*/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
