#!/bin/bash
# Round 2, visit K: the 2 x 2-blocked 4-wave pair variant (64 rows x 96 columns per wave, one workgroup per CU, 512 registers)
OUT=gpurun_out/r2_k
mkdir -p $OUT
export TMPDIR=/tmp
for v in 5 6 7; do
  ( AMP_STRIP_C128=$v timeout 600 python -m pytest tests/test_gpu_pair.py -m gpu -q -k "policy" --timeout 300 2>&1 | tail -3 ) > $OUT/pytest_c128_$v.txt; echo "variant $v: $(tail -1 $OUT/pytest_c128_$v.txt)"
done
for rep in 1 2; do
for v in 0 5 6 7; do
  ( AMP_STRIP_C128=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/bench_c128_${v}_$rep.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_k/bench_c128_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], round(d["ms_per_step"],3), "dom_us", round(r["launch_us"],1), "stages", [round(v,2) for v in r["mrf_stack"]["ms_per_stage"]])
    except Exception as e: print(f, "ERR", e)
PY
( timeout 900 python -m pytest tests/test_gpu_range_guard.py tests/test_gpu_pair.py tests/test_gpu_generator.py tests/test_gpu_full_size.py -m gpu -q --timeout 400 2>&1 | tail -3 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
