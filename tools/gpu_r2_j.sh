#!/bin/bash
# Round 2, visit J: in-forward A/B of the strip policy for the dominant k = 11, C = 128 pairs (bench.py, 20 steps each)
OUT=gpurun_out/r2_j
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 1 2 3 4; do
  ( AMP_STRIP_K11=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/bench_k11_${v}_$rep.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_j/bench_k11_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], round(d["ms_per_step"],3), "dom_us", round(r["launch_us"],1), "stages", [round(v,2) for v in r["mrf_stack"]["ms_per_stage"]])
    except Exception as e: print(f, "ERR", e)
PY
( timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_generator.py -m gpu -q --timeout 300 2>&1 | tail -3 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
