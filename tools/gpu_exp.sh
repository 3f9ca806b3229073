#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_generator.py -q -x 2>&1 | tail -4 ) > $OUT/pytest.txt
cat $OUT/pytest.txt
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision f32 2>> $OUT/err.txt | tail -1 ) > $OUT/bench_f32.json
python - <<PY
import json
d=json.load(open("$OUT/bench_f32.json")); r=d["roofline"]
print("f32", round(d["ms_per_step"],2), "ms  mrf", round(r["mrf_stack"]["ms"],2), [round(v,2) for v in r["mrf_stack"]["ms_per_stage"]], "dominant frac", round(r["frac"],3))
PY
