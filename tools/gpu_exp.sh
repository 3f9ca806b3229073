#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_inference_api.py tests/test_gpu_nsf.py -q -x 2>&1 | tail -4 ) > $OUT/pytest.txt
cat $OUT/pytest.txt
for f in 1 0 1 0; do
( AMP_CONCURRENT_RB=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.txt | tail -1 ) > $OUT/bench_$f.json
python - <<PY
import json
d=json.load(open("$OUT/bench_$f.json")); r=d["roofline"]
print("concurrent_rb=$f", round(d["ms_per_step"],2), "ms   (profiled sequential mrf", round(r["mrf_stack"]["ms"],2), ")")
PY
done
