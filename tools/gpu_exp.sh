#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_generator.py tests/test_gpu_bigvgan.py -q -x 2>&1 | tail -5 ) > $OUT/pytest.txt
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.txt | tail -1 ) > $OUT/bench.json
( timeout 300 python tools/conv_bench.py --precision f16x3 2>> $OUT/err.txt | grep "convT\|prec" > $OUT/conv_bench.csv )
cat $OUT/pytest.txt
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
r=d["roofline"]
print(round(d["ms_per_step"],2), "ms  mrf", round(r["mrf_stack"]["ms"],2), [round(v,2) for v in r["mrf_stack"]["ms_per_stage"]])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k not in ("mrf_stack","kernel","peak_note","traffic_note","sustained_note")})
PY
cat $OUT/conv_bench.csv; tail -3 $OUT/err.txt
