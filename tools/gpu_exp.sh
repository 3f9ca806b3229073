#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_conv.py tests/test_gpu_generator.py -q -s 2>&1 | grep -v "^C=\|Removing" | tail -25 ) > $OUT/pytest.txt
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/err.txt | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.txt
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(round(d["ms_per_step"],2), "ms  mrf", round(d["roofline"]["mrf_ms"],2), [round(v,2) for v in d["roofline"]["mrf_ms_per_stage"]])
PY
