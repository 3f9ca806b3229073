#!/bin/bash
# Round 2, visit H: range guard, lazy notices, new mel kernel in the whole suite; bench line; mel kernel timing
OUT=gpurun_out/r2_h
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_range_guard.py tests/test_gpu_mel.py tests/test_gpu_mel_loss.py -m gpu -q -s --timeout 300 2>&1 | tail -60 ) > $OUT/pytest_new.txt
grep "\[range\]\|passed\|failed\|Error" $OUT/pytest_new.txt | head -30
( timeout 1500 python -m pytest tests -m gpu -q -n 4 --timeout 600 2>&1 | tail -30 ) > $OUT/pytest_gpu.txt
tail -8 $OUT/pytest_gpu.txt
( timeout 600 python bench.py --steps 20 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench.json
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['launch_us']); print(json.dumps(d['other_configs']['mel_front_end'])); print([x['ms'] for x in d['other_configs']['latency']])"
