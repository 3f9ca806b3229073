#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_inference_api.py tests/test_gpu_bigvgan.py tests/test_gpu_generator.py -q -x 2>&1 | tail -15 ) > $OUT/pytest.txt
( timeout 600 python tools/bench_configs.py 2> $OUT/bench_configs.err ) > $OUT/bench_configs.jsonl
cat $OUT/pytest.txt; cat $OUT/bench_configs.jsonl; tail -5 $OUT/bench_configs.err
