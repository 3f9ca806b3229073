#!/bin/bash
# Round 2, visit AD: tiles beyond an utterance's valid length exit at once (ragged batches): parity + list-API timing
OUT=gpurun_out/r2_ad
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_inference_api.py tests/test_gpu_generator.py tests/test_gpu_bigvgan.py tests/test_gpu_fuzz.py tests/test_gpu_pair.py tests/test_gpu_c1_clips.py tests/test_gpu_melgan.py tests/test_gpu_nsf.py tests/test_gpu_apnet.py tests/test_gpu_vits.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
python tools/bench_configs.py --only list --reps 5 | tee $OUT/list.txt
python tools/bench_configs.py --only list --reps 5 | tee -a $OUT/list.txt
