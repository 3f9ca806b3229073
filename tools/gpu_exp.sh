#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_nsf.py tests/test_gpu_inference_api.py tests/test_gpu_melgan.py -q -x 2>&1 | tail -8 ) > $OUT/pytest.txt
cat $OUT/pytest.txt
