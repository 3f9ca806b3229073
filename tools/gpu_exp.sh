#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_mel.py -q -x 2>&1 | tail -15 ) > $OUT/pytest.txt
( timeout 120 ./tests/experiments/mfma_peak ) > $OUT/mfma_peak.txt 2>&1
cat $OUT/pytest.txt; cat $OUT/mfma_peak.txt
