#!/bin/bash
# Round 2, visit AA: whole-K kernel for k = 7 / 11 (the C = 256 stage of single utterances): parity + latency A/B
OUT=gpurun_out/r2_aa
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_generator.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_apnet.py tests/test_gpu_melgan.py tests/test_gpu_bigvgan.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for rep in 1 2; do for v in 1 0; do echo "== AMP_SMALL_CONV=$v"; AMP_SMALL_CONV=$v python tools/bench_configs.py --only lat --reps 20 | grep -v hipGraph; done; done
