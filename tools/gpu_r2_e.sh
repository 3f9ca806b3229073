#!/bin/bash
# Round 2, visit E: software-pipelined B fragments in the fused pair (timing-only harness), C1 16-clip test, full GPU suite
OUT=gpurun_out/r2_e
mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2 3; do
  for k in 3 7 11; do for m in 0 512 1024; do timeout 60 tests/experiments/pair_ablate_k${k}_$m 5; done; done
done > $OUT/pair_pipeline.txt 2>&1
sort $OUT/pair_pipeline.txt | grep "dil 5"
( timeout 600 python -m pytest tests/test_gpu_c1_clips.py -m gpu -q -x -s --timeout 300 2>&1 | tail -30 ) > $OUT/pytest_c1.txt
tail -12 $OUT/pytest_c1.txt
( timeout 1500 python -m pytest tests -m gpu -q -n 4 --timeout 600 2>&1 | tail -30 ) > $OUT/pytest_gpu.txt
tail -8 $OUT/pytest_gpu.txt
