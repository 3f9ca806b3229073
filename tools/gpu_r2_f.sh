#!/bin/bash
# Round 2, visit F: wide (8-wave, one workgroup per CU) fused-pair tiles / short strips; new host-side tests
OUT=gpurun_out/r2_f
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 200 python tools/pair_bench.py --reps 5 --C 128 64 --modes 0 ) > $OUT/tile.csv 2>&1
grep -v amdgpu $OUT/tile.csv
for steps in 1 2 4 0; do
  echo "== wide, steps per strip $steps (0 = planner)"
  AMP_STRIP_WIDE=1 AMP_STRIP_STEPS=$steps timeout 200 python tools/pair_bench.py --reps 5 --C 128 64 --modes 1 | grep -v "amdgpu\|^C,k"
done > $OUT/wide.txt 2>&1
cat $OUT/wide.txt
for steps in 1 2 4; do
  echo "== narrow (4-wave) strips, steps per strip $steps"
  AMP_STRIP_STEPS=$steps timeout 200 python tools/pair_bench.py --reps 5 --C 256 128 --modes 1 | grep -v "amdgpu\|^C,k"
done > $OUT/narrow.txt 2>&1
cat $OUT/narrow.txt
( timeout 900 python -m pytest tests/test_gpu_inference_api.py tests/test_gpu_nsf.py tests/test_gpu_pair.py -m gpu -q --timeout 300 2>&1 | tail -15 ) > $OUT/pytest_some.txt
tail -5 $OUT/pytest_some.txt
