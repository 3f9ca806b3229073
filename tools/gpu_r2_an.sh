#!/bin/bash
# Round 2, visit AN: ping-pong order also for the act1d launches (BigVGAN) -- parity, C3 A/B
OUT=gpurun_out/r2_an
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bigvgan.py -x -q 2>&1 | tail -4 ) > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
for m in 1 0 1 0 1 0; do
  echo "# AMP_PINGPONG=$m" >> $OUT/other.txt
  AMP_PINGPONG=$m timeout 200 python tools/bench_configs.py --only c3 --reps 20 >> $OUT/other.txt 2>> $OUT/other.err
done
cut -c1-130 $OUT/other.txt
