#!/bin/bash
# Round 2, visit AV: 128-row workgroups of the row-blocked kernel for M = 128 convs (AMP_CONV_BLK=4) -- parity of the four new shapes, per-layer A/B at C = 128
OUT=gpurun_out/r2_av
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 60 python -m pytest tests/test_gpu_f16x3_kernels.py -x -q -k "blocked and (128-128 or 100-384)" 2>&1 | tail -3 ) > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
for m in 3 4; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/conv_bench.txt
  AMP_CONV_BLK=$m timeout 40 python tools/conv_bench.py --precision f16x3 --reps 10 --batch 32 --only c128 2>> $OUT/err.txt | grep "^f16x3" >> $OUT/conv_bench.txt
done
cat $OUT/conv_bench.txt
