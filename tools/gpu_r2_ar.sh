#!/bin/bash
# Round 2, visit AR: are the small-C pairs bound by HBM?  The same launches at B = 64 (537-MB tensors: HBM) and B = 8 / 16 (67 / 134 MB: x and y stay in the
# 256-MB Infinity Cache across the repetitions); per-item time
OUT=gpurun_out/r2_ar
mkdir -p $OUT
export TMPDIR=/tmp
for b in 64 16 8 64 8; do
  echo "# batch $b" >> $OUT/pair_bench.txt
  timeout 200 python tools/pair_bench.py --reps 40 --batch $b --C 64 32 --k 3 7 11 --d 3 --modes -1 >> $OUT/pair_bench.txt 2>> $OUT/err.txt
done
cat $OUT/pair_bench.txt
