#!/bin/bash
# Round 2, visit AS: the row-blocked kernel with an A-fragment ring for k = 7 / 11 (AMP_CONV_BLK=3) against the default (2) -- parity, per-layer, bench
OUT=gpurun_out/r2_as
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_f16x3_kernels.py -x -q 2>&1 | tail -4 ) > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
for m in 2 3 2 3; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/conv_bench.txt
  AMP_CONV_BLK=$m timeout 200 python tools/conv_bench.py --precision f16x3 --reps 20 --only rg >> $OUT/conv_bench.txt 2>> $OUT/err.txt
done
grep -v convT $OUT/conv_bench.txt
for m in 3 2 3 2; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/bench.txt
  ( AMP_CONV_BLK=$m timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
cat $OUT/bench.txt
