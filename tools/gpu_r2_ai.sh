#!/bin/bash
# Round 2, visit AI: k = 3 fused pairs at C <= 64 with three workgroups per CU (pair_f16x3.hip, OCC = 3) -- parity, A/B
OUT=gpurun_out/r2_ai
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_f16x3_kernels.py -x -q 2>&1 | tail -6 ) > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for m in 0 1 0 1; do
  echo "# AMP_PAIR_OCC3=$m" >> $OUT/pair_bench.txt
  AMP_PAIR_OCC3=$m timeout 200 python tools/pair_bench.py --reps 20 --C 64 32 --k 3 --d 1 3 5 --modes 0 >> $OUT/pair_bench.txt 2>> $OUT/pair_bench.err
done
cat $OUT/pair_bench.txt
for m in 1 0 1 0; do
  echo "# AMP_PAIR_OCC3=$m" >> $OUT/bench.txt
  ( AMP_PAIR_OCC3=$m timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
cat $OUT/bench.txt
du -sh $OUT
