#!/bin/bash
# Round 2, visit AJ: row group as the fastest grid index of the multi-row-group convs -- parity, per-layer A/B, bench A/B, PMC traffic
OUT=gpurun_out/r2_aj
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_f16x3_kernels.py tests/test_gpu_conv.py tests/test_gpu_fuzz.py tests/test_gpu_generator.py tests/test_gpu_full_size.py tests/test_gpu_vits.py -x -q 2>&1 | tail -6 ) > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for m in 0 1 0 1; do
  echo "# AMP_CONV_RG_FAST=$m" >> $OUT/conv_bench.txt
  AMP_CONV_RG_FAST=$m timeout 200 python tools/conv_bench.py --precision f16x3 --reps 20 --only rg >> $OUT/conv_bench.txt 2>> $OUT/conv_bench.err
done
cat $OUT/conv_bench.txt
for m in 1 0 1 0; do
  echo "# AMP_CONV_RG_FAST=$m" >> $OUT/bench.txt
  ( AMP_CONV_RG_FAST=$m timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
cat $OUT/bench.txt
for m in 1 0; do
  echo "# AMP_CONV_RG_FAST=$m" >> $OUT/other.txt
  AMP_CONV_RG_FAST=$m timeout 200 python tools/bench_configs.py --only c3 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
  AMP_CONV_RG_FAST=$m timeout 200 python tools/bench_configs.py --only c5 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
done
cut -c1-200 $OUT/other.txt
du -sh $OUT
