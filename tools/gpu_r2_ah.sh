#!/bin/bash
# Round 2, visit AH: the row-blocked conv kernel (conv_blk_f16x3.hip) -- parity, per-layer A/B, bench A/B on one box
OUT=gpurun_out/r2_ah
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_f16x3_kernels.py tests/test_gpu_conv.py tests/test_gpu_fuzz.py tests/test_gpu_generator.py tests/test_gpu_full_size.py tests/test_gpu_bigvgan.py tests/test_gpu_inference_api.py -x -q 2>&1 | tail -15 ) > $OUT/pytest.txt
tail -4 $OUT/pytest.txt
for m in 0 1 2 0 2; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/conv_bench.txt
  AMP_CONV_BLK=$m timeout 200 python tools/conv_bench.py --precision f16x3 --reps 20 --only blk >> $OUT/conv_bench.txt 2>> $OUT/conv_bench.err
done
cat $OUT/conv_bench.txt
for m in 2 0 1 2 0; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/bench.txt
  ( AMP_CONV_BLK=$m timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
cat $OUT/bench.txt
for m in 2 0; do
  echo "# AMP_CONV_BLK=$m" >> $OUT/other.txt
  AMP_CONV_BLK=$m timeout 200 python tools/bench_configs.py --only c3 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
  AMP_CONV_BLK=$m timeout 200 python tools/bench_configs.py --only c5 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
done
cat $OUT/other.txt | cut -c1-300
du -sh $OUT
