#!/bin/bash
# Round 2, visit I: BigVGAN with the resblocks of a stage on concurrent streams (VALU-bound activations under MFMA-bound convs)
OUT=gpurun_out/r2_i
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py tests/test_gpu_inference_api.py tests/test_gpu_range_guard.py tests/test_gpu_mel.py tests/test_gpu_mel_loss.py -m gpu -q -s --timeout 400 2>&1 | grep "\[range\]\|\[mel-loss\]\|passed\|failed\|FAILED\|Error" | head -40 ) > $OUT/pytest.txt
cat $OUT/pytest.txt
for rep in 1 2; do
for sv in 0 1; do
  ( AMP_BIGVGAN_STREAMS=$sv timeout 200 python tools/bench_configs.py --only c3 --reps 10 2>&1 | tail -1 ) > $OUT/c3_streams${sv}_$rep.json
  echo "streams=$sv: $(cat $OUT/c3_streams${sv}_$rep.json)"
done; done
( cd /tmp && timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/tools/bench_configs.py --only c3 --reps 5 > /dev/null 2> $REPO/$OUT/prof.err )
head -8 $OUT/prof/kt_kernel_stats.csv | cut -c1-160
rm -f $OUT/prof/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
