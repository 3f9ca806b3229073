#!/bin/bash
# Round 2, visit X: whole-K kernel with the A-slice touch prefetch: parity + C5 kernel stats
OUT=gpurun_out/r2_x
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_vits.py tests/test_gpu_vits_infer.py -m gpu -q -x --timeout 300 2>&1 | tail -3 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c5 -o kt -- python $REPO/tools/bench_configs.py --only c5 --reps 10 > $REPO/$OUT/c5.json 2> $REPO/$OUT/c5.err )
cat $OUT/c5.json
grep "conv_small\|mask_kernel" $OUT/c5/kt_kernel_stats.csv | awk -F, '{print $1, $2, $(NF-5), $(NF-4)}' | cut -c1-160
rm -f $OUT/c5/kt_kernel_trace.csv
python tools/bench_configs.py --only c5 --reps 20
python tools/bench_configs.py --only c3 --reps 10
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
