#!/bin/bash
# Round 2, visit G: wave-per-frame radix-8 mel kernel -- parity + timing
OUT=gpurun_out/r2_g
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_mel.py tests/test_gpu_c1_clips.py tests/test_gpu_apnet.py -m gpu -q -s --timeout 300 2>&1 | tail -60 ) > $OUT/pytest_mel.txt
grep "\[mel\]\|\[c1\]\|passed\|failed\|Error\|error" $OUT/pytest_mel.txt | head -60
( timeout 200 python tools/bench_configs.py --only mel --reps 50 2>&1 | tail -1 ) > $OUT/mel_bench.json
cat $OUT/mel_bench.json
( cd /tmp && timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/tools/bench_configs.py --only mel --reps 20 > /dev/null 2> $REPO/$OUT/prof.err )
head -5 $OUT/prof/kt_kernel_stats.csv | cut -c1-200
rm -f $OUT/prof/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
