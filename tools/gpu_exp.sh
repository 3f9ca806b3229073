#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/tools/bench_configs.py --reps 3 --only c5 > $REPO/$OUT/c5.jsonl 2> $REPO/$OUT/prof.err )
cat $OUT/c5.jsonl; cut -c1-160 $OUT/prof/kt_kernel_stats.csv | head -40
