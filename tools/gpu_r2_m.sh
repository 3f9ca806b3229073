#!/bin/bash
# Round 2, visit M: streaming conv_post, device-pointer weights, per-handle range flags in the whole suite; v_sin_f32 accuracy probe
OUT=gpurun_out/r2_m
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 60 tests/experiments/vsin_accuracy > $OUT/vsin_accuracy.txt 2>&1; cat $OUT/vsin_accuracy.txt
( timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -6 ) > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
( timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > $OUT/smoke.txt; cat $OUT/smoke.txt
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
grep "conv_post" $OUT/prof/kt_kernel_stats.csv | cut -c1-200
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/bench.json
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['traffic'], d['roofline']['kernel'][:60])"
rm -f $OUT/prof/kt_kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
