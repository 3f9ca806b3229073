#!/bin/bash
# Round 2, visit AU: last validation of the final build (AMP_CONV_BLK default 3, ring of 4): whole GPU suite, bench line without the CPU / library legs, kernel stats
OUT=gpurun_out/r2_au
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 80 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $OUT/pytest_gpu.txt
tail -2 $OUT/pytest_gpu.txt
( timeout 40 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> $OUT/bench.err | tail -1 ) > $OUT/bench_nocpu.json
python -c "
import json;d=json.load(open('$OUT/bench_nocpu.json'));r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1), 'frac', round(r['frac'],4))"
( cd /tmp && timeout 40 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/prof.err )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info*" -delete
head -12 $OUT/prof/kt_kernel_stats.csv | cut -c1-120
