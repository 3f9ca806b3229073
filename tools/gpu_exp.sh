#!/bin/bash
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_inference_api.py -q -x 2>&1 | tail -4 ) > $OUT/pytest.txt
( timeout 600 python tools/bench_configs.py --reps 5 --only c3 2> $OUT/bench_configs.err ) > $OUT/bench_c3.jsonl
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/tools/bench_configs.py --reps 3 --only c3 > /dev/null 2> $REPO/$OUT/prof.err )
cat $OUT/pytest.txt; cat $OUT/bench_c3.jsonl; cut -c1-150 $OUT/prof/kt_kernel_stats.csv | head -24
