#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC traffic pass.
# usage (from repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.txt
( timeout 600 python bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench.json
REPO=$PWD
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o pf -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_fetch.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o pw -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_write.err )
# keep only the small summaries
find $OUT -name '*.db' -delete 2>/dev/null
du -sh $OUT
cat $OUT/pytest_gpu.txt | tail -5
cat $OUT/bench.json
