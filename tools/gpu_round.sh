#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC traffic passes, other configs.
# usage (from repo root on the GPU box): [PYTEST_N=4] [FULL=1] bash tools/gpu_round.sh <tag>
#   PYTEST_N  pytest-xdist workers for the GPU suite (the tests spend most of their time in the CPU oracle)
#   FULL=1    also the per-layer conv microbench and the SQ counter pass
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
NW=${PYTEST_N:-0}
if [ "$NW" != "0" ]; then XD="-n $NW"; else XD=""; fi
( timeout 1500 python -m pytest tests -m gpu -x -q $XD 2>&1 | tail -40 ) > $OUT/pytest_gpu.txt   # -x, one process: what the driver runs at round end
tail -5 $OUT/pytest_gpu.txt
( timeout 600 python bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/bench.json
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o pf -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_fetch.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o pw -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_write.err )
( timeout 300 python bench.py --steps 10 --warmup 3 --precision f32 --no-cpu-baseline 2> $OUT/bench_f32.err | tail -1 ) > $OUT/bench_f32.json
( timeout 400 python tools/bench_configs.py --reps 10 > $OUT/other_configs.jsonl 2> $OUT/other_configs.err )
cat $OUT/other_configs.jsonl
( timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > $OUT/smoke.txt; cat $OUT/smoke.txt
if [ "$FULL" = "1" ]; then
  ( timeout 600 python tools/conv_bench.py > $OUT/conv_bench.csv 2> $OUT/conv_bench.err )
  # SQ counter pass over one bench step (MFMA busy / stall split per kernel)
  ( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $REPO/$OUT/pmc_sq -o sq -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_sq.err )
fi
find $OUT -name "*.db" -delete 2>/dev/null; du -sh $OUT
