#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, PMC traffic pass.
# usage (from repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest_gpu.txt
( timeout 600 python tools/conv_bench.py > $OUT/conv_bench.csv 2> $OUT/conv_bench.err )
( timeout 300 python bench.py --steps 10 --warmup 3 --precision f32 --no-cpu-baseline 2> $OUT/bench_f32.err | tail -1 ) > $OUT/bench_f32.json
( timeout 600 python bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench.json
REPO=$PWD
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o pf -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_fetch.err )
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o pw -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_write.err )
# keep only the small summaries


tail -12 $OUT/pytest_gpu.txt
cat $OUT/conv_bench.csv; tail -3 $OUT/conv_bench.err
cat $OUT/bench.json
# SQ counter pass over one bench step (MFMA busy / stall split per kernel)
( cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $REPO/$OUT/pmc_sq -o sq -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_sq.err )
find $OUT -name "*.db" -delete 2>/dev/null; du -sh $OUT
