#!/bin/bash
# Round 2, visit D: why the strip kernel loses at k = 3 / 7 and C <= 64 -- de-phasing experiments; new bench.py line.
OUT=gpurun_out/r2_d
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 300 python tools/pair_bench.py --reps 5 ) > $OUT/pair_bench_base.csv 2>&1
cat $OUT/pair_bench_base.csv
for sm in "1 3" "1 6" "1 10" "2 8" "2 16"; do set -- $sm
  echo "== stagger mode $1 x $2"; AMP_STRIP_STAGGER_MODE=$1 AMP_STRIP_STAGGER=$2 timeout 200 python tools/pair_bench.py --reps 5 --C 128 --modes 1
done > $OUT/pair_bench_stagger.txt 2>&1
cat $OUT/pair_bench_stagger.txt
for m in 2 4; do echo "== spi x $m"; AMP_STRIP_SPI_MUL=$m timeout 200 python tools/pair_bench.py --reps 5 --C 128 64 32 --modes 1; done > $OUT/pair_bench_spimul.txt 2>&1
cat $OUT/pair_bench_spimul.txt
( AMP_PAIR_STRIP=0 timeout 600 python bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench_tile_full.json
python -c "
import json; d=json.load(open('$OUT/bench_tile_full.json')); print(d['ms_per_step']); print(json.dumps(d.get('other_configs'), indent=1)[:3000]); print(d.get('library_baseline')); print(d.get('cpu_baseline'))"
tail -5 $OUT/bench.err
