#!/bin/bash
# Round 2, visit C: strip-mined fused pair kernel -- parity, A/B against the per-tile kernel, kernel stats.
OUT=gpurun_out/r2_c
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_pair.py -m gpu -q -x --timeout 300 2>&1 | tail -30 ) > $OUT/pytest_pair.txt
tail -15 $OUT/pytest_pair.txt
for rep in 1 2; do
( AMP_PAIR_STRIP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_tile.err | tail -1 ) > $OUT/bench_tile_$rep.json
( AMP_PAIR_STRIP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_strip.err | tail -1 ) > $OUT/bench_strip_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_c/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, round(d["ms_per_step"],3), "dom_us", round(r["launch_us"],1), "stages", [round(v,2) for v in r["mrf_stack"]["ms_per_stage"]])
    except Exception as e: print(f, "ERR", e)
PY
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/prof -o kt -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
head -20 $OUT/prof/kt_kernel_stats.csv | cut -c1-180
rm -f $OUT/prof/kt_kernel_trace.csv
( timeout 1500 python -m pytest tests -m gpu -q -n 4 --timeout 600 2>&1 | tail -30 ) > $OUT/pytest_gpu.txt
tail -8 $OUT/pytest_gpu.txt
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*agent_info*" -delete; du -sh $OUT
