#!/bin/bash
# Round 2, visit AQ: hi / lo split through v_fma_mixlo/hi_f16 (new) against the packed form (base library through AMP_LIB_PATH) -- equivalence on the
# hardware, parity, same-box A/B
OUT=gpurun_out/r2_aq
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
./tests/experiments/split_mix > $OUT/split_mix.txt 2>&1; cat $OUT/split_mix.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
for rep in 1 2 3; do
for v in new base; do
  if [ $v = new ]; then unset AMP_LIB_PATH; else export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_$v.so; fi
  ( timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $rep:', round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
done
cat $OUT/bench.txt
for v in new base new base; do
  if [ $v = new ]; then unset AMP_LIB_PATH; else export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_$v.so; fi
  echo "# $v" >> $OUT/other.txt
  timeout 200 python tools/bench_configs.py --only c3 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
  timeout 200 python tools/bench_configs.py --only c5 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
done
unset AMP_LIB_PATH
cut -c1-130 $OUT/other.txt
