#!/bin/bash
# Round 2, visit AL: same-box A/B of library builds (AMP_LIB_PATH): base = flat_load / flat_store in the strip kernel's epilogue,
# new = global_* (laundered pointers cast back to address space 1), nts = new + nontemporal y stores, ntsl = nts + nontemporal staging loads
OUT=gpurun_out/r2_al
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_generator.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -4 ) > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
for rep in 1 2; do
for v in new base nts ntsl; do
  if [ $v = new ]; then unset AMP_LIB_PATH; else export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_$v.so; fi
  ( timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $rep:', round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
done
unset AMP_LIB_PATH
cat $OUT/bench.txt
for v in nts ntsl; do
  ( AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_$v.so timeout 300 python -m pytest tests/test_gpu_pair.py tests/test_gpu_generator.py -x -q 2>&1 | tail -2 ) > $OUT/pytest_$v.txt; tail -1 $OUT/pytest_$v.txt
done
du -sh $OUT
