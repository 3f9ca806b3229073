#!/bin/bash
# Round 2, visit AT: A-fragment ring of 3 taps (default build) vs 4 taps with the B fragments in two halves (libamphion_hip_ring4.so), AMP_CONV_BLK=3
OUT=gpurun_out/r2_at
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
export AMP_CONV_BLK=3
( AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_ring4.so timeout 300 python -m pytest tests/test_gpu_f16x3_kernels.py -x -q -k blocked 2>&1 | tail -2 ) > $OUT/pytest_ring4.txt; tail -1 $OUT/pytest_ring4.txt
for v in ring3 ring4 ring3 ring4; do
  if [ $v = ring3 ]; then unset AMP_LIB_PATH; else export AMP_LIB_PATH=$REPO/amphion_amd/lib/libamphion_hip_$v.so; fi
  echo "# $v" >> $OUT/conv_bench.txt
  timeout 200 python tools/conv_bench.py --precision f16x3 --reps 30 --only rg 2>> $OUT/err.txt | grep "conv,256,256,\(7\|11\)" >> $OUT/conv_bench.txt
done
cat $OUT/conv_bench.txt
