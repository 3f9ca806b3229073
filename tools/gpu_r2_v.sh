#!/bin/bash
# Round 2, visit V: act1d with the conflict-free `sl` layout; tiles per workgroup sweep (2 = straight-line, 8 / 16 / 32 = rolled strips)
OUT=gpurun_out/r2_v
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py tests/test_gpu_range_guard.py -m gpu -q -x --timeout 300 2>&1 | tail -4 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for v in 8; do
  ( AMP_ACT1D_TILES=$v timeout 600 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 300 -k "bigvgan or act" 2>&1 | tail -2 ) > $OUT/pytest_$v.txt; echo "tiles=$v: $(tail -1 $OUT/pytest_$v.txt)"
done
for v in 2 8 16 32; do
  ( cd /tmp && AMP_ACT1D_TILES=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c3_$v -o kt -- python $REPO/tools/bench_configs.py --only c3 --reps 5 > $REPO/$OUT/c3_$v.json 2> $REPO/$OUT/c3_$v.err )
  echo "== AMP_ACT1D_TILES=$v $(cat $OUT/c3_$v.json)"
  grep act1d $OUT/c3_$v/kt_kernel_stats.csv | cut -c1-40,150-260
  rm -f $OUT/c3_$v/kt_kernel_trace.csv
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
