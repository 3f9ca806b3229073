#!/bin/bash
# Round 2, visit AB: narrow fused-pair tiles for under-filled grids: parity + single-utterance latency A/B
OUT=gpurun_out/r2_ab
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_generator.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_inference_api.py tests/test_gpu_c1_clips.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) > $OUT/pytest.txt; cat $OUT/pytest.txt
for rep in 1 2; do for v in -1 0; do
  if [ $v = 0 ]; then export AMP_PAIR_NARROW=0; else unset AMP_PAIR_NARROW; fi
  echo "== AMP_PAIR_NARROW=${AMP_PAIR_NARROW:-policy}"; python tools/bench_configs.py --only lat --reps 20 | grep -v hipGraph
done; done 2>&1 | tee $OUT/lat.txt
unset AMP_PAIR_NARROW
python tools/bench_configs.py --only list --reps 5 | tee -a $OUT/lat.txt
