#!/bin/bash
# Same-box A/B of two builds of libamphion_hip.so through AMP_LIB_PATH (amphion_amd/_lib.py), runs alternating:
#   bash tools/lib_ab.sh <libA.so> <libB.so> [rounds] [command ...]      default command: bench.py --no-cpu-baseline, 10 steps
A=$1; B=$2; R=${3:-3}; shift 3
CMD=${@:-python bench.py --steps 10 --warmup 3 --no-cpu-baseline}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "
    AMP_LIB_PATH=$PWD/$L $CMD 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), [round(v,3) for v in d['roofline']['mrf_stack']['ms_per_stage']], round(d['roofline']['launch_us'],1))"
  done
done
