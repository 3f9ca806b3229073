#!/bin/bash
# Round 2, visit AX: generator-level check of the -DAMP_STRIP_RING experiment library
OUT=gpurun_out/r2_ax; mkdir -p $OUT
AMP_LIB_PATH=$PWD/amphion_amd/lib/libamphion_hip_ring.so timeout 24 python tests/experiments/strip_ring_gen_check.py > $OUT/gen.txt 2>&1; tail -10 $OUT/gen.txt
