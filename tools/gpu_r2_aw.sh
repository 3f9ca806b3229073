#!/bin/bash
# Round 2, visit AW (the round's last GPU seconds): first hardware run of the -DAMP_STRIP_RING experiment library
OUT=gpurun_out/r2_aw; mkdir -p $OUT
AMP_LIB_PATH=$PWD/amphion_amd/lib/libamphion_hip_ring.so timeout 30 python tests/experiments/strip_ring_check.py > $OUT/ring.txt 2>&1; cat $OUT/ring.txt | tail -16
