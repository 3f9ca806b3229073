#!/bin/bash
# Round 2, visit AO: late-stage batch groups (AMP_LATE_GROUP=<first stage>,<items>) -- waveform checksum and forward time per setting
OUT=gpurun_out/r2_ao
mkdir -p $OUT
export TMPDIR=/tmp
for s in - 2,16 2,8 3,16 3,8 - 2,32 3,32 2,4 1,16 -; do
  if [ "$s" = "-" ]; then unset AMP_LATE_GROUP; else export AMP_LATE_GROUP=$s; fi
  timeout 200 python tests/experiments/late_group.py 2>> $OUT/err.txt | tail -1 >> $OUT/late_group.txt
done
cat $OUT/late_group.txt
