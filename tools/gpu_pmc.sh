#!/bin/bash
# SQ counter passes over the per-layer conv microbench (f16x3)
OUT=gpurun_out/${1:-pmc}
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L > $REPO/$OUT/counters.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $REPO/$OUT/a -o a -- python $REPO/tools/conv_bench.py --precision f16x3 --reps 2 > $REPO/$OUT/a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 -d $REPO/$OUT/b -o b -- python $REPO/tools/conv_bench.py --precision f16x3 --reps 2 > $REPO/$OUT/b.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_FLAT -d $REPO/$OUT/c -o c -- python $REPO/tools/conv_bench.py --precision f16x3 --reps 2 > $REPO/$OUT/c.txt 2>&1
cd $REPO
grep -c . $OUT/counters.txt; tail -3 $OUT/a.txt; tail -3 $OUT/b.txt; tail -3 $OUT/c.txt; ls $OUT/a $OUT/b $OUT/c
