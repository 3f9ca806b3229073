#!/bin/bash
# Round 2, visit A: first execution of the never-run kernels (full VITS inference, fused AMP pair) + LDS/VMEM counters
# on the dominant fused-pair kernel.
OUT=gpurun_out/r2_a
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
( AMP_RUN_UNVERIFIED=1 timeout 400 python -m pytest tests/test_gpu_vits_infer.py -m gpu -q -x --timeout 120 2>&1 | tail -60 ) > $OUT/vits_infer_x.txt
( AMP_RUN_UNVERIFIED=1 timeout 600 python -m pytest tests/test_gpu_vits_infer.py -m gpu -q --timeout 120 2>&1 | tail -150 ) > $OUT/vits_infer_all.txt
tail -30 $OUT/vits_infer_all.txt
( AMP_FUSE_AMP=1 timeout 600 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_inference_api.py -m gpu -q --timeout 200 2>&1 | tail -80 ) > $OUT/fuse_amp_bigvgan.txt
tail -30 $OUT/fuse_amp_bigvgan.txt
( AMP_FUSE_AMP=1 timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -k bigvgan --timeout 400 2>&1 | tail -40 ) > $OUT/fuse_amp_full.txt
tail -8 $OUT/fuse_amp_full.txt
( timeout 200 python tools/bench_configs.py --only c3 --reps 10 2>&1 | tail -1 ) > $OUT/c3_unfused.json
( AMP_FUSE_AMP=1 timeout 200 python tools/bench_configs.py --only c3 --reps 10 2>&1 | tail -1 ) > $OUT/c3_fused.json
cat $OUT/c3_unfused.json $OUT/c3_fused.json
cd /tmp
P="--output-format csv --kernel-trace"
B="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 $P --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES -d $REPO/$OUT/pmc_lds -o l -- $B > /dev/null 2> $REPO/$OUT/pmc_lds.err
timeout 200 rocprofv3 $P --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $REPO/$OUT/pmc_vm -o v -- $B > /dev/null 2> $REPO/$OUT/pmc_vm.err
timeout 200 rocprofv3 $P --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $REPO/$OUT/pmc_l2 -o c -- $B > /dev/null 2> $REPO/$OUT/pmc_l2.err
timeout 200 rocprofv3 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $REPO/$OUT/pmc_sq -o s -- $B > /dev/null 2> $REPO/$OUT/pmc_sq.err
cd $REPO
tail -3 $OUT/pmc_*.err
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*agent_info*" -delete; du -sh $OUT
