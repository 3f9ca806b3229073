#!/bin/bash
# One GPU-box visit, assembled from named steps (round 3: replaces the 47 single-use visit scripts of round 2).
#   usage (repo root on the GPU box):  bash tools/gpu_round.sh <tag> [step ...]        default steps: tests bench prof pmc smoke
#   steps   tests     pytest -m gpu -x -q (what the driver runs at round end); PYTEST_N=<n> adds xdist workers
#           bench     the full bench.py line (CPU / library baselines, other configs)            -> bench.json
#           prof      rocprofv3 --kernel-trace --stats over bench.py --no-cpu-baseline           -> prof/kt_kernel_stats.csv, manifest.tsv, roofline_table.txt
#           pmc       FETCH_SIZE and WRITE_SIZE passes (separate runs, kernel-trace only)         -> pmc_fetch/, pmc_write/
#           sq        four SQ counter passes over ONE config-2 forward (or $SQCMD)                -> sq_table.csv, sq_summary.txt
#           configs   tools/bench_configs.py (C3, C5, mel, list API, latency)                     -> other_configs.jsonl
#           f32       bench.py --precision f32 --no-cpu-baseline                                  -> bench_f32.json
#           smoke     __graft_entry__.py --smoke
#           pairs     tools/pair_bench.py $PAIR_ARGS (policy kernels, --modes -1)                 -> pairs.csv
#           cmd       eval "$CMD" (free-form: A/B runs with environment switches)                 -> cmd.txt
TAG=${1:-run}; shift
STEPS=${@:-tests bench prof pmc smoke}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
[ -f .commit_stamp ] && cp .commit_stamp $OUT/commit.txt      # tools/gpu.sh: the commit this snapshot was cut from
PROF="rocprofv3 --output-format csv --kernel-trace"
# PROFCMD=<command>: what the prof / pmc steps profile (default: the config-2 bench line; e.g. "python $PWD/tools/bench_configs.py --only c3")
PROFCMD=${PROFCMD:-"python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"}
PMCCMD=${PROFCMD_PMC:-${PROFCMD/--steps 5 --warmup 2/--steps 2 --warmup 1}}
for STEP in $STEPS; do
  echo "== $STEP"
  case $STEP in
    tests)
      NW=${PYTEST_N:-0}; if [ "$NW" != "0" ]; then XD="-n $NW"; else XD=""; fi
      ( timeout 1500 python -m pytest tests -m gpu -x -q $XD 2>&1 | tail -40 ) > $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt ;;
    bench)
      ( timeout 600 python bench.py --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 ) > $OUT/bench.json; cut -c1-1500 $OUT/bench.json ;;
    prof)
      rm -f $REPO/$OUT/manifest.tsv     # the library's launch manifest of the same run (amp_internal.h), joined by tools/roofline_table.py
      ( cd /tmp && AMP_LAUNCH_MANIFEST=$REPO/$OUT/manifest.tsv timeout 600 $PROF --stats -d $REPO/$OUT/prof -o kt -- $PROFCMD > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
      ;;
    pmc)
      ( cd /tmp && timeout 600 $PROF --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o pf -- $PMCCMD > /dev/null 2> $REPO/$OUT/pmc_fetch.err )
      ( cd /tmp && timeout 600 $PROF --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o pw -- $PMCCMD > /dev/null 2> $REPO/$OUT/pmc_write.err ) ;;
    sq)
      SQCMD=${SQCMD:-"python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline"}     # SQCMD=<other command>: counters of another workload
      ( cd /tmp
        timeout 300 $PROF --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $REPO/$OUT/sq_a -o a -- $SQCMD > /dev/null 2> $REPO/$OUT/sq_a.err
        timeout 300 $PROF --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 -d $REPO/$OUT/sq_b -o b -- $SQCMD > /dev/null 2> $REPO/$OUT/sq_b.err
        timeout 300 $PROF --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $REPO/$OUT/sq_c -o c -- $SQCMD > /dev/null 2> $REPO/$OUT/sq_c.err
        timeout 300 $PROF --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 -d $REPO/$OUT/sq_d -o d -- $SQCMD > /dev/null 2> $REPO/$OUT/sq_d.err )
      python tools/pmc_table.py $OUT/sq_a/*counter_collection.csv $OUT/sq_b/*counter_collection.csv $OUT/sq_c/*counter_collection.csv $OUT/sq_d/*counter_collection.csv > $OUT/sq_table.csv 2> $OUT/sq_table.err
      python tools/pmc_table.py --summary $OUT/sq_table.csv > $OUT/sq_summary.txt 2>> $OUT/sq_table.err; head -30 $OUT/sq_summary.txt ;;
    configs)
      ( timeout 500 python tools/bench_configs.py --reps 10 > $OUT/other_configs.jsonl 2> $OUT/other_configs.err ); cat $OUT/other_configs.jsonl ;;
    f32)
      ( timeout 300 python bench.py --steps 10 --warmup 3 --precision f32 --no-cpu-baseline 2> $OUT/bench_f32.err | tail -1 ) > $OUT/bench_f32.json ;;
    smoke)
      ( timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > $OUT/smoke.txt; cat $OUT/smoke.txt ;;
    pairs)
      ( timeout 300 python tools/pair_bench.py ${PAIR_ARGS:---modes -1} > $OUT/pairs.csv 2> $OUT/pairs.err ); cat $OUT/pairs.csv ;;
    cmd)
      ( eval "$CMD" ) > $OUT/cmd.txt 2>&1; tail -60 $OUT/cmd.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
done
# the per-kernel roofline table of this visit, AFTER every step (it joins the prof step's trace + manifest with the pmc step's counters)
[ -f $OUT/prof/kt_kernel_trace.csv ] && python tools/roofline_table.py $OUT --title "${ROOFTITLE:-$PROFCMD}" > $OUT/roofline_table.txt 2> $OUT/roofline_table.err
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*agent_info*" -delete 2>/dev/null
find $OUT -path "*sq_*" -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $OUT
