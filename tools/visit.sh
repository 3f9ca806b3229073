timeout 900 python -m pytest tests/test_gpu_resblock.py tests/test_gpu_bigvgan.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -4
export CMD='
for i in 1 2; do
echo "## rb_fusion default"; timeout 200 python tools/bench_configs.py --only c3 --reps 10 | cut -c1-120
echo "## AMP_RB_FUSION=0"; AMP_RB_FUSION=0 timeout 200 python tools/bench_configs.py --only c3 --reps 10 | cut -c1-120
done
cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_z/prof -o kt -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only c3 --reps 3 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r3_z/prof/kt_kernel_stats.csv")))
for r in rows[:24]:
    print(r["Name"][:75].ljust(75), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
rm -rf gpurun_out/r3_z/prof
'
bash tools/gpu_round.sh r3_z cmd
