#!/bin/bash
# Round 2, visit AM: ping-pong tile order (AMP_PINGPONG=1: every other conv / pair launch walks its tiles backwards) -- parity, bench A/B, FETCH_SIZE
OUT=gpurun_out/r2_am
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_generator.py -x -q 2>&1 | tail -4 ) > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
( AMP_PINGPONG=1 timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_generator.py tests/test_gpu_pair.py tests/test_gpu_bigvgan.py -x -q 2>&1 | tail -4 ) > $OUT/pytest_pp.txt
tail -2 $OUT/pytest_pp.txt
for m in 1 0 1 0; do
  echo "# AMP_PINGPONG=$m" >> $OUT/bench.txt
  ( AMP_PINGPONG=$m timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],3),'ms/step  stages',[round(v,2) for v in r['mrf_stack']['ms_per_stage']],'dominant us',round(r['launch_us'],1))" ) >> $OUT/bench.txt
done
cat $OUT/bench.txt
for m in 1 0; do
  echo "# AMP_PINGPONG=$m" >> $OUT/other.txt
  AMP_PINGPONG=$m timeout 200 python tools/bench_configs.py --only c3 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
  AMP_PINGPONG=$m timeout 200 python tools/bench_configs.py --only c5 --reps 10 >> $OUT/other.txt 2>> $OUT/other.err
done
cut -c1-170 $OUT/other.txt
cd /tmp
for m in 0 1; do
  AMP_PINGPONG=$m timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc$m -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc$m.err
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for m in (0, 1):
    acc = collections.defaultdict(lambda: [0, 0.0]); tot = 0.0
    for f in glob.glob(f"gpurun_out/r2_am/pmc{m}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE": continue
            k = r["Kernel_Name"].replace("void amp::", "")[:44]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"]); tot += float(r["Counter_Value"])
    print("== AMP_PINGPONG =", m, " FETCH MB per launch (KiB x 2 gfx950 correction); total GB over the run:", round(tot * 2048 / 1e9, 2))
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  ", k, n, round(v / n * 2048 / 1e6, 1))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
