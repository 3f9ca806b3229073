#!/bin/bash
# Round 2, visit AK: FETCH_SIZE of the multi-row-group convs with the row group as blockIdx.y (0) vs the fastest grid index (1)
OUT=gpurun_out/r2_ak
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for m in 0 1 0 1; do
  echo "# AMP_CONV_RG_FAST=$m" >> $OUT/conv_bench.txt
  AMP_CONV_RG_FAST=$m timeout 200 python tools/conv_bench.py --precision f16x3 --reps 20 --only rg >> $OUT/conv_bench.txt 2>> $OUT/conv_bench.err
done
cat $OUT/conv_bench.txt
cd /tmp
for m in 0 1; do
  AMP_CONV_RG_FAST=$m timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc$m -o p -- python $REPO/tools/conv_bench.py --precision f16x3 --reps 2 --only rg > /dev/null 2> $REPO/$OUT/pmc$m.err
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for m in (0, 1):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/r2_ak/pmc{m}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE": continue
            k = (r["Kernel_Name"][:48], r["Grid_Size"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    print("== AMP_CONV_RG_FAST =", m, " (kernel, grid threads) launches, FETCH MB per launch (KiB x 2 gfx950 correction)")
    for k, (n, v) in sorted(acc.items()):
        if "conv" in k[0]: print("  ", k, n, round(v / n * 1024 * 2 / 1e6, 1))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info*" -delete
du -sh $OUT
