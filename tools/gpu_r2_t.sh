#!/bin/bash
# Round 2, visit T: full -m gpu suite with the whole-K conv kernel / fused WN on; where do C5's device copies come from
OUT=gpurun_out/r2_t
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -8 ) > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
( cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $REPO/$OUT/c5 -o kt -- python $REPO/tools/bench_configs.py --only c5 --reps 3 > $REPO/$OUT/c5.json 2> $REPO/$OUT/c5.err )
cat $OUT/c5.json; ls $OUT/c5
python - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/r2_t/c5/*memory_copy_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("memory copies:", len(rows), "columns:", list(rows[0].keys()) if rows else None)
    c = collections.Counter((r.get("Direction"), r.get("Size", r.get("Bytes", "?"))) for r in rows)
    for k, v in c.most_common(30): print(v, k)
kt = glob.glob("gpurun_out/r2_t/c5/*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt[0])))
    names = [r["Kernel_Name"] for r in rows]
    # what precedes each copyBuffer kernel
    prev = collections.Counter()
    for i, n in enumerate(names):
        if "copyBuffer" in n and i > 0: prev[(names[i-1][:60], names[i+1][:60] if i + 1 < len(names) else "")] += 1
    for k, v in prev.most_common(25): print(v, k)
PY
rm -f $OUT/c5/*kernel_trace.csv; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
