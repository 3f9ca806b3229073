#!/bin/bash
# Round 2, visit AG: the same SQ counter passes over act1d_kernel after the swizzled LDS layout + strips
OUT=gpurun_out/r2_ag
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES"
P2="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"
P3="SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $P -d $REPO/$OUT/p$i -o p -- python $REPO/tools/bench_configs.py --only c3 --reps 1 > $REPO/$OUT/p$i.txt 2>&1
  tail -1 $REPO/$OUT/p$i.txt
done
cd $REPO
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/r2_ag/p{i}/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "act1d" not in k: continue
            key = (k[:40], r["Grid_Size"])
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and (r["Dispatch_Id"]) not in seen:
                seen.add(r["Dispatch_Id"]); n[key] += 1
for key, d in acc.items():
    print("==", key, "launches", n[key])
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / max(1, n[key]):16.0f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
