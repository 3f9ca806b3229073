#!/bin/bash
# Round 2, visit AP: SQ counter passes over the fused pairs at every stage width (pair_bench, per-tile / policy kernels): what bounds the small-C pairs?
OUT=gpurun_out/r2_ap
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/tools/pair_bench.py --reps 2 --C 128 64 32 --k 3 7 11 --d 3 --modes -1"
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $REPO/$OUT/a -o a -- $CMD > $REPO/$OUT/a.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 -d $REPO/$OUT/b -o b -- $CMD > $REPO/$OUT/b.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $REPO/$OUT/c -o c -- $CMD > $REPO/$OUT/c.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 -d $REPO/$OUT/d -o d -- $CMD > $REPO/$OUT/d.txt 2>&1
cd $REPO
tail -2 $OUT/a.txt
python tools/pmc_table.py $OUT/a/*counter_collection.csv $OUT/b/*counter_collection.csv $OUT/c/*counter_collection.csv $OUT/d/*counter_collection.csv > $OUT/table.csv 2> $OUT/table.err
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r2_ap/table.csv")))
agg = collections.OrderedDict()
for r in rows:
    k = (r["kernel"], r["grid"])
    agg.setdefault(k, []).append(r)
cols = [c for c in rows[0].keys() if c not in ("kernel", "grid", "vgpr", "lds")]
print("kernel,grid,n," + ",".join(cols))
for k, rs in agg.items():
    out = []
    for c in cols:
        vals = [float(r[c]) for r in rs if r.get(c) not in (None, "", "0", "0.0")]
        out.append(f"{sum(vals) / len(vals):.0f}" if vals else "0")
    print(",".join([k[0], k[1], str(len(rs))] + out))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
