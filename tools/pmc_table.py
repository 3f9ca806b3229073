#!/usr/bin/env python
"""rocprofv3 --pmc counter_collection.csv files -> one row per dispatch (kernel, grid, counters), or a per-kernel summary.

    python tools/pmc_table.py a/*counter_collection.csv b/*counter_collection.csv ... > table.csv
    python tools/pmc_table.py --summary table.csv > summary.txt

The passes are separate runs of the same command, so dispatch ids line up.  --summary averages the dispatches of each
kernel instantiation and derives (units: MI355X_MICROARCH.md "rocprofv3 PMC slots"): MFMA busy % =
SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) (calibrated in round 2 on the k = 11 strips), the wave-time split
SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES, per wave: MFMA instructions (MOPS_F16 / 64), other VALU, LDS, VMEM read / write instructions, and the LDS bank-conflict share of the LDS-active cycles."""
import collections
import csv
import sys

STRIP = (("void amp::", ""), ("(amp::ConvArgs)", ""), ("(amp::PairArgs)", ""), ("(amp::RbArgs)", ""), (", ", ","))


def short(name):
    for a, b in STRIP:
        name = name.replace(a, b)
    return name.split("(")[0]


def table(paths):
    rows = collections.OrderedDict()
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                if "amp::" not in r["Kernel_Name"]:
                    continue
                d = rows.setdefault(int(r["Dispatch_Id"]), {
                    "kernel": short(r["Kernel_Name"]).replace(",", "."), "grid": r["Grid_Size"],
                    "us": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3, "vgpr": r.get("VGPR_Count", ""),
                    "lds": r.get("LDS_Block_Size", ""), "scratch": r.get("Scratch_Size", "")})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    names = []
    for d in rows.values():
        for k in d:
            if k not in names:
                names.append(k)
    print(",".join(names))
    for d in rows.values():
        print(",".join(str(round(d.get(k, 0), 1)) if isinstance(d.get(k, 0), float) else str(d.get(k, "")) for k in names))


def summary(path):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        agg.setdefault((r["kernel"], r["grid"]), []).append(r)
    def mean(rs, c):
        v = [float(r[c]) for r in rs if r.get(c) not in (None, "")]
        return sum(v) / len(v) if v else 0.0
    import os
    for c in (os.path.join(os.path.dirname(path), "commit.txt"), ".commit_stamp"):
        if os.path.exists(c):
            print("# tree " + " ".join(open(c).read().split()))
            break
    print(f"{'kernel':44s} {'grid':>9s} {'n':>3s} {'us':>8s} {'vgpr':>5s} {'scr':>5s} {'MFMA%':>6s} | {'wait%':>6s} {'stall%':>6s} {'issue%':>6s} |"
          f" {'mfma/wv':>8s} {'valu/wv':>8s} {'lds/wv':>7s} {'vmrd/wv':>7s} {'vmwr/wv':>7s} | {'v:m':>5s} {'ldsCf%':>6s}")
    order = sorted(agg.items(), key=lambda kv: -mean(kv[1], "us") * len(kv[1]))
    for (k, grid), rs in order:
        waves = mean(rs, "SQ_WAVES") or 1.0
        wc = mean(rs, "SQ_WAVE_CYCLES") or 1.0
        busy = mean(rs, "SQ_BUSY_CYCLES") or 1.0
        mf = mean(rs, "SQ_INSTS_VALU_MFMA_MOPS_F16") / 64.0 / waves            # 64 MOPS units per v_mfma_f32_32x32x16_f16
        valu = mean(rs, "SQ_INSTS_VALU") / waves
        lds_act = mean(rs, "SQ_ACTIVE_INST_LDS") or 1.0
        print(f"{k:44s} {grid:>9s} {len(rs):3d} {mean(rs, 'us'):8.1f} {rs[0].get('vgpr', ''):>5s} {rs[0].get('scratch', ''):>5s} "
              f"{100 * mean(rs, 'SQ_VALU_MFMA_BUSY_CYCLES') / (32 * busy):6.1f} | {100 * mean(rs, 'SQ_WAIT_ANY') / wc:6.1f} "
              f"{100 * mean(rs, 'SQ_WAIT_INST_ANY') / wc:6.1f} {100 * mean(rs, 'SQ_ACTIVE_INST_ANY') / wc:6.1f} | {mf:8.0f} {valu - mf:8.0f} "
              f"{mean(rs, 'SQ_INSTS_LDS') / waves:7.0f} {mean(rs, 'SQ_INSTS_VMEM_RD') / waves:7.0f} {mean(rs, 'SQ_INSTS_VMEM_WR') / waves:7.0f} | "
              f"{(valu - mf) / mf if mf else 0:5.2f} {100 * mean(rs, 'SQ_LDS_BANK_CONFLICT') / lds_act:6.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summary":
        summary(sys.argv[2])
    else:
        table(sys.argv[1:])
