#!/usr/bin/env python
"""Condense one tools/gpu_round.sh visit (gpurun_out/<tag>/) into the small files kept under profiles/.

    python tools/summarize_prof.py gpurun_out/<tag> profiles/<name>

Writes <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), <name>_bench.json and
<name>_hbm_traffic.csv: per kernel name, launches, mean duration and mean FETCH_SIZE / WRITE_SIZE per
launch.  rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts a 128-B request of a wide
coalesced read as 64 B, so the raw value is doubled (MI355X_MICROARCH.md "HBM"); WRITE_SIZE is left raw
(uncalibrated there).  Both columns are printed raw and corrected.
"""
import csv
import collections
import os
import shutil
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])  # n, sum counter, sum ns
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if key in seen:
                continue
            seen.add(key)
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    for a, b in (("prof/kt_kernel_stats.csv", "_kernel_stats.csv"), ("bench.json", "_bench.json")):
        p = os.path.join(src, a)
        if os.path.exists(p):
            shutil.copy(p, dst + b)
    stamp = ""
    for c in (os.path.join(src, "commit.txt"), ".commit_stamp"):
        if os.path.exists(c):
            stamp = " ".join(open(c).read().split())
            break
    if stamp:   # every summary of this visit carries the commit its snapshot was cut from (tools/gpu.sh)
        open(dst + "_tree.txt", "w").write(f"profiles {os.path.basename(dst)}_*: measured on a snapshot of tree {stamp}\n")
    fe = os.path.join(src, "pmc_fetch", "pf_counter_collection.csv")
    wr = os.path.join(src, "pmc_write", "pw_counter_collection.csv")
    if os.path.exists(fe) and os.path.exists(wr):
        F, W = per_kernel(fe, "FETCH_SIZE"), per_kernel(wr, "WRITE_SIZE")
        with open(dst + "_hbm_traffic.csv", "w") as out:
            if stamp:
                out.write(f"# tree {stamp}\n")
            out.write("kernel,launches,mean_us_under_pmc,fetch_MB_raw,fetch_MB_x2_gfx950,write_MB_raw,total_MB_corrected\n")
            tot = 0.0
            for k in sorted(F, key=lambda k: -F[k][1]):
                n, s, ns = F[k]
                w = W.get(k, [1, 0.0, 0.0])
                f_mb = s / n * 1024 / 1e6
                w_mb = w[1] / max(w[0], 1) * 1024 / 1e6
                out.write(f"\"{k}\",{n},{ns / n / 1e3:.1f},{f_mb:.2f},{2 * f_mb:.2f},{w_mb:.2f},{2 * f_mb + w_mb:.2f}\n")
                tot += (2 * s + w[1] * n / max(w[0], 1)) * 1024 / 1e6
            out.write(f"# sum over all dispatches in the profiled run (MB, corrected): {tot:.1f}\n")
        print(open(dst + "_hbm_traffic.csv").read())


if __name__ == "__main__":
    main()
