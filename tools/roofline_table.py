#!/usr/bin/env python
"""Per-kernel roofline table of one profiled run, keyed by (kernel instantiation, workgroups) -- no shape guessing.

Inputs, all from ONE tools/gpu_round.sh visit (gpurun_out/<tag>/):
  prof/kt_kernel_trace.csv      rocprofv3 --kernel-trace: per-dispatch name, grid, start / end
  manifest.tsv                  the library's own launch manifest of the same run (AMP_LAUNCH_MANIFEST, amp_internal.h): per launch the
                                kernel name with its template arguments, workgroups, algorithmic GFLOP and MB, and what it computes
  pmc_fetch/, pmc_write/        optional FETCH_SIZE / WRITE_SIZE passes (corrected as tools/summarize_prof.py does: FETCH x 2 on gfx950)
Rows: one per (name, workgroups).  A key that the manifest lists with several different works (two layers sharing an instantiation AND a
grid) shows their mean and says so.  Fractions are of the f16x3 MFMA peak (2516.6 / 3 TFLOP/s) and of 8 TB/s HBM; a fraction above 1
cannot be right and is printed as 'inconsistent' -- never as a number.  'ovl' = share of the key's dispatches that overlapped another
dispatch in time (concurrent streams): their durations are not exclusive and their fractions are lower bounds.

    python tools/roofline_table.py gpurun_out/<tag> [--title "config 3 ..."] > profiles/<name>_roofline_table.txt
"""
import collections
import csv
import os
import re
import sys

PEAK_TF, PEAK_GBS = 2516.6 / 3.0, 8000.0


def short(name):
    m = re.search(r"amp::([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else re.sub(r"\(.*", "", name)


def main():
    args = [a for a in sys.argv[1:]]
    title = ""
    if "--title" in args:
        i = args.index("--title")
        title = args[i + 1]
        del args[i:i + 2]
    src = args[0]
    trace = os.path.join(src, "prof", "kt_kernel_trace.csv")
    rows = list(csv.DictReader(open(trace)))
    disp = []
    for r in rows:
        wg = 1
        for ax in "XYZ":
            wg *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
        disp.append((short(r["Kernel_Name"]), wg, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    # overlap marks: a dispatch that shares more than 5 % of its duration with other dispatches (concurrent streams)
    order = sorted(range(len(disp)), key=lambda i: disp[i][2])
    shared = [0] * len(disp)
    active = []                                   # indices still running, by end time
    for i in order:
        s0, e0 = disp[i][2], disp[i][3]
        active = [j for j in active if disp[j][3] > s0]
        for j in active:
            o = min(e0, disp[j][3]) - s0
            if o > 0:
                shared[i] += o
                shared[j] += o
        active.append(i)
    ovl = [shared[i] > 0.05 * max(1, disp[i][3] - disp[i][2]) for i in range(len(disp))]
    agg = collections.OrderedDict()
    for i, (n, wg, s, e) in enumerate(disp):
        a = agg.setdefault((n, wg), [0, 0.0, 0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] += 1 if ovl[i] else 0
    man = collections.defaultdict(lambda: collections.Counter())
    mpath = os.path.join(src, "manifest.tsv")
    if os.path.exists(mpath):
        for line in open(mpath):
            p = line.rstrip("\n").split("\t")
            if len(p) >= 5:
                man[(p[0], int(p[1]))][(float(p[2]), float(p[3]), p[4])] += 1
    traffic = {}
    fe, wr = os.path.join(src, "pmc_fetch", "pf_counter_collection.csv"), os.path.join(src, "pmc_write", "pw_counter_collection.csv")
    if os.path.exists(fe) and os.path.exists(wr):
        for path, ctr, mul in ((fe, "FETCH_SIZE", 2.0), (wr, "WRITE_SIZE", 1.0)):
            seen = set()
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] != ctr or (r["Dispatch_Id"], ctr) in seen:
                    continue
                seen.add((r["Dispatch_Id"], ctr))
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
                t = traffic.setdefault(key, {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})[ctr]
                t[0] += 1
                t[1] += float(r["Counter_Value"]) * 1024 / 1e6 * mul
    stamp = ""
    for p in (os.path.join(src, "commit.txt"), ".commit_stamp"):
        if os.path.exists(p):
            stamp = " ".join(open(p).read().split())
            break
    print(f"# per-kernel roofline{', ' + title if title else ''}; from {src}/prof/kt_kernel_trace.csv + manifest.tsv" + (f"; tree {stamp}" if stamp else ""))
    print("# peak: f16x3 MFMA %.1f TFLOP/s (2516.6 / 3), HBM %.0f GB/s; GFLOP / 'alg MB' = the launcher's own statement of the launch's work "
          "(x read once + y written once + residual / running sum)" % (PEAK_TF, PEAK_GBS))
    print("%-46s %7s %5s %9s %4s %9s %8s %6s %8s %8s %7s %6s  %s" % ("kernel", "wgs", "calls", "avg us", "ovl", "GFLOP", "TFLOP/s", "frac", "alg MB",
                                                                       "PMC MB", "GB/s", "frac", "work"))
    tot = 0.0
    bad = 0
    for (n, wg), (calls, us, nov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if n.startswith("__amd") or us < 20.0:
            continue
        avg = us / calls
        works = man.get((n, wg))
        gf = mb = None
        what = "(not in the manifest)"
        if works:
            tw = sum(works.values())
            gf = sum(k[0] * c for k, c in works.items()) / tw
            mb = sum(k[1] * c for k, c in works.items()) / tw
            what = max(works.items(), key=lambda kv: kv[1])[0][2]
            if len({k[0] for k in works}) > 1:
                what += "  [mean of %d different works on this key]" % len({k[0] for k in works})
            elif len(works) > 1:
                what += "  [+ %d variants of it: dilation / residual / running sum]" % (len(works) - 1)
        tr = traffic.get((n, wg))
        pmc = (tr["FETCH_SIZE"][1] / max(1, tr["FETCH_SIZE"][0]) + tr["WRITE_SIZE"][1] / max(1, tr["WRITE_SIZE"][0])) if tr else None

        def frac(v, peak):
            nonlocal bad
            if v is None:
                return "     -"
            if v / peak > 1.0:
                bad += 1
                return "incons"
            return "%6.3f" % (v / peak)
        tf = gf / (avg * 1e-6) / 1e3 if gf else None
        gbs = (pmc if pmc else mb) / 1e3 / (avg * 1e-6) if (pmc or mb) else None
        print("%-46s %7d %5d %9.1f %4s %9s %8s %s %8s %8s %7s %s  %s" % (
            n[:46], wg, calls, avg, ("%3.0f%%" % (100.0 * nov / calls)) if nov else "  - ",
            ("%9.1f" % gf) if gf else "        -", ("%8.1f" % tf) if tf else "       -", frac(tf, PEAK_TF) if tf else "     -",
            ("%8.0f" % mb) if mb else "       -", ("%8.0f" % pmc) if pmc else "       -", ("%7.0f" % gbs) if gbs else "      -",
            frac(gbs, PEAK_GBS), what))
        tot += us
    print("# the launches listed cover %.1f ms of kernel time in the profiled run" % (tot / 1e3))
    if bad:
        print("# %d fraction(s) above 1 withheld: the manifest's work and the measured duration of that key do not describe the same launches" % bad)
        sys.exit(1)


if __name__ == "__main__":
    main()
