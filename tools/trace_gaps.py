#!/usr/bin/env python
"""rocprofv3 --kernel-trace (+ --memory-copy-trace) csv -> one line per generator forward (ending with a conv_post kernel):
wall span on the GPU, summed kernel time, the largest gap between consecutive GPU activities and what surrounds it.
    python tools/trace_gaps.py <dir with *_kernel_trace.csv [*_memory_copy_trace.csv]>"""
import csv, glob, os, sys

d = sys.argv[1]
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void amp::", "")[:60]))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
t0 = ev[0][0]
fwd_start, busy, biggest, prev_end, n = None, 0, (0, "", "", 0), None, 0
print("forward,start_ms,span_ms,kernel_busy_ms,largest_gap_ms,gap_at_ms_into_forward,gap_between")
for s, e, name in ev:
    if fwd_start is None:
        fwd_start, busy, biggest, prev_end, prev_name = s, 0, (0, "", "", 0), s, "(start)"
    gap = s - prev_end
    if gap > biggest[0]:
        biggest = (gap, prev_name, name, prev_end - fwd_start)
    busy += e - s
    prev_end, prev_name = max(prev_end, e), name
    if "conv_post" in name:
        print(f"{n},{(fwd_start - t0) / 1e6:.2f},{(e - fwd_start) / 1e6:.2f},{busy / 1e6:.2f},{biggest[0] / 1e6:.2f},{biggest[3] / 1e6:.2f},{biggest[1]} -> {biggest[2]}")
        n += 1
        fwd_start = None
