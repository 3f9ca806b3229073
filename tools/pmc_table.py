#!/usr/bin/env python
"""Pivot rocprofv3 --pmc counter_collection.csv files into one row per dispatch (kernel, grid, counters)."""
import csv, sys, collections
rows = collections.OrderedDict()
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            if "conv" not in r["Kernel_Name"] and "pair" not in r["Kernel_Name"]:
                continue
            key = (int(r["Dispatch_Id"]))
            d = rows.setdefault(key, {"kernel": r["Kernel_Name"].replace("void amp::", "").replace("(amp::ConvArgs)", "").replace("(amp::PairArgs)", "").replace(", ", "."), "grid": r["Grid_Size"],
                                      "us": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3, "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = []
for d in rows.values():
    for k in d:
        if k not in names:
            names.append(k)
print(",".join(names))
for d in rows.values():
    print(",".join(str(round(d.get(k, 0), 1)) if isinstance(d.get(k, 0), float) else str(d.get(k, "")) for k in names))
