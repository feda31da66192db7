#!/usr/bin/env python3
"""Join rocprofv3 --pmc counters with --kernel-trace durations per dispatch of one kernel and print them in dispatch order, averaged
over runs of `group` consecutive dispatches (scripts/placement_probe.py: one allocation = one run of launches).
    python scripts/pmc_by_allocation.py <dir> <kernel substring> <group>"""
import csv, glob, os, sys
d, sub, group = sys.argv[1], sys.argv[2], int(sys.argv[3])
cnt = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
trc = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in trc:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = {}
for f in cnt:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows, key=int)
names = sorted({n for v in rows.values() for n in v})
print("dispatches %d   counters %s" % (len(ids), names))
for i in range(0, len(ids), group):
    g = ids[i:i + group]
    line = "run %2d  n %3d  us %8.1f" % (i // group, len(g), sum(dur.get(k, 0.0) for k in g) / len(g))
    for n in names:
        line += "  %s %.4g" % (n, sum(rows[k].get(n, 0.0) for k in g) / len(g))
    print(line)
