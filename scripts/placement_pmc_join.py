#!/usr/bin/env python3
"""Join the rocprofv3 --pmc output of scripts/placement_pmc.py with its record: the last reps * 2 * group dispatches of the
dominant kernel are the labelled ones (fast, slow, fast, slow, ...).  Prints, per counter, the mean per launch on the fast and on
the slow allocation and slow / fast.
    python3 scripts/placement_pmc_join.py <rocprof dir> <record.json> <kernel substring>"""
import csv, glob, json, os, sys
d, rec, sub = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(rows)
g, order = rec["group"], rec["order"]
n = g * len(order)
if len(ids) < n:
    sys.exit("only %d dispatches of %s, need %d" % (len(ids), sub, n))
ids = ids[-n:]
acc = {"fast": {}, "slow": {}}
cnt = {"fast": 0, "slow": 0}
for gi, label in enumerate(order):
    for k in ids[gi * g + 1:(gi + 1) * g]:      # the first launch after a buffer switch is left out
        cnt[label] += 1
        for c, v in rows[k].items():
            acc[label][c] = acc[label].get(c, 0.0) + v
print("%s  map %s  fast %.4f ms  slow %.4f ms  (contrast %.3f by HIP events in the same process)  launches/label %d" % (
    rec["workload"], rec["map"], rec["fast_ms"], rec["slow_ms"], rec["contrast"], cnt["fast"]))
for c in sorted(acc["fast"]):
    a, b = acc["fast"][c] / cnt["fast"], acc["slow"].get(c, 0.0) / max(cnt["slow"], 1)
    print("  %-44s fast %14.6g   slow %14.6g   slow/fast %7.3f" % (c, a, b, b / a if a else float("nan")))
