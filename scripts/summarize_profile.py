#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/profile_bench.sh into the small committed summaries under profiles/<tag>/."""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "profiles", tag)
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(P, "trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))
try:
    shutil.copy(os.path.join(P, "trace", "trace_agent_info.csv"), os.path.join(out, "agent_info.csv"))
except OSError:
    pass
summ = {}
for name, f in (("FETCH_SIZE", "pmc_fetch/fetch_counter_collection.csv"), ("WRITE_SIZE", "pmc_write/write_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(P, f))):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "silent" in k:
            summ.setdefault(k, {"frames_per_dispatch": frames})[name] = {"dispatches": len(v), "mean_KiB_per_dispatch": sum(v) / len(v)}
sys.path.insert(0, root)
import bench  # noqa: E402  (csrc_revision: bench.py only trusts a summary taken at the current kernel sources)
summ["_csrc_revision"] = bench.csrc_revision()
json.dump(summ, open(os.path.join(out, "pmc_hbm_bytes.json"), "w"), indent=1)
rows = [r for r in csv.DictReader(open(os.path.join(P, "trace", "trace_kernel_trace.csv"))) if "silent" in r["Kernel_Name"]]
cols = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Start_Timestamp", "End_Timestamp"]
with open(os.path.join(out, "kernel_trace_silent.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(cols + ["Duration_ns"])
    for r in rows:
        w.writerow([r.get(c, "") for c in cols] + [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
for line in open(os.path.join(out, "kernel_stats.csv")):
    print(line.strip()[:150])
for k, v in summ.items():
    if not isinstance(v, dict):
        continue
    print(k[:70], {a: round(b["mean_KiB_per_dispatch"] * 1024 / 1e6, 1) for a, b in v.items() if isinstance(b, dict)}, "MB/dispatch (raw counters)")
