#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/profile_bench.sh into the small committed summaries under profiles/<tag>/.

    python scripts/summarize_profile.py <tag> <frames per dispatch> [run ...] [--kernel SUBSTR]

Every ``run`` is the tag of one profile_bench.sh call (gpurun_out/prof_<run>/), normally one per gpurun call = one box.  The
pool's boxes differ by up to 8 % on the same binary, so the summary that is committed is the one of the MEDIAN run by the
dominant kernel's average duration -- never the best -- and profiles/<tag>/README.md lists the average of every run.
Without runs: the single call gpurun_out/prof_<tag>/."""
import collections
import csv
import json
import os
import shutil
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
kernel_sub = "gray_stream_kernel"
if "--kernel" in sys.argv:
    kernel_sub = sys.argv[sys.argv.index("--kernel") + 1]
    args.remove(kernel_sub)
tag = args[0] if args else "r01"
frames = int(args[1]) if len(args) > 1 else 64
runs = args[2:] or [tag]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles", tag)
os.makedirs(out, exist_ok=True)


def stats_of(run):
    """Average duration (us) of the dominant kernel over the launches BEHIND bench.py's trace marker (the tuner's draws lie in front
    of it), and rocprofv3's own table over every launch."""
    d = os.path.join(root, "gpurun_out", "prof_" + run, "trace")
    rows = list(csv.DictReader(open(os.path.join(d, "trace_kernel_stats.csv"))))
    tr = sorted(csv.DictReader(open(os.path.join(d, "trace_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(tr) if "trace_marker_kernel" in r["Kernel_Name"]]
    if not marks:
        sys.exit("summarize_profile: no trace_marker_kernel in %s -- not a trace of this revision's bench.py" % d)
    dom = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr[marks[0] + 1:] if kernel_sub in r["Kernel_Name"]]
    if len(dom) < 20:
        sys.exit("summarize_profile: only %d launches of %s behind the marker in %s" % (len(dom), kernel_sub, d))
    return sum(dom) / len(dom) / 1e3, rows


per_run = {r: stats_of(r) for r in runs}
order = sorted(runs, key=lambda r: per_run[r][0])
median = order[len(order) // 2] if len(order) % 2 else order[len(order) // 2 - 1]      # (even count: the lower middle)
P = os.path.join(root, "gpurun_out", "prof_" + median)
# rocprofv3's own statistics cover every launch of the process -- also the ones bench.py's tuners issued on placements and stream pairs
# that were then dropped.  bench.py marks the end of tuning with a busy_wait_kernel launch: kernel_stats.csv = the same statistics
# over the launches AFTER that marker (what the step runs with); the tool's table is kept as kernel_stats_all_launches.csv.
shutil.copy(os.path.join(P, "trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats_all_launches.csv"))
trace_rows = sorted(csv.DictReader(open(os.path.join(P, "trace", "trace_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
# (the FIRST trace_marker_kernel: side workloads of the same process set markers of their own later on)
marks = [i for i, r in enumerate(trace_rows) if "trace_marker_kernel" in r["Kernel_Name"]]
if not marks:
    sys.exit("summarize_profile: no trace_marker_kernel in %s -- not a trace of this revision's bench.py" % P)
after = trace_rows[marks[0] + 1:]
agg = collections.OrderedDict()
for r in after:
    agg.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = float(sum(sum(v) for v in agg.values())) or 1.0
with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        mean = sum(v) / len(v)
        sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
        w.writerow([k, len(v), sum(v), "%.6f" % mean, "%.6f" % (100.0 * sum(v) / tot), min(v), max(v), "%.6f" % sd])
dom_after = [d for k, v in agg.items() if kernel_sub in k for d in v]
if len(dom_after) < 20:
    sys.exit("summarize_profile: only %d launches of %s behind the marker (the timed steps alone are more): wrong trace or wrong kernel" % (len(dom_after), kernel_sub))
if dom_after:
    per_run[median] = (sum(dom_after) / len(dom_after) / 1e3, per_run[median][1])
try:
    shutil.copy(os.path.join(P, "trace", "trace_agent_info.csv"), os.path.join(out, "agent_info.csv"))
except OSError:
    pass
summ = {}
for name, f in (("FETCH_SIZE", "pmc_fetch/fetch_counter_collection.csv"), ("WRITE_SIZE", "pmc_write/write_counter_collection.csv")):
    agg = collections.defaultdict(list)
    try:
        rows = csv.DictReader(open(os.path.join(P, f)))
    except OSError:
        continue
    for r in rows:
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
with open(os.path.join(out, "runs.json"), "w") as fh:
    json.dump({"dominant_kernel": kernel_sub, "average_us_per_run": {r: round(per_run[r][0], 2) for r in runs}, "median_run": median,
               "csrc_revision": bench.csrc_revision()}, fh, indent=1)
print("runs (average us of %s): %s -> median run %s" % (kernel_sub, {r: round(per_run[r][0], 1) for r in runs}, median))
for line in open(os.path.join(out, "kernel_stats.csv")):
    print(line.strip()[:150])
for k, v in summ.items():
    if not isinstance(v, dict):
        continue
    print(k[:70], {a: round(b["mean_KiB_per_dispatch"] * 1024 / 1e6, 1) for a, b in v.items() if isinstance(b, dict)}, "MB/dispatch (raw counters)")
