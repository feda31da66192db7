#!/usr/bin/env python3
"""Print the markdown tables of profiles/<tag>/README.md from the committed summaries (scripts/summarize_profile.py).

    python scripts/profile_readme.py <tag> [<tag> ...]        e.g.  r03 r03_config3 r03_config5 r03_reference_layout

For every tag: the kernel_stats table (kernels above 1 % of the trace, or the six largest), the HBM traffic table (FETCH_SIZE x 2,
WRITE_SIZE, both KiB -> GB: /opt/skills/guides/MI355X_MICROARCH.md) and, where runs.json lists more than one run, the per-run
averages of the dominant kernel.  The prose of the README is written by hand around these tables."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def tables(tag, algorithmic_gb=None):
    d = os.path.join(ROOT, "profiles", tag)
    out = []
    runs = json.load(open(os.path.join(d, "runs.json")))
    if len(runs["average_us_per_run"]) > 1:
        out.append("| gpurun call | `%s` average (µs) |%s" % (runs["dominant_kernel"], " algorithmic GB / t | of 8 TB/s |" if algorithmic_gb else ""))
        out.append("|---|---|" + ("---|---|" if algorithmic_gb else ""))
        for r, us in runs["average_us_per_run"].items():
            tail = ""
            if algorithmic_gb:
                tbs = algorithmic_gb * 1e9 / (us * 1e-6) / 1e12
                tail = " %.2f TB/s | %.1f %% |" % (tbs, 100 * tbs / 8.0)
            mark = " **(median: committed as `kernel_stats.csv`)**" if r == runs["median_run"] else ""
            out.append("| `%s`%s | %.1f |%s" % (r, mark, us, tail))
        out.append("")
    rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
    keep = [r for r in rows if float(r["Percentage"]) >= 1.0]
    if len(keep) < 6:
        keep = rows[:6]
    out.append("| kernel | calls | average µs | min | max | % |")
    out.append("|---|---|---|---|---|---|")
    for r in keep:
        out.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    out.append("")
    pmc = json.load(open(os.path.join(d, "pmc_hbm_bytes.json")))
    out.append("| kernel | fetched (× 2) | written | total GB per launch |")
    out.append("|---|---|---|---|")
    for k, v in pmc.items():
        if not isinstance(v, dict):
            continue
        f = 2 * v.get("FETCH_SIZE", {}).get("mean_KiB_per_dispatch", 0.0) * 1024 / 1e9
        w = v.get("WRITE_SIZE", {}).get("mean_KiB_per_dispatch", 0.0) * 1024 / 1e9
        if f + w >= 0.01:
            out.append("| `%s` | %.3f | %.3f | %.3f |" % (short(k), f, w, f + w))
    out.append("")
    out.append("(`_csrc_revision` %s)" % pmc.get("_csrc_revision"))
    return "\n".join(out)


if __name__ == "__main__":
    for t in sys.argv[1:]:
        alg = 3.89216256 if t.split("_")[0] == t else None          # the headline tag: gray_stream_kernel<4,4>'s algorithmic GB
        print("### %s\n" % t)
        print(tables(t, alg))
        print()
