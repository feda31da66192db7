#!/usr/bin/env python3
"""profiles/<tag>/boxes.md from the JSON lines scripts/box_identity.py left in gpurun_out/ (one per gpurun call that ran it).
    python scripts/boxes_md.py r04 gpurun_out/r4_box_*.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, files = sys.argv[1], sys.argv[2:]
rows = []
for f in files:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        continue
    rows.append((os.path.basename(f), d))
out = ["# The boxes of this round's gpurun calls (`scripts/box_identity.py`)", "",
       "One row per call that ran the script (a call gets whatever MI355X box is free).  Same binary everywhere; settled timings (settle loop,",
       "then 60 steps); clock and package power as `rocm-smi` reports them while 300 more steps are queued.", "",
       "| call | PCI / unique id | HBM vendor, VBIOS, partitions | device copy | config 2 step / kernel (ms) | clock, power | config 5 step / kernel (ms) | clock, power |",
       "|---|---|---|---|---|---|---|---|"]
for name, d in rows:
    smi = d.get("smi", {})
    c2, c5 = d.get("config2", {}), d.get("config5", {})
    out.append("| `%s` | %s / %s | %s, %s, %s / %s%s | %.2f TB/s | %.4f / %.4f | %s, %s W | %.4f / %.4f | %s, %s W |" % (
        name, d.get("pci", ""), smi.get("Unique ID", ""), smi.get("GPU memory vendor", ""), smi.get("VBIOS version", ""),
        smi.get("Compute Partition", ""), smi.get("Memory Partition", ""),
        (", max power %s W" % smi["Max Graphics Package Power (W)"]) if "Max Graphics Package Power (W)" in smi else "",
        d.get("copy_GBs", 0) / 1e3, c2.get("ms_per_step", 0), c2.get("kernel_ms", 0), c2.get("sclk", ""), c2.get("power_W", ""),
        c5.get("ms_per_step", 0), c5.get("kernel_ms", 0), c5.get("sclk", ""), c5.get("power_W", "")))
extra = os.path.join(ROOT, "profiles", tag, "boxes_notes.md")
if os.path.exists(extra):
    out += ["", open(extra).read().rstrip()]
open(os.path.join(ROOT, "profiles", tag, "boxes.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
