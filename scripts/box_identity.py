#!/usr/bin/env python3
"""What kind of box is this?  Prints one JSON line: device identity (name, CUs, PCI bus id, VBIOS, compute / memory partition mode as far
as rocm-smi tells), the device copy rate, and settled step / kernel times of configs 2 and 5 with the clock and package power
rocm-smi reports under each.  Run once per gpurun call: the pool's boxes differ by up to 10 % (config 5: 27 %) on the same binary,
and profiles/r04/README.md tabulates them.
    python scripts/box_identity.py"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd import distributed as D


def sh(cmd):
    try:
        return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=30).stdout
    except (OSError, subprocess.SubprocessError):
        return ""


def smi_json(*flags):
    try:
        d = json.loads(sh(["rocm-smi", "--json"] + list(flags)))
        return d.get("card0", next(iter(d.values())))
    except (ValueError, StopIteration):
        return {}


dev = torch.device("cuda", 0)
p = torch.cuda.get_device_properties(0)
out = {"device": p.name, "cus": p.multi_processor_count, "gcn": getattr(p, "gcnArchName", ""),
       "pci": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0) & 0xff, getattr(p, "pci_device_id", 0)),
       "total_mem_GiB": round(p.total_memory / 2 ** 30, 1)}
ident = smi_json("--showvbios", "--showcomputepartition", "--showmemorypartition", "--showperflevel", "--showmemvendor", "--showserial", "--showuniqueid",
                 "--showmaxpower")
out["smi"] = {k: v for k, v in ident.items() if any(s in k.lower() for s in ("vbios", "partition", "perf", "vendor", "unique", "max"))}
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
out["copy_GBs"] = round(2 * a.numel() * 4 / (np.median(ts[3:]) * 1e6), 0)
del a, b
for name in ("config2", "config5"):
    wl = bench.WORKLOADS[name]
    B = wl["frames"]
    pipe = bench.make_pipeline(wl, B, 0, None)
    frames = bench.make_frames(torch, D, wl, B, 0, 1, dev)
    bench.settle(torch, pipe, frames, dev)
    t0 = time.perf_counter()
    for _ in range(60):
        pipe.step(frames)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 60 * 1e3
    for _ in range(300):
        pipe.step(frames)
    s = smi_json("--showclocks", "--showpower", "--showtemp")
    torch.cuda.synchronize()
    dom = bench.dominant_kernel(torch, pipe, frames, wl, B, dev, launches=16)
    out[name] = {"ms_per_step": round(step, 4), "kernel_ms": round(dom["ms"], 4),
                 "sclk": s.get("sclk clock speed:", ""), "mclk": s.get("mclk clock speed:", ""),
                 "power_W": s.get("Current Socket Graphics Package Power (W)", ""), "T_mem": s.get("Temperature (Sensor memory) (C)", "")}
    del pipe, frames
    torch.cuda.empty_cache()
print(json.dumps(out))
