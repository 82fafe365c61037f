#!/usr/bin/env python3
"""Condenses a profiles/collect.sh output directory into the per-kernel summary committed under
profiles/: kernel-trace stats + PMC counter totals per dispatch of pt_trace_kernel, with the
gfx950 corrections of MI355X_MICROARCH.md §HBM applied (FETCH_SIZE x2 for wide coalesced reads is
NOT applied to this kernel's scattered 16-B gathers; both raw and KiB->bytes values are shown)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(d, "kt", "**", "*_kernel_stats.csv"), recursive=True):
    print("== kernel-trace stats (%s)" % os.path.relpath(f, d))
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith("pt_"):
            print("  %-24s calls %s avg %.3f ms min %.3f max %.3f  (%s %%)" % (
                row["Name"], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6,
                float(row["MaxNs"]) / 1e6, row["Percentage"]))
            out.setdefault("kernel_stats", {})[row["Name"]] = {
                "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
ctr = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        ctr[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for k in sorted(ctr):
    if not k.startswith("pt_trace"):
        continue
    print("== PMC per dispatch of %s (mean over dispatches; summed over XCDs/SEs by rocprofv3)" % k)
    vals = {}
    for name in sorted(ctr[k]):
        per = defaultdict(float)
        for disp, v in ctr[k][name]:
            per[disp] += v
        mean = sum(per.values()) / max(len(per), 1)
        vals[name] = mean
        print("  %-28s %.6g" % (name, mean))
    out["pmc"] = vals
    g = vals
    if "SQ_WAVE_CYCLES" in g and "SQ_ACTIVE_INST_VALU" in g:
        print("  -- derived")
        if g.get("SQ_BUSY_CYCLES"):
            print("  VALU-active / busy-cycles (per SE sum)     %.3f" % (g["SQ_ACTIVE_INST_VALU"] / g["SQ_BUSY_CYCLES"]))
        if g.get("SQ_INSTS_VALU") and g.get("SQ_THREAD_CYCLES_VALU") and g.get("SQ_ACTIVE_INST_VALU"):
            print("  lanes active per VALU instr (of 64)        %.2f" % (g["SQ_THREAD_CYCLES_VALU"] / g["SQ_ACTIVE_INST_VALU"] / 4 * 1.0))
        print("  VALU insts per wave                        %.4g" % (g["SQ_INSTS_VALU"] / max(g.get("SQ_WAVES", 1), 1)))
    if "FETCH_SIZE" in g:
        print("  FETCH_SIZE KiB %.6g -> bytes %.6g (x2 if wide coalesced: %.6g)" % (g["FETCH_SIZE"], g["FETCH_SIZE"] * 1024, g["FETCH_SIZE"] * 2048))
    if "WRITE_SIZE" in g:
        print("  WRITE_SIZE KiB %.6g -> bytes %.6g" % (g["WRITE_SIZE"], g["WRITE_SIZE"] * 1024))
# HBM traffic of pt_trace_kernel per launch, corrected as MI355X_MICROARCH.md §HBM prescribes:
# WRITE_SIZE (KiB) is exact for 16-B/lane stores and atomics; FETCH_SIZE (KiB) reads 1/2 of a wide
# coalesced stream -> x2 (upper bound for this kernel's small, partly scattered reads).
pm = out.get("pmc", {})
if "WRITE_SIZE" in pm and "FETCH_SIZE" in pm:
    traffic = pm["WRITE_SIZE"] * 1024 + 2 * pm["FETCH_SIZE"] * 1024
    out["pt_trace_kernel_hbm_bytes_per_launch"] = int(traffic)
    print("  HBM traffic per launch (WRITE_SIZE + 2*FETCH_SIZE): %.4g bytes" % traffic)
    json.dump({"pt_trace_kernel_hbm_bytes_per_launch": int(traffic),
               "workload": "bench.py --steps 16 --warmup 16 (config 2, 16 passes per launch)",
               "write_size_kib": pm["WRITE_SIZE"], "fetch_size_kib": pm["FETCH_SIZE"]},
              open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
