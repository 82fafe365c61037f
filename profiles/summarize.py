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

if sys.argv[1] == "--merge":
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    doc = json.load(open(dst)) if os.path.exists(dst) else {}
    recs = doc.get("records") or ([doc] if doc.get("kernel") else [])
    for src in sys.argv[2:]:
        rec = json.load(open(os.path.join(src, "pmc_traffic.json")))
        recs = [r for r in recs if not (r.get("kernel") == rec.get("kernel") and str(r.get("config", "2")) == str(rec.get("config", "2")))]
        recs.append(rec)
        print("merged %s config %s from %s" % (rec.get("kernel"), rec.get("config"), src))
    recs.sort(key=lambda r: str(r.get("config", "2")))
    json.dump({"note": "one record per (kernel, config): the counter-derived figures bench.py attaches (labelled PRIOR) to a line of the "
                       "same kernel, config and launch shape; written by profiles/summarize.py from a profiles/collect.sh run",
               "records": recs}, open(dst, "w"), indent=1)
    sys.exit(0)
d = sys.argv[1]
out = {}
# the bench line of the profiled command (kt.log holds its stdout): which config, which launch shape
bench = {}
try:
    for line in open(os.path.join(d, "kt.log")):
        if line.startswith("{") and '"metric"' in line:
            bench = json.loads(line)
except Exception:
    bench = {}
cfg_name = (bench.get("config", {}).get("workload", "config2").split(":")[0] or "config2")
cfg_key = cfg_name.replace("config", "")
ppl = int(bench.get("config", {}).get("passes_per_launch", 64))
spp_pass = int(bench.get("config", {}).get("spp_per_pass", 16))
for f in glob.glob(os.path.join(d, "kt", "**", "*_kernel_stats.csv"), recursive=True):
    print("== kernel-trace stats (%s)" % os.path.relpath(f, d))
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith("pt_"):
            print("  %-24s calls %s avg %.3f ms min %.3f max %.3f  (%s %%)" % (
                row["Name"], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6,
                float(row["MaxNs"]) / 1e6, row["Percentage"]))
            out.setdefault("kernel_stats", {})[row["Name"]] = {
                "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
# the timed launches in the kernel trace of the bench command: per trace kernel, the dispatches within 15 % of
# its longest (the autotune / first-frame legs launch the same kernels on fewer passes)
for f in glob.glob(os.path.join(d, "kt", "**", "*_kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("pt_trace")]
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("== full-size dispatches per trace kernel (%s)" % os.path.relpath(f, d))
    for k, v in sorted(by.items()):
        full = [x for x in v if x >= 0.85 * max(v)]
        print("  %-32s %d of %d dispatches full-size: avg %.3f ms  min %.3f  max %.3f" % (k, len(full), len(v), sum(full) / len(full), min(full), max(full)))
        out.setdefault("kernel_full", {})[k] = {"calls": len(full), "avg_ms": sum(full) / len(full)}
# the other configs: every trace-kernel dispatch in order (pt_tune's trials of each usable path, then
# the measured launch = the last dispatch before the next config's first)
for f in glob.glob(os.path.join(d, "kt_configs", "**", "*_kernel_trace.csv"), recursive=True):
    print("== trace-kernel dispatches of tools/config_sweep.py config3 config4 config5 default (%s)" % os.path.relpath(f, d))
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("pt_trace")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        print("  %-32s %9.3f ms  grid %s threads in workgroups of %s  lds %s B  vgpr %s+%s sgpr %s" % (
            r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", "?"),
            r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"),
            r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?")))
for f in glob.glob(os.path.join(d, "kt_configs.log")):
    print("== tools/config_sweep.py output under the profiler")
    for line in open(f):
        if line.startswith(("config", "default")):
            print("  " + line.rstrip())
# PMC passes: per kernel, only the FULL-SIZE dispatches count (bench.py's autotune and first-frame legs
# launch the same kernels on 8 passes; the timed launches are the long ones): a dispatch is kept when its
# duration is within 15 % of that kernel's longest in the same pass
ctr = defaultdict(lambda: defaultdict(list))
seen_in = defaultdict(set)  # a counter collected in two passes (two pmc directories) counts ONCE, not twice
for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*_counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f))]
    pass_dir = os.path.relpath(f, d).split(os.sep)[0]
    longest = defaultdict(float)
    for row in rows:
        dur = float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        longest[row["Kernel_Name"]] = max(longest[row["Kernel_Name"]], dur)
    for row in rows:
        dur = float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if dur >= 0.85 * longest[row["Kernel_Name"]]:
            key = (row["Kernel_Name"], row["Counter_Name"])
            if seen_in[key] and pass_dir not in seen_in[key]:
                continue  # (a counter collected in two passes: the first pass that holds it counts)
            seen_in[key].add(pass_dir)
            ctr[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
main_kernel = None
if ctr:
    cands = [k for k in ctr if k.startswith("pt_trace")]
    # the kernel bench.py times is the one dispatched most often (autotune trials run once each)
    main_kernel = max(cands, key=lambda k: max(len(v) for v in ctr[k].values())) if cands else None
for k in sorted(ctr):
    if not k.startswith("pt_trace"):
        continue
    print("== PMC per dispatch of %s (mean over dispatches; summed over XCDs/SEs by rocprofv3)%s" % (
        k, "  <-- timed kernel" if k == main_kernel else ""))
    vals = {}
    for name in sorted(ctr[k]):
        per = defaultdict(float)
        for disp, v in ctr[k][name]:
            per[disp] += v
        mean = sum(per.values()) / max(len(per), 1)
        vals[name] = mean
        print("  %-28s %.6g" % (name, mean))
    if k == main_kernel:
        out["pmc"] = vals
        out["pmc_kernel"] = k
    g = vals
    if "SQ_INSTS_VALU" in g and "GRBM_GUI_ACTIVE" in g:
        print("  -- derived: cycles per VALU instruction per SIMD = %.3f" % ((g["GRBM_GUI_ACTIVE"] / 8.0) * 1024.0 / g["SQ_INSTS_VALU"]))
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
    rec = {"pt_trace_kernel_hbm_bytes_per_launch": int(traffic), "kernel": out.get("pmc_kernel"),
           "config": cfg_key,
           "workload": "bench.py --config %s (%s; %d passes of %d spp per launch; pt_tune runs each usable path, then the timed launches use the fastest)"
                       % (cfg_key, bench.get("config", {}).get("workload", "?"), ppl, spp_pass),
           "spp_per_pass": spp_pass, "passes_per_launch": ppl, "hbm_bytes_per_pass": int(traffic / ppl),
           "profile": "profiles/" + os.path.basename(os.path.normpath(d)).replace("prof_", "") + "_summary.txt",
           "write_size_kib": pm["WRITE_SIZE"], "fetch_size_kib": pm["FETCH_SIZE"],
           # the build the counters belong to (the profiled bench line says what it ran on: ray_tracer_webgl_amd/_lib.py
           # build_identity); bench.py attaches the record only to lines of the same build
           "csrc_sha256": (bench.get("build") or {}).get("csrc_sha256"), "lib_sha256": (bench.get("build") or {}).get("lib_sha256")}
    # what the kernel actually issued (the hierarchy walk skips most of the algorithmic tests):
    # wave-level VALU instructions per launch and the share of the chip's VALU issue slots they
    # fill (one wave64 instruction occupies a SIMD's 32 lanes for 2 cycles; GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs)
    if "SQ_INSTS_VALU" in pm and "GRBM_GUI_ACTIVE" in pm:
        cycles = pm["GRBM_GUI_ACTIVE"] / 8.0
        rec["valu_insts_per_launch"] = pm["SQ_INSTS_VALU"]
        rec["kernel_cycles"] = cycles
        rec["valu_issue_frac"] = 2.0 * pm["SQ_INSTS_VALU"] / (1024.0 * cycles)
        for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_INT32",
                  "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_ADD_F16", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F64",
                  "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_LDS_BANK_CONFLICT",
                  "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES",
                  "SQ_INSTS_VMEM_RD", "TCC_HIT_sum", "TCC_MISS_sum"):
            if k in pm:
                rec[k.lower()] = pm[k]
    # the same kernel's average duration in the kernel-trace pass of the same command
    ks = out.get("kernel_full", {}).get(out.get("pmc_kernel")) or out.get("kernel_stats", {}).get(out.get("pmc_kernel"))
    if ks:
        rec["kernel_ms"] = ks["avg_ms"]
    json.dump(rec, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)

# merge the per-config records into the committed file bench.py reads:  summarize.py --merge <dir> [<dir> ...]
