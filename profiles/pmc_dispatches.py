#!/usr/bin/env python3
"""Per-dispatch PMC table for any program run under several `rocprofv3 --pmc` passes.

    python3 profiles/pmc_dispatches.py <dir with pmc*/ subdirectories> [kernel-name prefix]

Every pass of a deterministic program dispatches the same kernels in the same order, so the k-th
dispatch of a trace kernel in one pass is the k-th in every other: the counters of the separate passes
are joined on that ordinal.  Prints, per trace-kernel dispatch: name, launch shape, duration under the
profiler (ms, from the pass that carried GRBM_GUI_ACTIVE if any) and every counter, followed by the
derived figures DESIGN.md quotes:

    clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration          (GRBM counts on each of the 8 XCDs)
    cyc_per_valu     = (GRBM_GUI_ACTIVE / 8) * 1024 SIMDs / SQ_INSTS_VALU
    valu_issue_frac  = 2 / cyc_per_valu                          (a wave64 VALU op holds a SIMD 2 cycles)
    lane_util        = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
    cu_coverage      = SQ_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4)    (SQ_BUSY_CYCLES sums the 32 SEs' SQs; 4 per XCD)
    wave_residency   = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * SQ_WAVES)   (share of the launch a wave is alive; quad-cycles)
    wait_frac        = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
    salu_per_valu, branch_per_valu, lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "pt_trace"

passes = sorted(glob.glob(os.path.join(d, "pmc*")))
table = defaultdict(dict)   # ordinal -> {counter: value}
meta = {}                   # ordinal -> (name, grid, wg, lds, vgpr, sgpr)
dur = defaultdict(list)     # ordinal -> [ms per pass]
for pdir in passes:
    if not os.path.isdir(pdir):
        continue
    for f in glob.glob(os.path.join(pdir, "**", "*_counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        info = {}
        for row in csv.DictReader(open(f)):
            if not row["Kernel_Name"].startswith(prefix):
                continue
            did = int(row["Dispatch_Id"])
            per[did][row["Counter_Name"]] += float(row["Counter_Value"])
            info[did] = (row["Kernel_Name"], row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"],
                         row["VGPR_Count"], row["SGPR_Count"],
                         (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        for k, did in enumerate(sorted(per)):
            table[k].update(per[did])
            name = info[did][0]
            if k in meta and meta[k][0] != name:
                print("!! dispatch %d is %s in one pass and %s in another: joined anyway" % (k, meta[k][0], name))
            meta[k] = info[did][:6]
            dur[k].append(info[did][6])

for k in sorted(table):
    c = table[k]
    name, grid, wg, lds, vgpr, sgpr = meta[k]
    ms = sorted(dur[k])[len(dur[k]) // 2]
    print("== dispatch %d: %s  grid %s / wg %s  lds %s B  vgpr %s sgpr %s  %.3f ms (median of %d passes: %s)" % (
        k, name, grid, wg, lds, vgpr, sgpr, ms, len(dur[k]), " ".join("%.2f" % x for x in dur[k])))
    for n in sorted(c):
        print("   %-26s %.6g" % (n, c[n]))
    g = c.get("GRBM_GUI_ACTIVE")
    der = []
    if g:
        der.append("clock_ghz %.3f" % (g / 8.0 / (ms * 1e6)))
        if c.get("SQ_INSTS_VALU"):
            cpv = (g / 8.0) * 1024.0 / c["SQ_INSTS_VALU"]
            der.append("cyc_per_valu %.3f  valu_issue_frac %.3f" % (cpv, 2.0 / cpv))
        if c.get("SQ_BUSY_CYCLES"):
            der.append("cu_coverage %.3f" % (c["SQ_BUSY_CYCLES"] / (g * 4.0)))
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):  # SQ_WAVE_CYCLES counts quad-cycles
            der.append("wave_residency %.3f" % (4.0 * c["SQ_WAVE_CYCLES"] / ((g / 8.0) * c["SQ_WAVES"])))
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        der.append("lane_util %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])))
    if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES"):
        der.append("wait_frac %.3f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    if c.get("SQ_INSTS_VALU"):
        if c.get("SQ_INSTS_SALU"):
            der.append("salu_per_valu %.3f" % (c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]))
        if c.get("SQ_INSTS_BRANCH"):
            der.append("branch_per_valu %.3f" % (c["SQ_INSTS_BRANCH"] / c["SQ_INSTS_VALU"]))
        if c.get("SQ_INSTS_VALU_FMA_F32") is not None and c.get("SQ_INSTS_VALU_MUL_F32") is not None and \
                c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            lu = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            flop = (2.0 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"]) * 64.0 * lu
            der.append("fp32_tflops(fma+mul, lane-weighted) %.2f" % (flop / (ms * 1e-3) / 1e12))
    if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_LDS_BANK_CONFLICT") is not None:
        der.append("lds_conflict_frac %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
    if c.get("WRITE_SIZE") is not None and c.get("FETCH_SIZE") is not None:
        der.append("hbm_bytes(WRITE + 2*FETCH KiB) %.4g" % ((c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"]) * 1024.0))
    print("   -- " + "  ".join(der))
