#!/usr/bin/env python3
"""Dev tool (no GPU needed): WHAT does a trace kernel's instruction stream consist of, phase by phase?

    python tools/instruction_mix.py [--kernel pt_trace_kernel_grid] [--tallies profiles/r05_twin_tallies_config2.json]
                                    [--pmc-config 2] [--blocks]

1. builds ray_tracer_webgl_amd/csrc/pt_kernels.hip for gfx950 with line tables (-g: the instruction stream is the shipped
   one — the tool checks the instruction count against a build without -g), disassembles the kernel (llvm-objdump) and asks
   llvm-symbolizer for every instruction's INLINE STACK: the frame directly inside pt_trace_body is the phase (refill,
   start_sample, grid_walk, shade_segment, ...), the source line inside that function the sub-phase (anchors in the
   sources below, not line numbers);
2. splits the stream into basic blocks (branch targets and fall-throughs), tags each block with the region most of its
   instructions belong to, and classifies every instruction (fp32 arithmetic / transcendental, selects and moves, integer and
   address arithmetic, compares, cross-lane, LDS, global memory, scalar ALU, scalar loads, hazard s_nop, s_waitcnt,
   exec-mask regions, branches);
3. weights every block with how often its region runs — the measuring twin's tallies (loop trips: wave steps, cell steps,
   leaf rounds, exact evaluations; region counts: Tally::flag / collect in pt_scene.hpp), dumped by `WL_JSON=... python
   tools/wave_log.py` on the GPU box and committed under profiles/ — giving DYNAMIC counts per phase and class;
4. cross-checks the totals against the committed PMC record of the same kernel and launch (profiles/pmc_traffic.json:
   SQ_INSTS_VALU, _FMA_F32, _MUL_F32, _ADD_F32, _TRANS_F32, _INT32, _CVT, SQ_INSTS_SALU, _SMEM, _LDS, _BRANCH).

A region's weight is the number of wave steps in which ANY lane entered it; blocks inside a region that the wave skips
(s_cbranch_execz around a sub-branch no lane takes) are still counted, so the model is an upper bound by a few per cent.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracer_webgl_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden".split()


def build(tmp, tu, debug):
    out = os.path.join(tmp, "g" if debug else "n")
    os.makedirs(out, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + FLAGS + (["-g"] if debug else []) + [
        "-save-temps=obj", "--cuda-device-only", "-c", os.path.join(CSRC, tu + ".hip"), "-o", os.path.join(out, tu + ".o")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=CSRC)
    return os.path.join(out, tu + "-hip-amdgcn-amd-amdhsa-gfx950.out")


def disassemble(obj, kernel):
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", obj], check=True, stdout=subprocess.PIPE, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^\n|\Z)" % re.escape(kernel), txt, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not in %s" % (kernel, obj))
    insts = []
    for line in m.group(1).splitlines():
        mm = re.match(r"\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):(?:.*<[^>]*\+0x([0-9a-f]+)>)?", line)
        if mm:
            insts.append({"op": mm.group(1), "args": mm.group(2), "addr": int(mm.group(3), 16),
                          "target": int(mm.group(4), 16) if mm.group(4) else None})
    for k, i in enumerate(insts):  # what follows the last s_endpgm is padding (s_nop up to the next kernel's alignment)
        if i["op"] == "s_endpgm":
            last = k
    insts = insts[: last + 1]
    base = insts[0]["addr"]
    for i in insts:
        if i["target"] is not None and (i["op"].startswith("s_cbranch") or i["op"] == "s_branch"):
            i["target"] += base
        else:
            i["target"] = None
    return insts


def symbolize(obj, insts):
    inp = "".join("0x%x\n" % i["addr"] for i in insts)
    out = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + obj, "--inlines", "--functions=short"], input=inp, check=True,
                         stdout=subprocess.PIPE, text=True).stdout.strip().split("\n\n")
    assert len(out) == len(insts), (len(out), len(insts))
    for i, b in zip(insts, out):
        L = b.split("\n")
        fr = []
        for k in range(0, len(L) - 1, 2):
            loc = L[k + 1].rsplit(":", 2)
            fr.append((L[k].split("<")[0], os.path.basename(loc[0]), int(loc[1]) if loc[1].isdigit() else 0))
        i["stack"] = fr  # innermost first: (function, file, line)


def anchors(fname, spec):
    """[(region, first line, last line)] from pairs of anchor substrings in a source file."""
    lines = open(os.path.join(CSRC, fname)).read().split("\n")

    def find(sub, start=0):
        for n in range(start, len(lines)):
            if sub in lines[n]:
                return n + 1
        raise SystemExit("anchor %r not found in %s" % (sub, fname))
    out = []
    for region, a, b in spec:
        la = find(a)
        lb = find(b, la) if b else len(lines)
        out.append((region, la, lb))
    return out


GRID_REGIONS = anchors("pt_grid_walk.hpp", [
    ("walk: exact evaluation", "auto exact_group = [&]", "  };"),
    ("walk: always-tested group", "tally.always_group(true)", "tally.always_group(false)"),
    ("walk: per-ray constants", "per-ray constants of the walk", "entry: where does the half-line"),
    ("walk: entry (slab test)", "entry: where does the half-line", "if (enter) {"),
    ("walk: entry (first cell)", "if (enter) {", "tally.phase(3)"),
    ("walk: cell step", "#define PT_CELL_STEP", "#undef PT_CELL_STEP"),
    ("walk: leaf round", "tally.phase(4)", "tally.phase(5)"),
    ("walk: loop control", "tally.phase(5)", "carried = rem != 0u"),
])
SHADE_REGIONS = anchors("pt_shade.hpp", [
    ("shade: hit record + normal", "if (hit >= 0) {", "const bool sky = hit < 0"),
    ("shade: shared (1/|d|, seed hash)", "const bool sky = hit < 0", "if (hit < 0) {"),
    ("shade: sky", "if (hit < 0) {", "const V3 p = hp;"),
    ("shade: diffuse / metal (shared draw)", "if (mtype == 0 || mtype == 1) {", "if (mtype == 0) { // DIFFUSE"),
    ("shade: diffuse", "if (mtype == 0) { // DIFFUSE", "} else { // METAL"),
    ("shade: metal", "} else { // METAL", "      if (ok) {"),
    ("shade: diffuse / metal (outcome)", "      if (ok) {", "} else if (mtype == 2) {"),
    ("shade: glass", "} else if (mtype == 2) {", "} else if (mtype == 3) {"),
    ("shade: emissive / other", "} else if (mtype == 3) {", "if (!finished) {"),
    ("shade: depth bookkeeping", "if (!finished) {", "  if (finished) {"),
    ("shade: sample / item end", "  if (finished) {", "p.alive = alive; p.new_path = new_path; p.item_segs"),
])
REFILL_REGIONS = anchors("pt_refill.hpp", [
    ("refill: reservation", "if (pool_next == pool_end) {", "const uint32_t avail = pool_end - pool_next;"),
    ("refill: deal + item decode", "const uint32_t avail = pool_end - pool_next;", "refill_waited = (dealt ? 0u"),
])


def region_of(stack):
    """Region of one instruction from its inline stack (innermost first)."""
    idx = [k for k, f in enumerate(stack) if f[0] == "pt_trace_body"]
    if not idx:
        return "kernel prologue / epilogue" if stack and stack[-1][0].startswith("pt_trace_kernel") else None
    k = idx[0]
    if k == 0:
        return "step glue (ballots, loop)" if stack[0][2] else None
    phase = stack[k - 1][0]

    def line_in(fname):
        for f in stack[:k]:  # the innermost frame that lies in this file
            if f[1] == fname and f[2]:
                return f[2]
        return 0
    if phase == "grid_walk":
        ln = line_in("pt_grid_walk.hpp")
        hit = [r for r, a, b in GRID_REGIONS if a <= ln <= b]
        if hit and hit[0] == "walk: exact evaluation":
            # the lambda is inlined at both of its call sites: the frame that calls it says which copy this is
            outer = [f[2] for f in stack[:k] if f[1] == "pt_grid_walk.hpp" and f[2] and not (GRID_REGIONS[0][1] <= f[2] <= GRID_REGIONS[0][2])]
            site = [r for r, a, b in GRID_REGIONS if outer and a <= outer[0] <= b]
            return "walk: exact evaluation (always-tested)" if site and site[0] == "walk: always-tested group" else "walk: exact evaluation (cell entries)"
        return hit[0] if hit else ("walk: set-up / write-back" if ln else None)
    if phase == "shade_segment":
        ln = line_in("pt_shade.hpp")
        hit = [r for r, a, b in SHADE_REGIONS if a <= ln < b]
        return hit[0] if hit else ("shade: entry / write-back" if ln else None)
    if phase == "refill":
        ln = line_in("pt_refill.hpp")
        hit = [r for r, a, b in REFILL_REGIONS if a <= ln < b]
        return hit[0] if hit else ("refill: loop head / write-back" if ln else None)
    if phase == "start_sample":
        return "camera ray"
    if phase in ("park_store", "park_load"):
        return "park / unpark (LDS)"
    if phase in ("literal_loop", "tail_mode"):
        return "literal loop (rare)"
    if phase in ("stage", "pixel_div", "GridWalk", "BvhWalk", "Carry", "Path", "Queue", "Tally"):
        return "kernel prologue / epilogue"
    if phase in ("regular_ray",):
        return "step glue (ballots, loop)"
    if phase in ("atomicAdd", "flush"):
        return "kernel prologue / epilogue"
    return "step glue (ballots, loop)"


CLASSES = ["fp32 fma", "fp32 mul", "fp32 add/sub", "fp32 min/max/med/other", "transcendental", "select / move", "compare", "integer / bit / address",
           "convert", "cross-lane", "LDS", "global memory", "SALU", "scalar load", "s_nop", "s_waitcnt", "exec-mask region", "branch", "other scalar"]


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mad_f32", "v_mac_f32", "v_madmk_f32", "v_madak_f32", "v_pk_fma_f32")):
            return "fp32 fma"
        if op.startswith(("v_mul_f32", "v_mul_legacy_f32", "v_pk_mul_f32")):
            return "fp32 mul"
        if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32")):
            return "fp32 add/sub"
        if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
            return "transcendental"
        if op.startswith(("v_cndmask", "v_mov_b", "v_accvgpr", "v_swap")):
            return "select / move"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "compare"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute")) or "dpp" in op:
            return "cross-lane"
        if op.startswith("v_cvt") or op.startswith(("v_floor", "v_trunc", "v_rndne", "v_ceil", "v_fract")):
            return "convert"
        if op.startswith(("v_min_f32", "v_max_f32", "v_med3_f32", "v_min3_f32", "v_max3_f32", "v_ldexp", "v_frexp", "v_div_", "v_fma_mix")):
            return "fp32 min/max/med/other"
        return "integer / bit / address"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "global memory"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_store", "s_atomic")):
        return "scalar load"
    if "saveexec" in op:
        return "exec-mask region"
    if op.startswith("s_"):
        return "SALU" if not op.startswith(("s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_sendmsg", "s_setreg", "s_getreg", "s_icache")) else "other scalar"
    return "other scalar"


def blocks_of(insts):
    starts = {insts[0]["addr"]}
    for k, i in enumerate(insts):
        if i["target"] is not None:
            starts.add(i["target"])
            if k + 1 < len(insts):
                starts.add(insts[k + 1]["addr"])
    out, cur = [], []
    for i in insts:
        if i["addr"] in starts and cur:
            out.append(cur)
            cur = []
        cur.append(i)
    if cur:
        out.append(cur)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="pt_trace_kernel_grid")
    ap.add_argument("--tu", default="pt_kernels")
    ap.add_argument("--tallies", default=os.path.join(ROOT, "profiles", "r05_twin_tallies_config2.json"))
    ap.add_argument("--pmc-config", default="2")
    ap.add_argument("--blocks", action="store_true", help="list every basic block")
    args = ap.parse_args()

    with tempfile.TemporaryDirectory() as tmp:
        obj = build(tmp, args.tu, True)
        insts = disassemble(obj, args.kernel)
        symbolize(obj, insts)
        plain = disassemble(build(tmp, args.tu, False), args.kernel)
    same = [a["op"] for a in insts] == [b["op"] for b in plain]
    same_set = collections.Counter(a["op"] for a in insts) == collections.Counter(b["op"] for b in plain)
    moved = sum(1 for a, b in zip(insts, plain) if a["op"] != b["op"])
    print("# %s: %d instructions; against the shipped build (no -g, %d instructions) the line-table build has %s"
          % (args.kernel, len(insts), len(plain), "the same instruction stream" if same else
             ("the same instructions with %d of them scheduled a few slots away (register names aside)" % moved if same_set else "a DIFFERENT instruction mix")))

    for i in insts:
        i["region"] = region_of(i["stack"])
        i["cls"] = classify(i["op"])
    blocks = blocks_of(insts)
    prev = "kernel prologue / epilogue"
    for b in blocks:  # a block belongs to the region most of its attributed instructions name; line-0 code inherits
        votes = collections.Counter(i["region"] for i in b if i["region"])
        reg = votes.most_common(1)[0][0] if votes else prev
        for i in b:
            i["block_region"] = i["region"] or reg
        b[0]["block_tag"] = reg
        prev = reg

    # COLD blocks: the plain `/` and sqrtf expansions (v_div_scale / v_div_fmas / v_div_fixup; sqrtf's range scaling by 2^32 / 2^-16)
    # are the FALLBACK of the unscaled forms the kernels run (pt_arith.hpp div_core, sqrt_core, hit_root: same bits for operands in
    # a stated exponent range; a wave takes the plain operators only when one of its lanes is outside — practically never), and
    # what is reachable only through them.  They are laid out of line and get weight 0.
    index = {b[0]["addr"]: n for n, b in enumerate(blocks)}
    succ = []
    for n, b in enumerate(blocks):
        last = b[-1]
        out = []
        if last["target"] is not None and last["target"] in index:
            out.append(index[last["target"]])
        if last["op"] not in ("s_branch", "s_endpgm") and n + 1 < len(blocks):
            out.append(n + 1)
        succ.append(out)
    preds = [[] for _ in blocks]
    for n, out in enumerate(succ):
        for m in out:
            preds[m].append(n)
    # (GLASS divides and takes square roots with the plain operators on purpose — `1.0f / ri`, sqrt_rn, once per glass hit: hot code)
    hot_plain = ("shade: glass",)
    cold = [b[0]["block_tag"] not in hot_plain and
            any(i["op"] in ("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32") or
                (i["op"].startswith("v_mul_f32") and ("0x4f800000" in i["args"] or "0x37800000" in i["args"])) for i in b) for b in blocks]
    changed = True
    while changed:
        changed = False
        for n in range(len(blocks)):
            if not cold[n] and preds[n] and all(cold[m] for m in preds[n] if m != n):
                cold[n] = True
                changed = True
    n_cold = 0
    for n, b in enumerate(blocks):
        if cold[n]:
            n_cold += len(b)
            for i in b:
                i["block_region"] = "(cold: plain / and sqrtf fallbacks)"
            b[0]["block_tag"] = "(cold: plain / and sqrtf fallbacks)"
    print("# %d instructions in %d cold blocks (fallbacks of the unscaled division / square-root forms): weight 0" % (n_cold, sum(cold)))

    # the cell step exists twice (its first trip is written out in front of its loop, pt_grid_walk.hpp): the first contiguous run of
    # blocks is the written-out trip, what follows the loop's copy
    run, seen_gap = 0, True
    for b in blocks:
        if b[0]["block_tag"] == "walk: cell step":
            if seen_gap:
                run += 1
                seen_gap = False
            name = "walk: cell step (first trip)" if run == 1 else "walk: cell step (loop)"
            for i in b:
                if i["block_region"] == "walk: cell step":
                    i["block_region"] = name
        else:
            seen_gap = True

    # ---- static table -------------------------------------------------------------------------------------------
    regions = []
    for i in insts:
        if i["block_region"] not in regions:
            regions.append(i["block_region"])
    static = {r: collections.Counter() for r in regions}
    for i in insts:
        static[i["block_region"]][i["cls"]] += 1
    valu_cls = [c for c in CLASSES if c in ("fp32 fma", "fp32 mul", "fp32 add/sub", "fp32 min/max/med/other", "transcendental", "select / move", "compare",
                                            "integer / bit / address", "convert", "cross-lane")]

    def valu(cnt):
        return sum(cnt[c] for c in valu_cls)
    print("\n## static instructions per region (%d basic blocks)" % len(blocks))
    print("%-36s %6s %6s | %5s %5s %5s %5s %5s | %5s %5s %5s %5s %5s | %5s %5s %5s %5s %5s %5s" % (
        "region", "insts", "VALU", "fma", "mul", "add", "mnmx", "trans", "sel/mv", "cmp", "int", "cvt", "xlane", "LDS", "gmem", "SALU", "sload", "nop", "branch"))

    def row(name, cnt, scale=1.0, fmt="%5d"):
        v = [cnt[c] * scale for c in ("fp32 fma", "fp32 mul", "fp32 add/sub", "fp32 min/max/med/other", "transcendental", "select / move", "compare",
                                      "integer / bit / address", "convert", "cross-lane", "LDS", "global memory")]
        s = (cnt["SALU"] + cnt["exec-mask region"] + cnt["other scalar"]) * scale
        tail = [s, cnt["scalar load"] * scale, cnt["s_nop"] * scale, cnt["branch"] * scale]
        f6 = fmt.replace("5", "6")
        print(("%-36s " + f6 + " " + f6 + " | " + " ".join([fmt] * 5) + " | " + " ".join([fmt] * 5) + " | " + " ".join([fmt] * 6)) % (
            (name, sum(cnt.values()) * scale, valu(cnt) * scale) + tuple(v[:5]) + tuple(v[5:10]) + tuple(v[10:12]) + tuple(tail)))
    tot = collections.Counter()
    for r in regions:
        row(r, static[r])
        tot.update(static[r])
    row("TOTAL", tot)
    if args.blocks:
        print("\n## basic blocks")
        for b in blocks:
            c = collections.Counter(i["cls"] for i in b)
            print("  %08x  %4d insts  %-36s valu %3d  %s" % (b[0]["addr"], len(b), b[0].get("block_tag", "?"), valu(c), b[-1]["op"] + " " + b[-1]["args"]))

    # ---- dynamic: weight regions with the twin's tallies -----------------------------------------------------------
    if not os.path.exists(args.tallies):
        print("\n(no tallies file %s: static table only)" % args.tallies)
        return 0
    T = json.load(open(args.tallies))
    ctr = T["counters"]
    W = float(ctr[8 + 6])  # wave steps
    walk_it, leaf_it, exact_it = float(ctr[8 + 0]), float(ctr[8 + 2]), float(ctr[8 + 4])
    R = {k: float(ctr[32 + 2 * k]) for k in range(16)}
    RL = {k: float(ctr[32 + 2 * k + 1]) for k in range(16)}
    names = ["REFILL_DECODE", "REFILL_RESERVE", "CAMERA_RAY", "SHADE_HIT_RECORD", "SHADE_SKY", "SHADE_DIFFUSE", "SHADE_METAL", "SHADE_GLASS",
             "SHADE_GLASS_REFRACT", "SHADE_CONTINUES", "SHADE_FINISHED", "SHADE_ITEM_STORE", "WALK_ENTRY", "WALK_ENTER_CELL", "WALK_FAR_RAY", "SHADE_ANY"]
    reg = dict(zip(names, range(16)))
    n_waves = float(T["waves"])
    lit_steps = float(ctr[16 + 2])
    exact_always, walk_first = float(ctr[16 + 5]), float(ctr[16 + 7])
    n_always_groups = max(1.0, (float(T.get("grid_always", 4)) + 3) // 4)
    weight = {
        "kernel prologue / epilogue": n_waves,
        "step glue (ballots, loop)": W,
        "refill: loop head / write-back": W,
        "refill: reservation": R[reg["REFILL_RESERVE"]],
        "refill: deal + item decode": R[reg["REFILL_DECODE"]],
        "camera ray": R[reg["CAMERA_RAY"]],
        "park / unpark (LDS)": W,
        "walk: set-up / write-back": W,
        "walk: always-tested group": W * n_always_groups,
        "walk: per-ray constants": W,
        "walk: entry (slab test)": R[reg["WALK_ENTRY"]],
        "walk: entry (first cell)": R[reg["WALK_ENTER_CELL"]],
        "walk: cell step (first trip)": walk_first,
        "walk: cell step (loop)": walk_it - walk_first,
        "walk: cell step": walk_it,
        "walk: leaf round": leaf_it,
        "walk: loop control": leaf_it + W,  # every leaf round ends a trip; one more trip per step finds nothing left
        "walk: exact evaluation (always-tested)": exact_always,
        "walk: exact evaluation (cell entries)": exact_it - exact_always,
        "literal loop (rare)": lit_steps,
        "(cold: plain / and sqrtf fallbacks)": 0.0,
        "shade: entry / write-back": R[reg["SHADE_ANY"]],
        "shade: hit record + normal": R[reg["SHADE_HIT_RECORD"]],
        "shade: shared (1/|d|, seed hash)": R[reg["SHADE_ANY"]],
        "shade: sky": R[reg["SHADE_SKY"]],
        "shade: diffuse / metal (shared draw)": max(R[reg["SHADE_DIFFUSE"]], R[reg["SHADE_METAL"]]),
        "shade: diffuse": R[reg["SHADE_DIFFUSE"]],
        "shade: metal": R[reg["SHADE_METAL"]],
        "shade: diffuse / metal (outcome)": max(R[reg["SHADE_DIFFUSE"]], R[reg["SHADE_METAL"]]),
        "shade: glass": R[reg["SHADE_GLASS"]],
        "shade: emissive / other": 0.0,
        "shade: depth bookkeeping": R[reg["SHADE_CONTINUES"]],
        "shade: sample / item end": R[reg["SHADE_FINISHED"]],
    }
    print("\n## the twin's tallies (%s): %d waves, %.4g wave steps, %.4g segments" % (os.path.relpath(args.tallies, ROOT), n_waves, W, float(T["segments"])))
    print("   per wave step: cell steps %.2f, leaf rounds %.2f, exact evaluations %.2f; regions (share of steps in which any lane ran it, mean lanes when it ran):"
          % (walk_it / W, leaf_it / W, exact_it / W))
    print("   " + ", ".join("%s %.2f x %.1f" % (n.lower(), R[k] / W, RL[k] / max(R[k], 1)) for n, k in reg.items()))
    print("\n## DYNAMIC wave-instructions per wave step, by region (static count x region weight / wave steps)")
    print("%-36s %6s %6s | %5s %5s %5s %5s %5s | %5s %5s %5s %5s %5s | %5s %5s %5s %5s %5s %5s   weight" % (
        "region", "insts", "VALU", "fma", "mul", "add", "mnmx", "trans", "sel/mv", "cmp", "int", "cvt", "xlane", "LDS", "gmem", "SALU", "sload", "nop", "branch"))
    dyn_tot = collections.Counter()
    phase_tot = collections.OrderedDict()
    for r in regions:
        w = weight.get(r)
        if w is None:
            print("  (no weight for region %r: counted once per wave step)" % r)
            w = W
        scale = w / W
        sys.stdout.write("")
        row(r, static[r], scale, "%5.1f")
        print("   ^ x %.3f" % scale) if False else None
        for c, v in static[r].items():
            dyn_tot[c] += v * scale
            ph = r.split(":")[0]
            phase_tot.setdefault(ph, collections.Counter())[c] += v * scale
    row("TOTAL per wave step", dyn_tot, 1.0, "%5.1f")
    print("\n## by phase: share of the step's VALU wave-instructions, and what they are")
    vt = valu(dyn_tot)
    for ph, cnt in phase_tot.items():
        v = valu(cnt)
        if v < 0.05:
            continue
        arith = cnt["fp32 fma"] + cnt["fp32 mul"] + cnt["fp32 add/sub"] + cnt["fp32 min/max/med/other"] + cnt["transcendental"]
        print("  %-28s VALU %6.1f (%4.1f %%): fp32 arithmetic %4.1f %%, selects / moves %4.1f %%, compares %4.1f %%, integer / address %4.1f %%, "
              "convert %4.1f %%, cross-lane %4.1f %% | per VALU: SALU %.2f, s_nop %.3f, branches %.3f, exec regions %.3f"
              % (ph, v, 100 * v / vt, 100 * arith / max(v, 1e-9), 100 * cnt["select / move"] / max(v, 1e-9), 100 * cnt["compare"] / max(v, 1e-9),
                 100 * cnt["integer / bit / address"] / max(v, 1e-9), 100 * cnt["convert"] / max(v, 1e-9), 100 * cnt["cross-lane"] / max(v, 1e-9),
                 (cnt["SALU"] + cnt["exec-mask region"]) / max(v, 1e-9), cnt["s_nop"] / max(v, 1e-9), cnt["branch"] / max(v, 1e-9), cnt["exec-mask region"] / max(v, 1e-9)))
    arith = dyn_tot["fp32 fma"] + dyn_tot["fp32 mul"] + dyn_tot["fp32 add/sub"] + dyn_tot["fp32 min/max/med/other"] + dyn_tot["transcendental"]
    print("  %-28s VALU %6.1f: fp32 arithmetic %4.1f %% (fma %4.1f, mul %4.1f, add %4.1f, min/max %4.1f, trans %4.1f), selects / moves %4.1f %%, compares %4.1f %%, "
          "integer / address %4.1f %%, convert %4.1f %%, cross-lane %4.1f %%"
          % ("WHOLE STEP", vt, 100 * arith / vt, 100 * dyn_tot["fp32 fma"] / vt, 100 * dyn_tot["fp32 mul"] / vt, 100 * dyn_tot["fp32 add/sub"] / vt,
             100 * dyn_tot["fp32 min/max/med/other"] / vt, 100 * dyn_tot["transcendental"] / vt, 100 * dyn_tot["select / move"] / vt,
             100 * dyn_tot["compare"] / vt, 100 * dyn_tot["integer / bit / address"] / vt, 100 * dyn_tot["convert"] / vt, 100 * dyn_tot["cross-lane"] / vt))

    # ---- cross-check against the committed PMC record ---------------------------------------------------------------
    rec = None
    try:
        for r in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("records", []):
            if r.get("kernel") == args.kernel and str(r.get("config")) == str(args.pmc_config):
                rec = r
    except Exception:
        rec = None
    if rec:
        # the record is per launch of `passes_per_launch` passes; the tallies file says how many passes its launch had
        scale = W * float(rec.get("passes_per_launch", 64)) / float(T.get("passes", 64))
        print("\n## cross-check: model (per wave step x %.4g wave steps per %d-pass launch) against the PMC record %s" % (scale, rec.get("passes_per_launch", 64), rec.get("profile")))
        print("   (the model counts every block of a region that ran, also sub-branches the wave skipped: an upper bound, visibly so for the scalar and branch\n"
              "    instructions that wrap those sub-branches; SQ_INSTS_VALU_INT32 is the hardware's own class — add / mul / shift / logic on integers — and has no\n"
              "    model row: the tool's `int` column also holds v_mbcnt, v_bitop3, v_lshl_add_u64 ...)")
        pairs = [("SQ_INSTS_VALU", "valu_insts_per_launch", vt),
                 ("SQ_INSTS_VALU_FMA_F32", "sq_insts_valu_fma_f32", dyn_tot["fp32 fma"]),
                 ("SQ_INSTS_VALU_MUL_F32", "sq_insts_valu_mul_f32", dyn_tot["fp32 mul"]),
                 ("SQ_INSTS_VALU_ADD_F32", "sq_insts_valu_add_f32", dyn_tot["fp32 add/sub"]),
                 ("SQ_INSTS_VALU_TRANS_F32", "sq_insts_valu_trans_f32", dyn_tot["transcendental"]),
                 ("SQ_INSTS_VALU_INT32", "sq_insts_valu_int32", None),
                 ("SQ_INSTS_VALU_CVT", "sq_insts_valu_cvt", dyn_tot["convert"]),
                 ("SQ_INSTS_SALU", "sq_insts_salu", dyn_tot["SALU"] + dyn_tot["exec-mask region"]),
                 ("SQ_INSTS_SMEM", "sq_insts_smem", dyn_tot["scalar load"]),
                 ("SQ_INSTS_LDS", "sq_insts_lds", dyn_tot["LDS"]),
                 ("SQ_INSTS_BRANCH", "sq_insts_branch", dyn_tot["branch"])]
        for name, key, model in pairs:
            pm = rec.get(key)
            if pm is None:
                print("  %-26s PMC: not in the record%s" % (name, "" if model is None else "; model %.4g" % (model * scale)))
            elif model is None:
                print("  %-26s PMC %.4g per launch = %.1f per wave step (%.1f %% of the VALU stream)" % (name, pm, pm / scale, 100 * pm / rec["valu_insts_per_launch"]))
            else:
                print("  %-26s PMC %.4g per launch = %6.1f per wave step; model %6.1f  (model / PMC = %.3f)" % (name, pm, pm / scale, model, model * scale / pm))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
