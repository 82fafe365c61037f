#!/bin/bash
# Compact per-kernel resource table (VGPRs, SGPRs, spills, scratch, waves/SIMD) + static instruction
# counts of the kernels, from a cross-compile of the three device translation units (pt_kernels.hip: what every
# context uses; pt_kernels_small.hip: the small-list kernels; pt_kernels_extra.hip: Russian-roulette builds, measuring
# twins) for gfx950 (no GPU needed).
# Usage: tools/kernel_resources.sh [outdir]   (default /tmp/ptres)
set -e
OUT=${1:-/tmp/ptres}
mkdir -p "$OUT"
cd "$(dirname "$(readlink -f "$0")")/../ray_tracer_webgl_amd/csrc"
: > "$OUT/remarks.txt"
for TU in pt_kernels pt_kernels_small pt_kernels_extra; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize \
    -fvisibility=hidden -Wall -Wno-unused-function $PT_EXTRA_FLAGS -Rpass-analysis=kernel-resource-usage \
    -save-temps=obj -c $TU.hip -o "$OUT/$TU.o" 2>> "$OUT/remarks.txt" || { cat "$OUT/remarks.txt"; exit 1; }
done
grep -v "remark:" "$OUT/remarks.txt" | grep -E "warning|error" || true
python3 - "$OUT" <<'PY'
import re, sys, glob
out = sys.argv[1]
txt = open(out + "/remarks.txt").read()
rows = []
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split()[0]
    g = lambda k: re.search(k + r": (\d+)", blk)
    rows.append((name, g("VGPRs").group(1), g("TotalSGPRs").group(1), g("VGPRs Spill").group(1),
                 g("SGPRs Spill").group(1), g(r"ScratchSize \[bytes/lane\]").group(1), g(r"Occupancy \[waves/SIMD\]").group(1)))
asm = glob.glob(out + "/*gfx950*.s")
counts = {}
for path in asm:
    cur = None
    for line in open(path):
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and m.group(1).startswith("pt_"):
            cur = m.group(1); counts[cur] = [0, 0, 0, 0]; continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur and re.match(r"^\t[a-z]", line):
            op = line.split()[0]
            c = counts[cur]
            c[0] += 1
            if op.startswith("v_"): c[1] += 1
            elif op.startswith("s_"): c[2] += 1
            if op.startswith("s_cbranch") or op == "s_branch": c[3] += 1
print("%-36s %5s %5s %6s %6s %7s %5s | %6s %6s %6s %6s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "waves", "insts", "valu", "salu", "branch"))
for r in rows:
    c = counts.get(r[0], ["-"] * 4)
    print("%-36s %5s %5s %6s %6s %7s %5s | %6s %6s %6s %6s" % (r + tuple(c)))
PY
