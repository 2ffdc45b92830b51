#!/bin/bash
# Register / spill / occupancy table of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
#   scripts/diag/kernel_resources.sh foodrec_amd/csrc/m2d_catalogue_scan_bf16.hip [name filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wno-inline-asm -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur is not None: cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    if re.search(sys.argv[1], n):
        print("%-62s VGPR %3s AGPR %3s spill v%s s%s occ %s LDS %s" % (n[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy"), r.get("LDS Size")))
' "$filt"
