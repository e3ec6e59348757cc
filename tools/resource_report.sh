#!/bin/bash
# Compiler resource report of the column kernels (VGPRs, spills, scratch, occupancy), one line per instance.
# Usage: tools/resource_report.sh solve_lw|solve_sw [grep-pattern] [extra hipcc flags]
cd "$(dirname "$0")/../rrtmgp.jl_amd/csrc"
F=$1; PAT=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" \
  -Rpass-analysis=kernel-resource-usage -c $F.hip -o /tmp/rr_$F.o 2>&1 |
python3 -c '
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\]| \[waves/SIMD\])?: (\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"rrtmgp::|\(rrtmgp::\w+<\w+>\)|void ", "", name)
    print("%-58s VGPR %3s spill %3s  SGPR %3s spill %3s  scratch %4s B  waves/SIMD %s" % (
        name[:58], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))
' | grep -E "$PAT"
