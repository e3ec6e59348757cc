#!/bin/bash
# Runs on the GPU box: which copies of copy_engine_probe went through SDMA (memory-copy trace) and which through a shader
# copy kernel, optionally under an environment setting:  copy_engine_probe.sh [VAR=value ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/cep
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cep -o cep -- $R/tools/ubench/copy_engine_probe > /tmp/cep.log 2>&1
python - "$@" <<'PY'
import csv, glob, re, sys
print("== environment:", " ".join(sys.argv[1:]) or "(default)")
lab = {}
for ln in open('/tmp/cep.log'):
    m = re.match(r"(.*?)\s+(\d+) bytes$", ln)
    if m: lab[int(m.group(2))] = m.group(1).strip()
sd = {}
ms = glob.glob('/tmp/cep/**/*memory_copy_trace.csv', recursive=True)
if ms:
    rows = list(csv.DictReader(open(ms[0])))
    key = [k for k in rows[0].keys() if 'ytes' in k or 'ize' in k] if rows else []
    for r in rows:
        for k in key:
            try: sd[int(r[k])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            except Exception: pass
ks = glob.glob('/tmp/cep/**/*kernel_trace.csv', recursive=True)
nblit = sum(1 for r in csv.DictReader(open(ks[0])) if 'copyBuffer' in r['Kernel_Name']) if ks else -1
for sz, what in sorted(lab.items()):
    print("  %-86s %s" % (what, ("SDMA %.0f us (%.1f GB/s)" % (sd[sz], sz / sd[sz] / 1e3)) if sz in sd else "shader copy"))
print("  copyBuffer kernels in the trace:", nblit)
PY
