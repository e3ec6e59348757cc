#!/bin/bash
# Runs on the GPU box (round 5): GPU-side timeline of a short fused step on device arrays (512 columns x 72, aerosols; LW and SW on
# the workspace's two lanes): kernel durations and the gaps between them, from rocprofv3 --kernel-trace.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NCOL=${1:-512}
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --ncol $NCOL --nlay 72 --aerosols --fused-step --leg x --steps 20 --warmup 5 > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-200
python - <<'PY'
import csv, glob
k = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
K = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:28]) for r in csv.DictReader(open(k)))
sw = [i for i, e in enumerate(K) if 'sw_solve' in e[2]]
# three steps in the middle of the timed loop
i0 = sw[len(sw) // 2 - 2]
t0 = K[i0][1]
print('events after the end of an SW kernel, times in us relative to it:')
n = 0
for s, e, name in K[i0 + 1:]:
    print('  %-28s start %8.1f  end %8.1f  dur %7.1f' % (name, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    n += 1
    if n > 16: break
ends = [K[i][1] for i in sw]
per = [(ends[i + 1] - ends[i]) / 1e3 for i in range(len(ends) // 2, len(ends) - 1)]
print('SW end -> next SW end (us):', ' '.join('%.0f' % p for p in per[:10]))
PY
