#!/bin/bash
# Runs on the GPU box: kernel + memory-copy timeline of the host-array Layer-2 step (bench.py --l2 fused), last step:
# busy time per kind, kernel durations and the gaps between them, copy-queue occupancy per direction.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py ${TL_MODE:---l2 fused} --leg x --steps 3 --warmup 1 > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-300
python - <<'PY'
import csv, glob
k = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/tl/**/*memory_copy_trace.csv', recursive=True)[0]
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in csv.DictReader(open(k))]
C = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction']) for r in csv.DictReader(open(m))]
K.sort(); C.sort()
sw = [e for e in K if 'sw_solve' in e[2]]
nper = len(sw) // 4                       # 1 warm-up + 3 steps
last = sw[-nper:]
lw = [e for e in K if 'lw_solve' in e[2]][-nper:]
t0, t1 = lw[0][0], last[-1][1]
print('chunks per step', nper, ' first LW start -> last SW end: %.2f ms' % ((t1 - t0) / 1e6))
inwin = lambda ev: [e for e in ev if e[1] >= t0 - 5_000_000 and e[0] <= t1 + 5_000_000]
kw = inwin(K)
names = {}
for s, e, n in kw:
    names.setdefault(n, []).append(e - s)
for n, d in names.items():
    print('  kernel %-42s n %3d  busy %.2f ms  avg %.0f us' % (n, len(d), sum(d) / 1e6, sum(d) / len(d) / 1e3))
solve = sorted([e for e in kw if 'solve_kernel' in e[2]])
gaps = [(solve[i + 1][0] - solve[i][1]) / 1e3 for i in range(len(solve) - 1)]
print('  gaps between consecutive solve kernels (us):', ' '.join('%.0f' % g for g in gaps), ' sum %.2f ms' % (sum(gaps) / 1e3))
print('  LW durations (us):', ' '.join('%.0f' % ((e - s) / 1e3) for s, e, n in lw))
print('  SW durations (us):', ' '.join('%.0f' % ((e - s) / 1e3) for s, e, n in last))
cw = inwin(C)
for d in sorted(set(c[2] for c in cw)):
    iv = [c for c in cw if c[2] == d]
    print('  copies %-28s n %4d  busy %.2f ms  first start %+.2f ms  last end %+.2f ms (relative to first LW start / last SW end)' % (
        d, len(iv), sum(e - s for s, e, _ in iv) / 1e6, (iv[0][0] - t0) / 1e6, (iv[-1][1] - t1) / 1e6))
PY
