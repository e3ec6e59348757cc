#!/bin/bash
# Runs on the GPU box: kernel + memory-copy timeline of the host-array path (bench.py --host), summarised per step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --host --steps 3 --warmup 1 --cpu-sample 0 --no-legs > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-300
python - <<'PY'
import csv, glob
k = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob('/tmp/tl/**/*memory_copy_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    if 'solve_kernel' in r['Kernel_Name']:
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K_' + ('lw' if 'lw_solve' in r['Kernel_Name'] else 'sw')))
rows = list(csv.DictReader(open(m)))
print(rows[0].keys())
for r in rows:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'][:12]))
ev.sort()
# last step only: the last 32 solve kernels
ks = [e for e in ev if e[2].startswith('K_')][-32:]
t0, t1 = ks[0][0], ks[-1][1]
win = [e for e in ev if e[0] >= t0 - 3_000_000 and e[1] <= t1 + 3_000_000]
busy = {}
for s, e, n in win:
    busy.setdefault(n, []).append((s, e))
print('window ms', (t1 - t0) / 1e6)
for n, iv in busy.items():
    print(n, 'count', len(iv), 'busy ms %.2f' % (sum(e - s for s, e in iv) / 1e6), 'avg us %.1f' % (sum(e - s for s, e in iv) / len(iv) / 1e3))
# gaps between consecutive solve kernels
gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
print('kernel gaps us:', ' '.join('%.0f' % g for g in gaps))
# everything that happens around three chunk boundaries of the LW solve (times in us relative to the first kernel)
allk = []
for r in csv.DictReader(open(k)):
    allk.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'k:' + r['Kernel_Name'][:28]))
for r in rows:
    allk.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'c:' + r['Direction'][12:] + ':s' + r['Stream_Id']))
allk.sort()
a, b = ks[3][0], ks[6][1]
for s_, e_, n in allk:
    if e_ >= a and s_ <= b:
        print('%9.1f %9.1f  %7.1f  %s' % ((s_ - a) / 1e3, (e_ - a) / 1e3, (e_ - s_) / 1e3, n))
print('kernel durations us:', ' '.join('%.0f' % ((e - s) / 1e3) for s, e, n in ks))
PY
