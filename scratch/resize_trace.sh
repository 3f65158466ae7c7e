#!/bin/bash
# Per-launch durations of the pyramid kernels (one batch at a time), from a rocprofv3 kernel trace.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rt
rm -rf $O && mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --streams 1 --no-cpu-baseline --ransac-pairs 0 --steps 3 --warmup 1 > $O/bench.json 2> $O/log.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/rt/t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_resize_strip' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
g = [(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp'])) / 1e3 for i in range(len(rows) - 1)]
print('launches', len(d))
for k in range(7):
    v = d[k::7]
    print('level', k + 1, 'us', round(sum(v) / len(v), 1), 'grid', rows[k]['Grid_Size_X'] if 'Grid_Size_X' in rows[k] else '')
print('gaps inside a chain (us):', [round(x, 1) for x in g[:6]])
PY
rm -rf $O/t
