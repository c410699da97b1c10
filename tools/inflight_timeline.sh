#!/bin/bash
# usage: ktrace2.sh <workload> : two-stream timeline of steady-state frames (frames in flight = 2), from the timed region's middle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o a --output-format csv -- python3 bench.py --workload $1 --no-cpu-baseline --no-second --steps 40 --warmup 5 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/kt/**/a_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(rows[0].keys())
idx = [i for i, r in enumerate(rows) if 'k_frame_constants' in r['Kernel_Name']]
# timed region: frames 15+10(serial warm) ... take frame constants number 40..43 (inside the 40 timed steps)
a, b = idx[35], idx[38]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', '?')
    print(f"{(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:7.1f}  q{q}  {r['Kernel_Name'].replace('brmi::','').replace('void ','').split('(')[0][:50]}")
PY
