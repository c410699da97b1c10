#!/bin/bash
# tools/path_timeline.sh <tag> [bench args]: serial frames along the camera path under a kernel trace; per launch slot of the frame (same sequence every frame)
# the mean start offset, duration and the gap in front of it, over the last 100 frames.  Output: gpurun_out/path_timeline_<tag>.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_$TAG; cd $ROOT; mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace -d /tmp/pt_$TAG -o a --output-format csv -- python3 bench.py --no-cpu-baseline --no-second --no-third --no-dense --no-fourth --no-skinned --frames-in-flight 1 --steps 2 --warmup 1 --camera-path 120 "$@" > gpurun_out/path_timeline_$TAG.log 2>&1
python3 - /tmp/pt_$TAG > gpurun_out/path_timeline_$TAG.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/a_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'brmi::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_frame_constants' in r['Kernel_Name']]
frames = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])][-100:]
shape = collections.Counter(tuple(r['Kernel_Name'].split('(')[0] for r in fr) for fr in frames).most_common(1)[0][0]
frames = [fr for fr in frames if tuple(r['Kernel_Name'].split('(')[0] for r in fr) == shape]
print(f"{len(frames)} frames of the common launch sequence ({len(shape)} launches)")
tot_d = tot_g = 0.0
for k, name in enumerate(shape):
    st = sum(int(fr[k]['Start_Timestamp']) - int(fr[0]['Start_Timestamp']) for fr in frames) / len(frames) / 1e3
    du = sum(int(fr[k]['End_Timestamp']) - int(fr[k]['Start_Timestamp']) for fr in frames) / len(frames) / 1e3
    gp = sum(int(fr[k]['Start_Timestamp']) - int(fr[k - 1]['End_Timestamp']) for fr in frames) / len(frames) / 1e3 if k else 0.0
    tot_d += du; tot_g += gp
    print(f"{st:8.1f} us  gap {gp:6.1f}  dur {du:7.1f}  {name.replace('brmi::', '').replace('void ', '')[:60]}")
print(f"kernels {tot_d:.1f} us, gaps {tot_g:.1f} us")
PY
rm -rf /tmp/pt_$TAG
tail -40 gpurun_out/path_timeline_$TAG.txt
