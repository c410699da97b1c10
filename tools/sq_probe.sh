#!/bin/bash
# tools/sq_probe.sh <tag> [bench.py args]: three SQ counter passes over `python3 bench.py <args>`; per-kernel table in gpurun_out/<tag>/sq.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
python3 bench.py --no-cpu-baseline --no-second --no-dense --camera-path 0 --steps 100 "$@" > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $O/p1 -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-second --no-dense --camera-path 0 --steps 5 --warmup 2 "$@" > $O/p1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --kernel-trace -d $O/p2 -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-second --no-dense --camera-path 0 --steps 5 --warmup 2 "$@" > $O/p2.log 2>&1
python3 - $O <<'PY' > $O/sq.txt
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
for p in ("p1", "p2"):
    for f in glob.glob(f"{sys.argv[1]}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("brmi::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    g = lambda n, c=c, k=k: c.get(n, 0.0) / max(1, len(disp[k].get(n, ())))
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    rows.append((g("SQ_BUSY_CYCLES"), k, g))
print(f"{'kernel':40s} {'waves':>7s} {'VALU/w':>7s} {'SALU/w':>7s} {'SMEM/w':>7s} {'VMEM/w':>7s} {'LDS/w':>6s} {'trans/w':>7s} {'valu%':>6s} {'sca%':>5s} {'wait%':>6s} {'stall%':>6s} {'busyMcyc':>8s}")
for _, k, g in sorted(rows, key=lambda r: -r[0])[:14]:
    w = max(g("SQ_WAVES"), 1); wc = max(g("SQ_WAVE_CYCLES"), 1)
    print(f"{k[:40]:40s} {w:7.0f} {g('SQ_INSTS_VALU')/w:7.0f} {g('SQ_INSTS_SALU')/w:7.0f} {g('SQ_INSTS_SMEM')/w:7.0f} {g('SQ_INSTS_VMEM_RD')/w:7.0f} {g('SQ_INSTS_LDS')/w:6.0f} {g('SQ_INSTS_VALU_TRANS')/w:7.0f} "
          f"{100*g('SQ_ACTIVE_INST_VALU')/wc:6.1f} {100*g('SQ_ACTIVE_INST_SCA')/wc:5.1f} {100*g('SQ_WAIT_ANY')/wc:6.1f} {100*g('SQ_WAIT_INST_ANY')/wc:6.1f} {g('SQ_BUSY_CYCLES')/1e6:8.2f}")
PY
rm -rf $O/p1 $O/p2
cat $O/sq.txt; python3 -c "
import json; d = json.load(open('$O/bench.json')); print(d['ms_per_step'], d['stage_ms'])"
