#!/bin/bash
# tools/pmc_pass.sh <tag> "<counters>" [bench.py args]: ONE rocprofv3 counter pass over a short bench run; per-kernel means in gpurun_out/<tag>/pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SET=$2; shift 2
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $O/m -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 --steps 5 --warmup 2 "$@" > $O/m.log 2>&1
python3 - $O <<'PY' | tee -a $O/pmc.txt
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(f"{sys.argv[1]}/m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("brmi::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
names = sorted({n for c in acc.values() for n in c})
print("kernel".ljust(44) + "".join(n[-22:].rjust(24) for n in names))
for k in sorted(acc, key=lambda k: -max(acc[k].values()))[:12]:
    print(k[:43].ljust(44) + "".join(f"{acc[k].get(n, 0.0) / max(1, len(disp[k])):24.0f}" for n in names))
PY
rm -rf $O/m
