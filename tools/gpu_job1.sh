#!/bin/bash
# round-2 job 1: counter list, baseline bench lines, extended SQ counters on k_shade
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/job1; rm -rf $O; mkdir -p $O
cd $ROOT
rocprofv3 -L > $O/counters.txt 2>&1
python3 bench.py --no-cpu-baseline --steps 100 > $O/bench_sponza.json 2> $O/bench_sponza.err
python3 bench.py --no-cpu-baseline --steps 100 --workload bistro > $O/bench_bistro.json 2> $O/bench_bistro.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $O/sq2 -o sq2 --output-format csv -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace -d $O/sq3 -o sq3 --output-format csv -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/sq3.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM --kernel-trace -d $O/sq4 -o sq4 --output-format csv -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/sq4.log 2>&1
for p in sq2 sq3 sq4; do f=$O/$p/${p}_counter_collection.csv; [ -f $f ] && python3 - $f <<'PY' > $O/$p.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("brmi::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k[:48], len(disp[k]), {c: round(v / len(disp[k]), 1) for c, v in acc[k].items()})
PY
rm -rf $O/$p; done
tail -3 $O/*.log | head -60
cat $O/bench_sponza.json $O/bench_bistro.json
