#!/bin/bash
# tools/mem_probe.sh <tag> [bench.py args]: vector-memory path counters (TA / TD / TCP / TCC) of `python3 bench.py <args>`, per kernel, in gpurun_out/<tag>/mem.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
rocprofv3 -L > $O/avail.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/m$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-second --no-dense --camera-path 0 --steps 5 --warmup 2 "$@" > $O/m$i.log 2>&1
done
python3 - $O <<'PY' > $O/mem.txt
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set)); dur = collections.defaultdict(list)
for f in glob.glob(f"{sys.argv[1]}/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("brmi::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = sorted({n for c in acc.values() for n in c})
keep = [k for k in acc if any(s in k for s in ("k_shade<0>", "k_gbuffer<false", "k_raster_bins<false>", "k_raster<false>", "k_resolve_setup", "k_cull_hierarchy<false"))]
print("counter (mean per dispatch)".ljust(40) + "".join(k[:22].rjust(24) for k in keep))
for n in names:
    print(n.ljust(40) + "".join(f"{acc[k].get(n, 0.0) / max(1, len(disp[k].get(n, ()))):24.0f}" for k in keep))
PY
rm -rf $O/m[0-9]*/
cat $O/mem.txt
