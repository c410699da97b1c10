#!/bin/bash
# tools/profile.sh <tag> [bench.py args...]
# Profiles `python3 bench.py <args>` on the GPU box and writes the judged summaries to profiles/<tag>_*:
#   <tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats (per-kernel calls / average duration), frames back to back (--frames-in-flight 1):
#                           a kernel's own duration, what bench.py's roofline block is measured on
#   <tag>_inflight_kernel_stats.csv  the same for the default command (two frames in flight): durations while two frames share the chip
#   <tag>_pmc.txt           per-kernel SQ counters + FETCH_SIZE / WRITE_SIZE (three separate --pmc passes, as MI355X_MICROARCH.md prescribes)
#   <tag>_traffic.json      HBM-side bytes per launch for bench.py's roofline.traffic
#   <tag>_bench.json        the bench line of an unprofiled run of the same command
# Run through gpurun:  gpurun -- 'bash tools/profile.sh r01_sponza4k --workload sponza; cp -r profiles gpurun_out/'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $ROOT/profiles
cd $ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py "$@" --frames-in-flight 1 --no-cpu-baseline --steps 30 --warmup 5 --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $OUT/sq -o sq --output-format csv -- python3 bench.py "$@" --frames-in-flight 1 --no-cpu-baseline --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 --steps 5 --warmup 2 > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py "$@" --frames-in-flight 1 --no-cpu-baseline --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 --steps 5 --warmup 2 > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- python3 bench.py "$@" --frames-in-flight 1 --no-cpu-baseline --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 --steps 5 --warmup 2 > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats2 -o stats --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --steps 60 --warmup 5 --no-second --no-dense --no-third --no-fourth --no-skinned --camera-path 0 > $OUT/stats2.log 2>&1
python3 bench.py "$@" --no-cpu-baseline --no-second --no-dense --no-third --no-fourth --no-skinned > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/stats2/stats_kernel_stats.csv $ROOT/profiles/${TAG}_inflight_kernel_stats.csv
cp $OUT/stats/stats_kernel_stats.csv $ROOT/profiles/${TAG}_kernel_stats.csv
cp $OUT/bench.json $ROOT/profiles/${TAG}_bench.json
python3 tools/pmc_report.py $OUT $ROOT/profiles/${TAG}_traffic.json > $ROOT/profiles/${TAG}_pmc.txt
mkdir -p $ROOT/gpurun_out/profiles && cp $ROOT/profiles/${TAG}_* $ROOT/gpurun_out/profiles/
cat $ROOT/profiles/${TAG}_pmc.txt | head -16
