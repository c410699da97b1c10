cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in sponza bistro; do for m in 0 1 2; do
echo -n "$wl mode $m: "
BRMI_RASTER_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4_${wl}_$m -- python3 bench.py --steps 10 --warmup 2 --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms']['raster'], end=' ')"
f=$(find gpurun_out/prof4_${wl}_$m -name "*kernel_stats.csv" | head -1); grep raster $f | cut -d, -f1,4 | tr '\n' ' '; echo
done; done
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
