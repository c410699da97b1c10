timeout 800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for wl in sponza bistro; do echo -n "$wl: "; timeout 300 python bench.py --steps 20 --warmup 3 --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; done
