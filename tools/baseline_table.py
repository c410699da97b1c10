#!/usr/bin/env python3
"""tools/baseline_table.py <bench line .json>  --  the GPU rows of BASELINE.md section 5 from one `python bench.py` line (value, frame time in flight and serial, algorithmic GB/s, fraction)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def cells(x):
    wf = (x.get("roofline") or {}).get("whole_frame", {})
    ser = x.get("serial_frame_ms")
    return (f"{x['value']:,.0f}", f"{x['ms_per_step']:.3f}" + (f" ({ser:.3f})" if ser else ""), f"{wf['achieved_GBps']:,.0f}" if wf else "-", f"{wf['frac']:.2f}" if wf else "-")


rows = [("#2 Sponza-class 4K (`configs1`)", d["configs1"]), ("#3 Bistro-class 4K, the headline", d), ("#3 skinned 30 % (`skinned`)", d["skinned"]), ("#3 dense (`dense`)", d["dense"]),
        ("#4 San-Miguel-class 4K (`configs3`)", d["configs3"]), ("#5 Zorah-class 8K (`configs4`)", d["configs4"])]
for name, x in rows:
    v, ms, gb, fr = cells(x)
    print(f"| {name} | {v} | {ms} | {gb} | {fr} | stages {x.get('stage_ms')} |")
print(f"| #3 path / path_fast | {d['path']['value']:,.0f} / {d['path_fast']['value']:,.0f} | {d['path']['ms_per_step']:.3f} / {d['path_fast']['ms_per_step']:.3f} |")
print(f"| #4 path | {d['configs3']['path']['value']:,.0f} | {d['configs3']['path']['ms_per_step']:.3f} |")
print(f"| CPU 16 threads / 1 thread / configs0 | {d['cpu_baseline']['value']} / {d['cpu_baseline_1thread']['value']} / {d['configs0']['value']} |")
r = d["roofline"]
print("headline roofline:", r["kernel"], "frac", r["frac"], "launch_ms", r["launch_ms"], "valu/px", r["valu"]["insts_per_px"], "valu frac", r["valu"]["frac"], "in flight", r.get("launch_ms_in_flight"))
