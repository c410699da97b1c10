#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 passes made by tools/profile.sh (also writes the traffic JSON bench.py reads).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 128 B requests as 64 B (MI355X_MICROARCH.md, HBM section),
so the HBM-side read bytes are 2 x FETCH_SIZE; that factor was checked here on k_shade, whose line-level read volume is known
(44 B per shaded pixel -> 348 MiB at 4K against 2 x 175 MiB measured).
"""
import collections
import csv
import json
import sys

d, traffic_out = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, seen = collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    return acc, cnt


sq, n = load(f"{d}/sq/sq_counter_collection.csv")
fe, nf = load(f"{d}/fetch/fetch_counter_collection.csv")
wr, nw = load(f"{d}/write/write_counter_collection.csv")
st = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(f"{d}/stats/stats_kernel_stats.csv"))}
# frames of the stats run = launches of the kernel every frame starts with; a kernel's launches per frame follow from its own call count
frames = max(1, int(next((r["Calls"] for k, r in st.items() if "k_frame_constants" in k), "1")))
print(f"{'kernel':44s} {'avg_us':>8s} {'calls':>6s} {'waves':>7s} {'VALU/wave':>9s} {'valu%':>6s} {'wait%':>6s} {'stall%':>6s} {'fetchMiB(raw)':>13s} {'writeMiB':>9s}")
traffic = {}
for k in sorted(sq, key=lambda k: -float(st.get(k, {"TotalDurationNs": 0})["TotalDurationNs"])):
    s, c = sq[k], n[k]
    wc = max(s["SQ_WAVE_CYCLES"], 1)
    avg = float(st[k]["AverageNs"]) / 1e3 if k in st else 0.0
    fetch_kib = fe[k]["FETCH_SIZE"] / max(nf[k], 1)
    write_kib = wr[k]["WRITE_SIZE"] / max(nw[k], 1)
    name = k.replace("void ", "").replace("brmi::", "")
    print(f"{name[:44]:44s} {avg:8.1f} {st.get(k, {}).get('Calls', '?'):>6s} {s['SQ_WAVES'] / c:7.0f} {s['SQ_INSTS_VALU'] / max(s['SQ_WAVES'], 1):9.0f} "
          f"{100 * s['SQ_ACTIVE_INST_VALU'] / wc:6.1f} {100 * s['SQ_WAIT_ANY'] / wc:6.1f} {100 * s['SQ_WAIT_INST_ANY'] / wc:6.1f} {fetch_kib / 1024:13.2f} {write_kib / 1024:9.2f}")
    traffic[name] = {"avg_us": round(avg, 2), "fetch_size_kib_raw": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1),
                     "hbm_bytes_per_launch": int(2 * fetch_kib * 1024 + write_kib * 1024),
                     "valu_wave_insts_per_launch": int(s["SQ_INSTS_VALU"] / max(c, 1)), "waves_per_launch": int(s["SQ_WAVES"] / max(c, 1)),
                     # launches per frame in the --kernel-trace --stats run (Calls / frames): per-frame figures are per-launch averages x this
                     "calls": int(st[k]["Calls"]) if k in st else None, "calls_per_frame": round(int(st[k]["Calls"]) / frames, 3) if k in st else None}
if traffic_out:
    json.dump(traffic, open(traffic_out, "w"), indent=1, sort_keys=True)
