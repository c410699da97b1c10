# timeline of k_raster_bins' workgroups (instrumented build, BRMI_TUNING=raster_debug=1024): python3 tools/bins_timeline.py [workload]
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from basicrenderer_amd import Scene
from basicrenderer_amd.renderer import VisibilityRenderer
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "bistro"
preset, kw, feat = bench.WORKLOADS[wl]
sc = Scene(preset, 3840, 2160, point_lights=256, material_features=feat, **kw)
r = VisibilityRenderer(sc, occlusion=True, stats=True)
for _ in range(4): r.update(); r.execute()
torch.cuda.synchronize()
bins = 15 * 135
hdr = np.zeros(3, dtype=np.uint32)
buf = np.zeros(64 + 4 * 16384, dtype=np.uint64)
r.lib.brmi_debug_read_bin_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes | (1 << 63)) == 0
w = buf[64:].reshape(-1, 4)
w = w[w[:, 0] > 0].astype(np.int64)
w = w[w[:, 0] > w[:, 0].max() - 100000]      # the last frame's launch (1 ms)
t0 = w[:, 0].min()
start = (w[:, 0] - t0) / 100.0; end = (w[:, 1] - t0) / 100.0; dur = end - start
recs = w[:, 2] & 0xFFFFFFFF; item = w[:, 2] >> 32; walk = w[:, 3]
print(f"{wl}: items {len(w)}, launch span {end.max():.1f} us, sum of item time {dur.sum():.0f} us, mean {dur.mean():.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; records {recs.sum()}")
for lo in range(0, int(end.max()) + 1, 10):
    print(f"  t={lo:4d}..{lo+10:4d} us: items alive {((start < lo + 10) & (end > lo)).sum():5d}, started {((start >= lo) & (start < lo + 10)).sum():5d}, records of those started {recs[(start >= lo) & (start < lo + 10)].sum():7d}")
for i in np.argsort(-dur)[:8]:
    print(f"  bin {item[i] & 0xFFFF:5d} slice {(item[i] >> 16) & 0xFF} of {(item[i] >> 24) + 1}: {dur[i]:6.1f} us from {start[i]:6.1f}, records {recs[i]}, wave 0 walking {walk[i]} cycles")
print("  us per record (fit):", np.polyfit(recs, dur, 1))
