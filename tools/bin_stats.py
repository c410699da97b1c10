# statistics of the triangle-bin records of a steady frame: python3 tools/bin_stats.py [workload kwargs as python dict]
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from basicrenderer_amd import Scene
from basicrenderer_amd.renderer import VisibilityRenderer
wl = sys.argv[1] if len(sys.argv) > 1 else "bistro"
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else dict(unique_budget=True, lod_builder="own", relief_slope=1.5)
sc = Scene(wl, 3840, 2160, point_lights=256, directional=True, **kw)
r = VisibilityRenderer(sc, occlusion=True, stats=True)
for _ in range(3): r.update(); r.execute()
cap, bins = 8192, 15 * 135
r.lib.brmi_debug_read_bin_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
buf = np.zeros((bins, cap, 16), dtype=np.uint32)
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes | (1 << 62)) == 0
r.update(); r.execute()
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes) == 0
rows = buf[:, :, 1] >> 16
valid = rows > 0
cnt = valid.sum(axis=1)
print(f"{wl} {kw}: visible clusters {r.counters().visibleClusters}, records {valid.sum()}, bins with records {(cnt>0).sum()} of {bins}")
print("  records per bin: median %d p90 %d p99 %d max %d" % (np.median(cnt[cnt>0]), np.percentile(cnt[cnt>0], 90), np.percentile(cnt[cnt>0], 99), cnt.max()))
rv = rows[valid]; w = buf[:, :, 3][valid].astype(np.int64)
print("  rows per record: " + " ".join(f"{k}:{(rv==k).mean()*100:.1f}%" for k in range(1, 17)))
print("  mean rows %.2f; rect width: median %d p90 %d p99 %d max %d; mean %.1f" % (rv.mean(), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max(), w.mean()))
# clipped to the bin: the record's box against its 256 px strip
minx = buf[:, :, 2][valid].astype(np.int64).view(np.int64)
minx = buf[:, :, 2][valid].view(np.int32).astype(np.int64)
bx = np.repeat(np.arange(bins) % 15, cap).reshape(bins, cap)[valid]
lo = np.maximum(minx, bx * 256); hi = np.minimum(minx + w, bx * 256 + 256)
inw = np.maximum(hi - lo, 0)
print("  width inside the bin: mean %.1f median %d p90 %d; box pixels inside per record mean %.1f (sum %.1f M); steps from row start incl. outside-left part mean %.1f" % (inw.mean(), np.median(inw), np.percentile(inw, 90), (inw * rv).mean(), (inw * rv).sum() / 1e6, ((hi - minx).clip(0) ).mean()))
print("  lane utilisation at 16 lanes per record: %.2f; sum over records of max-row-length x 1 (serial steps per 16-lane group) %.1f M" % (rv.mean() / 16, (hi - minx).clip(0).sum() / 1e6))
# ---- model of the walk: a workgroup step takes 32 records (wave w: records 4w .. 4w+3 of the step, 16 lanes each, lane = row); a wave's step
# lasts as long as its longest row (pixels stepped from the box's left edge to the right end inside the bin)
Z = 8
tot_model = 0.0; tot_ideal = 0.0; worst = []
steps_w = (hi - minx).clip(0)          # per valid record: pixels a row's lane steps + walks
# rebuild per bin arrays
idx = np.nonzero(valid)
order = np.lexsort((idx[1], idx[0]))
b_of = idx[0][order]; slot = idx[1][order]; sw = steps_w[order]; rw = rv[order]; iw = inw[order]
starts = np.searchsorted(b_of, np.arange(bins)); ends = np.searchsorted(b_of, np.arange(bins), side="right")
for b in range(bins):
    n = ends[b] - starts[b]
    if n == 0: continue
    s_ = sw[starts[b]:ends[b]]; r_ = rw[starts[b]:ends[b]]; i_ = iw[starts[b]:ends[b]]
    sl = max(1024, ((n + Z - 1) // Z + 31) & ~31)
    for f in range(0, n, sl):
        ss = s_[f:f + sl]; m = len(ss)
        pad = (-m) % 32
        a4 = np.pad(ss, (0, pad)).reshape(-1, 8, 4).max(axis=2)        # steps x waves: the wave's longest row
        t_wave = a4.sum(axis=0).max()                                  # the slowest wave of the workgroup
        ideal = (np.pad(ss * r_[f:f + sl], (0, pad))).sum() / 512.0     # every lane busy
        worst.append((t_wave, ideal, m, b))
        tot_model += a4.sum(); tot_ideal += ideal * 8
worst.sort(reverse=True)
print("  model: pixel-steps of the slowest wave, 10 longest workgroups: " + ", ".join(f"{int(t)} (ideal {int(i)}, {m} rec, bin {b})" for t, i, m, b in worst[:10]))
print("  model: sum over waves of serial pixel-steps %.2f M; with every lane busy %.2f M (x%.1f)" % (tot_model / 1e6, tot_ideal / 1e6, tot_model / max(tot_ideal, 1)))
