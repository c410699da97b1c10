# Phase shares of k_raster_bins' and k_raster's wave-cycles (instrumented build -DBRMI_TILE_STAMPS -DBRMI_EXPERIMENTS; BRMI_TUNING=raster_debug=512 for the bins'
# phases, 256 for k_raster's):  BRMI_TUNING=raster_debug=768 BRMI_LIB_PATH=$PWD/scratch/variants/stamps/libbrmi.so python3 tools/bins_stamps.py <workload> [camera position on the preset's path]
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from basicrenderer_amd import Scene, capi
from basicrenderer_amd.renderer import VisibilityRenderer
import bench
wl = sys.argv[1]
preset, kw, feat = bench.WORKLOADS[wl]
sc = Scene(preset, 3840, 2160, point_lights=256, material_features=feat, **kw)
r = VisibilityRenderer(sc, occlusion=True, stats=True)
if len(sys.argv) > 2:      # the preset's camera at a position of its path
    cam_at = sc.camera_at(float(sys.argv[2]))
    r.set_camera_device(torch.from_numpy(cam_at[0]).to('cuda'), torch.from_numpy(cam_at[1]).to('cuda'), cam_at[0])
    torch.cuda.synchronize()
frames = 4
for _ in range(frames):
    r.execute()
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
r.lib.brmi_debug_read_bin_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes | (1 << 63)) == 0
ph = buf[32:40].astype(np.float64)
names = ["tile clear + barrier", "record walk: opaque rows (+ listing alpha records)", "alpha task list (scan of segment counts)", "alpha tasks: the walk (coverage, key, look at the tile, append)", "barrier before the merge (waiting for the slowest wave)", "merge into the visibility buffer", "alpha tasks: sampling 64 waiting pixels (texel fetch, filter, LDS min)", "alpha tasks: finding the record, loading it, segment setup"]
tot = ph.sum()
print(wl, "k_raster_bins phase shares (wave-cycles of every wave) over", frames, "frames; total %.3f G" % (tot / 1e9))
for n, v in zip(names, ph):
    print("  %5.1f %%  %s" % (100 * v / tot, n))
ph2 = buf[16:24].astype(np.float64); names2 = ["cluster fetch", "vertex stage", "triangle setup", "small boxes: global atomic-min (+ alpha test)", "bin records, few bins", "bin records, whole wave", "bin records, few bins, window too small (slot by slot)", "loop overhead"]
print(wl, "k_raster phase shares; total %.3f G" % (ph2.sum() / 1e9))
for n, v in zip(names2, ph2):
    if n != "-": print("  %5.1f %%  %s" % (100 * v / ph2.sum(), n))

lw = buf[40:48].astype(np.float64)
if lw.sum() > 0:
    print(wl, "k_raster: the longest wave of the four frames' launches, %.0f k cycles (~%.0f us at 2.3 GHz)" % (lw.sum() / 1e3, lw.sum() / 2300.0))
    for n, v in zip(names2, lw):
        print("  %5.1f %%  %s" % (100 * v / lw.sum(), n))
