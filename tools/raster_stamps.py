# Phase shares of k_raster's wave-cycles (instrumented build -DBRMI_TILE_STAMPS -DBRMI_EXPERIMENTS, BRMI_TUNING=raster_debug=256):
#   BRMI_TUNING=raster_debug=256 BRMI_LIB_PATH=$PWD/scratch/variants/stamps/libbrmi.so python3 tools/raster_stamps.py <workload>
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from basicrenderer_amd import Scene, capi
from basicrenderer_amd.renderer import VisibilityRenderer
import bench
wl = sys.argv[1]
preset, kw, feat = bench.WORKLOADS[wl]
W, H = bench.FRAME_SIZE.get(wl, (3840, 2160))
sc = Scene(preset, W, H, point_lights=bench.LIGHTS[wl], material_features=feat, **kw)
r = VisibilityRenderer(sc, occlusion=True, stats=True)
frames = 4
for _ in range(frames):
    r.execute()
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
r.lib.brmi_debug_read_bin_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes | (1 << 63)) == 0
ph = buf[16:24].astype(np.float64)
names = ["cluster fetch (setup -> view, object constants)", "vertex stage (positions -> LDS)", "triangle setup (indices, edge functions, vote)", "small boxes: rows re-dealt, global atomic-min", "bin records, a few bins per triangle", "bin records, whole-wave emission", "-", "loop overhead / idle"]
tot = ph.sum()
c = r.counters()
print(wl, "k_raster phase shares over", frames, "frames (both raster phases), visible clusters", c.visibleClusters, "; total wave-cycles %.3f G" % (tot / 1e9))
for n, v in zip(names, ph):
    if n != "-": print("  %5.1f %%  %s" % (100 * v / tot, n))
