# Phase stamps of the flat traversal (k_cull_hierarchy): wave-cycles per visible instance and phase.  Needs an instrumented build:
#   tools/mkvar.sh stamps "-DBRMI_TILE_STAMPS -DBRMI_EXPERIMENTS" brmi_raster brmi_cull;  BRMI_LIB_PATH=$PWD/scratch/variants/stamps/libbrmi.so python3 tools/hier_stamps.py [workload]
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
frames = 4
for _ in range(frames): r.execute()
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
r.lib.brmi_debug_read_bin_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert r.lib.brmi_debug_read_bin_records(r._h, buf.ctypes.data, buf.nbytes | (1 << 63)) == 0
ph = buf[48:56].astype(np.float64)
names = ["instance + walk record arrive", "object + K1 (instances that pass)", "nodes, leaves, tests, depth chain, page map", "reached, replay append, bucket records staged", "flush: bucket slots reserved, records stored"]
c = r.counters()
print(wl, "flat traversal, wave-cycles over", frames, "frames, instances visible", c.instancesVisible, "; per visible instance and frame:")
for n, v in zip(names, ph): print("  %8.0f cycles  %s" % (v / frames / max(c.instancesVisible, 1), n))
