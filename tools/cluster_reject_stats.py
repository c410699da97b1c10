import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
import orc, bench
from basicrenderer_amd import Scene
wl = sys.argv[1] if len(sys.argv) > 1 else "bistro"
margin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
preset, kw, feat = bench.WORKLOADS[wl]
W, H = bench.FRAME_SIZE.get(wl, (3840, 2160))
t0 = time.time()
sc = Scene(preset, W, H, point_lights=4, material_features=feat, **kw)
print("scene", time.time() - t0)
o = orc.OracleFrame(sc)
hz = None
for k in range(2):
    t0 = time.time(); hz = o.run_occlusion(hz); print("frame", k, time.time() - t0, "clusters", o.count)
n = o.count
flags = np.zeros(n, dtype=np.uint8)
t0 = time.time()
orc.lib().orc_cluster_facing(C.byref(o.sb), orc.P(o.clusters), orc.u32(n), orc.u32(W), orc.u32(H), C.c_float(margin), orc.P(flags), o.threads)
print("facing", time.time() - t0)
vis = o.vis
cov = vis != np.uint64(0xFFFFFFFFFFFFFFFF)
cid = ((vis[cov] >> np.uint64(7)) & np.uint64(0x3FFFFFF)).astype(np.int64)
own = np.zeros(n, dtype=bool); own[np.unique(cid)] = True
none, back, cone = (flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0
print(wl, "visible", n, "own a pixel", own.sum(), f"{100*own.sum()/n:.1f}%")
print("  no active triangle (fp32 setup):", none.sum(), f"{100*none.sum()/n:.1f}%")
print("  all back-facing exactly:", back.sum(), f"{100*back.sum()/n:.1f}%")
print("  cone reject (margin", margin, "):", cone.sum(), f"{100*cone.sum()/n:.1f}%")
print("  cone-rejected but some triangle active:", (cone & ~none).sum(), " cone-rejected and owns a pixel:", (cone & own).sum())
print("  exact-back but some triangle active:", (back & ~none).sum())
hz0, offs, nm = hz
for mt in (2, 4, 8, 16):
    f2 = np.zeros(n, dtype=np.uint8)
    orc.lib().orc_cluster_occlusion_stats(C.byref(o.sb), orc.P(o.clusters), orc.u32(n), orc.u32(W), orc.u32(H), orc.P(hz0), orc.P(offs), orc.u32(nm), C.c_int(mt), orc.P(f2), o.threads)
    ex, ab, sp = (f2 & 1) != 0, (f2 & 2) != 0, (f2 & 4) != 0
    print(f"  fine occlusion test, <= {mt}x{mt} texels: exact box {100*ex.sum()/n:.1f}%  object AABB {100*ab.sum()/n:.1f}%  sphere {100*sp.sum()/n:.1f}%   (rejected but owns a pixel: {(ex & own).sum()} {(ab & own).sum()} {(sp & own).sum()})")
