#!/usr/bin/env python3
"""Full-size parity sweeps against the CPU oracle (run on the GPU box: `gpurun -- python tests/stress_fullsize.py [frames|occlusion|bands]`).

The GPU suite (tests/test_parity_gpu.py::test_full_size_*) holds a fixed subset of these; the sweeps go wider: camera paths, seeds,
every material feature, 8K, the multi-GPU band partition at bench size.  They found the slice-boundary pixel and
the reference's vertical-extent-only HZB mip choice (section 4.2).  TEST INFRASTRUCTURE: imports the oracle.
"""
import sys
MODE = sys.argv[1] if len(sys.argv) > 1 else "frames"
import os, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # lives under tests/: the only place besides smoke() and the bench baseline that may use the oracle

def frames():
    import numpy as np, orc
    from conftest import Scene
    from basicrenderer_amd.renderer import VisibilityRenderer
    EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
    def check(tag, sc):
        t0 = time.time()
        r = VisibilityRenderer(sc, stats=True); r.execute()
        o = orc.OracleFrame(sc).run()
        bad = []
        if not np.array_equal(r.visible_clusters(), o.clusters[:o.count]): bad.append("clusters")
        vis = r.visibility()
        if not np.array_equal(vis, o.vis): bad.append(f"vis({int((vis != o.vis).sum())})")
        cov = o.vis != EMPTY
        if not np.array_equal(r.depth().view(np.uint32), o.depth.view(np.uint32)): bad.append("depth")
        g = r.gbuffer()
        if not np.array_equal(g["normals"].view(np.uint32)[cov], o.normals.view(np.uint32)[cov]): bad.append("normals")
        for k, ref in (("albedo", o.albedo), ("mr", o.mr), ("motion", o.motion), ("coat", o.coat), ("emissive", o.emissive), ("fuzz", o.fuzz)):
            if not np.array_equal(g[k][cov], ref[cov]): bad.append(k)
        d = np.abs(r.hdr().view(np.uint16).astype(np.int32) - o.hdr.view(np.uint16).astype(np.int32))
        if d.max() > 1: bad.append(f"hdr(max {d.max()}, n {(d > 1).sum()})")
        c = r.counters()
        if c.droppedRecords or c.droppedClusters: bad.append("dropped")
        r.close()
        print(tag, "OK" if not bad else "MISMATCH " + " ".join(bad), f"clusters {o.count} covered {cov.mean():.2f} {time.time()-t0:.1f}s", flush=True)
        return not bad
    ok = True
    cases = []
    for step in range(0, 6):
        cases.append((f"sponza4k step{step}", dict(preset="sponza", width=3840, height=2160, point_lights=64, camera_step=step)))
        cases.append((f"bistro4k step{step}", dict(preset="bistro", width=3840, height=2160, point_lights=256, camera_step=step)))
    for seed in range(1, 5):
        cases.append((f"sponza4k seed{seed} mf63", dict(preset="sponza", width=3840, height=2160, point_lights=64, seed=seed, material_features=63, spot_every=3)))
        cases.append((f"bistro4k seed{seed} mf59 skinned", dict(preset="bistro", width=3840, height=2160, point_lights=256, seed=seed, material_features=59, skinned_fraction=0.3)))
        cases.append((f"sanmiguel4k seed{seed} mf24", dict(preset="san_miguel", width=3840, height=2160, point_lights=256, seed=seed, material_features=24)))
    for seed in range(1, 4):
        cases.append((f"bistro4k seed{seed} mf255 parallax", dict(preset="bistro", width=3840, height=2160, point_lights=256, seed=seed, material_features=255)))
        cases.append((f"sanmiguel4k seed{seed} mf152 parallax alpha", dict(preset="san_miguel", width=3840, height=2160, point_lights=128, seed=seed, material_features=128 | 24)))
    cases.append(("sponza8k", dict(preset="sponza", width=7680, height=4320, point_lights=64)))
    cases.append(("bistro4k clod", dict(preset="bistro", width=3840, height=2160, point_lights=256, lod_builder="clusterlod", material_features=24)))
    cases.append(("zorah4k", dict(preset="zorah", width=3840, height=2160, point_lights=64, size_scale=0.02)))
    for tag, kw in cases:
        try:
            ok = check(tag, Scene(**kw)) and ok
        except Exception as e:
            print(tag, "ERROR", repr(e), flush=True); ok = False
    print("ALL OK" if ok else "SOME FAILED")
    

def occlusion():
    import numpy as np, orc
    from conftest import Scene
    from basicrenderer_amd.renderer import VisibilityRenderer
    EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
    ok = True
    for tag, kw in [("sponza4k", dict(preset="sponza", width=3840, height=2160, point_lights=64)),
                    ("bistro4k", dict(preset="bistro", width=3840, height=2160, point_lights=256)),
                    ("sanmiguel4k mf24", dict(preset="san_miguel", width=3840, height=2160, point_lights=256, material_features=24)),
                    ("bistro4k skinned mf63", dict(preset="bistro", width=3840, height=2160, point_lights=128, material_features=63, skinned_fraction=0.4, spot_every=2)),
                    ("zorah4k", dict(preset="zorah", width=3840, height=2160, point_lights=32, size_scale=0.02))]:
        hz, r = None, None
        for step in range(4):
            t0 = time.time()
            sc = Scene(camera_step=step, **kw)
            if r is None: r = VisibilityRenderer(sc, occlusion=True, stats=True)
            else: r.set_camera_from(sc, frame_index=step)
            r.execute()
            o = orc.OracleFrame(sc)
            hz = o.run_occlusion(hz)
            o.gbuffer(); o.light_cluster(); o.shade()
            bad = []
            c = r.counters()
            n = o.count
            if (c.visibleClusters, c.visibleClustersPhase2) != (o.count1, o.count2): bad.append(f"counts gpu {c.visibleClusters}+{c.visibleClustersPhase2} cpu {o.count1}+{o.count2}")
            elif not np.array_equal(r.visible_clusters(), o.clusters[:n]): bad.append("clusters")
            if not np.array_equal(r.visibility(), o.vis): bad.append("vis")
            cov = o.vis != EMPTY
            g = r.gbuffer()
            if not np.array_equal(g["normals"].view(np.uint32)[cov], o.normals.view(np.uint32)[cov]): bad.append("normals")
            if not np.array_equal(g["albedo"][cov], o.albedo[cov]): bad.append("albedo")
            d = np.abs(r.hdr().view(np.uint16).astype(np.int32) - o.hdr.view(np.uint16).astype(np.int32)).reshape(o.H, o.W, 4)[cov]   # uncovered pixels are not written (DeferredCSMain returns): they keep the previous frame
            if d.max() > 1: bad.append(f"hdr(max {d.max()}, n {(d > 1).sum()})")
            if c.droppedRecords or c.droppedClusters: bad.append("dropped")
            print(tag, "step", step, "OK" if not bad else "MISMATCH " + " ".join(bad), f"vis {o.count1}+{o.count2} {time.time()-t0:.1f}s", flush=True)
            ok = ok and not bad
        r.close()
    print("ALL OK" if ok else "SOME FAILED")
    

def bands():
    import numpy as np, orc
    from conftest import Scene
    from basicrenderer_amd.renderer import VisibilityRenderer
    from basicrenderer_amd import compose
    EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
    ok = True
    for preset, lights, n in (("sponza", 64, 2), ("bistro", 256, 4), ("san_miguel", 256, 4)):
        W, H = compose.frame_size(n)
        t0 = time.time()
        sc = Scene(preset, W, H, point_lights=lights)
        o = orc.OracleFrame(sc).run()
        fa, fb, fd = orc.canonical_ids(o.vis, o.clusters[:o.count])
        print(preset, n, "bands", (W, H), "oracle %.1fs" % (time.time() - t0), "clusters", o.count, flush=True)
        for rank in range(n):
            y0, y1 = compose.band_of(rank, n, H)
            r = VisibilityRenderer(sc, band=(y0, y1), occlusion=True, stats=True)
            r.execute(); r.execute()      # second frame tests against the band's own depth chain
            a, b, d = orc.canonical_ids(r.visibility(), r.visible_clusters())
            bad = []
            if not (np.array_equal(a[y0:y1], fa[y0:y1]) and np.array_equal(b[y0:y1], fb[y0:y1]) and np.array_equal(d[y0:y1], fd[y0:y1])): bad.append("ids")
            cov = (o.vis != EMPTY)[y0:y1]
            dd = np.abs(r.hdr().view(np.uint16).astype(np.int32).reshape(H, W, 4)[y0:y1][cov] - o.hdr.view(np.uint16).astype(np.int32).reshape(H, W, 4)[y0:y1][cov])
            if dd.max() > 1: bad.append(f"hdr {dd.max()}")
            c = r.counters()
            if c.droppedRecords or c.droppedClusters: bad.append("dropped")
            print("  rank", rank, "rows", (y0, y1), "OK" if not bad else "MISMATCH " + " ".join(bad), "vis", c.visibleClusters, "+", c.visibleClustersPhase2, flush=True)
            # San-Miguel-class 8K, top band: the meshlet the reference's occlusion test drops at the screen edge -- expected
            known = preset == "san_miguel" and rank == 0 and bad == ["ids"]
            ok = ok and (not bad or known)
            r.close()
    print("ALL OK" if ok else "SOME FAILED")
    

if __name__ == "__main__":
    {"frames": frames, "occlusion": occlusion, "bands": bands}[MODE]()
