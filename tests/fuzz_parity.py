#!/usr/bin/env python3
"""Randomised differential sweep: libbrmi.so against the CPU oracle on many small seeded frames (run on the GPU box:
`gpurun -- python tests/fuzz_parity.py [count] [seed] [size multiplier]`).  Every frame: random preset / size / lights / LOD depth / material feature mix /
skinning / LOD builder / occlusion culling with a camera path / row band; cluster lists, keys, depth, G-buffer exact, HDR <= 1 fp16 ULP on
covered pixels.  TEST INFRASTRUCTURE: imports the oracle."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from conftest import have_clodref
from conftest import Scene
from basicrenderer_amd.renderer import VisibilityRenderer
EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def compare(r, o, band, tag):
    bad = []
    y0, y1 = band if band != (0, 0) else (0, o.H)
    if band == (0, 0):
        if not np.array_equal(r.visible_clusters(), o.clusters[: o.count]): bad.append("clusters")
        if not np.array_equal(r.visibility(), o.vis): bad.append("vis")
        cov = o.vis != EMPTY
        g = r.gbuffer()
        if not np.array_equal(r.depth().view(np.uint32), o.depth.view(np.uint32)): bad.append("depth")
        if not np.array_equal(g["normals"].view(np.uint32)[cov], o.normals.view(np.uint32)[cov]): bad.append("normals")
        for k, ref in (("albedo", o.albedo), ("mr", o.mr), ("motion", o.motion), ("coat", o.coat), ("emissive", o.emissive), ("fuzz", o.fuzz)):
            if not np.array_equal(g[k][cov], ref[cov]): bad.append(k)
    else:
        a, b, d = orc.canonical_ids(r.visibility(), r.visible_clusters()); fa, fb, fd = orc.canonical_ids(o.vis, o.clusters[: o.count])
        if not (np.array_equal(a[y0:y1], fa[y0:y1]) and np.array_equal(b[y0:y1], fb[y0:y1]) and np.array_equal(d[y0:y1], fd[y0:y1])): bad.append("band ids")
        cov = o.vis != EMPTY
    hd = np.abs(r.hdr().view(np.uint16).astype(np.int32).reshape(o.H, o.W, 4) - o.hdr.view(np.uint16).astype(np.int32).reshape(o.H, o.W, 4))[y0:y1][cov[y0:y1]]
    if hd.size and hd.max() > 1: bad.append(f"hdr {hd.max()}")
    c = r.counters()
    if c.droppedRecords or c.droppedClusters: bad.append("dropped")
    return bad


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    fails, t0 = 0, time.time()
    for it in range(count):
        preset = rng.choice(["tiny", "tiny", "sponza", "bistro", "san_miguel", "zorah"])
        big = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # optional third argument: size multiplier (4 = up to 3600 x 2080)
        W, H = rng.randrange(64, 900 * big), rng.randrange(48, 520 * big)
        mf = rng.choice([0, 0, 3, 8, 24, 27, 32, 59, 63, 127, 4, 96 | 11, 128 | 8, 255, 128 | 91, 256 | 8, 256 | 27, 511, 256 | 128 | 64 | 11])
        kw = dict(seed=rng.randrange(1, 1 << 20), point_lights=rng.choice([0, 1, 7, 40, 150]), directional=rng.random() < 0.8, material_features=mf,
                  lod_levels=rng.choice([0, 1, 2, 3]), skinned_fraction=rng.choice([0.0, 0.0, 0.3, 1.0]), spot_every=rng.choice([0, 0, 2, 3]),
                  size_scale={"tiny": 1.0, "sponza": rng.choice([0.05, 0.2]), "bistro": rng.choice([0.05, 0.2]), "san_miguel": 0.02, "zorah": 0.003}[preset])
        if have_clodref() and preset in ("tiny", "sponza") and rng.random() < 0.2: kw["lod_builder"] = "clusterlod"
        occlusion = rng.random() < 0.5
        band = (0, 0)
        if not occlusion and rng.random() < 0.25 and H >= 64:
            y0 = 8 * rng.randrange(0, H // 16); band = (y0, min(H, y0 + 8 * rng.randrange(1, max(2, (H - y0) // 8 + 1))))
        tag = f"#{it} {preset} {W}x{H} mf={mf} " + " ".join(f"{k}={v}" for k, v in kw.items() if k not in ("material_features",)) + (" occ" if occlusion else "") + (f" band={band}" if band != (0, 0) else "")
        try:
            bad = []
            if occlusion and rng.random() < 0.4:
                # two frames in flight: linked passes alternate the frames on a geometry and a shading stream with no host synchronisation in
                # between; the last two frames (one per pass) are then compared with the oracle's sequential frames
                import torch
                steps = rng.choice([2, 3, 4, 5])
                ring = rng.choice([2, 3])
                tag += f" in-flight x{steps} ring {ring}"
                scenes = [Scene(preset, W, H, camera_step=s, **kw) for s in range(steps)]
                passes = [VisibilityRenderer(Scene(preset, W, H, camera_step=0, **kw), occlusion=True, stats=True) for _ in range(ring)]
                for k in range(ring): passes[k].set_history_source(passes[(k - 1) % ring])
                geometry, shading = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
                torch.cuda.synchronize()
                for s in range(steps):
                    with torch.cuda.stream(geometry):
                        passes[s % ring].set_camera_from(scenes[s], frame_index=s)
                        passes[s % ring].execute(shading)
                torch.cuda.synchronize()
                hz, oracles = None, []
                for s in range(steps):
                    o = orc.OracleFrame(scenes[s]); hz = o.run_occlusion(hz)
                    if s >= steps - 2: o.gbuffer(); o.light_cluster(); o.shade()
                    oracles.append(o)
                for s in (steps - 2, steps - 1):
                    c = passes[s % ring].counters()
                    if (c.visibleClusters, c.visibleClustersPhase2) != (oracles[s].count1, oracles[s].count2): bad.append(f"step{s} counts")
                    bad += [f"step{s} {b}" for b in compare(passes[s % ring], oracles[s], (0, 0), tag)]
                for p in passes: p.close()
                passes = oracles = scenes = None
            elif occlusion:
                hz, r = None, None
                for step in range(rng.choice([2, 3])):
                    sc = Scene(preset, W, H, camera_step=step, **kw)
                    if r is None: r = VisibilityRenderer(sc, occlusion=True, stats=True)
                    else: r.set_camera_from(sc, frame_index=step)
                    r.execute()
                    o = orc.OracleFrame(sc); hz = o.run_occlusion(hz); o.gbuffer(); o.light_cluster(); o.shade()
                    c = r.counters()
                    if (c.visibleClusters, c.visibleClustersPhase2) != (o.count1, o.count2): bad.append(f"step{step} counts")
                    bad += [f"step{step} {b}" for b in compare(r, o, (0, 0), tag)]
                r.close()
            else:
                sc = Scene(preset, W, H, **kw)
                r = VisibilityRenderer(sc, stats=True, band=band); r.execute()
                o = orc.OracleFrame(sc).run()
                bad = compare(r, o, band, tag)
                r.close()
        except Exception as e:      # noqa: BLE001
            bad = ["EXCEPTION " + repr(e)]
        r = o = sc = None
        if it % 20 == 19:
            import gc
            import torch
            gc.collect(); torch.cuda.empty_cache()
        if bad:
            fails += 1
            print("MISMATCH", tag, "->", " ".join(bad)[:300], flush=True)
        elif it % 25 == 0:
            print("ok", tag, f"({time.time() - t0:.0f}s)", flush=True)
    print(f"{count} frames, {fails} failing, {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
