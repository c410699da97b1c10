#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference (panthuncia/BasicRenderer) holds no golden vectors, images or known-answer tests for the
visibility-buffer path, and neither its HLSL nor its DX12 host code can run in this
environment, so these fixtures are outputs of OUR restatement (oracle/) on seeded procedural scenes.  They pin
the oracle against regressions and give the GPU tests a second, frozen reference; they do not pin the oracle to
the reference ("parity unpinned").

Usage:  python tests/golden/make_golden.py        (from the repository root, after `make scene oracle`)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN_CASES = {
    # name: (preset, W, H, Scene kwargs)
    "golden_tiny": ("tiny", 160, 90, dict(point_lights=5, seed=3)),
    "golden_tiny_lod_coat_fuzz": ("tiny", 160, 90, dict(point_lights=4, seed=5, lod_levels=2, material_features=3)),
    "golden_sponza": ("sponza", 192, 108, dict(point_lights=24, seed=1, size_scale=0.05)),
    # compute skinning + the 2-phase occlusion chain: second frame of a camera path (phase 1 tests against frame 0's chain)
    "golden_tiny_skinned_occlusion": ("tiny", 160, 90, dict(point_lights=3, seed=7, lod_levels=2, skinned_fraction=1.0)),
    # UV streams, alpha-tested rasterisation and texture-sampled materials through the software sampler
    "golden_tiny_textured_alpha": ("tiny", 160, 90, dict(point_lights=4, seed=9, lod_levels=2, material_features=24)),
    # every material feature at once: coat + fuzz with OpenPBR layer textures, mirrored instances, textures, alpha test, vertex colours, spot lights
    "golden_sponza_all_features": ("sponza", 192, 108, dict(point_lights=12, seed=4, size_scale=0.05, lod_levels=2, material_features=127, spot_every=3)),
    # contact-refinement parallax (height maps, with and without a normal map, rays that never hit) on top of textures + layer textures
    "golden_tiny_parallax": ("tiny", 160, 90, dict(point_lights=4, seed=13, lod_levels=2, material_features=128 | 64 | 8 | 3)),
    # three UV sets per page, texture slots (material, OpenPBR layer and height slots) spread over them
    "golden_tiny_uv_sets": ("tiny", 160, 90, dict(point_lights=4, seed=17, lod_levels=2, material_features=256 | 128 | 64 | 16 | 8 | 3)),
}


def render(name):
    import orc
    from basicrenderer_amd import Scene
    preset, W, H, kw = GOLDEN_CASES[name]
    if name.endswith("_occlusion"):
        hz = orc.OracleFrame(Scene(preset, W, H, camera_step=0, **kw), threads=1).run_occlusion(None)
        f = orc.OracleFrame(Scene(preset, W, H, camera_step=1, **kw), threads=1)
        hz = f.run_occlusion(hz)
        f.gbuffer(); f.light_cluster(); f.shade()
        c = f.counters
        mips = np.concatenate([hz[0][int(hz[1][m]): int(hz[1][m + 1]) if m + 1 < hz[2] else int(hz[1][m]) + 1] for m in range(1, hz[2])]).view(np.uint32)
        return dict(clusters=f.clusters[: f.count].copy(), vis=f.vis, depth=f.depth.view(np.uint32), normals=f.normals.view(np.uint32), motion=f.motion, hdr=f.hdr, hzb_mips=mips,
                    counters=np.array([c.nodesVisited, c.meshletsTested, f.count1, f.count2, f.n_replay_nodes.value, f.n_replay_meshlets.value], dtype=np.uint32))
    sc = Scene(preset, W, H, **kw)
    f = orc.OracleFrame(sc, threads=1).run()
    c = f.counters
    return dict(
        clusters=f.clusters[: f.count].copy(), vis=f.vis, depth=f.depth.view(np.uint32), normals=f.normals.view(np.uint32), albedo=f.albedo, coat=f.coat,
        emissive=f.emissive, fuzz=f.fuzz, mr=f.mr, motion=f.motion, hdr=f.hdr, light_clusters=f.light_clusters, light_pages=f.light_pages[: f.pages_used],
        counters=np.array([c.instancesTested, c.instancesVisible, c.nodesVisited, c.bucketRecords, c.meshletsTested, c.visibleClusters], dtype=np.uint32))


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name in (sys.argv[1:] or GOLDEN_CASES):
        data = render(name)
        path = os.path.join(out, name + ".npz")
        np.savez_compressed(path, **data)
        print(name, os.path.getsize(path), "bytes;", int((data["vis"] != np.uint64(0xFFFFFFFFFFFFFFFF)).sum()), "covered px;", len(data["clusters"]), "clusters")
