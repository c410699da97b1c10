#!/usr/bin/env python3
"""Render a Wavefront OBJ or a glTF 2.0 file (.gltf / .glb) through the whole path on an MI355X and write the lit frame as a PNG
(Reinhard + sRGB).

    python examples/render_obj.py model.obj|scene.glb out.png [--size 1920x1080] [--lights 32] [--features 0]
"""
import argparse
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj"); ap.add_argument("png")
    ap.add_argument("--size", default="1920x1080"); ap.add_argument("--lights", type=int, default=32); ap.add_argument("--features", type=int, default=0)
    a = ap.parse_args()
    from basicrenderer_amd import Scene
    from basicrenderer_amd.obj import frame_view, load_obj
    from basicrenderer_amd.renderer import VisibilityRenderer
    w, h = (int(x) for x in a.size.lower().split("x"))
    if a.obj.lower().endswith((".gltf", ".glb")):
        from basicrenderer_amd import gltf
        meshes, instances = gltf.load_gltf(a.obj)
        view = gltf.frame_view(meshes, instances)
    else:
        meshes = load_obj(a.obj)
        instances, view = [(k, np.eye(4, dtype=np.float32)) for k in range(len(meshes))], frame_view(meshes)
    sc = Scene(width=w, height=h, point_lights=a.lights, material_features=a.features, meshes=meshes, instances=instances, view=view)
    r = VisibilityRenderer(sc, occlusion=True)
    r.execute(); r.execute()
    hdr = r.hdr().view(np.float16).reshape(h, w, 4)[..., :3].astype(np.float32)
    ldr = hdr / (1.0 + hdr)
    srgb = np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * np.power(np.maximum(ldr, 1e-8), 1 / 2.4) - 0.055)
    write_png(a.png, (np.clip(srgb, 0, 1) * 255 + 0.5).astype(np.uint8))
    print(f"{a.obj}: {sc.stats['uniqueTriangles']} triangles, {sc.stats['meshletsTotal']} meshlets, {sc.stats['lodLevelsMax']} LOD levels -> {a.png}")


if __name__ == "__main__":
    main()
