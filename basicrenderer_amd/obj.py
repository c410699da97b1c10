"""Wavefront OBJ -> the mesh dictionaries Scene(meshes=...) takes (harness only; the product boundary is brmi_scene_create_from_meshes).

One mesh per `usemtl` / `o` / `g` run (materials are numbered in order of first use); faces are fan-triangulated; a corner is the
(position, texcoord, normal) index triple of the file, so seams keep their own vertices.  Normals are left to the library when the file has
none.  OBJ texcoords are bottom-up: v is flipped to the top-down convention of the path's textures."""
import numpy as np


def load_obj(path):
    pos, tex, nrm = [], [], []
    groups, order = {}, []              # material name -> {"corner": {(vi, ti, ni): local}, "tris": []}
    current = "default"

    def group(name):
        if name not in groups:
            groups[name] = dict(corner={}, keys=[], tris=[])
            order.append(name)
        return groups[name]

    def index(tok, n):
        i = int(tok)
        return i - 1 if i > 0 else n + i

    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if p[0] == "v":
                pos.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                tex.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
            elif p[0] == "vn":
                nrm.append([float(x) for x in p[1:4]])
            elif p[0] in ("usemtl", "o", "g") and len(p) > 1:
                current = p[1] if p[0] == "usemtl" else current
            elif p[0] == "f":
                g = group(current)
                corners = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    key = (index(parts[0], len(pos)),
                           index(parts[1], len(tex)) if len(parts) > 1 and parts[1] else -1,
                           index(parts[2], len(nrm)) if len(parts) > 2 and parts[2] else -1)
                    if key not in g["corner"]:
                        g["corner"][key] = len(g["keys"]); g["keys"].append(key)
                    corners.append(g["corner"][key])
                for k in range(1, len(corners) - 1):
                    g["tris"].append((corners[0], corners[k], corners[k + 1]))
    P, T, N = np.asarray(pos, dtype=np.float32).reshape(-1, 3), np.asarray(tex, dtype=np.float32).reshape(-1, 2), np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
    meshes = []
    for mi, name in enumerate(order):
        g = groups[name]
        if not g["tris"]:
            continue
        keys = np.asarray(g["keys"], dtype=np.int64)
        m = dict(positions=P[keys[:, 0]], indices=np.asarray(g["tris"], dtype=np.uint32).ravel(), material=mi, name=name)
        if len(T) and (keys[:, 1] >= 0).all():
            uv = T[keys[:, 1]].copy(); uv[:, 1] = 1.0 - uv[:, 1]
            m["uvs"] = uv
        if len(N) and (keys[:, 2] >= 0).all():
            m["normals"] = N[keys[:, 2]]
        meshes.append(m)
    return meshes


def frame_view(meshes, fov=60.0):
    """A view that looks at the meshes' bounding box from the front, slightly above."""
    lo = np.min([m["positions"].min(0) for m in meshes], 0); hi = np.max([m["positions"].max(0) for m in meshes], 0)
    c, ext = (lo + hi) * 0.5, float(max(hi - lo))
    dist = 0.5 * ext / np.tan(np.radians(fov) * 0.5) * 1.3
    return dict(eye=(float(c[0]), float(c[1] + 0.15 * ext), float(c[2] + dist + 0.5 * (hi[2] - lo[2]))), yaw=0.0, pitch=float(-np.arctan2(0.15 * ext, dist)), fov=fov,
                near=max(1e-3, 0.001 * ext), far=10.0 * ext + dist)
