"""glTF 2.0 (.gltf / .glb) -> the mesh dictionaries and instances Scene(meshes=..., instances=...) takes.

Harness only (the product boundary is brmi_scene_create_from_meshes; the reference's own importer is out of scope):
enough of the format to put real assets through the LOD builder and the path.  One mesh per triangle primitive (POSITION, NORMAL,
TEXCOORD_0, indices; integer accessors with `normalized` are converted as the specification says), one instance per (node, primitive)
with the node's world transform.  glTF stores matrices column-major for column vectors; read row by row they are the row-vector
matrices of the path (translation in the last row), so no transpose is needed.  glTF texcoords are top-down like the path's.
Not read: sparse accessors, morph targets, skins, cameras, textures (materials are numbered in file order; primitives without one get
the number after the last)."""
import base64
import json
import os
import struct

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfError(ValueError):
    pass


def _read_container(path):
    """Returns (json dict, binary chunk or None)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"glTF":
        return json.loads(data.decode("utf-8")), None
    magic, version, length = struct.unpack_from("<4sII", data, 0)
    if version != 2:
        raise GltfError(f"{path}: glb version {version} (2 is supported)")
    if length > len(data):
        raise GltfError(f"{path}: truncated glb ({len(data)} of {length} bytes)")
    doc, blob, off = None, None, 12
    while off + 8 <= length:
        n, kind = struct.unpack_from("<I4s", data, off)
        body = data[off + 8: off + 8 + n]
        if len(body) != n:
            raise GltfError(f"{path}: chunk at byte {off} runs past the end of the file")
        if kind == b"JSON":
            doc = json.loads(body.decode("utf-8"))
        elif kind == b"BIN\x00" and blob is None:
            blob = body
        off += 8 + ((n + 3) & ~3)
    if doc is None:
        raise GltfError(f"{path}: glb without a JSON chunk")
    return doc, blob


def _buffers(doc, blob, base_dir):
    out = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if blob is None or i != 0:
                raise GltfError(f"buffer {i} has no uri and the file has no binary chunk for it")
            data = blob
        elif uri.startswith("data:"):
            data = base64.b64decode(uri.split(",", 1)[1])
        else:
            with open(os.path.join(base_dir, uri), "rb") as f:
                data = f.read()
        if len(data) < b.get("byteLength", 0):
            raise GltfError(f"buffer {i}: {len(data)} bytes, byteLength says {b['byteLength']}")
        out.append(data)
    return out


def _accessor(doc, buffers, index):
    """The accessor as a [count, width] array of its component type (float32 for normalized integers)."""
    acc = doc["accessors"][index]
    if "sparse" in acc:
        raise GltfError(f"accessor {index}: sparse accessors are not supported")
    dtype, width, count = _COMPONENT.get(acc["componentType"]), _WIDTH.get(acc["type"]), acc["count"]
    if dtype is None or width is None:
        raise GltfError(f"accessor {index}: component type {acc['componentType']} / type {acc['type']}")
    if "bufferView" not in acc:
        return np.zeros((count, width), dtype=dtype)
    view = doc["bufferViews"][acc["bufferView"]]
    item = np.dtype(dtype).itemsize * width
    stride = view.get("byteStride", 0) or item
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    data = buffers[view["buffer"]]
    if count and start + stride * (count - 1) + item > len(data):
        raise GltfError(f"accessor {index}: reads past the end of buffer {view['buffer']}")
    if stride == item:
        a = np.frombuffer(data, dtype=dtype, count=count * width, offset=start).reshape(count, width)
    else:
        raw = np.frombuffer(data, dtype=np.uint8, count=stride * (count - 1) + item, offset=start) if count else np.zeros(0, np.uint8)
        rows = np.lib.stride_tricks.as_strided(raw, shape=(count, item), strides=(stride, 1))
        a = np.ascontiguousarray(rows).view(dtype).reshape(count, width)
    if acc.get("normalized") and dtype != np.float32:
        scale = {np.int8: 127.0, np.uint8: 255.0, np.int16: 32767.0, np.uint16: 65535.0}.get(dtype)
        if scale is None:
            raise GltfError(f"accessor {index}: normalized uint32")
        a = np.maximum(a.astype(np.float32) / np.float32(scale), np.float32(-1.0))
    return a


def _node_matrix(node):
    if "matrix" in node:
        return np.asarray(node["matrix"], dtype=np.float64).reshape(4, 4)          # column-major column-vector == row-major row-vector
    t = np.asarray(node.get("translation", (0, 0, 0)), dtype=np.float64)
    x, y, z, w = np.asarray(node.get("rotation", (0, 0, 0, 1)), dtype=np.float64)
    s = np.asarray(node.get("scale", (1, 1, 1)), dtype=np.float64)
    rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                    [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                    [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])   # column-vector rotation
    m = np.eye(4)
    m[:3, :3] = (rot * s[None, :]).T          # row vector: v * S * R^T, then + t
    m[3, :3] = t
    return m


def load_gltf(path):
    """Returns (meshes, instances): meshes as Scene(meshes=...) takes them, instances as (mesh index, 4x4 row-vector float32 world matrix)."""
    doc, blob = _read_container(path)
    if str(doc.get("asset", {}).get("version", "2.0")).split(".")[0] != "2":
        raise GltfError(f"{path}: glTF version {doc['asset'].get('version')} (2.x is supported)")
    buffers = _buffers(doc, blob, os.path.dirname(os.path.abspath(path)))
    n_materials = len(doc.get("materials", []))
    meshes, first_of = [], {}             # gltf mesh index -> [our mesh indices]
    for mi, mesh in enumerate(doc.get("meshes", [])):
        first_of[mi] = []
        for pi, prim in enumerate(mesh.get("primitives", [])):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue                  # points, lines, strips and fans are not geometry this path draws
            attr = prim["attributes"]
            pos = _accessor(doc, buffers, attr["POSITION"]).astype(np.float32)
            if pos.shape[1] != 3:
                raise GltfError(f"mesh {mi} primitive {pi}: POSITION is not VEC3")
            if "indices" in prim:
                idx = _accessor(doc, buffers, prim["indices"]).astype(np.uint32).ravel()
            else:
                idx = np.arange(len(pos), dtype=np.uint32)
            idx = idx[: len(idx) // 3 * 3]
            if len(idx) == 0:
                continue
            if int(idx.max()) >= len(pos):
                raise GltfError(f"mesh {mi} primitive {pi}: index {int(idx.max())} of {len(pos)} vertices")
            m = dict(positions=np.ascontiguousarray(pos), indices=np.ascontiguousarray(idx), material=prim.get("material", n_materials),
                     name=f"{mesh.get('name', 'mesh%d' % mi)}.{pi}")
            if "NORMAL" in attr:
                nrm = _accessor(doc, buffers, attr["NORMAL"]).astype(np.float32)
                if nrm.shape == pos.shape:
                    m["normals"] = np.ascontiguousarray(nrm)
            if "TEXCOORD_0" in attr:
                uv = _accessor(doc, buffers, attr["TEXCOORD_0"]).astype(np.float32)
                if uv.shape == (len(pos), 2):
                    m["uvs"] = np.ascontiguousarray(uv)
            first_of[mi].append(len(meshes))
            meshes.append(m)
    instances = []
    nodes = doc.get("nodes", [])
    scenes = doc.get("scenes", [])
    roots = scenes[doc.get("scene", 0)].get("nodes", []) if scenes else [i for i in range(len(nodes)) if not any(i in n.get("children", []) for n in nodes)]
    stack = [(r, np.eye(4)) for r in reversed(roots)]
    seen = 0
    while stack:
        ni, parent = stack.pop()
        seen += 1
        if seen > 4 * len(nodes) + 4:
            raise GltfError("node hierarchy has a cycle")
        node = nodes[ni]
        world = _node_matrix(node) @ parent          # row vectors: local first, then the parent's
        for k in first_of.get(node.get("mesh", -1), []):
            instances.append((k, world.astype(np.float32)))
        for c in reversed(node.get("children", [])):
            stack.append((c, world))
    if not meshes:
        raise GltfError(f"{path}: no triangle primitives")
    if not instances:                               # a file without a node hierarchy: every mesh once, untransformed
        instances = [(k, np.eye(4, dtype=np.float32)) for k in range(len(meshes))]
    return meshes, instances


def frame_view(meshes, instances, fov=60.0):
    """A view that looks at the instanced meshes' bounding box from the front (the -z side of a glTF scene faces +z), slightly above."""
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for k, m in instances:
        p = meshes[k]["positions"]
        corners = np.array([[x, y, z, 1.0] for x in (p[:, 0].min(), p[:, 0].max()) for y in (p[:, 1].min(), p[:, 1].max()) for z in (p[:, 2].min(), p[:, 2].max())]) @ np.asarray(m, dtype=np.float64)
        lo, hi = np.minimum(lo, corners[:, :3].min(0)), np.maximum(hi, corners[:, :3].max(0))
    c, ext = (lo + hi) * 0.5, float(max(hi - lo))
    dist = 0.5 * ext / np.tan(np.radians(fov) * 0.5) * 1.3
    return dict(eye=(float(c[0]), float(c[1] + 0.15 * ext), float(c[2] + dist + 0.5 * (hi[2] - lo[2]))), yaw=0.0, pitch=float(-np.arctan2(0.15 * ext, dist)), fov=fov,
                near=max(1e-3, 0.001 * ext), far=10.0 * ext + dist)
