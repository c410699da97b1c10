#!/usr/bin/env python3
"""An INDEPENDENT writer of the reference's CLodCache files (tests/golden/clodcache_tiny/), in plain struct.pack / numpy.

Nothing here calls libbrmi_scene.so: the files are assembled field by field in the order the reference serialises them --
    container  mesh_<i>.clodbin    BR/src/Import/CLodCache.cpp:252-259 (ContainerHeader {'CLOD', version 4, reserved, pageCount}),
                                   :309-374 (pageCount ClusterLODGroupDiskLocator {u64 blobOffset, u32 blobSizeBytes, u32 reserved}, then the blobs)
    metadata   mesh_<i>.clodmeta   SerializeMetadata, CLodCache.cpp:171-211, schema 47 (POD vectors = u64 count + elements, strings = u64 length + bytes)
    page blob                      BuildPackedTriangleMeshPageBlob, BR/src/Mesh/ClusterLODUtilities.cpp:2079-2311: CLodPageHeader (64 B),
                                   meshlet descriptors (64 B, clodStructs.hlsli:70-89), float3 positions, oct-snorm16 normals, 3 x u8 triangles
-- so that the reader of libbrmi_scene.so (brmi_scene_create_from_cache) is checked against bytes it did not write itself
(tests/test_oracle_cpu.py, and tests/test_parity_gpu.py renders from the loaded scene and compares with the oracle).

Three flat (single LOD depth) meshes, the mesh count of the `tiny` preset: a relief plane of 14 x 14 meshlets that needs TWO 256 KB pages,
a small dome and a cylinder wall.  Run it to regenerate the committed files:  python3 tests/golden/make_clod_cache.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "clodcache_tiny")
PAGE_SIZE = 256 * 1024
FLT_MAX = float(np.finfo(np.float32).max)


def oct_encode(n):
    n = n / np.linalg.norm(n)
    s = abs(n[0]) + abs(n[1]) + abs(n[2])
    ox, oy = n[0] / s, n[1] / s
    if n[2] < 0:
        ox, oy = (1 - abs(oy)) * (1 if ox >= 0 else -1), (1 - abs(ox)) * (1 if oy >= 0 else -1)
    q = lambda v: int(round(max(-1.0, min(1.0, v)) * 32767.0)) & 0xFFFF
    return q(ox) | (q(oy) << 16)


def enclose(spheres):
    c = np.mean([s[:3] for s in spheres], axis=0)
    r = max(np.linalg.norm(s[:3] - c) + s[3] for s in spheres)
    return np.array([c[0], c[1], c[2], r * (1 + 1e-5)], dtype=np.float64)


def surface(kind):
    """(position(u, v), meshlets in u, meshlets in v)"""
    if kind == "plane":
        return (lambda u, v: np.array([-2 + 4 * v, 0.12 * np.sin(7 * u) * np.cos(5 * v), -2 + 4 * u])), 14, 14
    if kind == "dome":
        def f(u, v):
            th, ph = -2 * np.pi * u, 0.15 + 1.2 * v
            return np.array([0.5 * np.sin(ph) * np.cos(th), 0.5 * np.cos(ph), 0.5 * np.sin(ph) * np.sin(th)])
        return f, 2, 1
    return (lambda u, v: np.array([0.25 * np.cos(-2 * np.pi * u), 1.5 * v, 0.25 * np.sin(-2 * np.pi * u)])), 1, 2


def build_meshlets(kind):
    f, nu, nv = surface(kind)
    NU, NV = nu * 8, nv * 8
    out = []
    for mj in range(nv):
        for mi in range(nu):
            pos = np.zeros((81, 3), dtype=np.float32)
            nrm = np.zeros(81, dtype=np.uint32)
            for lj in range(9):
                for li in range(9):
                    u, v = (mi * 8 + li) / NU, (mj * 8 + lj) / NV
                    p = f(u, v)
                    e = 1e-4
                    n = np.cross(f(u + e, v) - f(u - e, v), f(u, v + e) - f(u, v - e))
                    pos[lj * 9 + li] = p
                    nrm[lj * 9 + li] = oct_encode(n)
            tri = []
            for qj in range(8):
                for qi in range(8):
                    a = qj * 9 + qi
                    b, c, d = a + 1, a + 10, a + 9
                    tri += [a, b, c, a, c, d] if (qi + qj) % 2 == 0 else [a, b, d, b, c, d]
            lo, hi = pos.min(axis=0).astype(np.float64), pos.max(axis=0).astype(np.float64)
            c = (lo + hi) / 2
            r = float(np.max(np.linalg.norm(pos.astype(np.float64) - c, axis=1))) * (1 + 1e-5) + 1e-7
            out.append(dict(pos=pos, nrm=nrm, tri=bytes(tri), bounds=np.array([c[0], c[1], c[2], r])))
    return out


def page_blob(meshlets, group_of):
    """CLodPageHeader | descriptors | positions | normals | triangles; `group_of[k]` = mesh-local group of the page's k-th meshlet."""
    M = len(meshlets)
    V = sum(len(m["nrm"]) for m in meshlets)
    T = sum(len(m["tri"]) // 3 for m in meshlets)
    align4 = lambda x: (x + 3) & ~3
    desc_off = 64
    pos_off = align4(desc_off + M * 64)
    nrm_off = align4(pos_off + V * 12)
    bone_off = align4(nrm_off + V * 4)
    tri_off = align4(bone_off)
    size = align4(tri_off + T * 3)
    blob = bytearray(size)
    # meshletCount, positionFormat FLOAT3 (1), attributeMask NORMAL (1), uvSetCount, descriptorOffset, uvDescriptorOffset, positionBitstreamOffset, normalArrayOffset,
    # colorArrayOffset, jointArrayOffset, weightArrayOffset, uvBitstreamDirectoryOffset, triangleStreamOffset, boneIndexStreamOffset, reserved0, reserved1
    struct.pack_into("<16I", blob, 0, M, 1, 1, 0, desc_off, 0, pos_off, nrm_off, 0, 0, 0, 0, tri_off, bone_off, 0, 0)
    pc = ac = tc = 0
    for k, m in enumerate(meshlets):
        nv, nt = len(m["nrm"]), len(m["tri"]) // 3
        # positionBitOffset (bytes), vertexAttributeOffset (elements), triangleByteOffset, boneListOffset, minQ xyz, vertexCount << 24,
        # triangleCount | (refinedGroup + 1) << 16, boneCount, sourceGroupLocalIndex, reserved3, bounds
        struct.pack_into("<4I3i5I4f", blob, desc_off + k * 64, pc, ac, tc, 0, 0, 0, 0, nv << 24, nt | (0 << 16), 0, group_of[k], 0, *[float(np.float32(x)) for x in m["bounds"]])
        blob[pos_off + pc: pos_off + pc + nv * 12] = m["pos"].tobytes()
        blob[nrm_off + ac * 4: nrm_off + (ac + nv) * 4] = m["nrm"].tobytes()
        blob[tri_off + tc: tri_off + tc + nt * 3] = m["tri"]
        pc += nv * 12; ac += nv; tc += nt * 3
    assert size <= PAGE_SIZE
    return bytes(blob)


def vec(fmt_elem, items):
    return struct.pack("<Q", len(items)) + b"".join(struct.pack("<" + fmt_elem, *it) for it in items)


def string(s):
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def build_mesh(kind, index):
    meshlets = build_meshlets(kind)
    # groups of up to 16 meshlets, one terminal segment each; pages filled in order (a segment never straddles a page)
    per_meshlet = 64 + 81 * 16 + 384
    pages, cur, cur_bytes = [], [], 64
    groups, segments = [], []
    k = 0
    while k < len(meshlets):
        chunk = list(range(k, min(k + 16, len(meshlets))))
        need = per_meshlet * len(chunk)
        if cur_bytes + need > PAGE_SIZE - 256:
            pages.append(cur); cur, cur_bytes = [], 64
        segments.append(dict(refined=-1, first=len(cur), count=len(chunk), page=len(pages), meshlets=chunk))
        groups.append(dict(meshlets=chunk, segment=len(segments) - 1))
        cur += [(m, len(groups) - 1) for m in chunk]
        cur_bytes += need
        k += len(chunk)
    pages.append(cur)
    blobs = [page_blob([meshlets[m] for m, _ in pg], [g for _, g in pg]) for pg in pages]

    for g in groups:
        g["bounds"] = enclose([meshlets[m]["bounds"] for m in g["meshlets"]])
    seg_bounds = [groups[i]["bounds"] for i in range(len(segments))]
    # BVH: node 0 = super-root over the depth roots, node 1 = root of depth 0, 8-wide tiers below it, children contiguous
    leaves = [dict(leaf=True, seg=i, cull=seg_bounds[i], lod=groups[i]["bounds"], err=FLT_MAX) for i in range(len(segments))]
    tiers = [leaves]
    while len(tiers[-1]) > 1:
        below, up = tiers[-1], []
        for i in range(0, len(below), 8):
            kids = below[i:i + 8]
            up.append(dict(leaf=False, kids=kids, cull=enclose([c["cull"] for c in kids]), lod=enclose([c["lod"] for c in kids]), err=max(c["err"] for c in kids)))
        tiers.append(up)
    nodes = [None, None]
    root = tiers[-1][0]
    root["slot"] = 1
    queue = [root]
    while queue:
        n = queue.pop(0)
        if not n["leaf"]:
            n["first"] = len(nodes)
            for c in n["kids"]:
                c["slot"] = len(nodes); nodes.append(None)
            queue += n["kids"]
        nodes[n["slot"]] = n
    sup = dict(leaf=False, first=1, kids=[root], cull=enclose([root["cull"]]), lod=enclose([root["lod"]]), err=FLT_MAX)
    nodes[0] = sup

    def node_bytes(n):
        f32 = lambda a: [float(np.float32(x)) for x in a]
        if n["leaf"]:
            head = (2, n["seg"], segments[n["seg"]]["refined"] + 1, n["seg"])          # isLeaf = segment leaf, segment index, refinedGroup + 1, owner group (= segment here)
        else:
            head = (0, n["first"], len(n["kids"]) - 1, 0)
        return struct.pack("<4I4f4f4f", *head, *f32(n["cull"]), *f32(n["lod"]), float(np.float32(n["err"])), 0.0, 0.0, 0.0)

    first_meshlet = 0
    group_recs = []
    for g in groups:
        b = [float(np.float32(x)) for x in g["bounds"]]
        # centerAndRadius, error, firstMeshlet, meshletCount, depth, firstGroupVertex, groupVertexCount, firstSegment, segmentCount, terminalSegmentCount, flags,
        # pageMapBase, pageCount, parentGroupId, maxParentError, representationError
        group_recs.append(struct.pack("<5f2Ii8Ii2f", *b, 0.0, first_meshlet, len(g["meshlets"]), 0, 0, 0, g["segment"], 1, 1, 0, 0, 0, -1, FLT_MAX, 0.0))
        first_meshlet += len(g["meshlets"])
    seg_recs = [struct.pack("<i3I", s["refined"], s["first"], s["count"], s["page"]) for s in segments]

    # container
    header = struct.pack("<4I", 0x444F4C43, 4, 0, len(blobs))
    off = len(header) + 16 * len(blobs)
    locators = []
    for b in blobs:
        locators.append((off, len(b), 0)); off += len(b)
    container = header + b"".join(struct.pack("<QII", *l) for l in locators) + b"".join(blobs)

    meta = struct.pack("<IQ", 47, 0x7465737431)                                   # schema version, build config hash
    meta += struct.pack("<Q", len(group_recs)) + b"".join(group_recs)
    meta += struct.pack("<Q", len(seg_recs)) + b"".join(seg_recs)
    meta += vec("4f", [[float(np.float32(x)) for x in b] for b in seg_bounds])
    meta += struct.pack("<4f", *[float(np.float32(x)) for x in sup["cull"]])      # objectBoundingSphere
    meta += struct.pack("<B", 0)                                                   # no inline group chunks
    meta += struct.pack("<Q", 0)                                                   # groupDiskLocators: none (container layout)
    meta += vec("QII", locators)                                                   # pageDiskLocators
    refs, offs = [], [0]
    for g in groups:
        refs.append((segments[g["segment"]]["page"],)); offs.append(len(refs))
    meta += vec("I", refs) + vec("I", [(o,) for o in offs])
    meta += struct.pack("<3I", len(blobs), len(blobs), 0)                          # trianglePageCount, voxelPageBase, voxelPageCount
    meta += string("tests/golden/make_clod_cache.py") + string(f"/independent/mesh_{index}") + string("") + struct.pack("<Q", 0x7465737431) + string(f"mesh_{index}.clodbin")
    meta += struct.pack("<Q", len(nodes)) + b"".join(node_bytes(n) for n in nodes)
    meta += vec("2I", [(2, len(nodes) - 2)]) + vec("I", [(1,)])                    # lodNodeRanges of depth 0, lodLevelRoots
    meta += struct.pack("<2I", 0, len(tiers) + 1)                                  # maxDepth, maxTraversalDepth
    return container, meta, dict(meshlets=len(meshlets), pages=len(blobs), triangles=128 * len(meshlets), groups=len(groups), nodes=len(nodes))


def main():
    os.makedirs(OUT, exist_ok=True)
    summary = {}
    for i, kind in enumerate(["plane", "dome", "cylinder"]):
        container, meta, info = build_mesh(kind, i)
        open(os.path.join(OUT, f"mesh_{i}.clodbin"), "wb").write(container)
        open(os.path.join(OUT, f"mesh_{i}.clodmeta"), "wb").write(meta)
        summary[i] = info
        print(kind, info, len(container), len(meta))
    return summary


if __name__ == "__main__":
    main()
