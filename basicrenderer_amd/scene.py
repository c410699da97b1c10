"""Scene: numpy views over the arrays produced by libbrmi_scene.so, and their upload to HBM.

The generator emits the reference's GPU data contract (include/brmi_types.h); this wrapper only
moves bytes: host numpy views -> `SceneBuffers` with host pointers (for a CPU consumer) or with
device pointers (torch uint8 tensors, for libbrmi.so).
"""
import ctypes as C

import numpy as np

from . import capi

PRESETS = {"tiny": 0, "sponza": 1, "bistro": 2, "san_miguel": 3, "zorah": 4}

# SceneBuffers field <- (scene array name, count field or None)
_FIELD_MAP = [
    ("perObject", "perObject", "perObjectCount"), ("normalMatrices", "normalMatrices", None),
    ("perMesh", "perMesh", "perMeshCount"), ("perMeshInstance", "perMeshInstance", "perMeshInstanceCount"),
    ("clodOffsets", "clodOffsets", None), ("meshMetadata", "meshMetadata", "meshMetadataCount"),
    ("lodNodes", "lodNodes", "lodNodeCount"), ("lodGroups", "lodGroups", "lodGroupCount"),
    ("lodSegments", "lodSegments", "lodSegmentCount"), ("groupPageMap", "groupPageMap", "groupPageMapCount"),
    ("materials", "materials", "materialCount"), ("openpbrMaterials", "openpbrMaterials", "openpbrMaterialCount"),
    ("lights", "lights", "lightCount"), ("activeLightIndices", "activeLightIndices", None),
    ("cameras", "cameras", "cameraCount"), ("cullingCameras", "cullingCameras", None),
    ("viewRasterInfo", "viewRasterInfo", None), ("perFrame", "perFrame", None),
    ("activeDraws", "activeDraws", "activeDrawCount"), ("skinningMatrices", "skinningMatrices", "skinningMatrixCount"),
    ("lutOpaqueDielectricEnergyComplement", "lutOdE", None), ("lutOpaqueDielectricAvgEnergyComplement", "lutOdAvg", None),
    ("lutIdealMetalEnergyComplement", "lutImE", None), ("lutIdealMetalAvgEnergyComplement", "lutImAvg", None),
    ("lutFuzzLTC", "lutFuzzLTC", None),
    ("samplers", "samplerDescs", "samplerCount"), ("srgbToLinear", "srgbToLinear", None),
]
TEXTURE_DESC_BYTES = 96      # brmi_texture_desc; its first 8 bytes are the texel pointer (generator output: byte offset into `texels`)


class Scene:
    def __init__(self, preset="sponza", width=3840, height=2160, seed=0, point_lights=64, directional=True,
                 lod_levels=0, size_scale=1.0, material_features=0, camera_step=0, skinned_fraction=0.0, lod_builder="quadtree", spot_every=0, cache_dir=None, export_cache=None,
                 detail=1.0, unique_budget=False, relief_slope=0.0, dag_builder=None, meshes=None, instances=None, view=None):
        """lod_builder: "quadtree" (regular grid DAG) or "own" (the library's cluster-LOD builder); dag_builder = (build_fn, release_fn)
        addresses of a caller-supplied builder with the brmi_dag_build_fn / brmi_dag_release_fn signatures (lod_builder becomes "external").

        meshes / instances / view: a scene of the CALLER's geometry (brmi_scene_create_from_meshes) instead of a preset.  meshes = list of
        dicts {positions [V,3] f32, indices [T*3] u32, normals / uvs / colors optional, material}; instances = list of (mesh index, 4x4
        row-vector model matrix[, reverse_winding]); view = dict(eye, yaw, pitch, fov, near, far)."""
        lib = capi.scene_lib()
        p = capi.SceneParams()
        p.preset = PRESETS[preset] if isinstance(preset, str) else int(preset)
        p.seed, p.width, p.height = seed, width, height
        p.numPointLights, p.withDirectionalLight = point_lights, 1 if directional else 0
        p.lodLevels, p.sizeScale = lod_levels, size_scale
        p.materialFeatures = material_features
        p.cameraStep = camera_step
        p.spotLightEvery = spot_every
        if dag_builder is not None:
            lod_builder = "external"
        p.lodBuilder = capi.LOD_BUILDERS[lod_builder]
        p.detail = detail
        p.uniqueTriangleBudget = 1 if unique_budget else 0
        p.reliefSlope = relief_slope
        p.skinnedFraction1024 = int(round(skinned_fraction * 1024))
        self.preset, self.width, self.height = preset, width, height
        self._lib = lib
        self._params = p
        # cache_dir: take every mesh from CLodCache files (include/brmi_scene.h) instead of building it; export_cache: write them
        if meshes is not None:
            if lod_builder == "quadtree":
                p.lodBuilder = capi.LOD_BUILDERS["own"]
            keep = []
            mi = (capi.MeshInput * len(meshes))()
            for k, m in enumerate(meshes):
                pos = np.ascontiguousarray(m["positions"], dtype=np.float32).reshape(-1, 3); idx = np.ascontiguousarray(m["indices"], dtype=np.uint32).ravel()
                keep += [pos, idx]
                mi[k].positions, mi[k].vertexCount, mi[k].indices, mi[k].indexCount = pos.ctypes.data, len(pos), idx.ctypes.data, len(idx)
                for key, dt, comps in (("normals", np.float32, 3), ("uvs", np.float32, 2), ("colors", np.uint32, 1)):
                    if m.get(key) is not None:
                        a = np.ascontiguousarray(m[key], dtype=dt).reshape(len(pos), comps); keep.append(a)
                        setattr(mi[k], key, a.ctypes.data)
                mi[k].material = int(m.get("material", 0))
            ii = (capi.InstanceInput * len(instances))()
            for k, inst in enumerate(instances):
                ii[k].mesh, ii[k].reverseWinding = int(inst[0]), int(bool(inst[2])) if len(inst) > 2 else 0
                mm = np.asarray(inst[1], dtype=np.float32).reshape(4, 4)
                for r in range(4):
                    for c in range(4):
                        ii[k].model[r][c] = float(mm[r, c])
            vi = capi.ViewInput()
            vi.eye[0], vi.eye[1], vi.eye[2] = (float(x) for x in view["eye"])
            vi.yaw, vi.pitch, vi.fovYDegrees, vi.zNear, vi.zFar = float(view.get("yaw", 0.0)), float(view.get("pitch", 0.0)), float(view.get("fov", 60.0)), float(view.get("near", 0.1)), float(view.get("far", 1000.0))
            b = dag_builder if dag_builder is not None else (None, None)
            self._h = lib.brmi_scene_create_from_meshes(C.byref(p), mi, len(meshes), ii, len(instances), C.byref(vi), b[0], b[1], None)
            if not self._h:
                raise RuntimeError("brmi_scene_create_from_meshes failed: bad mesh / instance input")
            del keep
        elif cache_dir:
            self._h = lib.brmi_scene_create_from_cache(C.byref(p), str(cache_dir).encode())
        elif dag_builder is not None:
            self._h = lib.brmi_scene_create_with_dag_builder(C.byref(p), dag_builder[0], dag_builder[1], None)
        else:
            self._h = lib.brmi_scene_create(C.byref(p))
        if not self._h and cache_dir:
            raise RuntimeError(f"brmi_scene_create_from_cache({cache_dir}) failed: a cache file is missing, truncated or inconsistent")
        if not self._h:
            raise RuntimeError(f"brmi_scene_create failed (lod_builder={lod_builder})")
        if export_cache:
            if lib.brmi_scene_export_cache(self._h, str(export_cache).encode()) < 0:
                raise RuntimeError(f"brmi_scene_export_cache({export_cache}) failed")
        self.arrays, self.counts = {}, {}
        for i, name in enumerate(capi.SCENE_ARRAYS):
            ptr, nbytes, count = capi.vp(), capi.u64(), capi.u32()
            if lib.brmi_scene_array(self._h, i, C.byref(ptr), C.byref(nbytes), C.byref(count)) != 0:
                raise RuntimeError(f"brmi_scene_array({name}) failed")
            n = nbytes.value
            if n:
                buf = (C.c_uint8 * n).from_address(ptr.value)
                self.arrays[name] = np.frombuffer(buf, dtype=np.uint8).copy()   # own the bytes
            else:
                self.arrays[name] = np.zeros(0, dtype=np.uint8)
            self.counts[name] = count.value
        self.slabs = [None]
        for s in range(1, lib.brmi_scene_slab_count(self._h) + 1):
            ptr, nbytes = capi.vp(), capi.u64()
            lib.brmi_scene_slab(self._h, s, C.byref(ptr), C.byref(nbytes))
            buf = (C.c_uint8 * nbytes.value).from_address(ptr.value)
            self.slabs.append(np.frombuffer(buf, dtype=np.uint8).copy())
        st = capi.SceneStats()
        lib.brmi_scene_get_stats(self._h, C.byref(st))
        self.stats = {n: (list(getattr(st, n)) if n.startswith("scene") else getattr(st, n)) for n, _ in capi.SceneStats._fields_}
        # every (instance, meshlet) pair of the scene, all LOD levels: the bound on what a frame can list as visible (a pass sized by it keeps
        # its resolve arena complete: one G-buffer variant per frame instead of two launches, brmi_resolve.hip)
        try:
            md = self.arrays["meshMetadata"].view(np.uint32).reshape(-1, 10)
            seg = self.arrays["lodSegments"].view(np.uint32).reshape(-1, 4)
            offs = self.arrays["clodOffsets"].view(np.uint32)
            order = np.argsort(md[:, 1], kind="stable")
            ends = np.append(md[order, 1][1:], len(seg))
            csum = np.concatenate([[0], np.cumsum(seg[:, 2].astype(np.int64))])
            per_mesh = np.zeros(len(md), dtype=np.int64)
            per_mesh[order] = csum[ends] - csum[md[order, 1]]
            self.stats["instancedMeshlets"] = int(per_mesh[offs[: self.counts.get("clodOffsets", len(offs))]].sum())
        except (KeyError, ValueError, IndexError):
            self.stats["instancedMeshlets"] = 0
        lib.brmi_scene_destroy(self._h)
        self._h = None
        self._keep = []

    # -- host ------------------------------------------------------------------------------------
    def host_buffers(self):
        """SceneBuffers whose pointers address this object's numpy arrays (CPU consumers)."""
        sb = capi.SceneBuffers()
        slab_ptrs = (capi.vp * len(self.slabs))()
        for i, s in enumerate(self.slabs):
            slab_ptrs[i] = s.ctypes.data if s is not None else None
        self._keep.append(slab_ptrs)
        sb.slabs, sb.slabCount = C.cast(slab_ptrs, capi.vp), len(self.slabs)
        for field, arr, cnt in _FIELD_MAP:
            a = self.arrays[arr]
            setattr(sb, field, a.ctypes.data if a.size else None)
            if cnt:
                setattr(sb, cnt, self.counts[arr])
        if self.counts["textureDescs"]:
            descs = self._relocated_texture_descs(self.arrays["texels"].ctypes.data)
            self._keep.append(descs)
            sb.textures, sb.textureCount = descs.ctypes.data, self.counts["textureDescs"]
        return sb

    def _relocated_texture_descs(self, texel_base):
        descs = self.arrays["textureDescs"].copy()
        ptrs = descs.view(np.uint64).reshape(-1, TEXTURE_DESC_BYTES // 8)[:, 0]
        ptrs += np.uint64(texel_base)
        return descs

    def camera_host(self):
        return self.arrays["cameras"]

    def camera_at(self, step, prev_step=None):
        """(cameras, cullingCameras) byte arrays of the preset's camera at `step` of its path (brmi_scene_camera_at; fractions allowed) --
        what the reference's CameraManager writes between frames; the scene itself is not rebuilt."""
        cam, cull = np.zeros(len(self.arrays["cameras"]), dtype=np.uint8), np.zeros(len(self.arrays["cullingCameras"]), dtype=np.uint8)
        if self._lib.brmi_scene_camera_at(C.byref(self._params), float(step), float(step if prev_step is None else prev_step), cam.ctypes.data, cull.ctypes.data) != 0:
            raise RuntimeError("brmi_scene_camera_at failed: this scene has no preset camera path")
        return cam, cull

    def per_frame_host(self):
        return self.arrays["perFrame"]

    # -- device ----------------------------------------------------------------------------------
    def device_buffers(self, device="cuda"):
        """Upload every array to HBM; returns (SceneBuffers with device pointers, list of tensors to keep alive)."""
        import torch
        keep = []
        sb = capi.SceneBuffers()

        def up(a):
            if a is None or a.size == 0:
                return None
            t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
            keep.append(t)
            return t.data_ptr()

        ptrs = np.zeros(len(self.slabs), dtype=np.uint64)
        for i, s in enumerate(self.slabs):
            if s is not None:
                ptrs[i] = up(s)
        sb.slabs, sb.slabCount = up(ptrs.view(np.uint8)), len(self.slabs)
        self.device_arrays = {}
        for field, arr, cnt in _FIELD_MAP:
            n = len(keep)
            setattr(sb, field, up(self.arrays[arr]))
            if len(keep) > n:
                self.device_arrays[arr] = keep[-1]      # the camera manager's role in tests: rewrite a buffer in place
            if cnt:
                setattr(sb, cnt, self.counts[arr])
        if self.counts["textureDescs"]:
            sb.textures, sb.textureCount = up(self._relocated_texture_descs(up(self.arrays["texels"]))), self.counts["textureDescs"]
        return sb, keep
