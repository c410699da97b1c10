"""ctypes mirrors of include/brmi.h, include/brmi_scene.h and loaders for the in-tree libraries.

Harness-side plumbing only: the product is libbrmi.so (HIP kernels behind the C ABI).  This module
never imports anything from oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

u32, u64, f32, vp = C.c_uint32, C.c_uint64, C.c_float, C.c_void_p


class SceneParams(C.Structure):
    _fields_ = [("preset", u32), ("seed", u32), ("width", u32), ("height", u32), ("numPointLights", u32),
                ("withDirectionalLight", u32), ("lodLevels", u32), ("sizeScale", f32), ("skinnedFraction1024", u32),
                ("materialFeatures", u32), ("cameraStep", u32), ("lodBuilder", u32), ("spotLightEvery", u32), ("detail", f32), ("uniqueTriangleBudget", u32), ("reliefSlope", f32)]


class MeshInput(C.Structure):
    _fields_ = [("positions", vp), ("normals", vp), ("uvs", vp), ("colors", vp), ("vertexCount", C.c_size_t), ("indices", vp), ("indexCount", C.c_size_t),
                ("material", u32), ("reserved", u32 * 3)]


class InstanceInput(C.Structure):
    _fields_ = [("mesh", u32), ("reverseWinding", u32), ("model", (f32 * 4) * 4)]


class ViewInput(C.Structure):
    _fields_ = [("eye", f32 * 3), ("yaw", f32), ("pitch", f32), ("fovYDegrees", f32), ("zNear", f32), ("zFar", f32)]


class DagGroup(C.Structure):
    _fields_ = [("depth", C.c_int32), ("center", f32 * 3), ("radius", f32), ("error", f32), ("firstCluster", u32), ("clusterCount", u32)]


class DagCluster(C.Structure):
    _fields_ = [("group", C.c_int32), ("refined", C.c_int32), ("center", f32 * 3), ("radius", f32), ("error", f32),
                ("vertexCount", u32), ("triangleCount", u32), ("firstVertex", u32), ("firstTriangleByte", u32)]


class Dag(C.Structure):
    """brmi_dag (include/brmi_scene.h): a cluster-LOD DAG as flat arrays."""
    _fields_ = [("groups", C.POINTER(DagGroup)), ("groupCount", u32), ("clusters", C.POINTER(DagCluster)), ("clusterCount", u32),
                ("vertexRefs", C.POINTER(u32)), ("vertexRefCount", u32), ("triangles", C.POINTER(C.c_uint8)), ("triangleBytes", u32), ("owner", vp)]


LOD_BUILDERS = {"quadtree": 0, "external": 1, "own": 2}


class SceneStats(C.Structure):
    _fields_ = [("uniqueTriangles", u64), ("instancedTriangles", u64)] + \
               [(n, u32) for n in ("meshes instances meshletsTotal meshletsLod0 pages nodes groups segments "
                                   "lights materials maxBvhDepth lodLevelsMax").split()] + \
               [("sceneMin", f32 * 3), ("sceneMax", f32 * 3)]


# order == enum brmi_scene_array
SCENE_ARRAYS = ["perObject", "normalMatrices", "perMesh", "perMeshInstance", "clodOffsets", "meshMetadata", "lodNodes",
                "lodGroups", "lodSegments", "groupPageMap", "materials", "openpbrMaterials", "lights",
                "activeLightIndices", "cameras", "cullingCameras", "viewRasterInfo", "perFrame", "activeDraws",
                "skinningMatrices", "lutOdE", "lutOdAvg", "lutImE", "lutImAvg", "lutFuzzLTC",
                "textureDescs", "texels", "samplerDescs", "srgbToLinear"]


class SceneBuffers(C.Structure):
    """brmi_scene_buffers (include/brmi.h).  Pointers are host or device addresses."""
    _fields_ = [
        ("slabs", vp), ("slabCount", u32),
        ("perObject", vp), ("perObjectCount", u32),
        ("normalMatrices", vp),
        ("perMesh", vp), ("perMeshCount", u32),
        ("perMeshInstance", vp), ("perMeshInstanceCount", u32),
        ("clodOffsets", vp),
        ("meshMetadata", vp), ("meshMetadataCount", u32),
        ("lodNodes", vp), ("lodNodeCount", u32),
        ("lodGroups", vp), ("lodGroupCount", u32),
        ("lodSegments", vp), ("lodSegmentCount", u32),
        ("groupPageMap", vp), ("groupPageMapCount", u32),
        ("materials", vp), ("materialCount", u32),
        ("openpbrMaterials", vp), ("openpbrMaterialCount", u32),
        ("lights", vp), ("lightCount", u32),
        ("activeLightIndices", vp),
        ("cameras", vp), ("cameraCount", u32),
        ("cullingCameras", vp),
        ("viewRasterInfo", vp),
        ("perFrame", vp),
        ("activeDraws", vp), ("activeDrawCount", u32),
        ("skinningMatrices", vp), ("skinningMatrixCount", u32),
        ("lutOpaqueDielectricEnergyComplement", vp),
        ("lutOpaqueDielectricAvgEnergyComplement", vp),
        ("lutIdealMetalEnergyComplement", vp),
        ("lutIdealMetalAvgEnergyComplement", vp),
        ("lutFuzzLTC", vp),
        ("textures", vp), ("textureCount", u32),
        ("samplers", vp), ("samplerCount", u32),
        ("srgbToLinear", vp),
    ]


class Config(C.Structure):
    _fields_ = [("structSize", u32), ("width", u32), ("height", u32), ("maxVisibleClusters", u32),
                ("maxTraversalRecords", u32), ("enableOcclusionCulling", u32), ("enableClusteredLighting", u32),
                ("enablePunctualLights", u32), ("lightClusterSize", u32 * 3), ("phase2ExpansionFactor", u32),
                ("collectPassStatistics", u32), ("maxBvhLevels", u32), ("bandY0", u32), ("bandY1", u32),
                ("keepUniformLayerPlanes", u32), ("stripeRows", u32), ("stripeCount", u32), ("stripeIndex", u32), ("fullHeight", u32), ("dynamicBand", u32), ("reserved", u32 * 2)]


class ResourceDesc(C.Structure):
    _fields_ = [("id", u32), ("name", C.c_char_p), ("bytes", u64), ("usage", u32), ("width", u32), ("height", u32),
                ("bytesPerPixel", u32), ("tileW", u32), ("tileH", u32)]


class ResourceBinding(C.Structure):
    _fields_ = [("id", u32), ("ptr", vp), ("bytes", u64)]


class FrameUpdate(C.Structure):
    _fields_ = [("mainCameraHost", vp), ("perFrameHost", vp), ("frameIndex", u32)]


class Counters(C.Structure):
    _fields_ = [(n, u32) for n in ("instancesTested instancesVisible nodesVisited bucketRecords meshletsTested "
                                   "visibleClusters visibleClustersPhase2 droppedRecords droppedClusters "
                                   "lightPagesUsed replayNodes replayMeshlets").split()] + [("reserved", u32 * 6)]


class ComposeConfig(C.Structure):
    """brmi_compose_config (include/brmi_compose.h)."""
    _fields_ = [("structSize", u32), ("width", u32), ("bandY0", u32), ("bandY1", u32), ("bytesPerPixel", u32), ("transport", u32), ("depth", u32),
                ("rank", u32), ("nRanks", u32), ("device", C.c_int32), ("path", u32), ("waitTimeoutMs", u32), ("frameHeight", u32), ("reserved", u32 * 3)]


COMPOSE_EXPORTS = ["brmi_compose_unique_id", "brmi_compose_create", "brmi_compose_staging_bytes", "brmi_compose_output_bytes", "brmi_compose_bind",
                   "brmi_compose_submit", "brmi_compose_finish", "brmi_compose_destroy", "brmi_compose_last_error",
                   "brmi_compose_alloc_shared", "brmi_compose_export", "brmi_compose_import", "brmi_compose_last_wait_status", "brmi_compose_submit_rows", "brmi_compose_wait_source", "brmi_compose_set_bounds", "brmi_compose_balance_rows"]
COMPOSE_HANDLE_BYTES = 160
_compose_lib = None


def compose_lib():
    """libbrmi_compose.so: RCCL composition of the row-band partition behind a C ABI.  torch is imported first so that its RCCL / HIP
    runtime (same SONAMEs) are the ones the process uses."""
    global _compose_lib
    if _compose_lib is None:
        import torch  # noqa: F401
        path = os.path.join(LIB_DIR, "libbrmi_compose.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make compose` (or __graft_entry__.build())")
        try:      # torch ships librccl.so without the .1 suffix: map it under its SONAME before the library asks for it
            C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
        except OSError:
            pass
        lib = C.CDLL(path)
        lib.brmi_compose_unique_id.argtypes = [vp]
        lib.brmi_compose_create.argtypes = [C.POINTER(ComposeConfig), C.c_char_p, C.POINTER(vp)]
        lib.brmi_compose_staging_bytes.argtypes = [vp]
        lib.brmi_compose_staging_bytes.restype = u64
        lib.brmi_compose_output_bytes.argtypes = [vp]
        lib.brmi_compose_output_bytes.restype = u64
        lib.brmi_compose_bind.argtypes = [vp, vp, u64, vp, u64]
        lib.brmi_compose_submit.argtypes = [vp, vp, vp]
        lib.brmi_compose_finish.argtypes = [vp, vp, C.POINTER(vp)]
        lib.brmi_compose_destroy.argtypes = [vp]
        lib.brmi_compose_alloc_shared.argtypes = [vp]
        lib.brmi_compose_export.argtypes = [vp, C.c_char_p]
        lib.brmi_compose_import.argtypes = [vp, C.c_char_p, u32]
        lib.brmi_compose_last_wait_status.argtypes = [vp]
        lib.brmi_compose_submit_rows.argtypes = [vp, vp, u32, u32, vp]
        lib.brmi_compose_wait_source.argtypes = [vp, vp, vp]; lib.brmi_compose_wait_source.restype = C.c_int
        lib.brmi_compose_destroy.restype = None
        lib.brmi_compose_last_error.argtypes = [vp]
        lib.brmi_compose_last_error.restype = C.c_char_p
        _compose_lib = lib
    return _compose_lib


DECLARE_CB = C.CFUNCTYPE(None, vp, C.POINTER(ResourceDesc))

RES_NAMES = ["VISIBILITY", "LINEAR_DEPTH", "GBUF_NORMALS", "GBUF_ALBEDO", "GBUF_COAT", "GBUF_EMISSIVE", "GBUF_FUZZ",
             "GBUF_METALLIC_ROUGHNESS", "GBUF_MOTION_VECTORS", "HDR_COLOR", "VISIBLE_CLUSTERS", "LIGHT_CLUSTERS",
             "LIGHT_PAGES", "HZB", "WORKSPACE"]
RES = {n: i for i, n in enumerate(RES_NAMES)}
STAGE_NAMES = ["clear", "cull", "raster", "depth_copy", "hzb", "cull2", "raster2", "gbuffer", "light_cluster", "shade"]

_scene_lib = None
_brmi_lib = None


def scene_lib():
    """libbrmi_scene.so: the host-side procedural scene generator."""
    global _scene_lib
    if _scene_lib is None:
        path = os.path.join(LIB_DIR, "libbrmi_scene.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make scene` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.brmi_scene_create.restype = vp
        lib.brmi_scene_create.argtypes = [C.POINTER(SceneParams)]
        lib.brmi_scene_create_with_dag_builder.restype = vp
        lib.brmi_scene_create_with_dag_builder.argtypes = [C.POINTER(SceneParams), vp, vp, vp]
        lib.brmi_scene_create_from_meshes.restype = vp
        lib.brmi_scene_create_from_meshes.argtypes = [C.POINTER(SceneParams), C.POINTER(MeshInput), u32, C.POINTER(InstanceInput), u32, C.POINTER(ViewInput), vp, vp, vp]
        lib.brmi_lod_build.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.POINTER(Dag)]
        lib.brmi_lod_release.argtypes = [vp, C.POINTER(Dag)]
        lib.brmi_lod_release.restype = None
        lib.brmi_scene_create_from_cache.restype = vp
        lib.brmi_scene_create_from_cache.argtypes = [C.POINTER(SceneParams), C.c_char_p]
        lib.brmi_scene_export_cache.argtypes = [vp, C.c_char_p]
        lib.brmi_scene_destroy.argtypes = [vp]
        lib.brmi_scene_array.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u32)]
        lib.brmi_scene_slab_count.argtypes = [vp]
        lib.brmi_scene_slab_count.restype = u32
        lib.brmi_scene_slab.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
        lib.brmi_scene_get_stats.argtypes = [vp, C.POINTER(SceneStats)]
        lib.brmi_scene_camera_at.argtypes = [C.POINTER(SceneParams), C.c_double, C.c_double, vp, vp]
        _scene_lib = lib
    return _scene_lib


BRMI_EXPORTS = ["brmi_abi_version", "brmi_default_config", "brmi_create", "brmi_declare", "brmi_set_scene", "brmi_setup", "brmi_set_band",
                "brmi_update", "brmi_execute", "brmi_execute_split", "brmi_destroy", "brmi_last_error", "brmi_clear_visibility", "brmi_cull",
                "brmi_raster", "brmi_depth_copy", "brmi_build_hzb", "brmi_invalidate_hzb", "brmi_set_history_source", "brmi_gbuffer", "brmi_light_clustering",
                "brmi_shade", "brmi_set_shade_slabs", "brmi_read_counters", "brmi_stage_times", "brmi_set_timed_stages", "brmi_algorithmic_bytes", "brmi_algorithmic_bytes_launched", "brmi_debug_arith", "brmi_debug_arith_in_range", "brmi_debug_read_bin_records", "brmi_debug_wide_triangles", "brmi_debug_lean_clusters", "brmi_debug_read_lean_queue", "brmi_debug_read_held"]


def brmi_lib():
    """libbrmi.so: HIP kernels + C ABI.  Fails loudly when the extension has not been built."""
    global _brmi_lib
    if _brmi_lib is None:
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64 and owns
        # the device memory we are handed, so it must be loaded first; libbrmi.so's NEEDED libamdhip64.so.7
        # then resolves (by SONAME) to the copy already mapped instead of pulling in /opt/rocm's.
        import torch  # noqa: F401
        path = os.environ.get("BRMI_LIB_PATH") or os.path.join(LIB_DIR, "libbrmi.so")   # override: A/B builds of the same ABI
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension must be built (make hip); there is no CPU fallback")
        lib = C.CDLL(path)
        lib.brmi_abi_version.restype = u32
        lib.brmi_default_config.argtypes = [C.POINTER(Config), u32, u32]
        lib.brmi_default_config.restype = None
        lib.brmi_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        lib.brmi_declare.argtypes = [vp, DECLARE_CB, vp]
        lib.brmi_set_scene.argtypes = [vp, C.POINTER(SceneBuffers)]
        lib.brmi_setup.argtypes = [vp, C.POINTER(ResourceBinding), u32, vp]
        lib.brmi_update.argtypes = [vp, C.POINTER(FrameUpdate), vp]
        lib.brmi_execute.argtypes = [vp, vp]
        lib.brmi_execute_split.argtypes = [vp, vp, vp]
        lib.brmi_destroy.argtypes = [vp]
        lib.brmi_destroy.restype = None
        lib.brmi_last_error.argtypes = [vp]
        lib.brmi_last_error.restype = C.c_char_p
        for n in ("brmi_clear_visibility", "brmi_depth_copy", "brmi_build_hzb", "brmi_gbuffer", "brmi_light_clustering", "brmi_shade"):
            getattr(lib, n).argtypes = [vp, vp]
        lib.brmi_invalidate_hzb.argtypes = [vp]
        lib.brmi_set_history_source.argtypes = [vp, vp]
        lib.brmi_set_shade_slabs.argtypes = [vp, u32, vp, vp]
        lib.brmi_cull.argtypes = [vp, u32, vp]
        lib.brmi_raster.argtypes = [vp, u32, vp]
        lib.brmi_read_counters.argtypes = [vp, C.POINTER(Counters), vp]
        lib.brmi_stage_times.argtypes = [vp, C.POINTER(f32)]
        lib.brmi_set_timed_stages.argtypes = [vp, u32]
        lib.brmi_algorithmic_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        lib.brmi_algorithmic_bytes_launched.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        lib.brmi_debug_arith.argtypes = [vp, vp, vp, vp, vp, u32, vp]
        lib.brmi_debug_arith_in_range.argtypes = [vp, vp, vp, vp, u32, vp]
        _brmi_lib = lib
    return _brmi_lib
