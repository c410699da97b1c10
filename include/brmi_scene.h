/*
 * brmi_scene.h -- C ABI of the synthetic scene generator (libbrmi_scene.so, host only).
 *
 * The reference ships no assets (models/ and textures/ are git-ignored, .gitignore:17-18), so
 * the benchmark frames are procedural stand-ins emitted directly in the reference's GPU data
 * contract (include/brmi_types.h): 256 KB page slabs laid out like
 * BuildPackedTriangleMeshPageBlob (BR/src/Mesh/ClusterLODUtilities.cpp:2079-2311), one 8-wide
 * BVH per DAG depth under a super-root (ClusterLODUtilities.cpp:4606-4900), groups / segments /
 * page map, per-object / per-mesh / per-instance buffers, cameras, lights and materials.
 * Presets follow SURVEY.md section 8(d).
 */
#ifndef BRMI_SCENE_H
#define BRMI_SCENE_H

#include <stddef.h>
#include <stdint.h>
#include "brmi_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum brmi_scene_preset {
    BRMI_PRESET_TINY        = 0,  /* a handful of meshlets; unit tests */
    BRMI_PRESET_SPONZA      = 1,  /* atrium, ~262k tris flat LOD, 1 directional (+N point) lights */
    BRMI_PRESET_BISTRO      = 2,  /* street, ~3M tris, ~2000 instances of ~150 meshes, LOD DAG */
    BRMI_PRESET_SAN_MIGUEL  = 3,  /* ~10M tris, high depth complexity */
    BRMI_PRESET_ZORAH       = 4   /* massive instancing + deep LOD */
};

/* Who builds the cluster-LOD DAG of every mesh.
 *   QUADTREE  the generator's regular DAG (81-vertex / 128-triangle grid meshlets, 4x4-meshlet groups).
 *   EXTERNAL  a builder handed in by the caller (brmi_scene_create_with_dag_builder): the generator tessellates each mesh into one
 *             indexed triangle list and maps the DAG it gets back onto pages / groups / BVH.  The tests use this entry to run the
 *             scenes through the reference's own clodBuild (tests/clodref_bridge.py); libbrmi_scene.so itself never loads it.
 *   OWN       this library's cluster-LOD builder (basicrenderer_amd/csrc/scene/lod_builder.cpp, brmi_lod_build below): irregular
 *             meshlets, ~384-cluster groups, QEM simplification with locked group boundaries -- SURVEY.md section 8, row f-1. */
enum brmi_lod_builder { BRMI_LOD_BUILDER_QUADTREE = 0, BRMI_LOD_BUILDER_EXTERNAL = 1, BRMI_LOD_BUILDER_OWN = 2 };

/* A cluster-LOD DAG as flat arrays (what clusterlod.h's clodBuild reports through its callback, BR/include/ThirdParty/meshoptimizer/
 * clusterlod.h:118-160).  group.{center,radius,error} is clodGroup::simplified -- the sphere and error the two rendering rules test
 * (FLT_MAX error = terminal group); cluster.{center,radius} bounds the cluster's own geometry, cluster.error is the error of the
 * group whose simplification produced it, cluster.refined that group's index (-1 = input geometry).  vertexRefs lists, per cluster,
 * the input-mesh vertex of every local vertex; triangles holds 3 local indices per triangle. */
typedef struct brmi_dag_group   { int32_t depth; float center[3], radius, error; uint32_t firstCluster, clusterCount; } brmi_dag_group;
typedef struct brmi_dag_cluster { int32_t group, refined; float center[3], radius, error; uint32_t vertexCount, triangleCount, firstVertex, firstTriangleByte; } brmi_dag_cluster;
typedef struct brmi_dag {
    const brmi_dag_group* groups;     uint32_t groupCount;
    const brmi_dag_cluster* clusters; uint32_t clusterCount;
    const uint32_t* vertexRefs;       uint32_t vertexRefCount;
    const uint8_t* triangles;         uint32_t triangleBytes;
    void* owner;                      /* the builder's bookkeeping; released by its release function */
} brmi_dag;
/* positions: float3 per vertex; indices: 3 per triangle; normals: float3 per vertex or NULL.  Returns 0 on success. */
typedef int  (*brmi_dag_build_fn)(void* user, const float* positions, size_t vertexCount, const uint32_t* indices, size_t indexCount, const float* normals, brmi_dag* out);
typedef void (*brmi_dag_release_fn)(void* user, brmi_dag* dag);

typedef struct brmi_scene_params {
    uint32_t preset;
    uint32_t seed;
    uint32_t width, height;       /* render target size */
    uint32_t numPointLights;
    uint32_t withDirectionalLight;
    uint32_t lodLevels;           /* 0 = preset default; 1 = flat */
    float    sizeScale;           /* 1.0 = preset default triangle budget; <1 shrinks (tests) */
    uint32_t skinnedFraction1024; /* fraction (x/1024) of instances that are skinned; 0 = none */
    uint32_t materialFeatures;    /* bit 0: some materials carry an OpenPBR coat, bit 1: some carry fuzz, bit 2: every third instance is mirrored and drawn with reversed winding,
                                     bit 3: meshes carry a UV set and most materials sample textures (base colour, metallic / roughness, emissive, AO, normal map),
                                     bit 4: every third material is alpha tested against its base-colour / opacity texture (implies bit 3),
                                     bit 5: pages carry RGBA8 vertex colours (CLOD_PAGE_ATTRIBUTE_COLOR) that tint the base colour,
                                     bit 6: coat / fuzz materials (bits 0 / 1) also bind OpenPBR layer textures (needs bit 3),
                                     bit 7: about half of the textured materials (bit 3) carry a height map with MATERIAL_PARALLAX (default: none),
                                     bit 8: pages carry three UV sets and the materials' texture slots (bit 3, with 6 / 7 also the layer and height slots) are spread over them */
    uint32_t cameraStep;          /* frame number on the preset's camera path (0 = start); prevView is the view of step - 1 */
    uint32_t lodBuilder;          /* enum brmi_lod_builder */
    uint32_t spotLightEvery;      /* k > 0: every k-th punctual light is a spot light (0 = point lights only) */
    float    detail;              /* geometric detail: 0 / 1 = the smooth default surfaces; d > 1 adds seven octaves of relief below the base one, each half the
                                     wavelength and half the amplitude of the one before (first amplitude (d - 1) / 8 of the base), so that every LOD level keeps an
                                     error proportional to its edge length and the 1 px error test selects pixel-sized triangles */
    uint32_t uniqueTriangleBudget;/* street presets: 1 = the preset's triangle budget counts the triangles of the meshes (Bistro-class: ~3 M in 150 meshes) and all of
                                     its instances are placed (~2,000: ~19 M instanced triangles, SURVEY.md 8(d) config 3 / row a-3's meshlet counts);
                                     0 = the budget counts instanced triangles and caps the instance count (the frames of rounds 1-2 and the golden fixtures) */
    float    reliefSlope;         /* > 0: seven octaves of relief below every patch's base wavelength, octave k displacing by reliefSlope x its own wavelength
                                     (world units, the same for every patch; `detail` scales each patch's own amplitude instead).  A LOD level's error is then
                                     that fraction of its edge length, and the 1 px error test selects triangles of ~1 / reliefSlope pixels per edge */
} brmi_scene_params;

/* Arrays a scene exposes.  Element layouts are the brmi_types.h structs. */
enum brmi_scene_array {
    BRMI_ARR_PER_OBJECT = 0,        /* brmi_per_object[] */
    BRMI_ARR_NORMAL_MATRICES,       /* float[4][4][]  (Builtin::NormalMatrixBuffer) */
    BRMI_ARR_PER_MESH,              /* brmi_per_mesh[] */
    BRMI_ARR_PER_MESH_INSTANCE,     /* brmi_per_mesh_instance[] */
    BRMI_ARR_CLOD_OFFSETS,          /* brmi_mesh_instance_clod_offsets[] (per mesh instance) */
    BRMI_ARR_CLOD_MESH_METADATA,    /* brmi_clod_mesh_metadata[] */
    BRMI_ARR_LOD_NODES,             /* brmi_lod_node[] */
    BRMI_ARR_LOD_GROUPS,            /* brmi_lod_group[] */
    BRMI_ARR_LOD_SEGMENTS,          /* brmi_lod_segment[] */
    BRMI_ARR_GROUP_PAGE_MAP,        /* brmi_group_page_map_entry[] */
    BRMI_ARR_MATERIALS,             /* brmi_material_info[] */
    BRMI_ARR_OPENPBR_MATERIALS,     /* brmi_openpbr_material_info[] */
    BRMI_ARR_LIGHTS,                /* brmi_light_info[] */
    BRMI_ARR_ACTIVE_LIGHT_INDICES,  /* uint32_t[] */
    BRMI_ARR_CAMERAS,               /* brmi_camera[] */
    BRMI_ARR_CULLING_CAMERAS,       /* brmi_culling_camera[] */
    BRMI_ARR_VIEW_RASTER_INFO,      /* brmi_view_raster_info[] */
    BRMI_ARR_PER_FRAME,             /* brmi_per_frame[1] */
    BRMI_ARR_ACTIVE_DRAWS,          /* uint32_t[] per-mesh-instance indices to cull (draw set) */
    BRMI_ARR_SKINNING_MATRICES,     /* float[4][4][]: bone*invBind per (slot,joint); may be empty */
    BRMI_ARR_LUT_OD_ENERGY,         /* uint16[32][32][32]  opaque-dielectric energy complement (ior, alpha, cos) */
    BRMI_ARR_LUT_OD_AVG_ENERGY,     /* uint16[32][32]      its cosine-weighted average (ior, alpha) */
    BRMI_ARR_LUT_IM_ENERGY,         /* uint16[32][32]      ideal-metal energy complement (alpha, cos) */
    BRMI_ARR_LUT_IM_AVG_ENERGY,     /* uint16[32]          its average (alpha) */
    BRMI_ARR_LUT_FUZZ_LTC,          /* float[32][32][4]    fuzz LTC aInv, bInv, reflectance, 0 (rough, cos) */
    BRMI_ARR_TEXTURE_DESCS,         /* brmi_texture_desc[]; `texels` holds the BYTE OFFSET into BRMI_ARR_TEXELS: add the base after upload */
    BRMI_ARR_TEXELS,                /* uint8[]             RGBA8 texels of every texture, mip chains packed */
    BRMI_ARR_SAMPLER_DESCS,         /* brmi_sampler_desc[] */
    BRMI_ARR_SRGB_TO_LINEAR,        /* float[256] */
    BRMI_ARR_COUNT
};

typedef struct brmi_scene brmi_scene;

brmi_scene* brmi_scene_create(const brmi_scene_params* params);
/* The preset's camera anywhere on its path, without building the scene: what a CameraManager hands the passes between frames.  `step` is a
 * position on the path (cameraStep k of brmi_scene_params is step = k: the same bytes), `prevStep` the position of the frame before (prevView).
 * Only params->preset / width / height are read.  Returns 0, or -1 on a bad argument. */
int brmi_scene_camera_at(const brmi_scene_params* params, double step, double prevStep, brmi_camera* camera, brmi_culling_camera* cullingCamera);
/* lodBuilder = EXTERNAL: `build` is called once per mesh, `release` (may be NULL) once its DAG has been consumed. */
brmi_scene* brmi_scene_create_with_dag_builder(const brmi_scene_params* params, brmi_dag_build_fn build, brmi_dag_release_fn release, void* user);

/* A scene of the CALLER's meshes (row f-1: real content instead of the procedural stand-ins): every mesh goes through the cluster-LOD
 * builder (this library's, or `build` when lodBuilder = EXTERNAL) and the page / BVH packer; materials (params->materialFeatures), lights
 * (one directional + params->numPointLights inside the scene's bounds) and the lookup tables are the procedural ones.  params->preset is
 * ignored.  NULL on bad input (indices out of range, non-finite positions, no triangles, an instance naming a missing mesh). */
typedef struct brmi_mesh_input {
    const float*    positions;      /* vertexCount x 3 */
    const float*    normals;        /* vertexCount x 3, or NULL: area-weighted vertex normals are derived */
    const float*    uvs;            /* vertexCount x 2, or NULL: (0, 0) where the scene's materials want texcoords */
    const uint32_t* colors;         /* vertexCount x RGBA8, or NULL: white where materialFeatures bit 5 asks for vertex colours */
    size_t          vertexCount;
    const uint32_t* indices;        /* triangle list */
    size_t          indexCount;
    uint32_t        material;       /* the mesh's material (materials 0 .. max over the meshes are generated) */
    uint32_t        reserved[3];
} brmi_mesh_input;
typedef struct brmi_instance_input {
    uint32_t mesh, reverseWinding;
    float    model[4][4];           /* row-vector convention, as the reference's PerObjectCB: p' = p * M, translation in row 3 */
} brmi_instance_input;
typedef struct brmi_view_input { float eye[3], yaw, pitch, fovYDegrees, zNear, zFar; } brmi_view_input;   /* yaw about +y, pitch about the camera's x; looking down -z at 0, 0 */
brmi_scene* brmi_scene_create_from_meshes(const brmi_scene_params* params, const brmi_mesh_input* meshes, uint32_t meshCount,
                                          const brmi_instance_input* instances, uint32_t instanceCount, const brmi_view_input* view,
                                          brmi_dag_build_fn build, brmi_dag_release_fn release, void* user);

/* This library's cluster-LOD builder, with the brmi_dag_build_fn / brmi_dag_release_fn signatures (user is ignored).
 * Limits 128 vertices / 128 triangles per cluster, groups of ~384 clusters with <= 8 refined groups each, target ratio 0.5, stuck
 * threshold 0.85, error merge max(1.5 x previous, current): the settings of BR/src/Mesh/ClusterLODUtilities.cpp:5426-5458. */
int  brmi_lod_build(void* user, const float* positions, size_t vertexCount, const uint32_t* indices, size_t indexCount, const float* normals, brmi_dag* out);
void brmi_lod_release(void* user, brmi_dag* dag);

/* CLodCache: the reference's on-disk form of a mesh's cluster-LOD data (BR/src/Import/CLodCache.cpp).
 *   <dir>/mesh_<i>.clodbin   container v4: {magic 'CLOD', version 4, reserved, pageCount}, pageCount locators {u64 offset, u32 size,
 *                            u32 reserved}, the page blobs (SaveContainerPayload :309-374; checked like OpenContainerFile :1000-1020)
 *   <dir>/mesh_<i>.clodmeta  the metadata blob of SerializeMetadata (:171-211, schema 47): groups, segments, segment bounds, object
 *                            sphere, page locators, group -> page references, BVH nodes, per-depth node ranges and roots.
 *                            (The reference keeps this blob in the `clodBlob` attribute of a USD crate file; extracting it needs OpenUSD.)
 * brmi_scene_export_cache writes every mesh of a scene (returns the mesh count, < 0 on error).  brmi_scene_create_from_cache
 * builds the preset's scene but takes each mesh from the cache instead of building it: every cross reference of the loaded data is
 * validated (the kernels index it unchecked); NULL on a missing, truncated or inconsistent file. */
int         brmi_scene_export_cache(const brmi_scene* scene, const char* directory);
brmi_scene* brmi_scene_create_from_cache(const brmi_scene_params* params, const char* directory);
void        brmi_scene_destroy(brmi_scene* scene);

/* Returns 0 on success.  `count` = element count. */
int      brmi_scene_array(const brmi_scene* scene, uint32_t arrayId,
                          const void** ptr, uint64_t* bytes, uint32_t* count);
/* Page slabs: slab index s in [1, slabCount]; index 0 is the reference's "not resident". */
uint32_t brmi_scene_slab_count(const brmi_scene* scene);
int      brmi_scene_slab(const brmi_scene* scene, uint32_t slabIndex, const void** ptr, uint64_t* bytes);

/* Summary numbers for reporting. */
typedef struct brmi_scene_stats {
    uint64_t uniqueTriangles, instancedTriangles;
    uint32_t meshes, instances, meshletsTotal, meshletsLod0, pages, nodes, groups, segments;
    uint32_t lights, materials, maxBvhDepth, lodLevelsMax;
    float    sceneMin[3], sceneMax[3];
} brmi_scene_stats;
void brmi_scene_get_stats(const brmi_scene* scene, brmi_scene_stats* out);

#ifdef __cplusplus
}
#endif
#endif
