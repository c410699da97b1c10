/*
 * brmi.h -- C ABI of libbrmi.so: the MI355X-native visibility-buffer path of BasicRenderer.
 *
 * Drop-in boundary.  In the reference this path lives behind OpenRenderGraph's pass interface
 * (`ComputePass::{DeclareResourceUsages,Setup,Update,Execute,Cleanup}`, e.g.
 * BR/include/Render/GraphExtensions/ClusterLOD/ClusterSoftwareRasterizationPass.h:16-54) and the
 * graph extension that schedules the cull/raster chain
 * (`IRenderGraphExtension`, BR/include/Render/GraphExtensions/CLodExtension.h:20-31;
 * schedule BR/src/Render/GraphExtensions/CLodExtension.cpp:1411-2095).  Those interfaces hand
 * shaders *descriptor indices*; this ABI hands kernels *device pointers* and a hipStream_t in
 * place of the command list.  One `brmi_pass` object = the CLodExtension + VisUtil + light
 * clustering + deferred shading passes of one view.  Failures are int status codes plus
 * brmi_last_error() (the reference throws; BR/src/Renderer.cpp:2112-2122); fixed-capacity
 * appends drop + count and never fault (BR/shaders/ClusterLOD/workGraphCulling.hlsl:3094-3112).
 *
 * No torch / C++ types cross this boundary: plain pointers, sizes and PODs only.
 */
#ifndef BRMI_H
#define BRMI_H

#include <stdint.h>
#include "brmi_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BRMI_ABI_VERSION 1u

typedef enum brmi_status {
    BRMI_OK            = 0,
    BRMI_ERR_INVALID   = -1,   /* bad argument / missing binding */
    BRMI_ERR_HIP       = -2,   /* a HIP call failed; see brmi_last_error() */
    BRMI_ERR_CAPACITY  = -3,   /* a declared resource is too small */
    BRMI_ERR_STATE     = -4    /* phase called out of order (e.g. execute before setup) */
} brmi_status;

typedef struct brmi_pass brmi_pass;
typedef void* brmi_stream;     /* hipStream_t; NULL = the default stream */

/* ---- configuration: the SettingsManager keys this path reads ------------------------------ */
/* defaults: BR/src/Renderer.cpp:1108-1237 ; BR/include/Renderer.h:157,216-221 */
typedef struct brmi_config {
    uint32_t structSize;               /* sizeof(brmi_config), for forward compatibility */
    uint32_t width, height;            /* visibility / G-buffer / HDR target size */
    uint32_t maxVisibleClusters;       /* CLodExtension maxClusters (Renderer.cpp:2494: 30,000,000); at most 2^25 - 1 */
    uint32_t maxTraversalRecords;      /* frontier + bucket capacity (reference shares maxClusters) */
    uint32_t enableOcclusionCulling;   /* 2-phase HZB culling (Renderer.h:157 default true) */
    uint32_t enableClusteredLighting;  /* PSO_CLUSTERED_LIGHTING (default true) */
    uint32_t enablePunctualLights;     /* DeferredShadingPass root constant (default true) */
    uint32_t lightClusterSize[3];      /* {12,12,24}  (Renderer.cpp:1154) */
    uint32_t phase2ExpansionFactor;    /* meshlets per bucket record, CLodCommon.h:31 default 2 */
    uint32_t collectPassStatistics;    /* per-stage hipEvent timing (Renderer.cpp:1133-1134) */
    uint32_t maxBvhLevels;             /* level-loop bound; reference caps at 64 (HierarchicalDispatchCullingPass.cpp:57) */
    uint32_t bandY0, bandY1;           /* rows [bandY0,bandY1) this GPU owns (multi-GPU tile split); 0,0 = all */
    uint32_t keepUniformLayerPlanes;   /* default 0 (every plane is written every frame).  1: when every material of the scene packs to the same coat
                                          (fuzz) G-buffer word and binds no coat / fuzz texture, the plane is filled with that word once after brmi_setup
                                          and brmi_execute skips the per-frame stores to it (16 of 52 B per pixel).  Only for hosts that neither write,
                                          clear nor alias GBUF_COAT / GBUF_FUZZ between frames. */
    /* Interleaved screen partition, round 3.  stripeCount > 1: a frame `fullHeight` rows high is cut into chunks of `stripeRows`
     * rows (a multiple of 16), stripeCount consecutive chunks form a group, and this GPU owns one chunk of every group -- chunk stripeIndex of
     * the even groups, chunk stripeCount - 1 - stripeIndex of the odd ones (back and forth, so that a vertical cost gradient does not favour
     * one GPU) -- and renders them into COMPACT surfaces: `height` is then the number of rows it owns (fullHeight / stripeCount; fullHeight a
     * multiple of stripeRows * stripeCount), and row v of every surface is row (g * stripeCount + slot(g)) * stripeRows + v % stripeRows of the
     * frame, g = v / stripeRows.  Cameras, lights and perFrame.screenResY describe the FULL frame.  bandY0 / bandY1 stay 0.
     * 0 / 1: off (the contiguous band of bandY0 / bandY1, or the whole frame). */
    uint32_t stripeRows, stripeCount, stripeIndex, fullHeight;
    /* Round 6: 1 = the band may change from frame to frame (brmi_set_band; cost-balanced contiguous regions): per-band workspace tables are sized for
     * the whole frame.  bandY0 / bandY1 are then the first frame's band.  Not with the interleaved partition. */
    uint32_t dynamicBand;
    uint32_t reserved[2];
} brmi_config;

void brmi_default_config(brmi_config* cfg, uint32_t width, uint32_t height);

/* ---- scene providers (IResourceProvider keys, BR/include/Renderer.h:284-323) ----------------
 * Device pointers to the structured buffers the reference's managers own.  `slabs` is a device
 * array of device pointers indexed by slabDescriptorIndex (entry 0 unused = "not resident").   */
typedef struct brmi_scene_buffers {
    const uint8_t* const*                  slabs;            uint32_t slabCount;        /* incl. entry 0 */
    const brmi_per_object*                 perObject;        uint32_t perObjectCount;
    const float*                           normalMatrices;   /* float[4][4] per object */
    const brmi_per_mesh*                   perMesh;          uint32_t perMeshCount;
    const brmi_per_mesh_instance*          perMeshInstance;  uint32_t perMeshInstanceCount;
    const brmi_mesh_instance_clod_offsets* clodOffsets;
    const brmi_clod_mesh_metadata*         meshMetadata;     uint32_t meshMetadataCount;
    const brmi_lod_node*                   lodNodes;         uint32_t lodNodeCount;     /* nodes, groups and segments are topology: brmi_set_scene folds them into derived tables; call it again when they change */
    const brmi_lod_group*                  lodGroups;        uint32_t lodGroupCount;
    const brmi_lod_segment*                lodSegments;      uint32_t lodSegmentCount;
    const brmi_group_page_map_entry*       groupPageMap;     uint32_t groupPageMapCount;
    const brmi_material_info*              materials;        uint32_t materialCount;
    const brmi_openpbr_material_info*      openpbrMaterials; uint32_t openpbrMaterialCount;   /* at most 65536: the shading pass keeps 66 KB of folded table rows per record in the workspace (brmi_set_scene refuses more) */
    const brmi_light_info*                 lights;           uint32_t lightCount;
    const uint32_t*                        activeLightIndices;
    const brmi_camera*                     cameras;          uint32_t cameraCount;
    const brmi_culling_camera*             cullingCameras;
    const brmi_view_raster_info*           viewRasterInfo;
    const brmi_per_frame*                  perFrame;
    const uint32_t*                        activeDraws;      uint32_t activeDrawCount;  /* per-mesh-instance indices */
    const float*                           skinningMatrices; uint32_t skinningMatrixCount; /* float[4][4]: bone*invBind */
    /* OpenPBR lookup tables (BR/src/Render/OpenPBRLookupResources.cpp:34-77): R16_UNORM energy
     * tables 32x32(x32) and the 32x32 float LTC table, injected by the caller. */
    const uint16_t* lutOpaqueDielectricEnergyComplement;     /* [32 ior][32 alpha][32 cos] */
    const uint16_t* lutOpaqueDielectricAvgEnergyComplement;  /* [32 ior][32 alpha] */
    const uint16_t* lutIdealMetalEnergyComplement;           /* [32 alpha][32 cos] */
    const uint16_t* lutIdealMetalAvgEnergyComplement;        /* [32 alpha] */
    const float*    lutFuzzLTC;                               /* [32 rough][32 cos][4]: aInv,bInv,refl,0 */
    /* material textures (the descriptor-heap slots MaterialInfo indexes); all may be null / 0 for a scene of
     * constant-factor materials.  srgbToLinear: float[256], the sRGB decode of an 8-bit code (injected like the LUTs:
     * pow() differs between math libraries). */
    const brmi_texture_desc* textures;     uint32_t textureCount;
    const brmi_sampler_desc* samplers;     uint32_t samplerCount;
    const float*             srgbToLinear;
} brmi_scene_buffers;

/* ---- graph resources the pass declares (DeclareResourceUsages) ---------------------------- */
typedef enum brmi_resource_id {
    BRMI_RES_VISIBILITY = 0,          /* Builtin::PrimaryCamera::VisibilityTexture  u64/px, tiled 8x8 */
    BRMI_RES_LINEAR_DEPTH,            /* Builtin::PrimaryCamera::LinearDepthMap     f32/px, tiled */
    BRMI_RES_GBUF_NORMALS,            /* Builtin::GBuffer::Normals            float4  */
    BRMI_RES_GBUF_ALBEDO,             /* Builtin::GBuffer::Albedo             rgba8 unorm */
    BRMI_RES_GBUF_COAT,               /* Builtin::GBuffer::Coat               rgba16f */
    BRMI_RES_GBUF_EMISSIVE,           /* Builtin::GBuffer::Emissive           rgba16f */
    BRMI_RES_GBUF_FUZZ,               /* Builtin::GBuffer::Fuzz               rgba16f */
    BRMI_RES_GBUF_METALLIC_ROUGHNESS, /* Builtin::GBuffer::MetallicRoughness  rgba8 unorm */
    BRMI_RES_GBUF_MOTION_VECTORS,     /* Builtin::GBuffer::MotionVectors      rg16f */
    BRMI_RES_HDR_COLOR,               /* Builtin::Color::HDRColorTarget       rgba16f; like the G-buffer planes, pixels without geometry are NOT
                                         written (DeferredCSMain returns, BR/shaders/deferred.hlsl:33-37): the graph clears / the sky pass fills them */
    BRMI_RES_VISIBLE_CLUSTERS,        /* CLod visible-cluster buffer, 16 B records */
    BRMI_RES_LIGHT_CLUSTERS,          /* Builtin::Light::ClusterBuffer */
    BRMI_RES_LIGHT_PAGES,             /* Builtin::Light::PagesBuffer */
    BRMI_RES_HZB,                     /* linear-depth mip chain for occlusion culling */
    BRMI_RES_WORKSPACE,               /* frontiers, bucket records, bitmasks, counters, statistics */
    BRMI_RES_COUNT
} brmi_resource_id;

typedef enum brmi_usage {
    BRMI_USAGE_SHADER_RESOURCE  = 1u << 0,
    BRMI_USAGE_UNORDERED_ACCESS = 1u << 1,
    BRMI_USAGE_INTERNAL         = 1u << 2     /* no consumer outside this pass; may be aliased */
} brmi_usage;

typedef struct brmi_resource_desc {
    uint32_t    id;             /* brmi_resource_id */
    const char* name;           /* the reference's Builtin:: key this resource stands for */
    uint64_t    bytes;          /* required size */
    uint32_t    usage;          /* brmi_usage mask */
    uint32_t    width, height;  /* logical size in pixels (0 for buffers) */
    uint32_t    bytesPerPixel;
    uint32_t    tileW, tileH;   /* storage tiling (8x8, column-major inside a tile); pixel (x,y) lives at
                                   ((y/tileH)*tilesX + x/tileW)*tileW*tileH + (x%tileW)*tileH + y%tileH */
} brmi_resource_desc;

typedef void (*brmi_declare_cb)(void* user, const brmi_resource_desc* desc);

typedef struct brmi_resource_binding { uint32_t id; void* ptr; uint64_t bytes; } brmi_resource_binding;

/* per-frame host data (UpdateExecutionContext) */
typedef struct brmi_frame_update {
    const brmi_camera*    mainCameraHost;   /* host copy of cameras[perFrame.mainCameraIndex] */
    const brmi_per_frame* perFrameHost;     /* host copy of the per-frame constant buffer */
    uint32_t              frameIndex;
} brmi_frame_update;

/* counters read back after a frame (the reference's GPU telemetry, CLodTelemetry.h) */
typedef struct brmi_counters {
    uint32_t instancesTested, instancesVisible;
    uint32_t nodesVisited, bucketRecords;
    uint32_t meshletsTested, visibleClusters, visibleClustersPhase2;
    uint32_t droppedRecords, droppedClusters;
    uint32_t lightPagesUsed;
    uint32_t replayNodes, replayMeshlets;   /* occluded in phase 1, re-tested in phase 2 */
    /* Statistics of the rasteriser (no counterpart in CLodTelemetry.h).  [0] / [2]: vertex / triangle count sums of the visible clusters, both phases (32 bits
     * each since round 5: brmi_create refuses a cluster capacity whose sums could exceed them).  [4]: clusters handed to the rasteriser stage; [5]: raster records
     * that found their screen bin full.  Round 6, the draw list: [1] = clusters of the phase-1 visible list the culling HELD BACK as probably
     * hidden, [3] = how many of those the re-test could not prove hidden and the late pass drew; [1] - [3] clusters of the list were never rasterised, and the
     * image is the one of rasterising all of them.  Both 0 on frames that draw the list in order (no previous depth chain, multi-GPU partitions, hold_clusters=0). */
    uint32_t reserved[6];
} brmi_counters;

enum brmi_stage {
    BRMI_STAGE_CLEAR = 0, BRMI_STAGE_CULL, BRMI_STAGE_RASTER, BRMI_STAGE_DEPTH_COPY, BRMI_STAGE_HZB,
    BRMI_STAGE_CULL2, BRMI_STAGE_RASTER2, BRMI_STAGE_GBUFFER, BRMI_STAGE_LIGHT_CLUSTER, BRMI_STAGE_SHADE,
    BRMI_STAGE_COUNT
};

/* ---- lifecycle (ComputePass phases) -------------------------------------------------------- */
uint32_t    brmi_abi_version(void);
int         brmi_create(const brmi_config* cfg, brmi_pass** out);
int         brmi_declare(brmi_pass* pass, brmi_declare_cb cb, void* user);            /* DeclareResourceUsages */
int         brmi_set_scene(brmi_pass* pass, const brmi_scene_buffers* scene);          /* provider resolution   */
/* Setup also reads the resident pages once (round 6): the object-space box of every meshlet goes into a side table of the workspace (the draw list's tests, DESIGN.md 4.3c;
 * brmi_set_scene walked the page map for the table's size).  A host that rewrites page CONTENTS afterwards (streaming) calls brmi_set_scene + brmi_setup again; with
 * BRMI_TUNING=hold_clusters=0 the table is neither sized nor made. */
int         brmi_setup(brmi_pass* pass, const brmi_resource_binding* b, uint32_t n, brmi_stream stream); /* Setup */
int         brmi_update(brmi_pass* pass, const brmi_frame_update* upd, brmi_stream stream);               /* Update */
int         brmi_execute(brmi_pass* pass, brmi_stream stream);                          /* Execute: whole chain */
/* The same frame on two streams (the reference's graph owns a graphics, a compute and a copy queue and passes state a preference: `QueueKind`, `PreferQueue`; DeviceManager.cpp:178): culling, rasterisation
 * and the depth chain on `geometryStream`, G-buffer, light lists (when the culling launches did not carry them) and shading on
 * `shadingStream`; events order the two halves and the pass's next frame.  Outputs are ready when `shadingStream` is.  With
 * brmi_set_history_source and two passes alternating frames on the same geometry stream (the shading stream may be shared or one per
 * pass), frame k+1's geometry half -- latency-bound launches that leave most of the chip idle -- runs beside frame k's shading half;
 * give `geometryStream` the higher priority.
 * Per-frame inputs with frames in flight: the resolve + shading half reads the camera and the per-frame record from a copy the frame's
 * first launch makes in the pass's workspace, so the caller may rewrite `cameras`, `cullingCameras`, `viewRasterInfo` and `perFrame` for the
 * pass's NEXT frame as soon as this call has returned -- ON `geometryStream`, before the brmi_update of that frame (stream order then puts
 * the write behind this frame's geometry half and in front of the next frame's).  Passes that render frames in turn must not share these
 * buffers (each pass binds its own: brmi_set_scene), and buffers every frame reads but none owns (lights, materials, objects, geometry)
 * may only change while no frame that reads them is in flight.
 * The stage entry points that start a frame (brmi_clear_visibility, brmi_cull(1)) issue the same waits as this call, so a graph that runs
 * the stages itself after a split frame is ordered behind that frame's shading half. */
int         brmi_execute_split(brmi_pass* pass, brmi_stream geometryStream, brmi_stream shadingStream);
void        brmi_destroy(brmi_pass* pass);                                               /* Cleanup */
const char* brmi_last_error(const brmi_pass* pass);

/* Rows [bandY0, bandY1) of the frame this GPU renders FROM THE NEXT FRAME ON (multiples of 8; passes created with brmi_config::dynamicBand): call it between frames,
 * before the frame's brmi_update.  The screen-tile split of SURVEY.md 8(e) with regions whose boundaries follow the cost of the frames before, so that every GPU takes
 * the same time: a cluster is set up by the one GPU whose band holds it (two at a boundary), and the band test of the instance / node / cluster culling drops the rest
 * of the hierarchy.  Launches already enqueued keep the band they were issued with.  The depth chain's texels of rows that leave the band are reset to "empty" by the
 * next chain build, so a stale depth never occludes anything in phase 2. */
int brmi_set_band(brmi_pass* pass, uint32_t bandY0, uint32_t bandY1);

/* ---- stage-level entry points: one per compute pass the reference's graph schedules ---------
 * (order of BR/src/Render/GraphExtensions/CLodExtension.cpp:1580-2088 and
 *  BR/include/Render/RenderGraphBuildHelper.h:220-414) */
int brmi_clear_visibility(brmi_pass* pass, brmi_stream stream);   /* ClearVisibilityBufferPass */
int brmi_cull(brmi_pass* pass, uint32_t phase, brmi_stream stream);       /* HierarchicalCullingPass1/2 (K1-K3) */
int brmi_raster(brmi_pass* pass, uint32_t phase, brmi_stream stream);     /* SoftwareRasterizeClustersPass1/2 (K5) */
int brmi_depth_copy(brmi_pass* pass, brmi_stream stream);         /* LinearDepthCopyPass (K6) */
int brmi_build_hzb(brmi_pass* pass, brmi_stream stream);          /* LinearDepthDownsamplePass (SPD max-reduce; BR/shaders/downsample.hlsl) */
/* Drops the previous frame's depth chain (camera cut, resize): the next phase 1 runs without occlusion tests. */
int brmi_invalidate_hzb(brmi_pass* pass);
/* Frames in flight (the reference's `numFramesInFlight`, CLodStreamingSystem.cpp:956): two passes with their own resources render
 * alternate frames on two streams; phase 1 of a pass then tests against the depth chain the OTHER pass built for the frame before
 * (`source`), not against its own, which is two frames old.  Both passes must have the same size and band; for two frames in
 * flight each is the other's source.  brmi_execute orders the streams itself: a frame starts when the source's chain of the frame before is complete
 * (one event wait), everything after the source's chain build -- its G-buffer and shading -- overlaps this pass's culling and
 * rasterisation.  The images are those of one pass rendering the same frames in order.  NULL unlinks. */
int brmi_set_history_source(brmi_pass* pass, brmi_pass* source);
int brmi_gbuffer(brmi_pass* pass, brmi_stream stream);            /* MaterialHistogram..EvaluateMaterialGroups (K7,K8) */
int brmi_light_clustering(brmi_pass* pass, brmi_stream stream);   /* ClusterGenerationPass + LightCullingPass (K9,K10) */
int brmi_shade(brmi_pass* pass, brmi_stream stream);              /* DeferredShadingPass (K11) */
/* Shading in row slabs (multi-GPU composition overlapped with the frame's own shading): with slabs > 1 the deferred-shading launches of
 * brmi_shade / brmi_execute / brmi_execute_split cover the band in that many slabs of rows (multiples of 8 surface rows, top to bottom), and
 * `fn(user, row0, row1, stream)` is called on the host right after a slab's launches have been enqueued on `stream` -- the place to hand the rows
 * to brmi_compose_submit_rows (include/brmi_compose.h), whose stores then travel while the next slab is shaded.  Same pixels, same bytes as
 * one launch over the band.  slabs <= 1 or fn == NULL: one launch, no call. */
typedef void (*brmi_slab_fn)(void* user, uint32_t row0, uint32_t row1, brmi_stream stream);
int brmi_set_shade_slabs(brmi_pass* pass, uint32_t slabs, brmi_slab_fn fn, void* user);

/* ---- introspection ------------------------------------------------------------------------- */
int brmi_read_counters(brmi_pass* pass, brmi_counters* out, brmi_stream stream);   /* synchronises */
/* mean milliseconds per stage AND FRAME over the frames executed since the previous call (at most the last 32; the depth-chain stage, which
 * brmi_execute runs twice per frame, reports the sum of its two builds;
 * HIP events on the execute stream, collectPassStatistics); synchronises and resets the window */
int brmi_stage_times(brmi_pass* pass, float* msOut /* [BRMI_STAGE_COUNT] */);
/* Restricts the event pairs to the stages whose bit (1 << brmi_stage) is set; every event pair is a barrier on the stream,
 * so a benchmark times all stages once and then only the one it reports on.  Default: all stages. */
int brmi_set_timed_stages(brmi_pass* pass, uint32_t stageMask);
/* algorithmic bytes of the last frame per SURVEY.md 8(d): 140*P + sum(144+12V+3T) + 64*M + 16*Mvis + 64*N.  A read-back call: waits for
 * the device (whatever stream the frame ran on) before it reads the frame's counters. */
int brmi_algorithmic_bytes(brmi_pass* pass, uint64_t* perStage /* [BRMI_STAGE_COUNT] */, uint64_t* total);
/* The same count for the kernel variants the frame actually LAUNCHED: where a scene lets a stage move less than SURVEY.md 8(d)'s figure, this says how much it is
 * obliged to move -- today one case: no material of the scene has a coat or a fuzz layer, so the shading pass does not read those two G-buffer planes (44 instead of
 * 60 B per pixel).  brmi_algorithmic_bytes stays 8(d)'s definition; a roofline fraction against THIS count is the honest one for the launched kernel. */
int brmi_algorithmic_bytes_launched(brmi_pass* pass, uint64_t* perStage /* [BRMI_STAGE_COUNT] */, uint64_t* total);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* Evaluates the library's fp32 primitives on device data so a test can check the arithmetic contract
 * (correctly rounded a/b and sqrt(a), round-to-nearest-even float->half) against IEEE on the host. */
int brmi_debug_arith(const float* a, const float* b, float* outDiv, float* outSqrt, uint32_t* outHalfBits, uint32_t n, brmi_stream stream);

/* Triangles of the last frame whose bin records were many enough (more than BRMI_TUNING wide_entries = 128 bins; 16 for the triangles the lean rasteriser queues,
 * lean_wide_entries) to be handed to the cooperative emission pass (k_raster_wide): phase 1's draw pass, its late pass, phase 2.  Counted whether or not the pass
 * was launched (the host launches it while the frames before had such triangles).  Waits for the device. */
int brmi_debug_wide_triangles(brmi_pass* pass, uint32_t out[3]);
/* Round 6, the lean rasteriser: out[0] = 1 when the last frame's phase-1 main launch was the lean form of k_raster (frames of very many clusters;
 * BRMI_TUNING lean_min_clusters), out[1] = how many of its clusters it left to the general launch behind it (skinned vertices, a full triangle queue), out[2] / out[3] =
 * triangles it queued for k_raster_emit (large enough for the bins) and the runs they were queued in (one per wave and pass; requested, so beyond a full queue's
 * capacity).  Waits for the device. */
int brmi_debug_lean_clusters(brmi_pass* pass, uint32_t out[4]);
/* One of the 64 stripes of that queue as the last frame left it: counts = {entries, runs} (clamped to the stripe's capacity); runs = {first entry, count} pairs,
 * entries = 96 B records (brmi_raster.hip, WideTri: a 64 B bin record of the triangle's first row, then yHi, band0, band1, strip0, strip1).  Either buffer may be null. */
int brmi_debug_read_lean_queue(brmi_pass* pass, uint32_t stripe, uint32_t* runs, uint32_t maxRuns, void* entries, uint32_t maxEntries, uint32_t counts[2]);

/* The last frame's draw-list decisions, for tests (waits for the device): indices into the visible-cluster buffer of the phase-1 clusters the culling held back
 * (`held`, up to heldCapacity entries; *heldCount = how many there were) and of those the late pass drew (`late`).  held minus late was never rasterised. */
int brmi_debug_read_held(brmi_pass* pass, uint32_t* held, uint32_t heldCapacity, uint32_t* heldCount, uint32_t* late, uint32_t lateCapacity, uint32_t* lateCount);

/* Experiments only: the first `bytes` of the raster bin-record region of the workspace (instrumented builds park per-workgroup time stamps there). */
int brmi_debug_read_bin_records(brmi_pass* pass, void* dst, uint64_t bytes);

/* The shading pass's in-range forms of 1 / a, sqrt(a) and 1 / sqrt(a) (brmi_device.h: the IEEE expansions without their scaling prologue and
 * fix-up epilogue for 2^-63 <= a < 2^63, the general form elsewhere): a test compares them bit for bit with the host's IEEE results. */
int brmi_debug_arith_in_range(const float* a, float* outRcp, float* outSqrt, float* outRsqrt, uint32_t n, brmi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* BRMI_H */
