// brmi_api.hip -- the C ABI of libbrmi.so (include/brmi.h): pass lifecycle and stage scheduling.
//
// Host side of the drop-in.  Mirrors the five phases of the reference's ComputePass /
// IRenderGraphExtension (DeclareResourceUsages / Setup / Update / Execute / Cleanup) and the order
// in which CLodExtension + RenderGraphBuildHelper schedule the chain
// (BR/src/Render/GraphExtensions/CLodExtension.cpp:1580-2088, BR/include/Render/RenderGraphBuildHelper.h:220-414).
// The library never allocates device memory: the graph (caller) owns every resource, exactly as
// in the reference; kernels are enqueued on the caller's stream and nothing here synchronises
// except the explicit read-back calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "brmi_device.h"
#include "brmi_internal.h"

namespace brmi {

__global__ void k_debug_arith(const float* a, const float* b, float* outDiv, float* outSqrt, uint32_t* outHalf, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    outDiv[i] = a[i] / b[i];
    outSqrt[i] = sqrtf(fabsf(a[i]));
    outHalf[i] = f32_to_f16_bits(a[i]);
}

__global__ void k_debug_arith_in_range(const float* a, float* outRcp, float* outSqrt, float* outNorm, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    outRcp[i] = in_range_pow63(x) ? rcp_rn_in_range(x) : 1.0f / x;
    outSqrt[i] = in_range_pow63(x) ? sqrt_rn_in_range(x) : sqrtf(x);
    float len; const f3 v = normalize3_len(f3{x, 1.0f, 0.5f}, x, len);      // the guarded pair on the same operand
    outNorm[i] = v.y;                                                        // = 1 / sqrt(x)
}

int fail(brmi_pass* p, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (p) p->err = buf;
    return code;
}

static uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

static const char* kResNames[BRMI_RES_COUNT] = {
    "Builtin::PrimaryCamera::VisibilityTexture", "Builtin::PrimaryCamera::LinearDepthMap", "Builtin::GBuffer::Normals", "Builtin::GBuffer::Albedo",
    "Builtin::GBuffer::Coat", "Builtin::GBuffer::Emissive", "Builtin::GBuffer::Fuzz", "Builtin::GBuffer::MetallicRoughness", "Builtin::GBuffer::MotionVectors",
    "Builtin::Color::HDRColorTarget", "Builtin::CLod::VisibleClusters", "Builtin::Light::ClusterBuffer", "Builtin::Light::PagesBuffer",
    "Builtin::PrimaryCamera::LinearDepthMap(mips)", "brmi::Workspace"};
static const uint32_t kResBpp[BRMI_RES_COUNT] = {8, 4, 16, 4, 8, 8, 8, 4, 4, 8, 0, 0, 0, 4, 0};

static void compute_sizes(brmi_pass* p) {
    const brmi_config& c = p->cfg;
    p->tilesX = (c.width + 7) / 8; p->tilesY = (c.height + 7) / 8;
    p->paddedPixels = (uint64_t)p->tilesX * p->tilesY * 64;
    p->bandY0 = c.bandY0; p->bandY1 = (c.bandY1 == 0 || c.bandY1 > c.height) ? c.height : c.bandY1;
    if (p->bandY0 >= p->bandY1) p->bandY0 = 0;
    p->stripes = c.stripeCount > 1u ? StripeMap{c.stripeRows, c.stripeCount, c.stripeIndex, c.fullHeight} : StripeMap{0u, 0u, 0u, c.height};
    { const uint32_t t0 = p->bandY0 / 8, t1 = (p->bandY1 + 7) / 8; p->bandFirstPixel = (uint64_t)t0 * p->tilesX * 64; p->bandPixelCount = (uint64_t)(t1 - t0) * p->tilesX * 64; }
    p->numLightClusters = c.lightClusterSize[0] * c.lightClusterSize[1] * c.lightClusterSize[2];
    p->lightPagePool = p->numLightClusters * BRMI_LIGHT_PAGES_PER_CLUSTER;
    if (const long v = tuning("light_page_pool", 0)) p->lightPagePool = (uint32_t)std::max(1l, v);   // tests: exhaust the page pool
    for (int i = 0; i < BRMI_RES_COUNT; i++) p->resNeed[i] = 0;
    for (int i = BRMI_RES_VISIBILITY; i <= BRMI_RES_HDR_COLOR; i++) p->resNeed[i] = p->paddedPixels * kResBpp[i];
    p->resNeed[BRMI_RES_VISIBLE_CLUSTERS] = (uint64_t)c.maxVisibleClusters * 16;
    p->resNeed[BRMI_RES_LIGHT_CLUSTERS] = (uint64_t)p->numLightClusters * sizeof(brmi_light_cluster);
    p->resNeed[BRMI_RES_LIGHT_PAGES] = (uint64_t)p->lightPagePool * sizeof(brmi_light_page);
    // HZB: mip chain of the power-of-two padded linear depth (built only with occlusion culling)
    p->hzbMipOffsets.clear(); p->hzbMipW.clear(); p->hzbMipH.clear();
    uint64_t hzbFloats = 0;
    if (c.enableOcclusionCulling) {
        uint32_t w = 1, h = 1; while (w < c.width) w <<= 1; while (h < c.height) h <<= 1;
        for (uint32_t mip = 0;; mip++) {      // mip 0 is the depth map itself: no storage
            p->hzbMipOffsets.push_back(hzbFloats); p->hzbMipW.push_back(w); p->hzbMipH.push_back(h);
            if (mip > 0) hzbFloats += (uint64_t)w * h;
            if (w == 1 && h == 1) break;
            w = std::max(1u, w >> 1); h = std::max(1u, h >> 1);
        }
    }
    p->hzbMipCount = (uint32_t)p->hzbMipOffsets.size();
    p->resNeed[BRMI_RES_HZB] = std::max<uint64_t>(16, hzbFloats * 4);
    // workspace carve-up
    Workspace& w = p->ws; uint64_t off = 0;
    auto take = [&](uint64_t bytes) { uint64_t o = off; off = align_up(off + bytes, 256); return o; };
    // counters and both bitmasks are adjacent: one fill clears them at the start of a frame
    w.counters = take((uint64_t)(CNT_WORDS + 64) * 4);
    w.bitmask1 = take((uint64_t)p->totalWords * 4);
    w.bitmask2 = take((uint64_t)p->totalWords * 4);
    w.blockDirty = take((uint64_t)2 * (p->scanBlocks + 1));      // per phase and 2048-word block of the bitmask: some survivor set a bit there (the ranking skips the others)
    // phase 2's footprint for the second depth-chain build: a byte per 32 x 32 px block of the chain's head, from byte 4 (byte 0: "everything", stored by a triangle of more than 16 blocks)
    w.chainDirty = take(4ull + (uint64_t)((c.width + 31u) / 32u) * ((c.height + 31u) / 32u));
    w.frameClearBytes = off - w.counters;
    w.frontierA = take((uint64_t)c.maxTraversalRecords * sizeof(NodeRecord));
    w.frontierB = take((uint64_t)c.maxTraversalRecords * sizeof(NodeRecord));
    w.buckets = take((uint64_t)c.maxTraversalRecords * sizeof(BucketRecord));
    w.tempVisible = take((uint64_t)c.maxVisibleClusters * sizeof(TempVisible));
    w.wordPrefix = take((uint64_t)p->totalWords * 4);
    w.blockSums = take((uint64_t)(p->scanBlocks + 1) * 4);
    w.scanAgg = take(65 * 8);
    w.instanceBitBase = take((uint64_t)std::max<size_t>(1, p->hostInstanceBitBase.size()) * 4);
    w.segPrefix = take((uint64_t)std::max<size_t>(1, p->hostSegPrefix.size()) * 4);
    w.meshLevelWidth = take((uint64_t)std::max<size_t>(1, p->hostMeshLevelWidth.size()) * 4);
    w.flatNodes = take((uint64_t)std::max<size_t>(1, p->hostFlatNodes.size()) * sizeof(FlatNode));
    w.flatLeaves = take((uint64_t)std::max<size_t>(1, p->hostFlatLeaves.size()) * sizeof(FlatLeaf));
    w.instanceWalk = take((uint64_t)std::max<size_t>(1, p->hostInstanceWalk.size()) * sizeof(InstanceWalk));
    w.planes = take((uint64_t)2 * c.lightClusterSize[2] * 4);
    // phase-1 -> phase-2 hand-over (reference: clodStructs.hlsli:629-652); the replay bucket array doubles as the
    // phase-2 bucket array, so it has the full record capacity
    w.replayNodes = take(c.enableOcclusionCulling ? (uint64_t)c.maxTraversalRecords * sizeof(NodeRecord) : 16);
    w.replayBuckets = take(c.enableOcclusionCulling ? (uint64_t)c.maxTraversalRecords * sizeof(BucketRecord) : 16);
    w.lightVS = take((uint64_t)std::max(1u, p->scene.lightCount) * 16);
    w.lightMeta = take((uint64_t)std::max(1u, p->scene.lightCount) * 4);
    w.clusterPages = take((uint64_t)p->numLightClusters * 4);
    w.clusterHits = take((uint64_t)p->numLightClusters * 4);
    w.pageTotal = take(16);
    w.lightHitMasks = take((uint64_t)p->numLightClusters * ((std::max(1u, p->scene.lightCount) + 63u) / 64u) * 8);
    p->binsX = (c.width + 255) / 256; p->binsY = (c.height + 15) / 16;
    w.binCounts = take((uint64_t)p->binsX * p->binsY * 4 * BRMI_BIN_COUNT_STRIDE);
    w.binRecords = take((uint64_t)p->binsX * p->binsY * p->binCapacity * 64);
    w.binOverflow = take((uint64_t)CNT_STRIPE_COUNT * p->binOverflowPerStripe * 64);
    // the plan of a k_raster_bins launch: header, three words per bin, the (bin, slice) items; scratch tiles of 32 KB for the slices of bins
    // that several workgroups walk (2048 tiles = 64 MB: a 4K frame whose every bin is full would want 16 k; bins beyond merge with atomics)
    p->binItemCapacity = p->binsX * p->binsY * std::max(1u, (p->binCapacity + 31u) / 32u > 256u ? 256u : (p->binCapacity + 31u) / 32u);
    p->binItemCapacity = std::min<uint32_t>(p->binItemCapacity, 1u << 22);
    p->binScratchTiles = (uint32_t)std::max(0l, tuning("bin_scratch_tiles", p->binScratchTiles));
    w.binPlan = take((uint64_t)(16 + 3 * p->binsX * p->binsY) * 4);
    w.binItems = take((uint64_t)p->binItemCapacity * 4);
    w.binScratch = take((uint64_t)std::max(1u, p->binScratchTiles) * 4096 * 8);
    w.debugStamps = take(4096 + 1024 * 1024);      // instrumented builds (-DBRMI_TILE_STAMPS, possibly of one translation unit only) park per-phase cycle sums here
    w.clusterSetup = take((uint64_t)c.maxVisibleClusters * sizeof(ClusterSetup));
    // resolve arena: full tables (72 B per vertex + triangle slot) for up to 2^20 clusters = 9.7 GB of the 288; a configuration
    // that allows more visible clusters keeps the per-pixel path for the clusters that do not fit
    p->resolveCapacity = (uint32_t)std::min<uint64_t>((uint64_t)c.maxVisibleClusters, 1ull << 20) * BRMI_MESHLET_MAX_TRIS;
    if (const long v = tuning("resolve_capacity", 0)) p->resolveCapacity = (uint32_t)std::max(1l, v);   // tests: force the per-pixel fallback
    w.resolveVerts = take((uint64_t)p->resolveCapacity * sizeof(ResolveVertex));
    w.resolveTris = take((uint64_t)p->resolveCapacity * sizeof(ResolveTriangle));
    w.shadeTables = take(((uint64_t)2 * c.width + 2 * c.height + 64) * 4);
    w.matWords = take((uint64_t)std::max(1u, p->scene.materialCount) * sizeof(MaterialWords));
    w.layerUniform = take(sizeof(LayerUniform));
    w.frameConst = take(4 * 64);      // three matrix products and (round 6) the band's two planes
    w.frameSnapshot = take(sizeof(FrameSnapshot));
    w.matConst = take((uint64_t)std::max(1u, p->scene.openpbrMaterialCount) * sizeof(MatConst));
    w.objConst = take((uint64_t)std::max(1u, p->scene.perObjectCount) * OBJ_CONST_FLOATS * 4);
    // (a band that may change from frame to frame: sized for the whole frame)
    p->deferredStripeCapacity = (uint32_t)((((c.dynamicBand ? p->paddedPixels : p->bandPixelCount) / 4096 + CNT_STRIPE_COUNT) / CNT_STRIPE_COUNT) * 4096);   // 64-tile runs of a stripe x 4096 pixels
    w.deferredPixels = take((uint64_t)3 * CNT_STRIPE_COUNT * p->deferredStripeCapacity * 4);      // one set of striped lists per layered class
    w.lutF = take((uint64_t)(32768 + 1024 + 1024 + 32 + 256) * 4);
    w.shadeRows = take((uint64_t)std::max(1u, p->scene.openpbrMaterialCount) * 256 * 512);    // (OpenPBR material, roughness code) -> folded energy-table rows of the shading pass
    w.shadeAvgs = take((uint64_t)std::max(1u, p->scene.openpbrMaterialCount) * 256 * 8);
    w.ggxQuads = take(256 * 48);                                                                  // roughness code -> the GGX albedo fit as quadratics in N.V
    w.shadeLights = take((uint64_t)std::max(1u, p->scene.lightCount) * 64);                     // the shading pass's 64 B record per active light
    w.clusterList = take((uint64_t)p->numLightClusters * 8);                                    // per light cluster: first entry / length of its flat light list
    w.listEntries = take((uint64_t)p->lightPagePool * BRMI_LIGHTS_PER_PAGE * 4 + 256);
    w.listRecords = take((uint64_t)p->lightPagePool * BRMI_LIGHTS_PER_PAGE * 64 + 4096);      // ... and the lights' 64 B shading records in the same order        // the page contents once more, in the order the page walk visits them
    // textured / alpha-tested scenes only: where each visible cluster's UV set lives, the texcoords of the resolve arena's vertices,
    // and the alpha-test operands that travel with binned triangles
    const bool uvs = p->sceneHasTextures || p->sceneHasAlphaTest || p->sceneHasVertexColors;
    w.clusterUv = take(uvs ? (uint64_t)c.maxVisibleClusters * 32 : 16);
    w.resolveColors = take(p->sceneHasVertexColors ? (uint64_t)p->resolveCapacity * 4 : 16);
    w.resolveUVs = take(p->sceneHasTextures ? (uint64_t)p->resolveCapacity * 8 * p->sceneUvSets : 16);
    w.binAlpha = take(p->sceneHasAlphaTest ? (uint64_t)p->binsX * p->binsY * p->binCapacity * 48 : 16);
    w.overflowAlpha = take(p->sceneHasAlphaTest ? (uint64_t)CNT_STRIPE_COUNT * p->binOverflowPerStripe * 48 : 16);
    w.alphaMats = take(p->sceneHasAlphaTest ? (uint64_t)std::max(1u, p->scene.materialCount) * 128 : 16);
    // round 6, the draw list: per-meshlet boxes of the resident pages, the (slab, page) -> first box table, and the three lists of a frame
    w.meshletBoxes = take((uint64_t)std::max(1u, p->totalBoxes) * sizeof(MeshletBox));
    w.pageBoxBase = take((uint64_t)std::max<size_t>(1, p->hostPageBoxBase.size()) * 4);
    w.pageRefs = take((uint64_t)std::max<size_t>(1, p->hostPageRefs.size()) * sizeof(PageRef));
    w.drawList = take(p->holdEnabled ? (uint64_t)c.maxVisibleClusters * 4 : 16);
    w.generalList = take(p->leanMinClusters != 0u ? (uint64_t)c.maxVisibleClusters * 4 : 16);
    w.bigQueue = take(p->leanMinClusters != 0u ? (uint64_t)p->leanQueue * 96 : 16);      // WideTri
    w.bigRuns = take(p->leanMinClusters != 0u ? (uint64_t)p->leanQueue * 8 : 16);
    w.heldRecords = take(p->holdEnabled ? (uint64_t)c.maxVisibleClusters * sizeof(HeldRecord) : 16);
    w.lateList = take(p->holdEnabled ? (uint64_t)c.maxVisibleClusters * 4 : 16);
    w.wideQueue = take((uint64_t)std::max(1u, p->wideCapacity) * 96);      // WideTri (brmi_raster.hip)
    w.wideAlpha = take(p->sceneHasAlphaTest ? (uint64_t)std::max(1u, p->wideCapacity) * 48 : 16);
    w.total = off;
    p->resNeed[BRMI_RES_WORKSPACE] = w.total;
}

template <typename T>
static int read_back(brmi_pass* p, std::vector<T>& dst, const T* src, size_t n) {
    dst.resize(n);
    if (n == 0) return BRMI_OK;
    if (!src) return fail(p, BRMI_ERR_INVALID, "scene buffer missing");
    BRMI_HIP(p, hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return BRMI_OK;
}

}  // namespace brmi

brmi::HzbDesc brmi_pass::hzbDesc() const {
    brmi::HzbDesc d{};
    d.depth = static_cast<const float*>(res[BRMI_RES_LINEAR_DEPTH]); d.mips = static_cast<float*>(res[BRMI_RES_HZB]);
    d.width = cfg.width; d.height = cfg.height; d.tilesX = tilesX; d.mipCount = hzbMipCount; d.rowLo = bandY0; d.rowHi = bandY1; d.stripes = stripes;
    d.paddedW = hzbMipCount ? hzbMipW[0] : 1; d.paddedH = hzbMipCount ? hzbMipH[0] : 1;
    for (uint32_t i = 0; i < brmi::kMaxHzbMips; i++) d.mipOffset[i] = i < hzbMipCount ? (uint32_t)hzbMipOffsets[i] : 0u;
    return d;
}

using namespace brmi;

extern "C" {

uint32_t brmi_abi_version(void) { return BRMI_ABI_VERSION; }

void brmi_default_config(brmi_config* cfg, uint32_t width, uint32_t height) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->structSize = sizeof(brmi_config);
    cfg->width = width; cfg->height = height;
    cfg->maxVisibleClusters = 1u << 22;       // reference: 30,000,000 (Renderer.cpp:2494); callers size it to the scene
    cfg->maxTraversalRecords = 1u << 22;
    cfg->enableOcclusionCulling = 0;          // opt-in: 2-phase HZB occlusion culling (needs a previous frame to pay off)
    cfg->enableClusteredLighting = 1;
    cfg->enablePunctualLights = 1;
    cfg->lightClusterSize[0] = 12; cfg->lightClusterSize[1] = 12; cfg->lightClusterSize[2] = 24;
    cfg->phase2ExpansionFactor = 2;
    cfg->collectPassStatistics = 0;
    cfg->maxBvhLevels = 64;
    cfg->keepUniformLayerPlanes = 0;          // opt-in: every plane is written every frame unless the host says the planes are its to keep
}

}  // extern "C"
namespace brmi {
// BRMI_TUNING="key=value,key=value": looked up on every call (a test sets it around the creation of one pass; nothing here is on a frame's path)
static bool tuning_lookup(const char* key, long* out) {
    const char* e = std::getenv("BRMI_TUNING");
    if (!e) return false;
    const size_t n = std::strlen(key);
    for (const char* q = e; *q; ) {
        while (*q == ',' || *q == ' ') q++;
        const char* end = std::strchr(q, ',');
        const size_t len = end ? (size_t)(end - q) : std::strlen(q);
        if (len > n && std::strncmp(q, key, n) == 0 && q[n] == '=') { *out = std::strtol(q + n + 1, nullptr, 0); return true; }
        q += len;
    }
    return false;
}
long tuning(const char* key, long def) { long v; return tuning_lookup(key, &v) ? v : def; }
long experiment(const char* key, long def) {
#ifdef BRMI_EXPERIMENTS
    long v; return tuning_lookup(key, &v) ? v : def;
#else
    (void)key; return def;
#endif
}
}  // namespace brmi
extern "C" {

int brmi_create(const brmi_config* cfg, brmi_pass** out) {
    if (!cfg || !out || cfg->structSize != sizeof(brmi_config) || cfg->width == 0 || cfg->height == 0) return BRMI_ERR_INVALID;
    // (<= 2^25 - 1 clusters: the vertex and the triangle count sums of a frame's list share one 64-bit counter, 128 per cluster at most -- brmi_internal.h, CNT_SUM_VERTS_LO; the
    // reference's 30,000,000, Renderer.cpp:2494, fits)
    if (cfg->maxVisibleClusters == 0 || cfg->maxVisibleClusters > (1u << 25) - 1u || cfg->maxTraversalRecords == 0) return BRMI_ERR_INVALID;
    if (cfg->lightClusterSize[0] == 0 || cfg->lightClusterSize[1] == 0 || cfg->lightClusterSize[2] == 0) return BRMI_ERR_INVALID;
    if (cfg->lightClusterSize[2] > 62u) return BRMI_ERR_INVALID;
    if ((uint64_t)((cfg->width + 255u) / 256u) * ((cfg->height + 15u) / 16u) > 65535ull) return BRMI_ERR_INVALID;      // the raster bins' work items name a bin in 16 bits (16384 x 16368 px still fits)
    if (cfg->stripeCount > 1u) {      // interleaved partition: whole chunks of 16-row bin bands, the same number on every GPU, no band on top
        if (cfg->stripeRows == 0u || cfg->stripeRows % 16u || cfg->stripeIndex >= cfg->stripeCount || cfg->bandY0 != 0u || (cfg->bandY1 != 0u && cfg->bandY1 != cfg->height)) return BRMI_ERR_INVALID;
        if (cfg->fullHeight == 0u || cfg->fullHeight % (cfg->stripeRows * cfg->stripeCount) || cfg->height != cfg->fullHeight / cfg->stripeCount) return BRMI_ERR_INVALID;
        if (cfg->dynamicBand) return BRMI_ERR_INVALID;
    }
    if (cfg->dynamicBand && (cfg->bandY0 % 8u || (cfg->bandY1 % 8u && cfg->bandY1 < cfg->height))) return BRMI_ERR_INVALID;      // the slice-start table (workspace and the shading pass's LDS copy) holds 64 entries: gz + 2
    brmi_pass* p = new brmi_pass();
    p->cfg = *cfg;
    p->totalWords = 1; p->scanBlocks = 1;
    // sizes and switches a test (or a user) may set: BRMI_TUNING="key=value,..." 
    p->forceLevelKernels = tuning("cull_level_kernels", 0) != 0;
    p->packedFlat = tuning("flat_packed", 1) != 0;
    p->flatLevelsMinDraws = (uint32_t)std::max(0l, tuning("flat_levels_min_draws", p->flatLevelsMinDraws));      // (tests: 1 = the level-synchronous flat traversal for every scene)
    p->phase2DirectMax = (uint32_t)std::max(0l, tuning("phase2_direct_max", p->phase2DirectMax));
    p->resolveInlineMode = (int)tuning("resolve_inline", -1);
    p->wideCapacity = (uint32_t)std::min(1l << 20, std::max(0l, tuning("wide_capacity", p->wideCapacity)));
    p->wideMinTriangles = (uint32_t)std::max(0l, tuning("wide_min_triangles", p->wideMinTriangles));
    p->leanMinClusters = (uint32_t)std::max(0l, tuning("lean_min_clusters", p->leanMinClusters));
    if (p->leanMinClusters > cfg->maxVisibleClusters) p->leanMinClusters = 0u;      // a pass whose list cannot get that long never runs the lean rasteriser: no queue in its workspace (104 MB)
    p->leanMaxGeneralPct = (uint32_t)std::max(0l, std::min(100l, tuning("lean_max_general_pct", p->leanMaxGeneralPct)));
    p->leanGrid = (uint32_t)std::max(64l, tuning("lean_grid", p->leanGrid));
    p->leanQueue = (uint32_t)std::max(64l, std::min(1l << 24, tuning("lean_queue", p->leanQueue))) & ~63u;      // (64 stripes)
    p->leanWideEntries = (uint32_t)std::max(1l, tuning("lean_wide_entries", p->leanWideEntries));
    p->leanEmitGrid = (uint32_t)std::max(64l, std::min(65536l, tuning("lean_emit_grid", p->leanEmitGrid))) & ~63u;
    p->wideEntries = (uint32_t)std::max(1l, tuning("wide_entries", p->wideEntries));
    p->binMinSlice = (uint32_t)std::max(32l, tuning("bin_min_slice", p->binMinSlice));
    p->binSharedSlice = (uint32_t)std::max(32l, tuning("bin_shared_slice", p->binSharedSlice));
    p->binGrid = (uint32_t)std::min(65535l, std::max(1l, tuning("bin_grid", p->binGrid)));
    p->binOverflowPerStripe = (uint32_t)std::max(0l, tuning("bin_overflow", p->binOverflowPerStripe));
    p->binCapacity = (uint32_t)std::min(65536l, std::max(1l, tuning("bin_capacity", p->binCapacity)));   // 16-bit record indices inside a bin slice's alpha list; 65536 x 64 B x bins is far beyond any frame
    if (const long v = tuning("big_tri_area", 0)) p->bigTriArea = p->bigTriAreaAlpha = p->bigTriAreaDense = (int)std::max(1l, v);
    // A/B switches of the experiment logs: only in builds with -DBRMI_EXPERIMENTS
    p->rasterGrid = (uint32_t)std::max(64l, experiment("raster_grid", p->rasterGrid));
    p->shadeGridShared = (uint32_t)std::min(65535l, std::max(256l, experiment("shade_grid_shared", p->shadeGridShared)));
    p->gbufferGridShared = (uint32_t)std::min(65535l, std::max(256l, experiment("gbuffer_grid_shared", p->gbufferGridShared)));
    p->clearRiderBlocks = (uint32_t)std::min(65535l, std::max(64l, experiment("clear_rider_blocks", p->clearRiderBlocks)));
    p->wideFlat = experiment("flat_wide", 1) != 0;
    p->scanChained = experiment("scan_chained", 1) != 0;
    p->spillWidth = (uint32_t)std::min(1024l, std::max(128l, experiment("spill_width", p->spillWidth)));
    p->rasterDebug = (int)experiment("raster_debug", 0);
    if (const long v = experiment("big_tri_area_alpha", 0)) p->bigTriAreaAlpha = (int)std::max(1l, v);
    compute_sizes(p);
    *out = p;
    return BRMI_OK;
}

void brmi_destroy(brmi_pass* p) {
    if (!p) return;
    if (p->eventsCreated) for (int i = 0; i < BRMI_STAGE_COUNT; i++) for (uint32_t k = 0; k < brmi_pass::kEventRing; k++) { (void)hipEventDestroy(p->evStart[i][k]); (void)hipEventDestroy(p->evStop[i][k]); }
    (void)brmi_set_history_source(p, nullptr);
    for (brmi_pass* user : p->historyUsers) user->history = nullptr;
    if (p->chainReady) (void)hipEventDestroy(p->chainReady);
    if (p->geometryDone) (void)hipEventDestroy(p->geometryDone);
    if (p->cullDone) (void)hipEventDestroy(p->cullDone);
    if (p->frameDone) (void)hipEventDestroy(p->frameDone);
    if (p->phase2FeedbackHost) (void)hipHostFree(p->phase2FeedbackHost);
    delete p;
}

const char* brmi_last_error(const brmi_pass* p) { return p ? p->err.c_str() : "null pass"; }

// Provider resolution.  Also derives the static rank tables of the deterministic visible-cluster
// compaction (per-instance bit base, per-segment meshlet prefix) and the BVH level bound.
int brmi_set_scene(brmi_pass* p, const brmi_scene_buffers* scene) {
    if (!p || !scene) return BRMI_ERR_INVALID;
    p->scene = *scene; p->haveScene = false; p->setupDone = false;
    const brmi_scene_buffers& sc = p->scene;
    if (!sc.slabs || !sc.perObject || !sc.perMesh || !sc.perMeshInstance || !sc.clodOffsets || !sc.meshMetadata || !sc.lodNodes || !sc.lodGroups ||
        !sc.lodSegments || !sc.groupPageMap || !sc.materials || !sc.openpbrMaterials || !sc.cameras || !sc.cullingCameras || !sc.viewRasterInfo || !sc.perFrame ||
        !sc.normalMatrices || (sc.activeDrawCount && !sc.activeDraws) || (sc.lightCount && (!sc.lights || !sc.activeLightIndices)))
        return fail(p, BRMI_ERR_INVALID, "brmi_set_scene: a required scene buffer is null");
    if (!sc.lutOpaqueDielectricEnergyComplement || !sc.lutOpaqueDielectricAvgEnergyComplement || !sc.lutIdealMetalEnergyComplement || !sc.lutIdealMetalAvgEnergyComplement || !sc.lutFuzzLTC)
        return fail(p, BRMI_ERR_INVALID, "brmi_set_scene: OpenPBR lookup tables must be provided");
    std::vector<brmi_clod_mesh_metadata> md; std::vector<brmi_mesh_instance_clod_offsets> offs; std::vector<brmi_lod_node> nodes; std::vector<brmi_lod_segment> segs;
    int rc;
    if ((rc = read_back(p, md, sc.meshMetadata, sc.meshMetadataCount))) return rc;
    if ((rc = read_back(p, offs, sc.clodOffsets, sc.perMeshInstanceCount))) return rc;
    if ((rc = read_back(p, nodes, sc.lodNodes, sc.lodNodeCount))) return rc;
    if ((rc = read_back(p, segs, sc.lodSegments, sc.lodSegmentCount))) return rc;
    {   // which layered shading variants the scene can need (materials edited on the device later: call brmi_set_scene again)
        std::vector<brmi_openpbr_material_info> op;
        if ((rc = read_back(p, op, sc.openpbrMaterials, sc.openpbrMaterialCount))) return rc;
        p->sceneHasCoat = p->sceneHasFuzz = false;
        bool layerTextures = false;
        // UV sets the G-buffer pass must decode: 1 + the highest set an enabled texture slot names (AppendClodMaterialUvSample: an index >= 8 reads set 0)
        uint32_t uvSets = 1;
        auto names_set = [&](uint32_t setIndex) { if (setIndex < 8u && setIndex + 1u > uvSets) uvSets = setIndex + 1u; };
        for (size_t i = 0; i < op.size(); i++) {
            const auto& m = op[i];
            if (m.coatWeight > 0.0f) p->sceneHasCoat = true;
            if (m.fuzzWeight > 0.0f) p->sceneHasFuzz = true;
            // HasOpenPBRTexture (utilities.hlsli:643-646): a coat / fuzz slot with a valid texture AND sampler index is sampled
            for (int k = 0; k < 6; k++)
                if (m.textureBindings[2 * k] != 0xFFFFFFFFu && m.textureBindings[2 * k + 1] != 0xFFFFFFFFu) {
                    layerTextures = true;
                    names_set(m.textureBindings[26 + k]);
                }
        }
        // texture slots: the alpha-test variants of the rasteriser and the texture-sampling variant of the G-buffer pass are only
        // launched for scenes that need them
        if (sc.openpbrMaterialCount > 65536u) return fail(p, BRMI_ERR_CAPACITY, "brmi_set_scene: %u OpenPBR material records; the shading pass folds 66 KB of table rows per record and takes at most 65536", sc.openpbrMaterialCount);
        std::vector<brmi_material_info> mats;
        if ((rc = read_back(p, mats, sc.materials, sc.materialCount))) return rc;
        p->sceneHasAlphaTest = false; p->sceneHasTextures = layerTextures; p->sceneHasParallax = false;
        for (size_t i = 0; i < mats.size(); i++) {
            const brmi_material_info& m = mats[i];
            if (m.materialFlags & BRMI_MATERIAL_PARALLAX) names_set(m.heightUvSetIndex);
            if (m.materialFlags & BRMI_MATERIAL_ALPHA_TEST) p->sceneHasAlphaTest = true;
            if (m.materialFlags & BRMI_MATERIAL_PARALLAX) p->sceneHasParallax = true;
            if (m.materialFlags & BRMI_MATERIAL_ANY_TEXTURE) {
                p->sceneHasTextures = true;
                const uint32_t f = m.materialFlags;
                if (f & BRMI_MATERIAL_BASE_COLOR_TEXTURE) names_set(m.baseColorUvSetIndex);
                if (f & BRMI_MATERIAL_NORMAL_MAP) names_set(m.normalUvSetIndex);
                if (f & BRMI_MATERIAL_METALLIC_TEXTURE) names_set(m.metallicUvSetIndex);
                if (f & BRMI_MATERIAL_ROUGHNESS_TEXTURE) names_set(m.roughnessUvSetIndex);
                if (f & BRMI_MATERIAL_EMISSIVE_TEXTURE) names_set(m.emissiveUvSetIndex);
                if (f & BRMI_MATERIAL_AO_TEXTURE) names_set(m.aoUvSetIndex);
                // the opacity slot only feeds the alpha channel, which the G-buffer does not store; the rasteriser's alpha test reads UV set 0 as the reference's does
            }
        }
        p->sceneUvSets = uvSets;
        std::vector<brmi_per_mesh> pms;
        if ((rc = read_back(p, pms, sc.perMesh, sc.perMeshCount))) return rc;
        p->sceneHasVertexColors = false;
        for (const auto& pm : pms) if (pm.vertexFlags & 1u) p->sceneHasVertexColors = true;       // VERTEX_COLORS (BR/include/Mesh/VertexFlags.h)
        if (p->sceneHasTextures && (!sc.textures || !sc.samplers || !sc.srgbToLinear || sc.textureCount == 0 || sc.samplerCount == 0))
            return fail(p, BRMI_ERR_INVALID, "brmi_set_scene: materials sample textures but the texture / sampler tables or the sRGB decode table are missing");
    }
    {   // Cross references the kernels follow unchecked: validated once here, on the host copies (a bad index is an error message, not a
        // GPU fault).  Page contents (headers, descriptors, streams inside the slabs) are NOT walked: they are the builder's contract.
        std::vector<brmi_per_mesh_instance> insts; std::vector<uint32_t> draws; std::vector<brmi_per_mesh> pms; std::vector<brmi_material_info> mats;
        std::vector<brmi_group_page_map_entry> pmap; std::vector<brmi_lod_group> groups; std::vector<brmi_per_object> objs;
        if ((rc = read_back(p, insts, sc.perMeshInstance, sc.perMeshInstanceCount))) return rc;
        if ((rc = read_back(p, draws, sc.activeDraws, sc.activeDrawCount))) return rc;
        if ((rc = read_back(p, pms, sc.perMesh, sc.perMeshCount))) return rc;
        if ((rc = read_back(p, mats, sc.materials, sc.materialCount))) return rc;
        if ((rc = read_back(p, pmap, sc.groupPageMap, sc.groupPageMapCount))) return rc;
        if ((rc = read_back(p, groups, sc.lodGroups, sc.lodGroupCount))) return rc;
        if ((rc = read_back(p, objs, sc.perObject, sc.perObjectCount))) return rc;
        for (size_t i = 0; i < draws.size(); i++) if (draws[i] >= insts.size()) return fail(p, BRMI_ERR_INVALID, "activeDraws[%zu] = %u: no such mesh instance", i, draws[i]);
        for (size_t i = 0; i < insts.size(); i++) {
            if (insts[i].perMeshBufferIndex >= pms.size()) return fail(p, BRMI_ERR_INVALID, "mesh instance %zu: perMeshBufferIndex %u out of range", i, insts[i].perMeshBufferIndex);
            if (insts[i].perObjectBufferIndex >= sc.perObjectCount) return fail(p, BRMI_ERR_INVALID, "mesh instance %zu: perObjectBufferIndex %u out of range", i, insts[i].perObjectBufferIndex);
            if ((pms[insts[i].perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) && insts[i].skinningInstanceSlot != 0xFFFFFFFFu &&
                ((uint64_t)insts[i].skinningInstanceSlot + 1u) * 64u > sc.skinningMatrixCount)
                return fail(p, BRMI_ERR_INVALID, "mesh instance %zu: skinning slot %u lies outside the skinning matrix buffer", i, insts[i].skinningInstanceSlot);
        }
        for (size_t i = 0; i < pms.size(); i++) if (pms[i].materialDataIndex >= mats.size()) return fail(p, BRMI_ERR_INVALID, "mesh %zu: materialDataIndex %u out of range", i, pms[i].materialDataIndex);
        for (size_t i = 0; i < mats.size(); i++) if (mats[i].openPBRMaterialDataIndex >= sc.openpbrMaterialCount) return fail(p, BRMI_ERR_INVALID, "material %zu: openPBRMaterialDataIndex %u out of range", i, mats[i].openPBRMaterialDataIndex);
        for (size_t i = 0; i < pmap.size(); i++) {
            if (pmap[i].slabDescriptorIndex >= sc.slabCount) return fail(p, BRMI_ERR_INVALID, "page map entry %zu: slab %u out of range", i, pmap[i].slabDescriptorIndex);
            if (pmap[i].slabByteOffset % BRMI_PAGE_SIZE) return fail(p, BRMI_ERR_INVALID, "page map entry %zu: offset %u is not a page boundary", i, pmap[i].slabByteOffset);
        }
        for (size_t m = 0; m < md.size(); m++) if (md[m].groupsBase > groups.size() || md[m].segmentsBase > segs.size() || md[m].pageMapBase > pmap.size()) return fail(p, BRMI_ERR_INVALID, "mesh metadata %zu: base index out of range", m);
        for (size_t i = 0; i < offs.size(); i++) if (offs[i].clodMeshMetadataIndex >= md.size()) return fail(p, BRMI_ERR_INVALID, "instance %zu: bad mesh metadata index", i);
        // segments -> pages and refined groups, relative to the mesh that owns them (a mesh's segments end where the next mesh's begin)
        std::vector<uint32_t> order(md.size());
        for (size_t m = 0; m < md.size(); m++) order[m] = (uint32_t)m;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return md[a].segmentsBase < md[b].segmentsBase; });
        for (size_t k = 0; k < order.size(); k++) {
            const brmi_clod_mesh_metadata& mm = md[order[k]];
            const size_t segEnd = k + 1 < order.size() ? md[order[k + 1]].segmentsBase : segs.size();
            for (size_t si = mm.segmentsBase; si < segEnd; si++) {
                if ((uint64_t)mm.pageMapBase + segs[si].pageIndex >= pmap.size()) return fail(p, BRMI_ERR_INVALID, "segment %zu: page %u lies outside the group page map", si, segs[si].pageIndex);
                if (segs[si].refinedGroup >= 0 && (uint64_t)mm.groupsBase + (uint32_t)segs[si].refinedGroup >= groups.size()) return fail(p, BRMI_ERR_INVALID, "segment %zu: refined group %d out of range", si, segs[si].refinedGroup);
            }
        }
    }
    // per mesh: walk the BVH, collect the segments its leaves reference, depth of the tree
    p->hostSegPrefix.assign(segs.size(), 0);
    std::vector<uint32_t> meshBits(md.size(), 0);
    uint32_t maxDepth = 1;
    p->maxLevelWidth = 1; p->minLevelWidth = 0xFFFFFFFFu; p->hostMeshLevelWidth.assign(md.size(), 1u);
    p->spillLevels = 0;
    std::vector<uint32_t> levelWidth;
    std::vector<std::pair<uint32_t, uint32_t>> stack;
    for (size_t m = 0; m < md.size(); m++) {
        uint32_t segCount = 0;
        levelWidth.assign(66, 0);
        stack.clear(); stack.push_back({md[m].rootNode, 1});
        while (!stack.empty()) {
            auto [n, d] = stack.back(); stack.pop_back();
            if ((uint64_t)md[m].lodNodesBase + n >= nodes.size()) return fail(p, BRMI_ERR_INVALID, "mesh %zu: node %u out of range", m, n);
            const brmi_lod_node& nd = nodes[md[m].lodNodesBase + n];
            maxDepth = std::max(maxDepth, d);
            if (d < levelWidth.size()) p->hostMeshLevelWidth[m] = std::max(p->hostMeshLevelWidth[m], ++levelWidth[d]);
            if (nd.isLeaf != BRMI_NODE_INTERNAL) {
                // a leaf names its group and (countMinusOne - 1) the group that refines it: the traversal reads both records unchecked
                if ((uint64_t)md[m].groupsBase + nd.ownerGroupId >= sc.lodGroupCount) return fail(p, BRMI_ERR_INVALID, "mesh %zu: leaf node %u names group %u, which does not exist", m, n, nd.ownerGroupId);
                if (nd.countMinusOne != 0u && (uint64_t)md[m].groupsBase + (nd.countMinusOne - 1u) >= sc.lodGroupCount) return fail(p, BRMI_ERR_INVALID, "mesh %zu: leaf node %u names refined group %u, which does not exist", m, n, nd.countMinusOne - 1u);
                segCount = std::max(segCount, nd.indexOrOffset + 1); continue;
            }
            if (d > 64) return fail(p, BRMI_ERR_INVALID, "mesh %zu: BVH deeper than 64 levels", m);
            const uint32_t cc = std::min(nd.countMinusOne + 1u, BRMI_BVH_MAX_CHILDREN);
            for (uint32_t k = 0; k < cc; k++) stack.push_back({nd.indexOrOffset + k, d + 1});
        }
        uint64_t run = 0;
        for (uint32_t s = 0; s < segCount; s++) {
            const size_t gi = (size_t)md[m].segmentsBase + s;
            if (gi >= segs.size()) return fail(p, BRMI_ERR_INVALID, "mesh %zu: segment %u out of range", m, s);
            p->hostSegPrefix[gi] = (uint32_t)run; run += segs[gi].meshletCount;
        }
        if (run > 0xFFFFFFFFull) return fail(p, BRMI_ERR_CAPACITY, "mesh %zu has too many meshlets", m);
        meshBits[m] = (uint32_t)run;
        p->maxLevelWidth = std::max(p->maxLevelWidth, p->hostMeshLevelWidth[m]); p->minLevelWidth = std::min(p->minLevelWidth, p->hostMeshLevelWidth[m]);
        if (p->hostMeshLevelWidth[m] > p->spillWidth) {
            // a mesh too wide for the LDS walk: the walk hands its frontier to the level kernels at the first level that holds more than 128 nodes
            // (brmi_cull.hip, spill mode) -- never above the first level that CAN hold that many; what is left below bounds the level launches
            uint32_t depth = 1, first = 0;
            for (uint32_t d = 1; d < levelWidth.size(); d++) { if (levelWidth[d]) depth = d; if (!first && levelWidth[d] > 128u) first = d; }
            if (first) p->spillLevels = std::max(p->spillLevels, depth - first + 1u);
        }
    }
    p->hostInstanceBitBase.assign(offs.size(), 0);
    uint64_t bits = 0;
    for (size_t i = 0; i < offs.size(); i++) {
        if (offs[i].clodMeshMetadataIndex >= md.size()) return fail(p, BRMI_ERR_INVALID, "instance %zu: bad mesh metadata index", i);
        p->hostInstanceBitBase[i] = (uint32_t)bits; bits += meshBits[offs[i].clodMeshMetadataIndex];
        if (bits > 0xFFFFFFF0ull) return fail(p, BRMI_ERR_CAPACITY, "scene exceeds 2^32 (instance, meshlet) pairs");
    }
    {   // flat traversal tables: the BVH of a mesh with at most 8192 nodes, breadth-first, with what its leaves' groups and segments say folded in
        std::vector<brmi_lod_group> groups; std::vector<brmi_per_mesh_instance> insts; std::vector<brmi_per_mesh> pms;
        int rc2;
        if ((rc2 = read_back(p, groups, sc.lodGroups, sc.lodGroupCount))) return rc2;
        if ((rc2 = read_back(p, insts, sc.perMeshInstance, sc.perMeshInstanceCount))) return rc2;
        if ((rc2 = read_back(p, pms, sc.perMesh, sc.perMeshCount))) return rc2;
        p->hostFlatNodes.clear(); p->hostFlatLeaves.clear(); p->flatMaxDepth = 1;
        std::vector<uint32_t> flatBase(md.size(), 0), flatCount(md.size(), 0);
        const bool flatOn = tuning("flat_traversal", 1) != 0;
        std::vector<std::pair<uint32_t, uint32_t>> bfs;      // (node id, parent position)
        for (size_t m = 0; flatOn && m < md.size(); m++) {
            bfs.clear(); bfs.push_back({md[m].rootNode, 0u});
            bool fits = true;
            for (size_t k = 0; k < bfs.size() && fits; k++) {
                const brmi_lod_node& nd = nodes[md[m].lodNodesBase + bfs[k].first];
                if (nd.isLeaf != BRMI_NODE_INTERNAL) continue;
                const uint32_t cc = std::min(nd.countMinusOne + 1u, BRMI_BVH_MAX_CHILDREN);
                for (uint32_t c = 0; c < cc; c++) { bfs.push_back({nd.indexOrOffset + c, (uint32_t)k}); if (bfs.size() > 8192) { fits = false; break; } }
            }
            if (!fits) continue;
            flatBase[m] = (uint32_t)p->hostFlatNodes.size(); flatCount[m] = (uint32_t)bfs.size();
            // (breadth-first: the children of a node are pushed together, so they sit side by side; position of the first and depth of every node)
            std::vector<uint32_t> firstChild(bfs.size(), 0u), childCount(bfs.size(), 0u), depthOf(bfs.size(), 1u);
            uint32_t meshDepth = 1u;
            for (size_t k = 1; k < bfs.size(); k++) { const uint32_t par = bfs[k].second; if (childCount[par]++ == 0u) firstChild[par] = (uint32_t)k; depthOf[k] = depthOf[par] + 1u; meshDepth = std::max(meshDepth, depthOf[k]); }
            // the level-synchronous traversal keeps a frontier size per level in counters[CNT_FRONTIER0 + level] (96 words up to the stripe statistics) and launches a kernel per
            // level: a hierarchy deeper than that, or than the configured level bound, is left to the level walk
            if (meshDepth > std::min<uint32_t>(std::max(1u, p->cfg.maxBvhLevels), CNT_STRIPES - CNT_FRONTIER0 - 1u)) { flatCount[m] = 0; continue; }
            p->flatMaxDepth = std::max(p->flatMaxDepth, meshDepth);
            for (size_t k = 0; k < bfs.size(); k++) {
                const brmi_lod_node& nd = nodes[md[m].lodNodesBase + bfs[k].first];
                FlatNode f{}; FlatLeaf l{};
                f.children = firstChild[k] | (childCount[k] << 16);
                std::memcpy(f.cull, nd.cullCenterAndRadius, 16); std::memcpy(f.lod, nd.lodCenterAndRadius, 16); f.maxQuadricError = nd.maxQuadricError;
                f.nodeId = bfs[k].first; f.info = bfs[k].second << 8;
                if (nd.isLeaf == BRMI_NODE_INTERNAL) f.info |= 1u;
                else {
                    const brmi_lod_group& g = groups[md[m].groupsBase + nd.ownerGroupId];
                    std::memcpy(l.group, g.centerAndRadius, 16);
                    if (nd.countMinusOne != 0u) { const brmi_lod_group& cg = groups[md[m].groupsBase + (nd.countMinusOne - 1u)]; std::memcpy(l.child, cg.centerAndRadius, 16); l.childParentError = cg.maxParentError; f.info |= 1u << 1; }
                    const size_t si = (size_t)md[m].segmentsBase + nd.indexOrOffset;
                    if (si >= segs.size()) return fail(p, BRMI_ERR_INVALID, "mesh %zu: leaf node %u names segment %u, which does not exist", m, bfs[k].first, nd.indexOrOffset);
                    if (segs[si].meshletCount != 0u) f.info |= 1u << 2;
                    if (segs[si].meshletCount > 0xFFFFu || segs[si].firstMeshletInPage > 0xFFFFu) { flatCount[m] = 0; break; }      // (the bucket record packs both in 16 bits; leave such a mesh to the level walk)
                    f.ownerGroup = nd.ownerGroupId; f.segFirstCount = segs[si].firstMeshletInPage | (segs[si].meshletCount << 16);
                    f.pageMapIndex = md[m].pageMapBase + segs[si].pageIndex; f.firstBitRel = p->hostSegPrefix[si];
                }
                p->hostFlatNodes.push_back(f); p->hostFlatLeaves.push_back(l);
            }
            if (flatCount[m] == 0) { p->hostFlatNodes.resize(flatBase[m]); p->hostFlatLeaves.resize(flatBase[m]); }
        }
        p->anyWideFlat = false; p->allMeshesFlat = flatOn;
        for (size_t m = 0; m < md.size(); m++) { if (flatCount[m] > 256u) p->anyWideFlat = true; if (flatCount[m] == 0u) p->allMeshesFlat = false; }
        p->hostInstanceWalk.assign(offs.size(), InstanceWalk{0, 0, 0, 0});
        for (size_t i = 0; i < offs.size() && i < insts.size(); i++) {
            const uint32_t m = offs[i].clodMeshMetadataIndex;
            const bool skinned = insts[i].perMeshBufferIndex < pms.size() && (pms[insts[i].perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) != 0;
            p->hostInstanceWalk[i] = InstanceWalk{flatBase[m], flatCount[m], p->hostInstanceBitBase[i], skinned ? 1u : 0u};
        }
    }
    {   // Round 6, the draw list: every resident page of the page map, and where its meshlets' boxes start in the side table (k_meshlet_boxes fills it in brmi_setup).
        // A page's meshlet count is the first word of its header (the builder's contract; one small read-back per page, not per frame).
        std::vector<brmi_group_page_map_entry> pmap;
        if ((rc = read_back(p, pmap, sc.groupPageMap, sc.groupPageMapCount))) return rc;
        std::vector<const uint8_t*> slabPtrs;
        if ((rc = read_back(p, slabPtrs, reinterpret_cast<const uint8_t* const*>(sc.slabs), sc.slabCount))) return rc;
        p->hostPageRefs.clear(); p->totalBoxes = 0;
        p->hostPageBoxBase.assign((size_t)std::max(1u, sc.slabCount) * 1024u, 0xFFFFFFFFu);
        const bool boxesOn = tuning("hold_clusters", 1) != 0 && p->cfg.enableOcclusionCulling && sc.slabCount <= 4096u;
        for (size_t i = 0; boxesOn && i < pmap.size(); i++) {
            const uint32_t slab = pmap[i].slabDescriptorIndex, page = pmap[i].slabByteOffset / BRMI_PAGE_SIZE;
            if (slab == 0u || page >= 1024u || !slabPtrs[slab]) continue;      // not resident (or beyond the 10 page bits of the packed cluster: such a page is never named)
            uint32_t& base = p->hostPageBoxBase[(size_t)slab * 1024u + page];
            if (base != 0xFFFFFFFFu) continue;                                  // several groups share a page
            uint32_t meshlets = 0;
            BRMI_HIP(p, hipMemcpy(&meshlets, slabPtrs[slab] + pmap[i].slabByteOffset, 4, hipMemcpyDeviceToHost));
            meshlets = std::min(meshlets, (BRMI_PAGE_SIZE - 64u) / 64u);
            if ((uint64_t)p->totalBoxes + meshlets > 0x7FFFFFFFull) break;
            base = p->totalBoxes;
            p->hostPageRefs.push_back(PageRef{slab, pmap[i].slabByteOffset, base, meshlets});
            p->totalBoxes += meshlets;
        }
        // (the interleaved partition keeps the whole list: its compact surfaces hold this GPU's chunks only -- the re-test would have to map rows.  A contiguous band
        // lives in frame rows: the tests clamp their rectangles to it, and the chain's texels beyond it read "empty")
        p->holdEnabled = boxesOn && p->totalBoxes != 0u && p->stripes.count <= 1u;
        p->holdMinClusters = (uint32_t)std::max(0l, tuning("hold_min_clusters", p->holdMinClusters));
        p->lateDirectMax = (uint32_t)std::max(0l, tuning("late_direct_max", p->lateDirectMax));
        p->holdStillMax = (uint32_t)std::max(0l, tuning("hold_still_max", p->holdStillMax));
        p->holdFloor = (uint32_t)std::max(0l, tuning("hold_floor", p->holdFloor));
        p->holdMaxTexels = (uint32_t)std::min(16l, std::max(1l, tuning("hold_max_texels", p->holdMaxTexels)));
        p->retestMaxTexels = (uint32_t)std::min(16l, std::max(1l, tuning("retest_max_texels", p->retestMaxTexels)));
    }
    p->totalBits = bits;
    p->totalWords = (uint32_t)std::max<uint64_t>(1, (bits + 31) / 32);
    p->scanBlocks = (p->totalWords + 2047u) / 2048u;
    p->maxLevels = std::min(std::max(1u, maxDepth), std::max(1u, p->cfg.maxBvhLevels));
    compute_sizes(p);
    p->haveScene = true;
    p->updateSerial++;
    return BRMI_OK;
}

int brmi_set_band(brmi_pass* p, uint32_t bandY0, uint32_t bandY1) {
    if (!p) return BRMI_ERR_INVALID;
    if (!p->cfg.dynamicBand) return fail(p, BRMI_ERR_STATE, "brmi_set_band: the pass was created without brmi_config::dynamicBand");
    if (bandY1 > p->cfg.height) bandY1 = p->cfg.height;
    if (bandY0 % 8u || (bandY1 % 8u && bandY1 != p->cfg.height) || bandY0 >= bandY1) return fail(p, BRMI_ERR_INVALID, "brmi_set_band: rows [%u, %u) -- multiples of 8 inside the frame's %u rows", bandY0, bandY1, p->cfg.height);
    p->bandY0 = bandY0; p->bandY1 = bandY1;
    const uint32_t t0 = bandY0 / 8u, t1 = (bandY1 + 7u) / 8u;
    p->bandFirstPixel = (uint64_t)t0 * p->tilesX * 64; p->bandPixelCount = (uint64_t)(t1 - t0) * p->tilesX * 64;
    p->updated = false;      // the band planes of the culling are made by brmi_update
    return BRMI_OK;
}

int brmi_declare(brmi_pass* p, brmi_declare_cb cb, void* user) {
    if (!p || !cb) return BRMI_ERR_INVALID;
    if (!p->haveScene) return fail(p, BRMI_ERR_STATE, "brmi_declare: call brmi_set_scene first (workspace size depends on the scene)");
    for (uint32_t i = 0; i < BRMI_RES_COUNT; i++) {
        brmi_resource_desc d{};
        d.id = i; d.name = kResNames[i]; d.bytes = p->resNeed[i];
        const bool image = i <= BRMI_RES_HDR_COLOR;
        d.usage = BRMI_USAGE_UNORDERED_ACCESS | BRMI_USAGE_SHADER_RESOURCE | ((i == BRMI_RES_WORKSPACE || i == BRMI_RES_HZB) ? BRMI_USAGE_INTERNAL : 0u);
        d.width = image ? p->cfg.width : 0; d.height = image ? p->cfg.height : 0; d.bytesPerPixel = kResBpp[i];
        d.tileW = image ? 8 : 0; d.tileH = image ? 8 : 0;
        cb(user, &d);
    }
    return BRMI_OK;
}

int brmi_setup(brmi_pass* p, const brmi_resource_binding* b, uint32_t n, brmi_stream stream) {
    if (!p || (!b && n)) return BRMI_ERR_INVALID;
    if (!p->haveScene) return fail(p, BRMI_ERR_STATE, "brmi_setup: call brmi_set_scene first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (uint32_t i = 0; i < n; i++) {
        if (b[i].id >= BRMI_RES_COUNT) return fail(p, BRMI_ERR_INVALID, "brmi_setup: unknown resource id %u", b[i].id);
        p->res[b[i].id] = b[i].ptr; p->resBytes[b[i].id] = b[i].bytes;
    }
    for (uint32_t i = 0; i < BRMI_RES_COUNT; i++) {
        if (!p->res[i]) return fail(p, BRMI_ERR_INVALID, "brmi_setup: resource %s not bound", kResNames[i]);
        if (p->resBytes[i] < p->resNeed[i]) return fail(p, BRMI_ERR_CAPACITY, "brmi_setup: resource %s is %llu B, needs %llu B", kResNames[i], (unsigned long long)p->resBytes[i], (unsigned long long)p->resNeed[i]);
        if ((reinterpret_cast<uintptr_t>(p->res[i]) & 15u) != 0) return fail(p, BRMI_ERR_INVALID, "brmi_setup: resource %s must be 16-byte aligned", kResNames[i]);
    }
    BRMI_HIP(p, hipMemsetAsync(p->res[BRMI_RES_WORKSPACE], 0, p->ws.total, s));
    // the depth chain starts out "empty" everywhere; a multi-GPU band only ever rewrites the texels its rows reach
    if (p->cfg.enableOcclusionCulling && p->resNeed[BRMI_RES_HZB] >= 4) BRMI_HIP(p, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->res[BRMI_RES_HZB]), (int)BRMI_DEPTH_EMPTY_BITS, p->resNeed[BRMI_RES_HZB] / 4, s));
    p->hzbValid = false;
    if (!p->hostInstanceBitBase.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<uint32_t>(p->ws.instanceBitBase), p->hostInstanceBitBase.data(), p->hostInstanceBitBase.size() * 4, hipMemcpyHostToDevice, s));
    if (!p->hostSegPrefix.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<uint32_t>(p->ws.segPrefix), p->hostSegPrefix.data(), p->hostSegPrefix.size() * 4, hipMemcpyHostToDevice, s));
    if (!p->hostMeshLevelWidth.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<uint32_t>(p->ws.meshLevelWidth), p->hostMeshLevelWidth.data(), p->hostMeshLevelWidth.size() * 4, hipMemcpyHostToDevice, s));
    if (!p->hostFlatNodes.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<FlatNode>(p->ws.flatNodes), p->hostFlatNodes.data(), p->hostFlatNodes.size() * sizeof(FlatNode), hipMemcpyHostToDevice, s));
    if (!p->hostFlatLeaves.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<FlatLeaf>(p->ws.flatLeaves), p->hostFlatLeaves.data(), p->hostFlatLeaves.size() * sizeof(FlatLeaf), hipMemcpyHostToDevice, s));
    if (!p->hostInstanceWalk.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<InstanceWalk>(p->ws.instanceWalk), p->hostInstanceWalk.data(), p->hostInstanceWalk.size() * sizeof(InstanceWalk), hipMemcpyHostToDevice, s));
    if (!p->hostPageBoxBase.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<uint32_t>(p->ws.pageBoxBase), p->hostPageBoxBase.data(), p->hostPageBoxBase.size() * 4, hipMemcpyHostToDevice, s));
    if (!p->hostPageRefs.empty()) BRMI_HIP(p, hipMemcpyAsync(p->wsPtr<PageRef>(p->ws.pageRefs), p->hostPageRefs.data(), p->hostPageRefs.size() * sizeof(PageRef), hipMemcpyHostToDevice, s));
    { int rc = launch_expand_luts(p, s); if (rc) return rc; }
    { int rc = launch_meshlet_boxes(p, s); if (rc) return rc; }
    BRMI_HIP(p, hipStreamSynchronize(s));   // host vectors may be reused
    if (p->cfg.collectPassStatistics && !p->eventsCreated) {
        for (int i = 0; i < BRMI_STAGE_COUNT; i++) for (uint32_t k = 0; k < brmi_pass::kEventRing; k++) { BRMI_HIP(p, hipEventCreate(&p->evStart[i][k])); BRMI_HIP(p, hipEventCreate(&p->evStop[i][k])); }
        p->eventsCreated = true;
    }
    p->setupDone = true;
    p->layerPlanesDirty = true;      // new (or re-bound) G-buffer planes: the uniform coat / fuzz words have to be filled in again
    p->constantsSerial = 0;      // the workspace was cleared: the frame constants have to be evaluated again
    return BRMI_OK;
}

// Per-frame host work.  The slice plane depths of the light-cluster grid are evaluated here with
// logf/expf (clustering.hlsl:66-90 evaluates them per cluster on the GPU).
int brmi_update(brmi_pass* p, const brmi_frame_update* u, brmi_stream stream) {
    if (!p || !u || !u->mainCameraHost || !u->perFrameHost) return BRMI_ERR_INVALID;
    if (!p->setupDone) return fail(p, BRMI_ERR_STATE, "brmi_update: call brmi_setup first");
    p->camHost = *u->mainCameraHost; p->pfHost = *u->perFrameHost;
    const brmi_per_frame& pf = p->pfHost;
    if (pf.lightClusterGridSizeX != p->cfg.lightClusterSize[0] || pf.lightClusterGridSizeY != p->cfg.lightClusterSize[1] || pf.lightClusterGridSizeZ != p->cfg.lightClusterSize[2])
        return fail(p, BRMI_ERR_INVALID, "brmi_update: per-frame light cluster grid differs from the configured lightClusterSize");
    if (pf.screenResX != p->cfg.width || pf.screenResY != p->frameHeight()) return fail(p, BRMI_ERR_INVALID, "brmi_update: per-frame screen size differs from the configured target size");
    const float zNear = p->camHost.zNear, zFar = p->camHost.zFar, zSplit = pf.clusterZSplitDepth;
    const uint32_t gz = pf.lightClusterGridSizeZ, nearSlices = pf.nearClusterCount;
    p->planesHost.resize(2 * gz);
    for (uint32_t sliceZ = 0; sliceZ < gz; sliceZ++) {
        float pn, pfar;
        if (sliceZ < nearSlices) {
            const float sliceSize = (zSplit - zNear) / (float)nearSlices;
            pn = -(zNear + (float)sliceZ * sliceSize);
            pfar = -(zNear + (float)(sliceZ + 1) * sliceSize);
        } else {
            const float logStart = std::log(zSplit / zNear), logEnd = std::log(zFar / zNear);
            const float t0 = (float)(sliceZ - nearSlices) / (float)(gz - nearSlices);
            const float t1 = (float)(sliceZ + 1 - nearSlices) / (float)(gz - nearSlices);
            pn = -zNear * std::exp(logStart + t0 * (logEnd - logStart));
            pfar = -zNear * std::exp(logStart + t1 * (logEnd - logStart));
        }
        p->planesHost[2 * sliceZ] = pn; p->planesHost[2 * sliceZ + 1] = pfar;
    }
    // sliceStart[s] = smallest view depth whose cluster slice (ComputeClusterID, lighting.hlsli:166-196) is >= s: the formula is monotone in
    // depth, so the shading pass finds a pixel's slice by comparing against this table instead of two divisions and a logarithm.  Bisection
    // over the positive floats with the shader's own formula; `log` is the correctly rounded fp32 logarithm (through fp64).  Re-evaluated
    // only when the depth range or the grid changes.
    if (p->sliceStartHost.empty() || p->sliceKey[0] != zNear || p->sliceKey[1] != zFar || p->sliceKey[2] != zSplit || p->sliceKeyN[0] != nearSlices || p->sliceKeyN[1] != gz) {
        auto log_cr = [](float x) { return (float)std::log((double)x); };
        const float logStart = log_cr(zSplit / zNear), logEnd = log_cr(zFar / zNear);
        auto slice_of = [&](float z) -> uint32_t {
            if (z < zSplit) { const float t = (z - zNear) / (zSplit - zNear); return t > 0.0f ? (uint32_t)std::min(t * (float)nearSlices, 4294967040.0f) : 0u; }
            const float logZ = log_cr(z / zNear);
            const float u = (logZ - logStart) / (logEnd - logStart);
            return nearSlices + (u > 0.0f ? (uint32_t)std::min(u * (float)(gz - nearSlices), 4294967040.0f) : 0u);
        };
        auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        auto flt = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
        p->sliceStartHost.assign(gz + 2, 0.0f);
        p->sliceStartHost[gz + 1] = flt(0x7F800000u);
        for (uint32_t i = 1; i <= gz; i++) {
            // lo fails, hi passes.  The search stops at 1e30 (z / zNear must stay finite for the float -> uint conversion of the formula to
            // be defined); a slice that starts beyond it starts at +inf.
            uint32_t lo = 0u, hi = bits(1.0e30f);
            if (slice_of(1.0e30f) < i) lo = hi = 0x7F800000u;
            while (hi - lo > 1u) { const uint32_t mid = lo + ((hi - lo) >> 1); if (slice_of(flt(mid)) >= i) hi = mid; else lo = mid; }
            p->sliceStartHost[i] = flt(hi);
        }
        p->sliceKey[0] = zNear; p->sliceKey[1] = zFar; p->sliceKey[2] = zSplit; p->sliceKeyN[0] = nearSlices; p->sliceKeyN[1] = gz;
    }
    {   // band planes of the screen-tile split, 2 px of slack (view space, through the eye)
        const float projY = p->camHost.projection[1][1], H = (float)p->frameHeight();
        const float T = 1.0f - 2.0f * ((float)p->bandY0 - 2.0f) / H, B = 1.0f - 2.0f * ((float)p->bandY1 + 2.0f) / H;
        const float lt = std::sqrt(projY * projY + T * T), lb = std::sqrt(projY * projY + B * B);
        p->bandPlaneTop[0] = 0.0f; p->bandPlaneTop[1] = -projY / lt; p->bandPlaneTop[2] = -T / lt;
        p->bandPlaneBottom[0] = 0.0f; p->bandPlaneBottom[1] = projY / lb; p->bandPlaneBottom[2] = B / lb;
    }
    (void)stream;      // nothing is uploaded: the slice planes are kernel arguments of the light clustering, everything else is derived on the device
    p->updated = true;
    p->updateSerial++;       // the per-frame constants are re-evaluated by the next stage call
    return BRMI_OK;
}

#define STAGE_BEGIN(p, st, s) do { if ((p)->eventsCreated && (((p)->timedStages >> (st)) & 1u)) { (void)hipEventRecord((p)->evStart[st][(p)->evCount[st] % brmi_pass::kEventRing], (s)); } } while (0)
#define STAGE_END(p, st, s) do { if ((p)->eventsCreated && (((p)->timedStages >> (st)) & 1u)) { (void)hipEventRecord((p)->evStop[st][(p)->evCount[st] % brmi_pass::kEventRing], (s)); (p)->evCount[st]++; } } while (0)
#define CHECK_READY(p) do { if (!(p)) return BRMI_ERR_INVALID; if (!(p)->setupDone || !(p)->updated) return brmi::fail((p), BRMI_ERR_STATE, "%s: setup/update not done", __func__); } while (0)

// What a frame's first launch has to wait for when frames are in flight (brmi_execute_split, and the stage entry points that start a frame:
// a graph that schedules the stages itself after a split frame gets the same ordering).  No-ops when nothing was recorded.
// flags of the events that order the two halves of split frames (experiments: BRMI_EVENT_FLAGS, e.g. 0x2 | 0x40000000 = no timing, device-scope release)
static unsigned sync_event_flags() { static const unsigned f = (unsigned)experiment("event_flags", (long)hipEventDisableTiming); return f; }
static int dbg_events() { static const int m = (int)experiment("debug_events", 0); return m; }      // (builds with -DBRMI_EXPERIMENTS only: events NOT issued)
static int wait_for_frames_in_flight(brmi_pass* p, brmi_stream stream) {
    if (p->frameWaitsIssued) return BRMI_OK;      // brmi_execute_split has issued them for this frame: the stage entry points it calls do not repeat them
    // this pass's previous frame may still be resolving / shading on the other stream: its visibility buffer and tables are about to be rewritten
    if (p->frameDoneRecorded && !(dbg_events() & 16)) BRMI_HIP(p, hipStreamWaitEvent(static_cast<hipStream_t>(stream), p->frameDone, 0));
    p->frameDoneRecorded = false;
    // frames in flight: this frame's phase 1 reads the chain the source pass built for the frame before, possibly on another stream
    // (recorded on this very stream -- the passes of a ring share their geometry stream --: stream order already says so, and every wait
    // is a barrier packet worth a few us on the geometry half's critical path)
    if (p->history && p->history->chainRecorded && p->history->chainStream != stream) BRMI_HIP(p, hipStreamWaitEvent(static_cast<hipStream_t>(stream), p->history->chainReady, 0));
    // ... and this frame rewrites the chain a pass that has THIS one as its source may still be reading in its phase 1 (the same event:
    // it is recorded after that pass's culling)
    for (brmi_pass* user : p->historyUsers) if (user != p->history && user->chainRecorded && user->chainStream != stream) BRMI_HIP(p, hipStreamWaitEvent(static_cast<hipStream_t>(stream), user->chainReady, 0));
    return BRMI_OK;
}

int brmi_clear_visibility(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    if (int w = wait_for_frames_in_flight(p, stream)) return w;
    STAGE_BEGIN(p, BRMI_STAGE_CLEAR, s); int rc = launch_clear(p, s); STAGE_END(p, BRMI_STAGE_CLEAR, s); return rc;
}
int brmi_cull(brmi_pass* p, uint32_t phase, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    if (phase == 1) if (int w = wait_for_frames_in_flight(p, stream)) return w;
    const int st = phase == 2 ? BRMI_STAGE_CULL2 : BRMI_STAGE_CULL;
    STAGE_BEGIN(p, st, s); int rc = launch_cull(p, phase, s); STAGE_END(p, st, s); return rc;
}
int brmi_raster(brmi_pass* p, uint32_t phase, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    const int st = phase == 2 ? BRMI_STAGE_RASTER2 : BRMI_STAGE_RASTER;
    STAGE_BEGIN(p, st, s); int rc = launch_raster(p, phase, s); STAGE_END(p, st, s); return rc;
}
int brmi_depth_copy(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    STAGE_BEGIN(p, BRMI_STAGE_DEPTH_COPY, s); int rc = launch_depth_copy(p, s); STAGE_END(p, BRMI_STAGE_DEPTH_COPY, s); return rc;
}
int brmi_build_hzb(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    if (!p->cfg.enableOcclusionCulling) return brmi::fail(p, BRMI_ERR_STATE, "brmi_build_hzb: the pass was created without enableOcclusionCulling");
    STAGE_BEGIN(p, BRMI_STAGE_HZB, s); int rc = launch_hzb(p, s, false, false); STAGE_END(p, BRMI_STAGE_HZB, s);
    if (rc == BRMI_OK) p->hzbValid = true;
    return rc;
}
// brmi_execute's variants: (1) LinearDepthCopyPass1 + LinearDepthDownsamplePass1 in one kernel (the chain is built straight
// from the visibility keys, the depth map is written on the way); (2) LinearDepthDownsamplePass2 skipped on the device when
// phase 2 rasterised nothing, because the final depth then equals the phase-1 depth the chain was just built from.
static int build_hzb_fused(brmi_pass* p, hipStream_t s, bool fromVisibility, bool onlyIfPhase2Drew) {
    STAGE_BEGIN(p, BRMI_STAGE_HZB, s); int rc = launch_hzb(p, s, fromVisibility, onlyIfPhase2Drew); STAGE_END(p, BRMI_STAGE_HZB, s);
    if (rc == BRMI_OK) p->hzbValid = true;
    return rc;
}
int brmi_invalidate_hzb(brmi_pass* p) {
    if (!p) return BRMI_ERR_INVALID;
    p->hzbValid = false;
    if (p->history) p->history->hzbValid = false;      // the chain the next phase 1 would test against is the source's
    return BRMI_OK;
}
int brmi_set_history_source(brmi_pass* p, brmi_pass* source) {
    if (!p) return BRMI_ERR_INVALID;
    if (source == p) source = nullptr;
    if (source) {
        if (!p->setupDone || !source->setupDone) return brmi::fail(p, BRMI_ERR_STATE, "brmi_set_history_source: both passes need brmi_setup first");
        if (!p->cfg.enableOcclusionCulling || !source->cfg.enableOcclusionCulling) return brmi::fail(p, BRMI_ERR_STATE, "brmi_set_history_source: both passes need enableOcclusionCulling (there is no history otherwise)");
        if (p->cfg.width != source->cfg.width || p->cfg.height != source->cfg.height || ((p->bandY0 != source->bandY0 || p->bandY1 != source->bandY1) && !(p->cfg.dynamicBand && source->cfg.dynamicBand)))
            return brmi::fail(p, BRMI_ERR_INVALID, "brmi_set_history_source: the passes differ in size or band (%ux%u rows %u-%u against %ux%u rows %u-%u)", p->cfg.width, p->cfg.height, p->bandY0, p->bandY1,
                              source->cfg.width, source->cfg.height, source->bandY0, source->bandY1);
    }
    if (p->history) { auto& u = p->history->historyUsers; u.erase(std::remove(u.begin(), u.end(), p), u.end()); }
    p->history = source;
    if (source) source->historyUsers.push_back(p);
    return BRMI_OK;
}
int brmi_gbuffer(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    STAGE_BEGIN(p, BRMI_STAGE_GBUFFER, s); int rc = launch_gbuffer(p, s); STAGE_END(p, BRMI_STAGE_GBUFFER, s); return rc;
}
int brmi_light_clustering(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    STAGE_BEGIN(p, BRMI_STAGE_LIGHT_CLUSTER, s); int rc = launch_light_clustering(p, s); STAGE_END(p, BRMI_STAGE_LIGHT_CLUSTER, s); return rc;
}
int brmi_set_shade_slabs(brmi_pass* p, uint32_t slabs, brmi_slab_fn fn, void* user) {
    if (!p) return BRMI_ERR_INVALID;
    p->shadeSlabs = (fn && slabs > 1u) ? slabs : 0u; p->shadeSlabFn = p->shadeSlabs ? fn : nullptr; p->shadeSlabUser = p->shadeSlabs ? user : nullptr;
    return BRMI_OK;
}
int brmi_shade(brmi_pass* p, brmi_stream stream) {
    CHECK_READY(p); hipStream_t s = static_cast<hipStream_t>(stream);
    STAGE_BEGIN(p, BRMI_STAGE_SHADE, s); int rc = launch_shade(p, s); STAGE_END(p, BRMI_STAGE_SHADE, s); return rc;
}

// The whole chain in graph order.  K6 is folded into the G-buffer kernel (one visibility read).
int brmi_execute(brmi_pass* p, brmi_stream stream) { return brmi_execute_split(p, stream, stream); }

// Geometry half (culling, rasterisation, depth chain: latency-bound launches that leave most of the chip idle) on `stream`, resolve +
// shading half (VALU-bound, fills the chip) on `shadeStream`.  With two linked passes alternating frames on the SAME two streams -- the
// geometry stream created with a higher priority -- frame k+1's geometry runs beside frame k's shading and gets CU slots first.
int brmi_execute_split(brmi_pass* p, brmi_stream stream, brmi_stream shadeStream) {
    CHECK_READY(p);
    int rc;
    p->executesSinceTimes++;
    const bool split = shadeStream != stream;
    p->splitFrame = split;
    // whatever way this call returns, the frame's shortcuts do not outlive it: stand-alone stage calls behind a failed frame must not find `splitFrame`
    // (which disables the wide flat traversal), `shadeSharesChip` or the issued-waits mark still set
    struct FrameScope { brmi_pass* p; ~FrameScope() { p->splitFrame = false; p->shadeSharesChip = false; p->frameWaitsIssued = false; p->clearFrameStateWithConstants = false; p->fuseFrameClear = false; p->seedInHzbTail = false; } } frameScope{p};
    p->resolveSetupDone = false; p->depthFinal = false;      // (a frame that failed half-way must not leave its shortcuts to the stage entry points)
    if ((rc = wait_for_frames_in_flight(p, stream))) return rc;
    p->frameWaitsIssued = true;
    // When the phase-1 traversal is the one-launch LDS walk and the frame constants are due anyway, the frame needs no clear launch: the
    // constants kernel zeroes the culling state and the walk's launch carries the visibility clear (brmi_cull.hip, SideClear).
    static const bool rideEnv = experiment("clear_rides", 1) != 0;
    const bool rides = rideEnv && p->constantsSerial != p->updateSerial && p->minLevelWidth <= 1024u /* the one-launch walk runs (brmi_cull.hip: HIER_CAP_MAX) */ && !p->forceLevelKernels && p->scene.activeDrawCount != 0u;
    p->lightGridDone = false;
    // A split frame (round 4): the riders move to where they fit -- the visibility clear onto k_cull_clusters' launch (brmi_cull.hip: ClearRide), the
    // light clustering onto the shading stream, which is idle until this frame's pixel pass (below, behind the same event as the early resolve setup).
    static const bool sideEnv = experiment("side_riders", 1) != 0;
    // where the per-cluster resolve tables of a split frame are made (BRMI_EARLY_RESOLVE_SETUP): 0 = at the end of the geometry stream, 1 = for the phase-1
    // clusters on the shading stream beside the rasteriser (an event after the culling), 2 = on the shading stream in front of the pixel pass (no event)
    static const int setupWhere = (int)experiment("early_resolve_setup", 1);
    const bool earlySetup = split && setupWhere == 1;      // (a frame that resolves without tables launches nothing there: launch_resolve_setup)
    const bool lateSetup = split && setupWhere == 2;
    const bool sideRiders = rides && split && sideEnv && (p->bandPixelCount & 1ull) == 0ull;
    if (sideRiders) {
        p->clearFrameStateWithConstants = true; p->clearVisibilityWithClusterCull = true;
        rc = brmi_cull(p, 1, stream);
        p->clearFrameStateWithConstants = false;
        if (rc == BRMI_OK && p->clearVisibilityWithClusterCull) { p->clearVisibilityWithClusterCull = false; return brmi::fail(p, BRMI_ERR_STATE, "brmi_execute: the cluster culling did not carry the visibility clear"); }
        if (rc) { p->clearVisibilityWithClusterCull = false; return rc; }
    } else if (rides) {
        p->clearFrameStateWithConstants = true; p->clearVisibilityWithTraversal = true;
        rc = brmi_cull(p, 1, stream);
        p->clearFrameStateWithConstants = false;
        if (rc == BRMI_OK && p->clearVisibilityWithTraversal) { p->clearVisibilityWithTraversal = false; return brmi::fail(p, BRMI_ERR_STATE, "brmi_execute: the traversal did not carry the visibility clear"); }
        if (rc) { p->clearVisibilityWithTraversal = false; return rc; }
    } else {
    p->fuseFrameClear = true;
    rc = brmi_clear_visibility(p, stream);
    p->fuseFrameClear = false;
    if (rc) return rc;
    if ((rc = brmi_cull(p, 1, stream))) return rc;
    }
    // The per-cluster resolve tables need the cluster list, not the keys: for the phase-1 clusters they are made NOW, on the shading stream (idle until
    // this frame's pixel pass), beside the rasteriser -- the geometry half is a chain of latency-bound launches and this one was 40 us at its end.
    // (Not on frames of more triangles than pixels, whose setup skips clusters that own no pixel and so needs the final keys.)
    if (earlySetup) {
        if (!p->cullDone) BRMI_HIP(p, hipEventCreateWithFlags(&p->cullDone, sync_event_flags()));
        if (!(dbg_events() & 1)) {
        BRMI_HIP(p, hipEventRecord(p->cullDone, static_cast<hipStream_t>(stream)));
        BRMI_HIP(p, hipStreamWaitEvent(static_cast<hipStream_t>(shadeStream), p->cullDone, 0));
        }
        if (sideRiders) { if ((rc = brmi::launch_light_clustering(p, static_cast<hipStream_t>(shadeStream)))) return rc; p->lightGridDone = true; }
        if ((rc = launch_resolve_setup(p, static_cast<hipStream_t>(shadeStream), 1u))) return rc;
    }
    if ((rc = brmi_raster(p, 1, stream))) return rc;
    if (p->cfg.enableOcclusionCulling) {
        // reference graph: LinearDepthCopyPass1 -> LinearDepthDownsamplePass1 -> HierarchicalCullingPass2 -> ...RasterizeClustersPass2
        // -> LinearDepthCopyPass2 -> LinearDepthDownsamplePass2 (CLodExtension.cpp:1920-2088)
        p->seedInHzbTail = true;
        rc = build_hzb_fused(p, static_cast<hipStream_t>(stream), true, false);
        p->seedInHzbTail = false;
        if (rc) return rc;
        if ((rc = brmi_cull(p, 2, stream))) return rc;
        if ((rc = brmi_raster(p, 2, stream))) return rc;
        // LinearDepthCopyPass2 + LinearDepthDownsamplePass2 straight from the final visibility keys, skipped on the device when phase 2 drew
        // nothing: the depth map and the chain are final BEFORE the G-buffer kernel (which then skips its depth store), so a pass that
        // renders the next frame on another stream (brmi_set_history_source) can start while this frame is resolved and shaded.
        if ((rc = build_hzb_fused(p, static_cast<hipStream_t>(stream), true, true))) return rc;
        // recorded every frame (a pass may be linked to this one later, from another stream)
        if (!p->chainReady) BRMI_HIP(p, hipEventCreateWithFlags(&p->chainReady, sync_event_flags()));
        if (!(dbg_events() & 2)) { BRMI_HIP(p, hipEventRecord(p->chainReady, static_cast<hipStream_t>(stream))); p->chainRecorded = true; p->chainStream = stream; }
    }
    if (split) {
        if (!lateSetup) {
            if ((rc = launch_resolve_setup(p, static_cast<hipStream_t>(stream), earlySetup ? 2u : 0u))) return rc;
            p->resolveSetupDone = true;
        }
        if (!p->geometryDone) BRMI_HIP(p, hipEventCreateWithFlags(&p->geometryDone, sync_event_flags()));
        if (!p->frameDone) BRMI_HIP(p, hipEventCreateWithFlags(&p->frameDone, sync_event_flags()));
        if (!(dbg_events() & 4)) {
        BRMI_HIP(p, hipEventRecord(p->geometryDone, static_cast<hipStream_t>(stream)));
        BRMI_HIP(p, hipStreamWaitEvent(static_cast<hipStream_t>(shadeStream), p->geometryDone, 0));
        }
        stream = shadeStream;
    }
    const bool lightsDone = p->lightGridDone;      // the culling pass's launches carried the light clustering
    p->lightGridDone = false;
    p->depthFinal = p->cfg.enableOcclusionCulling != 0;
    p->shadeSharesChip = split;
    // (builds with -DBRMI_EXPERIMENTS only: debug_skip bit 0 drops the shading launch, bit 1 the G-buffer launch of brmi_execute -- what the other half costs without them)
    static const int skipDbg = (int)experiment("debug_skip", 0);
    rc = (skipDbg & 2) ? BRMI_OK : brmi_gbuffer(p, stream);
    p->shadeSharesChip = false;
    p->depthFinal = false;
    if (rc) return rc;
    if (!lightsDone && (rc = brmi_light_clustering(p, stream))) return rc;
    p->shadeSharesChip = split;
    rc = (skipDbg & 1) ? BRMI_OK : brmi_shade(p, stream);
    p->shadeSharesChip = false;
    if (rc) return rc;
    if (split && !(dbg_events() & 8)) { BRMI_HIP(p, hipEventRecord(p->frameDone, static_cast<hipStream_t>(stream))); p->frameDoneRecorded = true; }
    p->splitFrame = false;          // (the stage entry points, called on their own, are not part of a split frame)
    return BRMI_OK;
}

int brmi_read_counters(brmi_pass* p, brmi_counters* out, brmi_stream stream) {
    if (!p || !out) return BRMI_ERR_INVALID;
    if (!p->setupDone) return brmi::fail(p, BRMI_ERR_STATE, "brmi_read_counters: setup not done");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t c[CNT_WORDS];
    BRMI_HIP(p, hipMemcpyAsync(c, p->counters(), sizeof(c), hipMemcpyDeviceToHost, s));
    BRMI_HIP(p, hipStreamSynchronize(s));
    std::memset(out, 0, sizeof(*out));
    out->instancesTested = c[CNT_INSTANCES_TESTED]; out->instancesVisible = c[CNT_INSTANCES_VISIBLE];
    out->nodesVisited = c[CNT_NODES_VISITED];
    uint32_t overflowQueued = 0;
    for (uint32_t st = 0; st < CNT_STRIPE_COUNT; st++) {
        const uint32_t* sp = c + CNT_STRIPES + st * CNT_STRIPE_WORDS;
        out->instancesTested += sp[0]; out->instancesVisible += sp[1]; out->nodesVisited += sp[2];
        overflowQueued += sp[STRIPE_OVERFLOW];
    } out->bucketRecords = c[CNT_BUCKETS]; out->meshletsTested = c[CNT_MESHLETS_TESTED];
    for (uint32_t st = 0; st < CNT_STRIPE_COUNT; st++) out->meshletsTested += c[CNT_STRIPES + st * CNT_STRIPE_WORDS + STRIPE_MESHLETS_TESTED];
    out->visibleClusters = c[CNT_VISIBLE]; out->visibleClustersPhase2 = c[CNT_VISIBLE2];
    out->droppedRecords = c[CNT_DROPPED_RECORDS]; out->droppedClusters = c[CNT_DROPPED_CLUSTERS]; out->lightPagesUsed = c[CNT_LIGHT_PAGES];
    out->reserved[0] = c[CNT_SUM_VERTS_LO]; out->reserved[2] = c[CNT_SUM_VERTS_HI];      // (vertex sum | triangle sum << 32 in one word, brmi_internal.h)
    out->reserved[1] = std::min(c[CNT_HELD1], p->cfg.maxVisibleClusters); out->reserved[3] = std::min(c[CNT_LATE1], p->cfg.maxVisibleClusters);      // round 6: the draw list (include/brmi.h)
    out->reserved[4] = c[CNT_RASTER_CLUSTERS]; out->reserved[5] = c[CNT_BIN_OVERFLOW] + overflowQueued;
    out->replayNodes = c[CNT_REPLAY_NODES]; out->replayMeshlets = c[CNT_REPLAY_MESHLETS];
    return BRMI_OK;
}

int brmi_set_timed_stages(brmi_pass* p, uint32_t stageMask) {
    if (!p) return BRMI_ERR_INVALID;
    p->timedStages = stageMask;
    return BRMI_OK;
}

int brmi_stage_times(brmi_pass* p, float* ms) {
    if (!p || !ms) return BRMI_ERR_INVALID;
    if (!p->eventsCreated) return brmi::fail(p, BRMI_ERR_STATE, "brmi_stage_times: collectPassStatistics is off");
    // mean PER FRAME over the recordings since the previous call (at most the last kEventRing), then reset.  brmi_execute records the
    // depth-chain stage twice per frame (before phase 2 and after the G-buffer pass): its two recordings are one frame's cost.
    for (int i = 0; i < BRMI_STAGE_COUNT; i++) {
        ms[i] = 0.0f;
        const uint32_t perFrame = (i == BRMI_STAGE_HZB && p->executesSinceTimes != 0u && p->evCount[i] == 2u * p->executesSinceTimes) ? 2u : 1u;
        uint32_t n = std::min(p->evCount[i], brmi_pass::kEventRing);
        n -= n % perFrame;
        if (n == 0) continue;
        double sum = 0.0;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t slot = (p->evCount[i] - 1 - k) % brmi_pass::kEventRing;
            float t = 0.0f;
            BRMI_HIP(p, hipEventSynchronize(p->evStop[i][slot]));
            BRMI_HIP(p, hipEventElapsedTime(&t, p->evStart[i][slot], p->evStop[i][slot]));
            sum += t;
        }
        ms[i] = (float)(sum / (n / perFrame));
        p->evCount[i] = 0;
    }
    p->executesSinceTimes = 0;
    return BRMI_OK;
}

// bytes_frame = 140*P + sum_clusters(144 + 12V + 3T) + 64*M_tested + 16*M_visible + 64*N_nodes
int brmi_algorithmic_bytes(brmi_pass* p, uint64_t* perStage, uint64_t* total) {
    if (!p || !perStage || !total) return BRMI_ERR_INVALID;
    // a read-back call: waits for whatever stream the frame was executed on (a non-blocking stream does not order with the NULL stream)
    BRMI_HIP(p, hipDeviceSynchronize());
    brmi_counters c; int rc = brmi_read_counters(p, &c, nullptr); if (rc) return rc;
    const uint64_t P = (uint64_t)p->cfg.width * (p->bandY1 - p->bandY0);
    uint64_t sumV = c.reserved[0], sumT = c.reserved[2], nClusters = c.reserved[4];
    if (c.reserved[1] != 0u) {      // a frame that held clusters back: what was rasterised is the draw list, the late list and phase 2 (the held clusters that stayed hidden moved no vertex or index byte)
        uint32_t vt[2];
        BRMI_HIP(p, hipMemcpy(vt, p->counters() + CNT_DRAWN_VT, 8, hipMemcpyDeviceToHost));
        sumV = vt[0]; sumT = vt[1]; nClusters -= c.reserved[1] - c.reserved[3];
    }
    for (int i = 0; i < BRMI_STAGE_COUNT; i++) perStage[i] = 0;
    perStage[BRMI_STAGE_CLEAR] = 8 * P;
    perStage[BRMI_STAGE_CULL] = 64ull * c.meshletsTested + 16ull * (c.visibleClusters + c.visibleClustersPhase2) + 64ull * c.nodesVisited;
    perStage[BRMI_STAGE_RASTER] = 8 * P + 144ull * nClusters + 12ull * sumV + 3ull * sumT;
    if (p->cfg.enableOcclusionCulling) {
        // depth copy: 8 B key in, 4 B depth out; chain: every depth texel read once, 1/3 of that written, built twice per frame;
        // phase-2 cull / raster traffic is counted in the cull / raster rows (their counters accumulate over both phases)
        perStage[BRMI_STAGE_DEPTH_COPY] = 12 * P;
        perStage[BRMI_STAGE_HZB] = 2 * (4 * P + 4 * P / 3);
    }
    perStage[BRMI_STAGE_GBUFFER] = (8 + 52 + 4) * P;
    perStage[BRMI_STAGE_SHADE] = (4 + 48 + 8) * P;
    *total = 0; for (int i = 0; i < BRMI_STAGE_COUNT; i++) *total += perStage[i];
    return BRMI_OK;
}

int brmi_algorithmic_bytes_launched(brmi_pass* p, uint64_t* perStage, uint64_t* total) {
    if (int rc = brmi_algorithmic_bytes(p, perStage, total)) return rc;
    if (!p->sceneHasCoat && !p->sceneHasFuzz) {      // k_shade<0>: depth 4 + normals 16 + albedo 4 + metallic / roughness 4 + emissive 8 read, HDR 8 written; the coat and fuzz planes (8 + 8) stay unread
        const uint64_t P = (uint64_t)p->cfg.width * (p->bandY1 - p->bandY0);
        *total -= perStage[BRMI_STAGE_SHADE]; perStage[BRMI_STAGE_SHADE] = 44ull * P; *total += perStage[BRMI_STAGE_SHADE];
    }
    return BRMI_OK;
}

int brmi_debug_read_held(brmi_pass* p, uint32_t* held, uint32_t heldCapacity, uint32_t* heldCount, uint32_t* late, uint32_t lateCapacity, uint32_t* lateCount) {
    if (!p || !heldCount || !lateCount || !p->setupDone) return BRMI_ERR_INVALID;
    BRMI_HIP(p, hipDeviceSynchronize());
    uint32_t nh = 0, nl = 0;
    BRMI_HIP(p, hipMemcpy(&nh, p->counters() + CNT_HELD1, 4, hipMemcpyDeviceToHost));
    BRMI_HIP(p, hipMemcpy(&nl, p->counters() + CNT_LATE1, 4, hipMemcpyDeviceToHost));
    if (!p->holdEnabled) nh = nl = 0;
    nh = std::min(nh, p->cfg.maxVisibleClusters); nl = std::min(nl, p->cfg.maxVisibleClusters);
    *heldCount = nh; *lateCount = nl;
    if (held && nh) {
        std::vector<HeldRecord> rec(std::min(nh, heldCapacity));
        if (!rec.empty()) BRMI_HIP(p, hipMemcpy(rec.data(), p->wsPtr<HeldRecord>(p->ws.heldRecords), rec.size() * sizeof(HeldRecord), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < rec.size(); i++) held[i] = rec[i].clusterIndex;
    }
    if (late && nl && lateCapacity) BRMI_HIP(p, hipMemcpy(late, p->wsPtr<uint32_t>(p->ws.lateList), (size_t)std::min(nl, lateCapacity) * 4, hipMemcpyDeviceToHost));
    return BRMI_OK;
}

int brmi_debug_wide_triangles(brmi_pass* p, uint32_t out[3]) {
    if (!p || !out || !p->setupDone) return BRMI_ERR_INVALID;
    BRMI_HIP(p, hipDeviceSynchronize());
    uint32_t c[3] = {0u, 0u, 0u};
    BRMI_HIP(p, hipMemcpy(&c[0], p->counters() + CNT_WIDE1, 4, hipMemcpyDeviceToHost));
    BRMI_HIP(p, hipMemcpy(&c[1], p->counters() + CNT_WIDE1B, 4, hipMemcpyDeviceToHost));
    BRMI_HIP(p, hipMemcpy(&c[2], p->counters() + CNT_WIDE2, 4, hipMemcpyDeviceToHost));
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
    return BRMI_OK;
}

int brmi_debug_lean_clusters(brmi_pass* p, uint32_t out[4]) {
    if (!p || !out || !p->setupDone) return BRMI_ERR_INVALID;
    BRMI_HIP(p, hipDeviceSynchronize());
    uint32_t c = 0u, heads[64 * 32];
    BRMI_HIP(p, hipMemcpy(&c, p->counters() + CNT_GENERAL1, 4, hipMemcpyDeviceToHost));
    BRMI_HIP(p, hipMemcpy(heads, p->counters() + CNT_BIG1, sizeof(heads), hipMemcpyDeviceToHost));
    out[0] = p->leanLastLaunch ? 1u : 0u; out[1] = c; out[2] = out[3] = 0u;
    for (uint32_t s = 0; s < 64u; s++) { out[2] += heads[s * 32u]; out[3] += heads[s * 32u + 1u]; }
    return BRMI_OK;
}

int brmi_debug_read_lean_queue(brmi_pass* p, uint32_t stripe, uint32_t* runs, uint32_t maxRuns, void* entries, uint32_t maxEntries, uint32_t counts[2]) {
    if (!p || !counts || !p->setupDone || stripe >= 64u || p->leanMinClusters == 0u) return BRMI_ERR_INVALID;
    BRMI_HIP(p, hipDeviceSynchronize());
    uint32_t head[2];
    BRMI_HIP(p, hipMemcpy(head, p->counters() + CNT_BIG1 + stripe * 32u, 8, hipMemcpyDeviceToHost));
    const uint32_t cap = p->leanQueue / 64u;
    counts[0] = std::min(head[0], cap); counts[1] = std::min(head[1], cap);
    if (runs) BRMI_HIP(p, hipMemcpy(runs, p->wsPtr<uint8_t>(p->ws.bigRuns) + (size_t)stripe * cap * 8u, (size_t)std::min(maxRuns, counts[1]) * 8u, hipMemcpyDeviceToHost));
    if (entries) BRMI_HIP(p, hipMemcpy(entries, p->wsPtr<uint8_t>(p->ws.bigQueue) + (size_t)stripe * cap * 96u, (size_t)std::min(maxEntries, counts[0]) * 96u, hipMemcpyDeviceToHost));
    return BRMI_OK;
}

int brmi_debug_read_bin_records(brmi_pass* p, void* dst, uint64_t bytes) {
    if (!p || !dst || !p->setupDone) return BRMI_ERR_INVALID;
    BRMI_HIP(p, hipDeviceSynchronize());
    // (bytes with bit 63 set: the stamp region instead, where instrumented builds of k_raster park their phase sums)
    const bool overflowRegion = (bytes >> 63) != 0; bytes &= ~(1ull << 63);
    if ((bytes >> 62) & 1ull) { bytes &= ~(1ull << 62); BRMI_HIP(p, hipMemset(p->wsPtr<uint8_t>(p->ws.binRecords), 0, bytes)); return BRMI_OK; }   // (bit 62: zero the records instead, so that a following frame's records can be told from older ones)
    BRMI_HIP(p, hipMemcpy(dst, overflowRegion ? p->wsPtr<uint8_t>(p->ws.debugStamps) : p->wsPtr<uint8_t>(p->ws.binRecords), bytes, hipMemcpyDeviceToHost));
    return BRMI_OK;
}

int brmi_debug_arith(const float* a, const float* b, float* outDiv, float* outSqrt, uint32_t* outHalfBits, uint32_t n, brmi_stream stream) {
    if (!a || !b || !outDiv || !outSqrt || !outHalfBits) return BRMI_ERR_INVALID;
    if (n == 0) return BRMI_OK;
    hipLaunchKernelGGL(k_debug_arith, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, outDiv, outSqrt, outHalfBits, n);
    return hipGetLastError() == hipSuccess ? BRMI_OK : BRMI_ERR_HIP;
}

int brmi_debug_arith_in_range(const float* a, float* outRcp, float* outSqrt, float* outRsqrt, uint32_t n, brmi_stream stream) {
    if (!a || !outRcp || !outSqrt || !outRsqrt) return BRMI_ERR_INVALID;
    if (n == 0) return BRMI_OK;
    hipLaunchKernelGGL(k_debug_arith_in_range, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a, outRcp, outSqrt, outRsqrt, n);
    return hipGetLastError() == hipSuccess ? BRMI_OK : BRMI_ERR_HIP;
}

}  // extern "C"
