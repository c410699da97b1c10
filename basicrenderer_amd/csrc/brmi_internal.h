// brmi_internal.h -- host-side state of one brmi_pass and the workspace carve-up.
#ifndef BRMI_INTERNAL_H
#define BRMI_INTERNAL_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "brmi.h"
#include "brmi_device.h"

#ifndef BRMI_BIN_COUNT_STRIDE
#define BRMI_BIN_COUNT_STRIDE 32      // words between the record counters of two raster bins (brmi_raster.hip)
#endif
namespace brmi {

// device-visible counters block (u32 words) at the head of the workspace
enum CounterIndex : uint32_t {
    CNT_VISIBLE = 3,          // phase-1 visible clusters after compaction (min(capacity))
    CNT_VISIBLE2,             // phase-2 visible clusters after compaction
    CNT_DROPPED_RECORDS,
    CNT_DROPPED_CLUSTERS,
    CNT_INSTANCES_TESTED,
    CNT_INSTANCES_VISIBLE,
    CNT_NODES_VISITED,
    CNT_MESHLETS_TESTED,
    CNT_LIGHT_PAGES,
    CNT_SUM_VERTS_LO = 14, CNT_SUM_VERTS_HI,     // ONE 64-bit word: sum of the vertex counts of the rasterised clusters (low half) | of their triangle counts (high half): one atomic
                                                 // reserves both runs of the resolve arena (<= 30 M clusters x 128 fits 32 bits)
    CNT_SUM_TRIS_LO, CNT_SUM_TRIS_HI,            // (unused since round 5)
    CNT_RASTER_CLUSTERS,
    CNT_BIN_OVERFLOW,         // raster records that found their screen bin full (rasterised in place with global atomics)
    CNT_DEFERRED_PIXELS_B,    // second deferred-pixel counter: shading calls alternate, each clears the other one for the next call
    CNT_DEFERRED_PIXELS,      // pixels the specialised shading kernel left to the general one
    CNT_RESOLVE_SPILL,        // some visible cluster found the resolve arena full (its pixels decode their vertices in place)
    CNT_RESERVED_37,          // (round 4's "clusters marked" flag; the slot keeps the enumeration's layout)
    CNT_DEFERRED_DROPPED,     // layered pixels that found their deferred-list stripe full (impossible by construction; counted anyway)
    CNT_TILE_OVERFLOW,        // (cluster, tile) pairs of the tile rasteriser that found the tile's list full (folded into CNT_BIN_OVERFLOW per phase)
    CNT_FRONTIER0 = 32,       // frontier sizes per BFS level: [CNT_FRONTIER0 + level]
    CNT_STRIPES = 128,        // 64 stripes x 32 words: per-stripe {instances tested, instances visible, nodes visited}
    CNT_STRIPE_COUNT = 64, CNT_STRIPE_WORDS = 32,
    STRIPE_DEFERRED_A = 4, STRIPE_DEFERRED_B = 8,   // words of a stripe: deferred-pixel list lengths (3 classes each) of alternating shading calls
    STRIPE_OVERFLOW = 3,      // word of a stripe: records in the stripe's raster overflow queue
    STRIPE_MESHLETS_TESTED = 12,   // word of a stripe: meshlets the cluster cull looked at (statistics; one shared word cost a same-line atomic per wave)
    // the append counters every wave of the culling kernels hits (atomics with return) each on a 128 B line of its own: on one line they
    // serialise against each other and against the statistics (~100 wave-level atomics per microsecond and line)
    CNT_BUCKETS = 128 + 64 * 32,            // bucket records appended by the traversal
    CNT_TEMP_VISIBLE = CNT_BUCKETS + 32,    // survivors appended by the cluster cull (phase 1)
    CNT_TEMP_VISIBLE2 = CNT_BUCKETS + 64,   // survivors of phase 2
    CNT_REPLAY_NODES = CNT_BUCKETS + 96,
    CNT_REPLAY_MESHLETS = CNT_BUCKETS + 128,
    // round 6, the draw list (brmi_raster.hip): clusters of the phase-1 visible list the rasteriser takes first / holds back for the re-test / draws late
    CNT_DRAW1 = CNT_BUCKETS + 160,          // entries of the draw list
    CNT_HELD1 = CNT_BUCKETS + 192,          // held records
    CNT_LATE1 = CNT_BUCKETS + 224,          // held clusters the re-test could not prove hidden (drawn by the late pass)
    CNT_DRAWN_VT = CNT_BUCKETS + 256,       // ONE 64-bit word (even index): vertex | triangle << 32 sums of the clusters that ARE rasterised in phase 1 (draw list + late list)
    // round 6: triangles whose records a workgroup emits (k_raster_wide): queue lengths of the phase-1 draw pass, the late pass and phase 2
    CNT_WIDE1 = CNT_BUCKETS + 288, CNT_WIDE1B = CNT_BUCKETS + 320, CNT_WIDE2 = CNT_BUCKETS + 352,
    CNT_GENERAL1 = CNT_BUCKETS + 384,       // round 6: clusters the lean rasteriser left to the general launch behind it (k_raster<false, true>)
    CNT_BIG1 = CNT_BUCKETS + 416,           // ... and its queue of triangles large enough for the bins (k_raster_emit): 64 stripes x 32 words, a 64-bit head (entries | runs << 32) in the first two
    CNT_WORDS = CNT_BUCKETS + 416 + 64 * 32
};
static_assert((CNT_DRAWN_VT & 1u) == 0u, "64-bit counter");

// Phase 2 starts from the replay buffers: the replayed meshlets become the first bucket records, the replayed nodes the level-0 frontier;
// per-level frontier counters start from zero; the raster overflow queues of phase 1 are counted and emptied.  Thread t of (at least) 128.
__device__ inline void seed_phase2(uint32_t* counters, uint32_t capacity, uint32_t t) {
    if (t == 0) { counters[CNT_REPLAY_NODES] = min(counters[CNT_REPLAY_NODES], capacity); counters[CNT_BUCKETS] = min(counters[CNT_REPLAY_MESHLETS], capacity); counters[CNT_TEMP_VISIBLE2] = 0; counters[CNT_VISIBLE2] = 0; }
    if (t < CNT_STRIPES - CNT_FRONTIER0) counters[CNT_FRONTIER0 + t] = 0;
    if (t < CNT_STRIPE_COUNT) {
        uint32_t* q = &counters[CNT_STRIPES + t * CNT_STRIPE_WORDS + STRIPE_OVERFLOW];
        if (*q) atomicAdd(&counters[CNT_BIN_OVERFLOW], *q);
        *q = 0u;
    }
}

// internal records (ours; the reference's 12 B / 24 B records plus the rank bookkeeping)
// A mesh whose whole BVH has at most 8192 nodes (brmi_set_scene walks it) is evaluated flat by the traversal kernels: one lane per node (a wave for
// up to 256 nodes, eight draws to a wave up to 8, a 1024-thread workgroup beyond 256), the
// records below instead of the node -> group / segment chain (static topology: node, group and segment contents as brmi_set_scene read them;
// what the host may rewrite between frames -- instances, objects, the page map -- is still read from its buffers).
struct FlatNode {                                                                     // 64 B, breadth-first (a parent's position is below its children's)
    float cull[4], lod[4]; float maxQuadricError;
    uint32_t nodeId;            // relative to the mesh's lodNodesBase (replay records name nodes by it)
    uint32_t info;              // internal | has a refined group << 1 | segment holds meshlets << 2 | parent position << 8 (breadth-first: below the node's own)
    uint32_t ownerGroup;        // leaf: mesh-local group
    uint32_t segFirstCount;     // leaf: firstMeshletInPage | meshletCount << 16
    uint32_t pageMapIndex;      // leaf: absolute index of the segment's page-map entry
    uint32_t firstBitRel;       // leaf: the segment's first bit relative to the instance's
    uint32_t children;          // internal: position of the first child | child count << 16 (breadth-first: a node's children sit side by side)
};
struct FlatLeaf { float group[4], child[4]; float childParentError, pad[3]; };      // 48 B: the leaf's group sphere, its refined group's sphere and error
struct InstanceWalk { uint32_t flatBase, flatCount /* 0: the level walk */, bitBase, skinned; };                                    // 16 B per mesh instance
static_assert(sizeof(FlatNode) == 64 && sizeof(FlatLeaf) == 48 && sizeof(InstanceWalk) == 16, "flat traversal records");
struct NodeRecord { uint32_t instanceIndex, nodeIdPacked; };                       // 8 B (single view)
struct BucketRecord {                                                                 // 32 B
    uint32_t instanceIndex, groupIdPacked, meshletIndexAndCount, pageSlabDescriptorIndex;
    uint32_t pageSlabByteOffset, firstBit, pad0, pad1;
};
struct TempVisible { uint4 packed; uint32_t bit, pad0, pad1, pad2; };                // 32 B
// Round 6: what the library keeps per meshlet BESIDE the reference's descriptor (whose only bound is a sphere): the object-space box of its vertices, made once per
// brmi_setup from the page contents (k_meshlet_boxes).  Not part of the data contract; indexed by pageBoxBase[slab * 1024 + page] + meshlet.
struct MeshletBox { float lo[3]; uint32_t valid; float hi[3]; uint32_t pad; };       // 32 B
struct PageRef { uint32_t slab, byteOffset, boxBase, meshletCount; };                 // one resident page (brmi_set_scene walks the page map)
struct HeldRecord { uint32_t clusterIndex, boxIndex, perObjectIndex, vertsTris /* vertices | triangles << 16 */; };   // 16 B: a visible cluster the phase-1 rasteriser does not take until the re-test has looked at it
static_assert(sizeof(MeshletBox) == 32 && sizeof(PageRef) == 16 && sizeof(HeldRecord) == 16, "draw-list records");
constexpr uint32_t OBJ_CONST_FLOATS = 56;      // per object: MVP (16), objectToClip (16), modelViewZ (4), previous frame's MVP (16) and modelViewZ (4)

// Everything the rasteriser and the G-buffer pass need to start on a visible cluster, resolved once by the compaction
// kernel (one lane per cluster, all in flight): the per-cluster chain cluster -> slab -> page header -> meshlet descriptor
// -> instance -> mesh / object is 6 dependent HBM round trips when every consumer walks it itself.
struct ClusterSetup {                 // 64 B
    const uint8_t* posBase; const uint8_t* triBase; const uint8_t* nrmBase;
    uint32_t counts;                  // vertCount | triCount << 8 | positionFormat << 16 | reverseWinding << 24 | skinned mesh << 25 |
                                      // page has joints << 26 | page has weights << 27
    uint32_t perObjectIndex, instanceIndex, viewId, materialDataIndex, normalMatrixIndex;
    uint32_t vertBase, triBase32;     // first ResolveVertex / ResolveTriangle of the cluster in the resolve arena (BRMI_ARENA_NONE: none)
    int32_t jointDelta, weightDelta;  // byte offsets of the cluster's joint / weight arrays (32 B per vertex) relative to nrmBase
};
constexpr uint32_t BRMI_CS_SKINNED = 1u << 25, BRMI_CS_JOINTS = 1u << 26, BRMI_CS_WEIGHTS = 1u << 27;
constexpr uint32_t BRMI_CS_ALPHA = 1u << 28, BRMI_CS_TEXTURED = 1u << 29, BRMI_CS_COLOR = 1u << 30;    // the cluster's material is alpha tested / samples textures
// per-frame tables and per-material constants of the shading pass (brmi_frame.hip fills them, brmi_light.hip reads them)
struct AxisEntry { float uv; uint32_t tile; };       // per column / row: (i + 0.5) / res and the light-cluster tile index
struct ShadeTables { AxisEntry* x; AxisEntry* y; float* sliceStart; };
struct MatConst {
    float baseWeight, specularWeight, specR, specG, specB, weightedSpecularIor, dielF0Scalar, coatF0Scalar, coatIor, coatDarkening, baseDiffuseRoughness, f90Diel;
    // OpenPBRDiffuseEON with the material's diffuse roughness folded in (tolerance-level re-association, brmi_light.hip):
    // fon_dir_albedo(mu) = fonA + mc (fonK[0] + mc (fonK[1] + mc (fonK[2] + mc fonK[3]))), mc = 1 - mu
    float fonA, fonK[4], eonSingleScale /* A / pi */, eonAvgE, eonOneMinusAvgE, eonInvDen /* 1 / max(1e-4, 1 - avgE) */, dielF0[3];
    // dielF0 = sat(specularColor * dielF0Scalar), f90Diel = sat(dot(dielF0, 50 * 0.33)): the dielectric lobe's Fresnel ends depend on the material alone (round 5)
};
static_assert(sizeof(MatConst) == 96, "six float4");
constexpr uint32_t BRMI_ARENA_NONE = 0xFFFFFFFFu;
// resolve arena: per-vertex and per-triangle tables of the visible clusters (brmi_resolve.hip)
struct ResolveVertex { float px, py, pz, nx, ny, nz; };                                   // 24 B: object-space position, decoded normal
struct ResolveTriangle { float n0x, n0y, invW0, ddx[3], ddy[3], ddxSum, ddySum; uint32_t indices; };   // 48 B: triangle part of CalcFullBary
struct MaterialWords { uint32_t albedo, metallicRoughness; unsigned long long coat, emissive, fuzz; float opIndexF; uint32_t pad; };   // 40 B
static_assert(sizeof(ResolveVertex) == 24 && sizeof(ResolveTriangle) == 48 && sizeof(MaterialWords) == 40, "table layouts");
static_assert(sizeof(ClusterSetup) == 64, "one cache line");

// Linear-depth mip chain as the occlusion test sees it.  Mip 0 is the tiled LinearDepthMap itself (texels outside
// width x height read as "empty"), mips >= 1 are row-major arrays inside BRMI_RES_HZB at mipOffset[mip] floats.
constexpr uint32_t kMaxHzbMips = 16;
struct HzbDesc {
    const float* depth; float* mips;
    uint32_t width, height, tilesX, mipCount, paddedW, paddedH;
    uint32_t rowLo, rowHi;          // rows this GPU renders (multi-GPU band); every other row reads as empty
    StripeMap stripes;              // interleaved partition: the chain lives in surface rows, the occlusion test maps the frame's rows onto them
    uint32_t mipOffset[kMaxHzbMips];
};

// ---- Round 6: is everything a meshlet can draw behind the depth chain `hzb`?  (brmi_cull.hip: the draw list's prediction, against the previous frame's chain with the
// previous frame's matrices; brmi_raster.hip: the re-test that decides, against the chain of the keys the draw list left, with the frame's own.)  CONSERVATIVE for the
// keys the chain was built from: `true` means no triangle of the meshlet can win a pixel against them.
//   * rectangle: floor(min - g) .. floor(max + g) of the eight box corners' pixel coordinates (a vertex lies inside the box, the projection of a box in front of the
//     eye plane inside the hull of its corners; g = a quarter pixel plus 2^-20 of the largest clip-space term in pixels at the nearest corner covers the rounding of
//     the rasteriser's evaluation and of this one, whatever the association), clamped like the rasteriser's boxes (scissor, surface);
//   * depth: the nearest corner's -viewZ less 2^-17 of itself and 2^-18 of the largest term of the dot product (vertex depths are convex combinations of corner
//     depths; a key's depth a convex combination of vertex depths, one mantissa bit dropped);
//   * every texel of the rectangle at the finest mip (>= 1) where it is at most maxTexels x maxTexels texels; a texel is the FARTHEST key depth of its pixels ("empty"
//     where one has no key), so "all texels nearer than the box" is "all keys nearer than any key of the meshlet";
//   * a box that reaches the eye plane (or any NaN) says false; an empty rectangle (off screen) says true: nothing to draw.
#ifndef BRMI_BOX_ROWS
#define BRMI_BOX_ROWS 2      // (1 / 2 / 4 / 8: Bistro-class in flight 0.4788 / 0.4721 / 0.473 / 0.475 ms, dense 0.6307 / 0.6232 / 0.623 / 0.6252, Zorah-class 2.403 / 2.399 / 2.398 / 2.397)
#endif
struct BoxViewport { float width, height, minX, minY; int x0, y0, x1, y1; };      // (ndc + 1) / 2 * width + minX as the rasteriser has it; the clamp in pixels
__device__ __forceinline__ bool box_behind_chain(const HzbDesc& hzb, const MeshletBox& bx, const float* mvpRows /* 16 floats, object -> clip */, const float* mvzRow /* 4 floats, -viewZ = -dot */,
                                                 const BoxViewport& vp, uint32_t maxTexels) {
    // per-axis products: corner (i, j, k) = X[i] + Y[j] + Z[k] + row 3 in the columns x, y, w of the matrix, and in the depth's dot product
    float X[2][4], Y[2][4], Z[2][4];
    const int col[3] = {0, 1, 3};
    float magC = 0.0f, mag = fabsf(mvzRow[3]);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const float px = s ? bx.hi[0] : bx.lo[0], py = s ? bx.hi[1] : bx.lo[1], pz = s ? bx.hi[2] : bx.lo[2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            X[s][c] = px * mvpRows[0 + col[c]]; Y[s][c] = py * mvpRows[4 + col[c]]; Z[s][c] = pz * mvpRows[8 + col[c]] + mvpRows[12 + col[c]];
            magC = fmaxf(magC, fmaxf(fmaxf(fabsf(X[s][c]), fabsf(Y[s][c])), fmaxf(fabsf(pz * mvpRows[8 + col[c]]), fabsf(mvpRows[12 + col[c]]))));
        }
        X[s][3] = px * mvzRow[0]; Y[s][3] = py * mvzRow[1]; Z[s][3] = pz * mvzRow[2] + mvzRow[3];
        mag = fmaxf(mag, fmaxf(fmaxf(fabsf(X[s][3]), fabsf(Y[s][3])), fabsf(pz * mvzRow[2])));
    }
    float sx0 = 3.0e38f, sy0 = 3.0e38f, sx1 = -3.0e38f, sy1 = -3.0e38f, d = 3.0e38f, w = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int i = k & 1, j = (k >> 1) & 1, l = k >> 2;
        const float cx = (X[i][0] + Y[j][0]) + Z[l][0], cy = (X[i][1] + Y[j][1]) + Z[l][1], cw = (X[i][2] + Y[j][2]) + Z[l][2];
        const float inv = __builtin_amdgcn_rcpf(cw);      // (1 ulp: inside the guard)
        const float sx = (cx * inv + 1.0f) * 0.5f * vp.width + vp.minX, sy = (1.0f - cy * inv) * 0.5f * vp.height + vp.minY;
        sx0 = fminf(sx0, sx); sx1 = fmaxf(sx1, sx); sy0 = fminf(sy0, sy); sy1 = fmaxf(sy1, sy);
        d = fminf(d, -((X[i][3] + Y[j][3]) + Z[l][3])); w = fminf(w, cw);
    }
    if (!(w > 1e-6f && d > 0.0f && fabsf(sx0) < 1.0e9f && fabsf(sx1) < 1.0e9f && fabsf(sy0) < 1.0e9f && fabsf(sy1) < 1.0e9f)) return false;
    const float guard = 0.25f + magC * __builtin_amdgcn_rcpf(w) * 9.5367431640625e-7f * (vp.width + vp.height);
    const int x0 = max((int)floorf(fmaxf(sx0 - guard, -1.0e9f)), vp.x0), y0 = max((int)floorf(fmaxf(sy0 - guard, -1.0e9f)), vp.y0);
    const int x1 = min((int)floorf(fminf(sx1 + guard, 1.0e9f)), vp.x1), y1 = min((int)floorf(fminf(sy1 + guard, 1.0e9f)), vp.y1);
    const float nearSafe = d - (d * 7.62939453125e-6f + mag * 3.814697265625e-6f);
    if (!(nearSafe > 0.0f)) return false;
    if (x0 > x1 || y0 > y1) return true;
    uint32_t mip = 1u;
    while (mip + 1u < hzb.mipCount && (((x1 >> mip) - (x0 >> mip) + 1) > (int)maxTexels || ((y1 >> mip) - (y0 >> mip) + 1) > (int)maxTexels)) mip++;
    if (mip >= hzb.mipCount) return false;
    const uint32_t mw = max(hzb.paddedW >> mip, 1u), mh = max(hzb.paddedH >> mip, 1u);
    if (((x1 >> mip) - (x0 >> mip) + 1) > (int)maxTexels || ((y1 >> mip) - (y0 >> mip) + 1) > (int)maxTexels) return false;      // (not even the last mip: a chain shorter than the surface)
    const float* m = hzb.mips + hzb.mipOffset[mip];
    const int tx0 = x0 >> mip, tx1 = min(x1 >> mip, (int)mw - 1), ty1 = min(y1 >> mip, (int)mh - 1);
    // BRMI_BOX_ROWS rows of texels at a time (their loads side by side); a cluster that shows usually says so in its first rows
    for (int y = y0 >> mip; y <= ty1; y += BRMI_BOX_ROWS) {
        float farthest = 0.0f;
#pragma unroll
        for (int r = 0; r < BRMI_BOX_ROWS; r++) {
            const int yy = min(y + r, ty1);
            for (int x = tx0; x <= tx1; x++) farthest = fmaxf(farthest, m[(size_t)yy * mw + (uint32_t)x]);
        }
        if (!(farthest < nearSafe)) return false;
    }
    return true;
}

// Scenes whose materials all pack to the same coat (fuzz) G-buffer word: the word and whether the plane currently holds it everywhere
// (brmi_frame.hip: job_layer_uniform, k_fill_layer_planes; brmi_resolve.hip skips the plane's stores)
// The frame's main camera and per-frame record as the constants kernel found them (brmi_frame.hip).  The resolve + shading half of a frame reads
// THESE, never the caller's buffers: with frames in flight the host rewrites its camera buffer for frame k + n while frame k is still being
// shaded on the other stream; the constants kernel of the pass's next frame runs behind the frameDone wait, so the copy is stable for as
// long as anything reads it.  perFrame.mainCameraIndex of the copy is 0 (the copy holds that one camera).
struct FrameSnapshot { brmi_per_frame perFrame; brmi_camera camera; };

struct LayerUniform { unsigned long long coatWord, fuzzWord, coatFilledWord, fuzzFilledWord; uint32_t coatUniform, fuzzUniform, coatFilled, fuzzFilled; };

struct Workspace {     // byte offsets into BRMI_RES_WORKSPACE
    uint64_t counters, frontierA, frontierB, buckets, tempVisible, bitmask1, bitmask2, blockDirty, chainDirty, wordPrefix, blockSums,
             instanceBitBase, segPrefix, meshLevelWidth, scanAgg, flatNodes, flatLeaves, instanceWalk, planes, replayNodes, replayBuckets, lightVS, lightMeta, clusterPages, clusterHits, pageTotal, lightHitMasks, binCounts, binRecords, binOverflow, binPlan, binItems, binScratch, clusterSetup, resolveVerts, resolveTris, matWords, shadeTables, lutF, frameConst, objConst, matConst, deferredPixels,
             frameSnapshot, debugStamps, clusterUv, binAlpha, overflowAlpha, resolveUVs, resolveColors, alphaMats, shadeRows, shadeAvgs, ggxQuads, shadeLights, clusterList, listEntries, listRecords, layerUniform,
             meshletBoxes, pageBoxBase, pageRefs, drawList, heldRecords, lateList, wideQueue, wideAlpha, generalList, bigQueue, bigRuns, frameClearBytes, total;
};

}  // namespace brmi

struct brmi_pass {
    brmi_config cfg{};
    brmi_scene_buffers scene{};
    bool haveScene = false, setupDone = false, updated = false;
    void* res[BRMI_RES_COUNT] = {};
    uint64_t resBytes[BRMI_RES_COUNT] = {};
    uint64_t resNeed[BRMI_RES_COUNT] = {};
    uint32_t tilesX = 0, tilesY = 0;
    uint64_t paddedPixels = 0;
    uint32_t bandY0 = 0, bandY1 = 0;
    brmi::StripeMap stripes{0, 0, 0, 0};     // brmi_config::stripe*: count > 1 = interleaved partition, surfaces hold the owned rows only
    uint32_t frameHeight() const { return stripes.count > 1u ? stripes.fullHeight : cfg.height; }   // rows of the frame the cameras describe
    float bandPlaneTop[3] = {0, 0, 0}, bandPlaneBottom[3] = {0, 0, 0};
    uint64_t bandFirstPixel = 0, bandPixelCount = 0;   // tiled index range covering the band's tile rows
    uint32_t maxLevels = 1;
    uint32_t spillWidth = 1024;      // meshes with a BVH level wider than this go to the level kernels below their top (<= the widest LDS frontier; BRMI_SPILL_WIDTH)
    uint32_t spillLevels = 0;        // level-kernel launches the widest meshes still need below the point where the LDS walk hands them over
    uint32_t minLevelWidth = 0;      // narrowest such width over the meshes
    std::vector<uint32_t> hostMeshLevelWidth;   // per mesh metadata entry
    uint32_t maxLevelWidth = 0;      // widest BVH level of any mesh (decides between the per-instance and the per-level traversal)
    // Round 6, the draw list: per-meshlet boxes of the scene's resident pages and whether this pass holds clusters back (brmi_raster.hip)
    std::vector<brmi::PageRef> hostPageRefs; std::vector<uint32_t> hostPageBoxBase; uint32_t totalBoxes = 0;
    bool holdEnabled = false;        // the pass can hold clusters back: occlusion culling on, no band / interleaved partition, boxes available (BRMI_TUNING hold_clusters=0: off)
    bool holdThisFrame = false;      // this frame's phase 1 made a draw list (launch_cull -> launch_raster)
    uint32_t holdMinClusters = 16384; // hold only on frames whose last known visible-cluster count is at least this (BRMI_TUNING hold_min_clusters): below it the re-test, the late
                                     // pass and the chain's second look cost what the rasteriser saves (Bistro-class, 10 k clusters: +25 us; profiles/r06_experiments.md)
    uint32_t lateDirectMax = 128;    // the late pass walks every triangle directly (no records, plan, bins) while the last known late count is at most this (BRMI_TUNING late_direct_max;
                                     // 1024: the fast camera path's raster stage 0.338 -> 0.466 ms -- late clusters are the ones a moving view uncovers, near and large)
    uint32_t holdFloor = 512;        // ... and at least this many clusters were visible (hold_floor)
    uint32_t holdStillMax = 8;       // frames of fewer than holdMinClusters clusters hold clusters back all the same while phase 2 of the last frame the host has seen drew fewer clusters than
                                     // this (a still camera: the prediction is then exact and the late pass empty -- Bistro-class 0.505 -> 0.4815 ms in flight, San-Miguel-class 0.824 -> 0.776,
                                     // Sponza-class unchanged, the skinned leg + 2.7 %; with the camera moving the late pass costs more than the rasteriser saves: path 0.545 -> 0.594); 0: never
    uint32_t holdMaxTexels = 8;      // the prediction reads at most this many texels per axis of the previous chain (BRMI_TUNING hold_max_texels; Zorah-class: 4 / 6 / 8 hold 41.6 / 47.7 / 49.6 %
                                     // of the list, serial frame 2.574 / 2.53 / 2.50 ms against 2.948 without; a prediction finer than the re-test sends the difference to the late pass: 2.75)
    uint32_t retestMaxTexels = 8;    // ... and the re-test of this frame's
    uint32_t chainStripLo = 0xFFFFFFFFu, chainStripHi = 0u;   // 32-row strips of the chain that hold texels of a band (a band that moves: the next full build also resets the strips it left)
    uint32_t wideCapacity = 16384;   // triangles the wide queue holds (BRMI_TUNING wide_capacity; 0 = every triangle is emitted by its own wave); beyond it the wave emits them itself
    uint32_t wideEntries = 128;      // ... and, while it is launched, triangles of more bin entries than this go to it (BRMI_TUNING wide_entries; without it a lane keeps up to 512)
    // the lean rasteriser (brmi_raster.hip, k_raster<false, true>): phase 1's main launch of frames whose last such launch the host has seen had at least leanMinClusters
    // clusters (0 = never), while at most leanMaxGeneralPct % of them come back for the general launch (else off for 64 frames).  BRMI_TUNING lean_min_clusters,
    // lean_max_general_pct, lean_grid.
    uint32_t leanMinClusters = 32768, leanMaxGeneralPct = 40, leanGrid = 24576, leanEmitGrid = 4096, leanWideEntries = 16, leanQueue = 1u << 20;      // leanQueue: entries of the binned-triangle queue (96 B each; lean_queue)
    bool leanActive = false, leanLastLaunch = false; uint32_t leanRetryIn = 0;
    uint32_t wideMinTriangles = 4;   // the wide pass is launched while the last frame the host has seen queued at least this many (BRMI_TUNING wide_min_triangles)
    bool chainBuiltInRaster = false; // this frame's phase-1 rasteriser stage built the chain itself (before its re-test): the build that follows redoes the late pass's blocks only
    bool sceneHasVertexColors = false;                           // some mesh's pages carry vertex colours (perMesh.vertexFlags bit 0)
    uint32_t sceneUvSets = 1;      // UV sets the texture slots of the scene's materials name (brmi_set_scene)
    bool sceneHasAlphaTest = false, sceneHasTextures = false, sceneHasParallax = false;   // some material is alpha tested / samples a texture (brmi_set_scene)
    bool sceneHasCoat = true, sceneHasFuzz = true;   // some OpenPBR material has a coat / fuzz layer (brmi_set_scene)
    std::vector<float> sliceStartHost; float sliceKey[3] = {0, 0, 0}; uint32_t sliceKeyN[2] = {0, 0};   // slice starts of the light-cluster grid and the inputs they were made from
    // Phase 2 of a frame usually draws nothing or a few dozen clusters; then its triangles all take the row re-deal with global atomics (one
    // launch instead of k_raster + plan + bins).  Which it is, the host learns from the frames before: the ranking kernel of phase 2 also stores
    // the survivor count in a host-mapped word that launch_raster reads without waiting (any value is safe: both paths draw the same keys).
    uint32_t* phase2FeedbackHost = nullptr; uint32_t* phase2FeedbackDev = nullptr;      // word 0: phase-2 survivors; word 1: frames like the last ones should resolve without the per-cluster tables (resolve_inline_frame)
    bool ensureFeedback() {          // the 64 B host-mapped block, made at first use; false: none (the callers then take their always-safe paths)
        if (phase2FeedbackHost) return true;
        if (hipHostMalloc(reinterpret_cast<void**>(&phase2FeedbackHost), 64, hipHostMallocMapped) != hipSuccess) { phase2FeedbackHost = nullptr; return false; }
        phase2FeedbackHost[0] = 0xFFFFFFFFu; phase2FeedbackHost[1] = 0u;           // unknown: the general paths (word 1: the per-cluster resolve tables)
        for (int k = 2; k < 16; k++) phase2FeedbackHost[k] = 0u;                  // word 2: phase 1's bucket records, word 3: phase 1's visible clusters (launch sizes of the frames that follow)
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&phase2FeedbackDev), phase2FeedbackHost, 0) != hipSuccess) { (void)hipHostFree(phase2FeedbackHost); phase2FeedbackHost = nullptr; phase2FeedbackDev = nullptr; return false; }
        return true;
    }
    uint32_t phase2DirectMax = 128;   // BRMI_PHASE2_DIRECT_MAX: direct rasterisation while the last known phase-2 count is at most this (0: always bins).  Round 5: 256 -> 128 -- a camera that
                                      // moves fast leaves phase 2 two to four hundred clusters of large near triangles, which the row re-deal walks in few lanes: path_fast raster2 0.12 -> 0.065 ms
                                      // (0.648 -> 0.625 ms per frame); the slow path (30 - 80 clusters) and the still camera keep the one-launch form (sweep: profiles/r05_experiments.md)
    uint32_t clearRiderBlocks = 8192; // single-wave workgroups of the visibility clear that ride on the traversal launch (BRMI_CLEAR_RIDER_BLOCKS)
    bool splitFrame = false;         // brmi_execute_split with two streams: this frame's launches share the chip with another frame's
    uint32_t shadeSlabs = 0; brmi_slab_fn shadeSlabFn = nullptr; void* shadeSlabUser = nullptr;      // brmi_set_shade_slabs
    bool frameWaitsIssued = false;   // brmi_execute_split has issued this frame's cross-stream waits (the stage entry points it calls skip theirs)
    bool wideFlat = true;            // BRMI_FLAT_WIDE=0: hierarchies of more than 256 nodes take the level walk
    bool anyWideFlat = false, allMeshesFlat = false;      // brmi_set_scene: some mesh has 257 .. 8192 nodes / every mesh has flat tables
    uint32_t flatMaxDepth = 1;       // levels of the deepest flat hierarchy (launches of the level-synchronous flat traversal, brmi_cull.hip: k_cull_flat_level)
    uint32_t flatLevelsMinDraws = 16384;   // BRMI_FLAT_LEVELS_MIN_DRAWS: scenes with at least this many draws (all hierarchies flat) take the level-synchronous flat traversal in phase 1
    bool scanChained = true; uint32_t scanEpoch = 0;      // the survivor ranking as one launch (BRMI_SCAN_CHAINED=0: three)
    bool packedFlat = true;          // BRMI_FLAT_PACKED=0: one draw per wave of the traversal
    uint32_t shadeGridShared = 10240; // workgroups of k_shade<0, 3> (BRMI_SHADE_GRID_SHARED): shorter-lived than the stand-alone 8192 so that the other frame's small geometry launches find slots sooner (Bistro-class period 6144 / 8192 / 10240 / 12288: 0.547 / 0.539 / 0.530 / 0.531 ms; Sponza-class, whose geometry half is short: 0.386 / 0.398 / 0.398 / 0.397)
    uint32_t gbufferGridShared = 4096; // workgroups of the lean k_gbuffer in a split frame (BRMI_GBUFFER_GRID_SHARED)
    bool shadeSharesChip = false;    // brmi_execute_split with two streams: the shading half runs beside another frame's geometry half
    bool lightGridDone = false;      // this frame's light clustering ran inside the culling pass's launches
    bool clearVisibilityWithClusterCull = false;      // brmi_execute_split on two streams: the clear rides on k_cull_clusters instead (brmi_cull.hip: ClearRide)
    bool clearFrameStateWithConstants = false, clearVisibilityWithTraversal = false;   // brmi_execute: no clear launch (brmi_frame.hip, brmi_cull.hip)
    bool seedInHzbTail = false, phase2Seeded = false;           // brmi_execute: the tail of the phase-1 depth-chain build also seeds phase 2 (k_seed_phase2's work)
    bool fuseFrameClear = false, frameStateCleared = false;   // brmi_execute: the visibility clear also clears the culling pass's frame state
    bool forceLevelKernels = false;  // BRMI_CULL_LEVEL_KERNELS=1: always use the per-level kernels (tests, very wide hierarchies)
    uint64_t totalBits = 0; uint32_t totalWords = 0, scanBlocks = 0;
    uint32_t numLightClusters = 0, lightPagePool = 0;
    uint32_t binOverflowPerStripe = 1u << 14;       // 64 stripes x 16384 records x 64 B = 64 MB
    uint32_t binMinSlice = 1024, binSharedSlice = 512, binGrid = 1024;   // k_raster_bins: records one workgroup walks alone / per slice of a larger bin, workgroups of the pool (BRMI_BIN_MIN_SLICE, BRMI_BIN_SHARED_SLICE, BRMI_BIN_GRID)
    uint32_t binScratchTiles = 2048, binItemCapacity = 0;   // k_raster_bins: scratch tiles for bins several workgroups share (BRMI_BIN_SCRATCH_TILES), work items
    uint32_t binsX = 0, binsY = 0, binCapacity = 16384;  // raster bins: 256 px x 16 rows, binCapacity records of 64 B each (BRMI_BIN_CAPACITY): 2 GB at 4K, walked in slices of 1024.
                                                         // Round 5: 8192 -> 16384 -- rank 0 of the 8-GPU San-Miguel-class frame owns a double chunk on the horizon whose bins take
                                                         // > 8192 records of alpha-tested leaves; the overflow queues' global-atomic walk made it 1.67 ms against the others' 1.05 (1.18 now)
    uint32_t deferredStripeCapacity = 0;   // entries per deferred-pixel stripe
    uint32_t resolveCapacity = 0;   // vertices (and triangles) the resolve arena holds
    uint32_t rasterGrid = 8192;  // single-wave workgroups of k_raster (BRMI_RASTER_GRID)
    int rasterDebug = 0;         // BRMI_RASTER_DEBUG (experiments; non-zero gives wrong images)
    int bigTriAreaDense = 32; uint32_t denseClusterCount = 6144;   // frames with that many visible clusters bin from this area on (BRMI_BIG_TRI_AREA sets both)
    int bigTriArea = 32, bigTriAreaAlpha = 32;        // clamped-bbox pixels above which a triangle is binned (BRMI_BIG_TRI_AREA; 64 until the bins kernel walked sorted slices: Sponza-class raster 0.096 -> 0.090 ms at 32)
    uint32_t hzbMipCount = 0; std::vector<uint64_t> hzbMipOffsets; std::vector<uint32_t> hzbMipW, hzbMipH;   // [mip]; offsets in floats, mip 0 unused
    bool hzbValid = false;       // a chain built from a finished frame exists (phase 1 of the next frame tests against it)
    brmi_pass* history = nullptr;    // brmi_set_history_source: the pass whose chain phase 1 tests against (frames in flight); null = this pass's own
    hipEvent_t chainReady = nullptr; // recorded by brmi_execute after the frame's last chain build when another pass may be reading it
    bool chainRecorded = false;
    brmi_stream chainStream = nullptr;   // the stream chainReady was last recorded on
    hipEvent_t cullDone = nullptr;      // brmi_execute_split: the phase-1 cluster list is final (the resolve tables of those clusters are made beside the rasteriser)
    hipEvent_t geometryDone = nullptr, frameDone = nullptr;   // brmi_execute_split: geometry half -> shading half, and the frame's end on the shading stream
    bool frameDoneRecorded = false;
    std::vector<brmi_pass*> historyUsers;   // passes whose `history` is this pass (unlinked when it is destroyed)
    bool layerPlanesDirty = false;   // brmi_setup: the next constants launch is followed by k_fill_layer_planes
    bool layerPlanesUniform = false; // ... which found one coat and one fuzz word for the whole scene and filled both planes (read back once)
    bool chainDirtyTracked = false;  // this frame's phase-2 rasteriser recorded the blocks it may have touched (the second chain build skips the others)
    int resolveInlineMode = -1;      // BRMI_TUNING=resolve_inline: -1 by the frame, 0 never, 1 always (tests run scenes both ways)
    bool inlineResolve = false;      // this frame's G-buffer pass derives a pixel's triangle from the cluster's own data (no per-cluster tables, no setup launch): frames of more
                                     // triangles than pixels, decided when the frame's culling starts (resolve_inline_frame)
    bool resolveSetupDone = false;   // brmi_execute_split: the per-cluster tables were made on the geometry stream
    bool depthFinal = false;         // brmi_execute: the depth map is final before the G-buffer kernel runs (it skips its depth store)
    const brmi_pass* chainOwner(uint32_t phase) const { return (phase == 1 && history) ? history : this; }
    brmi::HzbDesc hzbDesc() const;
    std::vector<uint32_t> hostInstanceBitBase, hostSegPrefix;
    std::vector<brmi::FlatNode> hostFlatNodes; std::vector<brmi::FlatLeaf> hostFlatLeaves; std::vector<brmi::InstanceWalk> hostInstanceWalk;   // flat traversal tables (brmi_set_scene)
    brmi::Workspace ws{};
    brmi_camera camHost{};
    brmi_per_frame pfHost{};
    std::vector<float> planesHost;
    static constexpr uint32_t kEventRing = 32;     // per-stage event pairs of the last kEventRing frames
    hipEvent_t evStart[BRMI_STAGE_COUNT][kEventRing] = {}, evStop[BRMI_STAGE_COUNT][kEventRing] = {};
    uint32_t evCount[BRMI_STAGE_COUNT] = {};       // recordings since the last brmi_stage_times()
    uint32_t executesSinceTimes = 0;               // brmi_execute calls since then (a stage recorded twice per frame reports its per-frame cost)
    uint32_t updateSerial = 1, constantsSerial = 0;  // brmi_update / brmi_set_scene bump updateSerial; the frame constants follow
    uint32_t shadeSerial = 0;                        // parity selects the deferred-pixel counter of a shading call
    bool eventsCreated = false;
    uint32_t timedStages = 0xFFFFFFFFu;            // brmi_set_timed_stages
    std::string err;

    template <typename T> T* wsPtr(uint64_t off) const { return reinterpret_cast<T*>(static_cast<uint8_t*>(res[BRMI_RES_WORKSPACE]) + off); }
    uint32_t* counters() const { return wsPtr<uint32_t>(ws.counters); }
};

namespace brmi {

int fail(brmi_pass* p, int code, const char* fmt, ...);
// ONE environment variable, BRMI_TUNING="key=value,key=value" (DESIGN.md 6b).  tuning(): the keys a user or a test may set (sizes that force the rare
// paths); experiment(): keys that only exist in builds with -DBRMI_EXPERIMENTS (A/B switches of the measurements in profiles/*_experiments.md:
// they can drop events or launches and give wrong images) -- a product build returns the default whatever the environment says.
long tuning(const char* key, long def);
long experiment(const char* key, long def);
#define BRMI_HIP(p, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return brmi::fail((p), BRMI_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define BRMI_LAUNCH_CHECK(p, what) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return brmi::fail((p), BRMI_ERR_HIP, "launch %s: %s", (what), hipGetErrorString(e_)); } while (0)

// stage launchers (one translation unit each)
int ensure_frame_constants(brmi_pass* p, hipStream_t s);
brmi_scene_buffers shading_scene_of(const brmi_pass* p);      // p->scene with cameras / perFrame pointing at the pass's FrameSnapshot
ShadeTables shade_tables_of(const brmi_pass* p);
int launch_clear(brmi_pass* p, hipStream_t s);
int launch_cull(brmi_pass* p, uint32_t phase, hipStream_t s);
int launch_raster(brmi_pass* p, uint32_t phase, hipStream_t s);
int launch_depth_copy(brmi_pass* p, hipStream_t s);
int launch_hzb(brmi_pass* p, hipStream_t s, bool fromVisibility, bool onlyIfPhase2Drew);
int launch_gbuffer(brmi_pass* p, hipStream_t s);
int launch_resolve_setup(brmi_pass* p, hipStream_t s, uint32_t part);
uint32_t resolve_inline_ratio(const brmi_pass* p);
bool resolve_inline_frame(brmi_pass* p);
int launch_light_clustering(brmi_pass* p, hipStream_t s);
int launch_expand_luts(brmi_pass* p, hipStream_t s);
int launch_meshlet_boxes(brmi_pass* p, hipStream_t s);      // brmi_setup: the per-meshlet boxes of the draw list's tests, from the page contents
int launch_shade(brmi_pass* p, hipStream_t s);

}  // namespace brmi
#endif
