// brmi_cull.hip -- hierarchical culling (K1-K3) for gfx950.
//
// What it computes is the reference's chain
//   PureComputeObjectCullCS / PureComputeTraverseFrontierCS   BR/shaders/ClusterLOD/computeCulling.hlsl:103-531
//   ClusterCullBody                                           BR/shaders/ClusterLOD/workGraphCulling.hlsl:2398-3330
// but not how the reference schedules it.  MI355X-first differences:
//   * HIP has no ExecuteIndirect, and a dependent chain of tiny dispatches is latency-bound
//     (SURVEY.md 8a-2).  The default path walks each instance's BVH inside ONE launch with the
//     frontier in LDS (k_cull_hierarchy: one wave64 per draw / replayed node).  Hierarchies whose
//     widest level exceeds the LDS frontier fall back to one fixed-size grid-stride launch per
//     level that reads its record count from HBM (k_cull_instances / k_traverse); either way the
//     chain is a static launch sequence with no host round trip.
//   * The reference appends survivors with wave ballots into one buffer, so the cluster index
//     that ends up in the visibility key depends on atomic ordering.  Here survivors set one bit
//     in a per-(instance, segment, meshlet) bitmask, a popcount scan ranks the bits, and a scatter
//     places each survivor at its rank: the visible-cluster list is identical run to run
//     (canonical order: instance, segment, meshlet) with one atomic per wave, not per survivor.
//   * Frontier / bucket appends are wave-aggregated (one atomic per wave64).
#include <type_traits>
#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_lightgrid.h"
#include "brmi_texture.h"

namespace brmi {

struct CullArgs {
    brmi_scene_buffers sc;
    uint32_t* counters;
    const uint32_t* instanceBitBase;
    const uint32_t* segPrefix;
    uint32_t recordCapacity, visibleCapacity, factor, phase;
    // occlusion culling: phase 1 tests against the previous frame's chain with the previous transforms and appends what it
    // rejects to the replay buffers; phase 2 re-tests those against the chain of the depth phase 1 just rasterised
    uint32_t occlusion;
    uint32_t frontier0Counter, bucketCounter;   // counter words of the level-0 frontier and of the bucket array in use
    NodeRecord* replayNodes; BucketRecord* replayBuckets;
    HzbDesc hzb;
    // multi-GPU row band: two view-space planes through the eye bounding the band (1 = active)
    uint32_t bandActive; float bandTop[3], bandBottom[3];
    const float4* bandPlanes;        // the same two planes in memory (frameConst[3]): what the instance / node tests read
    StripeMap stripes;      // interleaved partition: ownership test of the cluster cull
    // mixed traversal: meshes whose widest BVH level fits the LDS frontier are walked by k_cull_hierarchy (one wave per instance, one launch),
    // the few wider ones by the level-per-launch kernels (all lanes of the chip on one level); the latter skip instances narrower than this
    const uint32_t* meshLevelWidth; uint32_t levelKernelsWidthLo;
    const FlatNode* flatNodes; const FlatLeaf* flatLeaves; const InstanceWalk* instanceWalk;     // flat traversal of small hierarchies (brmi_internal.h)
    unsigned long long* debugStamps;     // instrumented builds (-DBRMI_TILE_STAMPS)
    uint32_t wideFlat;                   // phase 1: hierarchies of 257 .. 8192 nodes are k_cull_flat_wide's (the walk skips them)
    uint32_t packedFlat;                 // phase 1: the launch's first ceil(draws / 8) waves take eight draws each (hierarchies of <= 8 nodes)
    uint32_t* feedback;                  // host-mapped words (brmi_pass::ensureFeedback) or null: word 2 = phase 1's bucket records (the host sizes the next frames' launches by it)
};

BRMI_DEV f3 to_view_space(f3 c, const m4& model, const m4& view) { return xyz(mul_vm(mul_point(c, model), view)); }

BRMI_DEV bool sphere_outside_frustum(f3 c, float r, const float (*planes)[4]) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float d = dot3(f3{planes[i][0], planes[i][1], planes[i][2]}, c) + planes[i][3];
        if (d < -r) return true;
    }
    return false;
}

BRMI_DEV float projected_error(f3 worldCenter, float worldRadius, float errMesh, float errScale, f3 camPos, float zNear, bool ortho) {
    const float wsErr = errMesh * errScale;
    if (ortho) return wsErr;
    const float dist = length3(worldCenter - camPos);
    const float denom = max2(dist - worldRadius, zNear);
    return wsErr / denom;
}

BRMI_DEV bool refined_child_suppresses(const brmi_scene_buffers& sc, uint32_t groupsBase, uint32_t childLocal, bool hasChild, const m4& model, float scale,
                                       f3 camPos, float zNear, float threshold, bool ortho) {
    if (!hasChild) return false;
    const brmi_lod_group* g = sc.lodGroups + (groupsBase + childLocal);
    const f3 c = xyz(mul_point(f3{g->centerAndRadius[0], g->centerAndRadius[1], g->centerAndRadius[2]}, model));
    const float r = g->centerAndRadius[3] * scale;
    const float eod = projected_error(c, r, g->maxParentError, scale, camPos, zNear, ortho);
    return !(eod < threshold);   // resident: static frame
}

// ceil(log2(x)) clamped to [0, maxMip], evaluated on the float's bits (the oracle's definition: stable next to powers of two)
BRMI_DEV uint32_t ceil_log2_clamped(float x, uint32_t maxMip) {
    if (!(x > 1.0f)) return 0u;
    const uint32_t u = __float_as_uint(x);
    const uint32_t m = ((u >> 23) & 0xFFu) - 127u + ((u & 0x7FFFFFu) ? 1u : 0u);
    return m < maxMip ? m : maxMip;
}

BRMI_DEV float hzb_load(const HzbDesc& h, uint32_t mip, uint32_t x, uint32_t y, uint32_t mipW) {
    if (mip == 0u) return (x < h.width && y >= h.rowLo && y < h.rowHi) ? h.depth[tiled_index(x, y, h.tilesX)] : __uint_as_float(BRMI_DEPTH_EMPTY_BITS);
    return h.mips[h.mipOffset[mip] + (size_t)y * mipW + x];
}

// sphere_screen_extents (Misc/sphereScreenExtents.hlsli:14-31) + OcclusionCullingPerspectiveTexture2D
// (occlusionCulling.hlsli:165-212): screen rectangle of the sphere -> mip whose texels cover it -> four point loads.
BRMI_DEV bool occlusion_culled(const HzbDesc& hzb, const brmi_camera* cam, float p00, float p11, f3 centerVS, float sphereDepth, float radius) {
    const float viewW = (float)cam->depthResX, viewH = (float)cam->depthResY;
    const float px = centerVS.x, py = -centerVS.y, pz = centerVS.z;
    const float rad2 = radius * radius, d = pz * radius;
    const float hv = sqrtf(px * px + pz * pz - rad2);
    const float ha = px * hv, hb = px * radius, hc = pz * hv;
    float L = (ha - d) * p00 / (hc + hb);
    float R = (ha + d) * p00 / (hc - hb);
    const float vv = sqrtf(py * py + pz * pz - rad2);
    const float va = py * vv, vb = py * radius, vc = pz * vv;
    const float B = (va - d) * p11 / (vc + vb);
    const float T = (va + d) * p11 / (vc - vb);
    L = -L; R = -R;
    const float u0 = sat(L * 0.5f + 0.5f), v0 = sat(T * -0.5f + 0.5f), u1 = sat(R * 0.5f + 0.5f), v1 = sat(B * -0.5f + 0.5f);
    const float ax0 = u0 * viewW, ay0 = v0 * viewH, ax1 = u1 * viewW, ay1 = v1 * viewH;
    const float ex = ax1 - ax0, ey = ay1 - ay0;
    const uint32_t mip = ceil_log2_clamped(max2(ex, ey), cam->numDepthMips - 1u);
    const float sx = cam->UVScaleToNextPowerOf2[0], sy = cam->UVScaleToNextPowerOf2[1];
    const float pu0 = u0 * sx, pv0 = v0 * sy, pu1 = u1 * sx, pv1 = v1 * sy;
    const float ssx = max2(sx, 1e-6f), ssy = max2(sy, 1e-6f);
    uint32_t hzbW = (uint32_t)rintf(viewW / ssx), hzbH = (uint32_t)rintf(viewH / ssy);
    hzbW = max(hzbW, 1u); hzbH = max(hzbH, 1u);
    const uint32_t mw = max(hzbW >> mip, 1u), mh = max(hzbH >> mip, 1u);
    uint32_t x0 = min((uint32_t)floorf(pu0 * (float)mw), mw - 1u), y0 = min((uint32_t)floorf(pv0 * (float)mh), mh - 1u);
    uint32_t x1 = min((uint32_t)floorf(pu1 * (float)mw), mw - 1u), y1 = min((uint32_t)floorf(pv1 * (float)mh), mh - 1u);
    if (stripe_on(hzb.stripes)) {
        // Interleaved partition (no counterpart in the reference): the chain is built from this GPU's compact depth surface, so the rectangle's
        // frame rows are mapped onto the surface rows this GPU owns among them -- consecutive surface rows -- and the mip is chosen from THAT
        // extent; a rectangle of extent <= 2^mip spans at most two texels per axis, so the four corner texels cover it.  A rectangle that
        // holds none of this GPU's rows says nothing here (the ownership test of the cluster cull drops such clusters for good).
        const uint32_t H = hzb.stripes.fullHeight;
        const uint32_t r0 = min((uint32_t)floorf(ay0), H - 1u), r1 = min((uint32_t)floorf(ay1), H - 1u);
        const uint32_t f = stripe_first_owned(hzb.stripes, r0), l = stripe_last_owned(hzb.stripes, r1);
        if (l == 0xFFFFFFFFu || f > l) return false;
        const uint32_t vy0 = stripe_vrow(hzb.stripes, f), vy1 = stripe_vrow(hzb.stripes, l);
        const uint32_t smip = ceil_log2_clamped(max2(fabsf(ex) + 1.0f, (float)(vy1 - vy0 + 1u)), hzb.mipCount - 1u);      // (|ex|: the reference's horizontal extents come out swapped; here the test has to be conservative)
        const uint32_t smw = max(hzb.paddedW >> smip, 1u), smh = max(hzb.paddedH >> smip, 1u);
        const uint32_t px0 = min((uint32_t)floorf(min2(ax0, ax1)), hzb.width - 1u), px1 = min((uint32_t)floorf(max2(ax0, ax1)), hzb.width - 1u);
        x0 = min(px0 >> smip, smw - 1u); x1 = min(px1 >> smip, smw - 1u); y0 = min(vy0 >> smip, smh - 1u); y1 = min(vy1 >> smip, smh - 1u);
        const float e0 = hzb_load(hzb, smip, x0, y0, smw), e1 = hzb_load(hzb, smip, x1, y0, smw), e2 = hzb_load(hzb, smip, x1, y1, smw), e3 = hzb_load(hzb, smip, x0, y1, smw);
        return max2(max2(e0, e1), max2(e2, e3)) < sphereDepth - radius;
    }
    if (mip >= hzb.mipCount) return false;
    const float d0 = hzb_load(hzb, mip, x0, y0, mw), d1 = hzb_load(hzb, mip, x1, y0, mw), d2 = hzb_load(hzb, mip, x1, y1, mw), d3 = hzb_load(hzb, mip, x0, y1, mw);
    const float mx = max2(max2(d0, d1), max2(d2, d3));
    return mx < sphereDepth - radius;
}

// phase 1: previous frame's transforms (the chain is the previous frame's depth); phase 2 / replay: current ones
BRMI_DEV bool occlusion_test(const CullArgs& a, const brmi_camera* cam, bool replay, f3 localCenter, float localRadius, f3 currentVS, float currentRadius,
                             const brmi_per_object* obj) {
    if (replay) return occlusion_culled(a.hzb, cam, cam->projection[0][0], cam->projection[1][1], currentVS, -currentVS.z, currentRadius);
    const m4 prevModel = load_m4(&obj->prevModel[0][0]);
    const f3 pc = to_view_space(localCenter, prevModel, load_m4(&cam->prevView[0][0]));
    return occlusion_culled(a.hzb, cam, cam->prevUnjitteredProjection[0][0], cam->prevUnjitteredProjection[1][1], pc, -pc.z, localRadius * max_axis_scale(prevModel));
}

// phase 1's test with the previous model matrix already in registers (the flat traversal requests it together with the current one: behind the
// node tests it was a memory round trip of its own in front of the depth chain's)
BRMI_DEV bool occlusion_test_prev(const CullArgs& a, const brmi_camera* cam, f3 localCenter, float localRadius, const m4& prevModel) {
    const f3 pc = to_view_space(localCenter, prevModel, load_m4(&cam->prevView[0][0]));
    return occlusion_culled(a.hzb, cam, cam->prevUnjitteredProjection[0][0], cam->prevUnjitteredProjection[1][1], pc, -pc.z, localRadius * max_axis_scale(prevModel));
}

// Interleaved partition: does the sphere's screen rectangle (the vertical extents of sphere_screen_extents, two rows of slack) hold a row
// this GPU owns?  Spheres that reach the near plane are kept (the extents are not defined there).
BRMI_DEV bool stripe_rejects(const StripeMap& m, const brmi_camera* cam, f3 centerVS, float radius) {
    const float pz = centerVS.z;
    if (!(-pz - radius > cam->zNear)) return false;
    const float py = -centerVS.y, p11 = cam->projection[1][1];
    const float rad2 = radius * radius, d = pz * radius;
    const float vv = sqrtf(py * py + pz * pz - rad2);
    const float va = py * vv, vb = py * radius, vc = pz * vv;
    const float B = (va - d) * p11 / (vc + vb), T = (va + d) * p11 / (vc - vb);
    const float v0 = sat(T * -0.5f + 0.5f), v1 = sat(B * -0.5f + 0.5f);
    if (!(v0 <= v1)) return false;
    const float H = (float)m.fullHeight;
    const int r0 = max(to_int_sat(floorf(v0 * H)) - 2, 0), r1 = min(to_int_sat(floorf(v1 * H)) + 2, (int)m.fullHeight - 1);
    return stripe_first_owned(m, (uint32_t)r0) > (uint32_t)r1;
}

// Frustum test of an instance's or a hierarchy node's sphere, and -- interleaved partition, round 4 -- the ownership test on top of it: a sphere whose
// screen rows (two rows of slack) hold none of this GPU's rows is not descended.  The cluster cull drops exactly such meshlets anyway
// (stripe_rejects on the meshlet's own sphere, inside the node's); before this every rank walked every node and tested every meshlet of the
// N-times-taller frame, which is where the render-side 0.78 of profiles/r03_rank_balance.md came from.  Dropped, not replayed.
// Round 6: the same for the contiguous band (brmi_config::bandY0 / bandY1, brmi_set_band) -- its two view-space planes through the eye, which the cluster cull already tested
// per meshlet: every rank of the 8-GPU San-Miguel-class frame visited all 37,600 nodes and tested 40 k meshlets for bands that show 24 .. 27 k clusters.
BRMI_DEV bool sphere_culled(const CullArgs& a, const brmi_camera* cam, f3 c, float r) {
    if (sphere_outside_frustum(c, r, cam->clippingPlanes)) return true;
    if (a.bandActive) {      // (the planes from memory, like the camera's: brmi_frame.hip)
        const float4 top = a.bandPlanes[0], bottom = a.bandPlanes[1];
        if ((top.x * c.x + top.y * c.y) + top.z * c.z < -r || (bottom.x * c.x + bottom.y * c.y) + bottom.z * c.z < -r) return true;
    }
    return stripe_on(a.stripes) && stripe_rejects(a.stripes, cam, c, r);
}

// K1 -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cull_instances(CullArgs a, NodeRecord* frontier0) {
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const m4 view = load_m4(&cam->view[0][0]);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < ((sc.activeDrawCount + 63u) & ~63u); d += gridDim.x * blockDim.x) {
        bool visible = false;
        uint32_t ii = 0, root = 0;
        if (d < sc.activeDrawCount && (a.levelKernelsWidthLo == 0u || a.meshLevelWidth[sc.clodOffsets[sc.activeDraws[d]].clodMeshMetadataIndex] >= a.levelKernelsWidthLo)) {
            ii = sc.activeDraws[d];
            const brmi_per_mesh_instance inst = sc.perMeshInstance[ii];
            const m4 model = load_m4(&sc.perObject[inst.perObjectBufferIndex].model[0][0]);
            const f3 c = to_view_space(f3{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}, model, view);
            const float r = inst.boundingSphere[3] * max_axis_scale(model);
            const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
            visible = !bad && !sphere_culled(a, cam, c, r);
            root = sc.meshMetadata[sc.clodOffsets[ii].clodMeshMetadataIndex].rootNode;
            atomicAdd(&a.counters[CNT_INSTANCES_TESTED], 1u);
            if (visible) atomicAdd(&a.counters[CNT_INSTANCES_VISIBLE], 1u);
        }
        const uint32_t slot = wave_append(&a.counters[CNT_FRONTIER0], visible);
        if (visible) {
            if (slot < a.recordCapacity) frontier0[slot] = NodeRecord{ii, (1u << 30) | (root & 0x3FFFFFFFu)};
            else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
        }
    }
}

// K2: one BFS level ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_traverse(CullArgs a, uint32_t level, const NodeRecord* frontierIn, NodeRecord* frontierOut, BucketRecord* buckets) {
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t inputCount = min(a.counters[level == 0 ? a.frontier0Counter : CNT_FRONTIER0 + level], a.recordCapacity);
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* lodCam = sc.cullingCameras + viewId;
    const bool ortho = cam->isOrtho != 0;
    const f3 camPos{lodCam->positionWorldSpace[0], lodCam->positionWorldSpace[1], lodCam->positionWorldSpace[2]};
    const float zNear = lodCam->zNear, threshold = lodCam->errorOverDistanceThreshold;
    const m4 view = load_m4(&cam->view[0][0]);
    uint32_t* nextCount = &a.counters[CNT_FRONTIER0 + level + 1];
    const uint32_t rounded = (inputCount + 63u) & ~63u;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rounded; idx += gridDim.x * blockDim.x) {
        bool have = idx < inputCount;
        // phase 2 of a mixed traversal: the replay buffer also holds nodes of the narrow instances, which k_cull_hierarchy<true> walks
        if (have && level == 0u && a.phase == 2u && a.levelKernelsWidthLo != 0u &&
            a.meshLevelWidth[sc.clodOffsets[frontierIn[idx].instanceIndex].clodMeshMetadataIndex] < a.levelKernelsWidthLo) have = false;
        // per-record state
        bool isInternal = false, emitLeaf = false, replay = false, occluded = false;
        uint32_t occludedNode = 0;
        uint32_t instIndex = 0, childBase = 0, childCount = 0, lodNodesBase = 0;
        uint32_t segFirst = 0, segCount = 0, ownerGroup = 0, slabDesc = 0, slabOff = 0, firstBit = 0;
        bool skinned = false;
        m4 model{}; float scale = 0.0f;
        f3 instC{0, 0, 0}; float instR = 0.0f;
        if (have) {
            const NodeRecord rec = frontierIn[idx];
            instIndex = rec.instanceIndex;
            replay = (rec.nodeIdPacked >> 31) != 0;
            const bool allowRefine = ((rec.nodeIdPacked >> 30) & 1u) != 0;
            const uint32_t nodeId = rec.nodeIdPacked & 0x3FFFFFFFu;
            const brmi_per_mesh_instance inst = sc.perMeshInstance[instIndex];
            const brmi_clod_mesh_metadata md = sc.meshMetadata[sc.clodOffsets[instIndex].clodMeshMetadataIndex];
            skinned = (sc.perMesh[inst.perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) != 0;
            model = load_m4(&sc.perObject[inst.perObjectBufferIndex].model[0][0]);
            scale = max_axis_scale(model);
            instC = f3{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; instR = inst.boundingSphere[3];
            lodNodesBase = md.lodNodesBase;
            const brmi_lod_node node = sc.lodNodes[md.lodNodesBase + nodeId];
            const f3 cullC = skinned ? instC : f3{node.cullCenterAndRadius[0], node.cullCenterAndRadius[1], node.cullCenterAndRadius[2]};
            const float cullR = skinned ? instR : node.cullCenterAndRadius[3];
            const f3 cVS = to_view_space(cullC, model, view);
            const float rW = cullR * scale;
            const bool culled = !replay && sphere_culled(a, cam, cVS, rW);
            if (!culled) {
                if (node.isLeaf != BRMI_NODE_INTERNAL) {
                    const brmi_lod_group* g = sc.lodGroups + (md.groupsBase + node.ownerGroupId);
                    const f3 gc = xyz(mul_point(f3{g->centerAndRadius[0], g->centerAndRadius[1], g->centerAndRadius[2]}, model));
                    const float gr = g->centerAndRadius[3] * scale;
                    const float eod = projected_error(gc, gr, node.maxQuadricError, scale, camPos, zNear, ortho);
                    bool ok = allowRefine && (eod >= threshold);
                    if (ok && refined_child_suppresses(sc, md.groupsBase, node.countMinusOne - 1u, node.countMinusOne != 0u, model, scale, camPos, zNear, threshold, ortho)) ok = false;
                    if (ok) {
                        const brmi_lod_segment seg = sc.lodSegments[md.segmentsBase + node.indexOrOffset];
                        const brmi_group_page_map_entry pe = sc.groupPageMap[md.pageMapBase + seg.pageIndex];
                        if (seg.meshletCount != 0u && pe.slabDescriptorIndex != 0u) {
                            emitLeaf = true;
                            segFirst = seg.firstMeshletInPage; segCount = seg.meshletCount; ownerGroup = node.ownerGroupId;
                            slabDesc = pe.slabDescriptorIndex; slabOff = pe.slabByteOffset;
                            firstBit = a.instanceBitBase[instIndex] + a.segPrefix[md.segmentsBase + node.indexOrOffset];
                        }
                    }
                } else {
                    const f3 lc = xyz(mul_point(f3{node.lodCenterAndRadius[0], node.lodCenterAndRadius[1], node.lodCenterAndRadius[2]}, model));
                    const float lr = node.lodCenterAndRadius[3] * scale;
                    const float nodeEod = projected_error(lc, lr, node.maxQuadricError, scale, camPos, zNear, ortho);
                    if (allowRefine && (nodeEod >= threshold)) {
                        if (a.occlusion && occlusion_test(a, cam, replay, cullC, cullR, cVS, rW, sc.perObject + inst.perObjectBufferIndex)) {
                            occluded = !replay; occludedNode = nodeId;     // a node rejected in phase 2 is simply dropped
                        } else {
                            isInternal = true;
                            childBase = node.indexOrOffset;
                            childCount = min(node.countMinusOne + 1u, BRMI_BVH_MAX_CHILDREN);
                        }
                    }
                }
            }
        }
        {   // statistics: one atomic per wave on one of 64 stripes
            const uint64_t hm = __ballot(have);
            if (hm != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&a.counters[CNT_STRIPES + (blockIdx.x & (CNT_STRIPE_COUNT - 1u)) * CNT_STRIPE_WORDS + 2u], (uint32_t)__popcll(hm));
        }
        if (a.occlusion && a.phase == 1u) {   // hand the rejected node to phase 2 (workGraphCulling.hlsl:3094-3112: drop + count when full)
            const uint32_t slot = wave_append(&a.counters[CNT_REPLAY_NODES], occluded);
            if (occluded) {
                if (slot < a.recordCapacity) a.replayNodes[slot] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (occludedNode & 0x3FFFFFFFu)};
                else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
            }
        }
        // leaf: chunk the segment into bucket records of `factor` meshlets (computeCulling.hlsl:385-406)
        // wave-cooperative emission: iterate chunk index k over the widest leaf in the wave
        {
            const uint32_t nChunks = emitLeaf ? (segCount + a.factor - 1u) / a.factor : 0u;
            uint32_t waveMax = nChunks;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) waveMax = max(waveMax, (uint32_t)__shfl_xor((int)waveMax, o));
            for (uint32_t k = 0; k < waveMax; k++) {
                const bool emit = k < nChunks;
                const uint32_t slot = wave_append(&a.counters[a.bucketCounter], emit);
                if (emit) {
                    if (slot < a.recordCapacity) {
                        const uint32_t first = segFirst + k * a.factor;
                        const uint32_t cnt = min(a.factor, segCount - k * a.factor);
                        BucketRecord b;
                        b.instanceIndex = instIndex; b.groupIdPacked = (replay ? 0x80000000u : 0u) | (ownerGroup & 0x7FFFFFFFu);
                        b.meshletIndexAndCount = (cnt << 16) | (first & 0xFFFFu);
                        b.pageSlabDescriptorIndex = slabDesc; b.pageSlabByteOffset = slabOff;
                        b.firstBit = firstBit + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                        buckets[slot] = b;
                    } else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                }
            }
        }
        // internal: pre-filter children, append survivors to the next frontier (computeCulling.hlsl:477-530)
        {
            uint32_t waveMax = isInternal ? childCount : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) waveMax = max(waveMax, (uint32_t)__shfl_xor((int)waveMax, o));
            for (uint32_t k = 0; k < waveMax; k++) {
                bool emit = false;
                uint32_t childId = 0;
                if (isInternal && k < childCount) {
                    childId = childBase + k;
                    const brmi_lod_node* ch = sc.lodNodes + (lodNodesBase + childId);
                    const f3 cc = skinned ? instC : f3{ch->cullCenterAndRadius[0], ch->cullCenterAndRadius[1], ch->cullCenterAndRadius[2]};
                    const float cr = skinned ? instR : ch->cullCenterAndRadius[3];
                    const f3 ccVS = to_view_space(cc, model, view);
                    emit = replay || !sphere_culled(a, cam, ccVS, cr * scale);
                    if (emit && ch->isLeaf == BRMI_NODE_INTERNAL) {
                        const f3 wc = xyz(mul_point(f3{ch->lodCenterAndRadius[0], ch->lodCenterAndRadius[1], ch->lodCenterAndRadius[2]}, model));
                        const float e = projected_error(wc, ch->lodCenterAndRadius[3] * scale, ch->maxQuadricError, scale, camPos, zNear, ortho);
                        if (e < threshold) emit = false;
                    }
                }
                const uint32_t slot = wave_append(nextCount, emit);
                if (emit) {
                    if (slot < a.recordCapacity) frontierOut[slot] = NodeRecord{instIndex, (replay ? 0x80000000u : 0u) | (1u << 30) | (childId & 0x3FFFFFFFu)};
                    else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                }
            }
        }
    }
}

// K1 + K2 in one launch: one wave64 per draw (phase 1) or per replayed node (phase 2) walks that instance's BVH breadth-first
// with the frontier in LDS.  Instances are independent, so the level-by-level kernel sequence above -- one launch and ~8
// dependent HBM round trips per level for every instance -- becomes one launch in which the per-instance state (instance,
// mesh metadata, model matrix) is fetched once and a level costs node -> group / segment -> page map.  Used when every mesh's
// BVH level fits the LDS frontier (brmi_set_scene checks); same tests, same operation order as k_cull_instances / k_traverse.
#ifndef BRMI_HIER_STAGE_WIDE
#define BRMI_HIER_STAGE_WIDE 128
#endif
constexpr uint32_t HIER_CAP_MAX = 1024;   // widest BVH level the LDS frontier variants cover
// HIER_CAP nodes per frontier, HIER_STAGE bucket records staged in LDS: (256, 128) = 6 KB keeps ~20 workgroups per CU in flight
// (scenes of many small instances), (1024, 128) = 12 KB covers wide hierarchies.  Meshes wider than that (a street's ground and facades
// tessellated to pixel-sized triangles: 1,600 leaf segments on one level) go through the level-per-launch kernels, which put every
// lane of the chip on one level -- a single wave walking such a mesh alone took 0.4 ms (tried with a 4096-node variant).
// SIDE (brmi_execute, phase 1): the launch carries extra workgroups behind the traversal's that clear the visibility buffer.  The walk is a
// chain of dependent loads on ~1.3 waves per SIMD; the 66 MB of stores disappear in its shadow instead of costing a launch of their own.
// Behind those, one workgroup per light cluster runs the first half of the light clustering (AABB + hit masks + page demand: it depends on
// the frame constants alone); the second half rides on k_cull_clusters.  Frame: three launches and ~25 us less.
struct NoSide {};
struct SideJobs { ulonglong2* vis2; uint64_t n2; uint32_t walkBlocks, clearBlocks; ClusterArgs lc; };
template <bool REPLAY, uint32_t HIER_CAP, uint32_t HIER_STAGE, bool SIDE = false>
// Spill mode (meshes wider than `spillAbove` nodes per level): the walk keeps such an instance only while its frontier is small enough for the
// next level to fit the LDS frontier whatever the fan-out (<= HIER_CAP / 8 nodes) and then appends the frontier to `spillOut`, the level-0 input
// of the level kernels.  The top levels of a wide hierarchy hold a handful of nodes each; as level launches they cost 12 us apiece.
__global__ void __launch_bounds__(64) k_cull_hierarchy(CullArgs a, BucketRecord* buckets, const uint32_t* meshLevelWidth, uint32_t widthLo, uint32_t widthHi, uint32_t spillAbove, NodeRecord* spillOut,
                                                    typename std::conditional<SIDE, SideJobs, NoSide>::type sj) {
    wave_prio<PRIO_CULL>();
    uint32_t walkBlocks = gridDim.x;
    if constexpr (SIDE) {
        walkBlocks = sj.walkBlocks;
        if (blockIdx.x >= sj.walkBlocks + sj.clearBlocks) { lc_count_wave(sj.lc, blockIdx.x - sj.walkBlocks - sj.clearBlocks, threadIdx.x); return; }
        if (blockIdx.x >= sj.walkBlocks) {
            const uint64_t stride = (uint64_t)sj.clearBlocks * 64u;
            for (uint64_t i = (uint64_t)(blockIdx.x - sj.walkBlocks) * 64u + threadIdx.x; i < sj.n2; i += stride) sj.vis2[i] = make_ulonglong2(BRMI_VIS_EMPTY, BRMI_VIS_EMPTY);
            return;
        }
    }
    __shared__ uint32_t frontier[2][HIER_CAP];
    __shared__ uint32_t counts[2];
    __shared__ uint32_t childOff[65], childFirst[64];
    __shared__ BucketRecord stage[HIER_STAGE];     // bucket records of the instance being walked: one global reservation per flush
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t lane = threadIdx.x;
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* lodCam = sc.cullingCameras + viewId;
    const bool ortho = cam->isOrtho != 0;
    const f3 camPos{lodCam->positionWorldSpace[0], lodCam->positionWorldSpace[1], lodCam->positionWorldSpace[2]};
    const float zNear = lodCam->zNear, threshold = lodCam->errorOverDistanceThreshold;
    const m4 view = load_m4(&cam->view[0][0]);
    const uint32_t seeds = REPLAY ? min(a.counters[CNT_REPLAY_NODES], a.recordCapacity) : sc.activeDrawCount;
    uint32_t nTested = 0, nVisible = 0, nNodes = 0;
    uint32_t staged = 0;                            // wave-uniform
    auto flush = [&]() {
        if (staged == 0u) return;
        uint32_t baseSlot = 0;
        if (lane == 0) baseSlot = atomicAdd(&a.counters[a.bucketCounter], staged);
        baseSlot = (uint32_t)__shfl((int)baseSlot, 0);
        __syncthreads();
        for (uint32_t k = lane; k < staged; k += 64u) {
            if (baseSlot + k < a.recordCapacity) buckets[baseSlot + k] = stage[k];
            else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
        }
        __syncthreads();
        staged = 0u;
    };
    // ---- eight instances to a wave.  The typical hierarchy of these scenes has five nodes (median; 90 % have <= 9): a wave per instance leaves
    // 59 lanes idle AND queues two atomics with return per instance on two counters that serve ~90 per microsecond (2,017 instances: 19 us
    // for the last wave, the launch's length).  The first ceil(draws / 8) waves take eight consecutive draws each, eight lanes per draw --
    // object matrices per lane --, evaluate those whose hierarchy has <= 8 nodes exactly as the one-instance path below does, and make ONE
    // reservation per counter for all of them; the waves behind them take one draw each and skip what was handled here.
    uint32_t firstSeed = blockIdx.x, seedStride = walkBlocks;
    if (!REPLAY && a.packedFlat) {
        const uint32_t packedWaves = (seeds + 7u) >> 3;
        if (blockIdx.x < packedWaves) {
            const uint32_t g8 = lane & ~7u, j = lane & 7u;
            const uint32_t seed = blockIdx.x * 8u + (lane >> 3);
            const bool haveSeed = seed < seeds;
            const uint32_t instIndex = haveSeed ? sc.activeDraws[seed] : 0u;
            brmi_per_mesh_instance inst{}; InstanceWalk iw{0u, 0u, 0u, 0u};
            if (haveSeed) { inst = sc.perMeshInstance[instIndex]; iw = a.instanceWalk[instIndex]; }
            const bool small = haveSeed && iw.flatCount >= 1u && iw.flatCount <= 8u;
            const brmi_per_object* obj = sc.perObject + (small ? inst.perObjectBufferIndex : 0u);
            const bool mine = small && j < iw.flatCount;
            FlatNode fn{}; FlatLeaf fl{};
            if (mine) { fn = a.flatNodes[iw.flatBase + j]; fl = a.flatLeaves[iw.flatBase + j]; }
            m4 model = load_m4(&obj->model[0][0]);
            m4 prevModel = model;
            if (a.occlusion && mine && ((fn.info & 1u))) prevModel = load_m4(&obj->prevModel[0][0]);       // (internal nodes: the occlusion test's matrix, requested now)
            const float scale = max_axis_scale(model);
            const f3 instC{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; const float instR = inst.boundingSphere[3];
            bool instVisible = false;
            {   // K1 (PureComputeObjectCullCS), by every lane of the draw's group alike
                const f3 c = to_view_space(instC, model, view);
                const float r = instR * scale;
                const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
                instVisible = small && !bad && !sphere_culled(a, cam, c, r);
            }
            nTested += (uint32_t)__popcll(__ballot(small && j == 0u)); nVisible += (uint32_t)__popcll(__ballot(instVisible && j == 0u));
            const bool skinned = iw.skinned != 0u;
            const bool internal = (fn.info & 1u);
            const f3 cullC = skinned ? instC : f3{fn.cull[0], fn.cull[1], fn.cull[2]};
            const float cullR = skinned ? instR : fn.cull[3];
            const f3 cVS = to_view_space(cullC, model, view);
            const float rW = cullR * scale;
            const bool inFrustum = mine && instVisible && !sphere_culled(a, cam, cVS, rW);
            bool pre = inFrustum, expand = false, hidden = false, leafOk = false;
            uint32_t slabDesc = 0, slabOff = 0;
            if (inFrustum && internal) {
                const f3 lc = xyz(mul_point(f3{fn.lod[0], fn.lod[1], fn.lod[2]}, model));
                const float e = projected_error(lc, fn.lod[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                pre = e >= threshold;
                if (pre) { hidden = a.occlusion && occlusion_test_prev(a, cam, cullC, cullR, prevModel); expand = !hidden; }
            } else if (inFrustum) {
                const f3 gc = xyz(mul_point(f3{fl.group[0], fl.group[1], fl.group[2]}, model));
                const float eod = projected_error(gc, fl.group[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                bool ok = eod >= threshold;
                if (ok && ((fn.info >> 1) & 1u)) {      // refined_child_suppresses
                    const f3 cc = xyz(mul_point(f3{fl.child[0], fl.child[1], fl.child[2]}, model));
                    const float ce = projected_error(cc, fl.child[3] * scale, fl.childParentError, scale, camPos, zNear, ortho);
                    if (!(ce < threshold)) ok = false;
                }
                if (ok && ((fn.info >> 2) & 1u)) {
                    const brmi_group_page_map_entry pe = sc.groupPageMap[fn.pageMapIndex];
                    slabDesc = pe.slabDescriptorIndex; slabOff = pe.slabByteOffset;
                    leafOk = slabDesc != 0u;
                }
            }
            const uint64_t expandM = __ballot(expand);
            const uint32_t parentLane = g8 | ((fn.info >> 8) & 7u);
            uint64_t reached = __ballot(mine && instVisible && j == 0u);
            for (;;) {
                const bool r = mine && instVisible && (j == 0u || (pre && ((reached >> parentLane) & 1ull) && ((expandM >> parentLane) & 1ull)));
                const uint64_t next = __ballot(r);
                if (next == reached) break;
                reached = next;
            }
            const bool here = (reached >> lane) & 1ull;
            nNodes += here ? 1u : 0u;
            const bool replayIt = a.occlusion && here && hidden;
            const uint64_t replayM = __ballot(replayIt);
            const bool emitLeaf = here && leafOk;
            const uint32_t segFirst = fn.segFirstCount & 0xFFFFu, segCount = fn.segFirstCount >> 16;
            const uint32_t nChunks = emitLeaf ? (segCount + a.factor - 1u) / a.factor : 0u;
            uint32_t incl = nChunks;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
            const uint32_t nBuckets = (uint32_t)__shfl((int)incl, 63), nReplay = (uint32_t)__popcll(replayM);
            uint32_t replayBase = 0, bucketBase = 0;
            if (lane == 0) {
                if (nReplay != 0u) replayBase = atomicAdd(&a.counters[CNT_REPLAY_NODES], nReplay);
                if (nBuckets != 0u) bucketBase = atomicAdd(&a.counters[a.bucketCounter], nBuckets);
            }
            replayBase = (uint32_t)__shfl((int)replayBase, 0); bucketBase = (uint32_t)__shfl((int)bucketBase, 0);
            if (replayIt) {
                const uint32_t slot = replayBase + (uint32_t)__popcll(replayM & ((1ull << lane) - 1ull));
                if (slot < a.recordCapacity) a.replayNodes[slot] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (fn.nodeId & 0x3FFFFFFFu)};
                else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
            }
            for (uint32_t k = 0; k < nChunks; k++) {
                const uint32_t slot = bucketBase + (incl - nChunks) + k;
                if (slot >= a.recordCapacity) { atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u); continue; }
                BucketRecord b;
                b.instanceIndex = instIndex; b.groupIdPacked = fn.ownerGroup & 0x7FFFFFFFu;
                b.meshletIndexAndCount = (min(a.factor, segCount - k * a.factor) << 16) | ((segFirst + k * a.factor) & 0xFFFFu);
                b.pageSlabDescriptorIndex = slabDesc; b.pageSlabByteOffset = slabOff;
                b.firstBit = iw.bitBase + fn.firstBitRel + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                buckets[slot] = b;
            }
            firstSeed = seeds;                                  // nothing else for this wave
        } else { firstSeed = blockIdx.x - packedWaves; seedStride = walkBlocks - packedWaves; }
    }
    for (uint32_t seed = firstSeed; seed < seeds; seed += seedStride) {
        uint32_t instIndex, startNode;
        if (REPLAY) { const NodeRecord rec = a.replayNodes[seed]; instIndex = rec.instanceIndex; startNode = rec.nodeIdPacked & 0x3FFFFFFFu; }
        else instIndex = sc.activeDraws[seed];
        const brmi_per_mesh_instance inst = sc.perMeshInstance[instIndex];
        if (!REPLAY) {
            // ---- a hierarchy of at most 64 nodes: all of it at once, one lane per node.  The level walk below is a chain of ~10 dependent
            // memory round trips per instance (instance -> mesh metadata -> root -> occlusion -> children -> group / segment -> page map ->
            // bucket slots) and the launch lasts as long as one such chain; here the nodes, the leaves' groups and segments (FlatNode /
            // FlatLeaf, folded by brmi_set_scene) and the object arrive together, the depth chain and the page map together after them.
            // Same tests, same arithmetic, same records; a node is reached iff every ancestor let its children through.
            const InstanceWalk iw = a.instanceWalk[instIndex];
            if (a.packedFlat && iw.flatCount >= 1u && iw.flatCount <= 8u) continue;      // one of the eight draws of a packed wave
#ifdef BRMI_TILE_STAMPS
            unsigned long long hph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hprev = __builtin_amdgcn_s_memtime();
#define HSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); hph[k] += now_ - hprev; hprev = now_; } while (0)
            { uint32_t probe_ = iw.flatCount + inst.perObjectBufferIndex; asm volatile("" :: "v"(probe_)); }
            HSTAMP(0);
#else
#define HSTAMP(k) do { } while (0)
#endif
            if (iw.flatCount > 256u && a.wideFlat) continue;      // k_cull_flat_wide's
            if (iw.flatCount != 0u && iw.flatCount <= 256u) {
                constexpr uint32_t FLAT_CHUNKS = 4;      // 64 nodes each (brmi_set_scene: hierarchies of up to 256 nodes)
                const uint32_t chunks = (iw.flatCount + 63u) >> 6;
                const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
                // (what the later phases need of a node; the spheres are used at once)
                uint32_t nodeIdA[FLAT_CHUNKS] = {}, parentA[FLAT_CHUNKS] = {}, ownerGroupA[FLAT_CHUNKS] = {}, segFirstCountA[FLAT_CHUNKS] = {}, firstBitRelA[FLAT_CHUNKS] = {};
                const m4 model = load_m4(&obj->model[0][0]);
                const m4 prevModel = a.occlusion ? load_m4(&obj->prevModel[0][0]) : model;      // (the occlusion test's matrix, requested with the current one)
                const float scale = max_axis_scale(model);
                const f3 instC{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; const float instR = inst.boundingSphere[3];
                {   // K1 (PureComputeObjectCullCS)
                    const f3 c = to_view_space(instC, model, view);
                    const float r = instR * scale;
                    const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
                    nTested++;
                    if (bad || sphere_culled(a, cam, c, r)) continue;
                    nVisible++;
                }
                HSTAMP(1);
                const bool skinned = iw.skinned != 0u;
                uint64_t preM[FLAT_CHUNKS] = {}, expandM[FLAT_CHUNKS] = {}, hiddenM[FLAT_CHUNKS] = {}, leafM[FLAT_CHUNKS] = {}, reached[FLAT_CHUNKS] = {};
                uint32_t slabDescA[FLAT_CHUNKS] = {}, slabOffA[FLAT_CHUNKS] = {};
#pragma unroll
                for (uint32_t c = 0; c < FLAT_CHUNKS; c++) if (c < chunks) {
                    const bool mine = c * 64u + lane < iw.flatCount;
                    FlatNode fn{}; FlatLeaf fl{};
                    if (mine) { fn = a.flatNodes[iw.flatBase + c * 64u + lane]; fl = a.flatLeaves[iw.flatBase + c * 64u + lane]; }
                    nodeIdA[c] = fn.nodeId; parentA[c] = fn.info >> 8; ownerGroupA[c] = fn.ownerGroup; segFirstCountA[c] = fn.segFirstCount; firstBitRelA[c] = fn.firstBitRel;
                    const bool internal = (fn.info & 1u);
                    const f3 cullC = skinned ? instC : f3{fn.cull[0], fn.cull[1], fn.cull[2]};
                    const float cullR = skinned ? instR : fn.cull[3];
                    const f3 cVS = to_view_space(cullC, model, view);
                    const float rW = cullR * scale;
                    const bool inFrustum = mine && !sphere_culled(a, cam, cVS, rW);
                    // internal node: children pass when its projected error is above the threshold and the depth chain does not hide it;
                    // as a child it was let through on the same two conditions (frustum, error)
                    bool pre = inFrustum, expand = false, hidden = false, leafOk = false;
                    if (inFrustum && internal) {
                        const f3 lc = xyz(mul_point(f3{fn.lod[0], fn.lod[1], fn.lod[2]}, model));
                        const float e = projected_error(lc, fn.lod[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                        pre = e >= threshold;
                        if (pre) { hidden = a.occlusion && occlusion_test_prev(a, cam, cullC, cullR, prevModel); expand = !hidden; }
                    } else if (inFrustum) {
                        const f3 gc = xyz(mul_point(f3{fl.group[0], fl.group[1], fl.group[2]}, model));
                        const float eod = projected_error(gc, fl.group[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                        bool ok = eod >= threshold;
                        if (ok && ((fn.info >> 1) & 1u)) {      // refined_child_suppresses
                            const f3 cc = xyz(mul_point(f3{fl.child[0], fl.child[1], fl.child[2]}, model));
                            const float ce = projected_error(cc, fl.child[3] * scale, fl.childParentError, scale, camPos, zNear, ortho);
                            if (!(ce < threshold)) ok = false;
                        }
                        if (ok && ((fn.info >> 2) & 1u)) {
                            const brmi_group_page_map_entry pe = sc.groupPageMap[fn.pageMapIndex];
                            slabDescA[c] = pe.slabDescriptorIndex; slabOffA[c] = pe.slabByteOffset;
                            leafOk = slabDescA[c] != 0u;
                        }
                    }
                    preM[c] = __ballot(pre); expandM[c] = __ballot(expand); hiddenM[c] = __ballot(hidden); leafM[c] = __ballot(leafOk);
                }
                HSTAMP(2);
                // reached: the root, or a node that passed as a child of a reached node that lets its children through
                reached[0] = 1ull;
                for (bool changed = true; changed; ) {
                    changed = false;
#pragma unroll
                    for (uint32_t c = 0; c < FLAT_CHUNKS; c++) if (c < chunks) {
                        const uint32_t parent = parentA[c], pc = parent >> 6, pb = parent & 63u;
                        uint64_t through = 0;      // reached parents that let their children through, the parent's chunk
#pragma unroll
                        for (uint32_t q = 0; q < FLAT_CHUNKS; q++) if (q == pc) through = reached[q] & expandM[q];
                        const bool r = (c == 0u && lane == 0u) || (c * 64u + lane < iw.flatCount && ((preM[c] >> lane) & 1ull) && ((through >> pb) & 1ull) && !(c == 0u && lane == 0u));
                        const uint64_t next = __ballot(r);
                        if (next != reached[c]) { reached[c] = next; changed = true; }
                    }
                }
                // Two reservations per instance -- replay nodes, bucket records -- and every wave of the launch reaches them at about the same
                // time: ~2,000 atomics with return on each counter are served at ~90 per microsecond, 19 us for the last wave, and one after
                // the other they were most of this kernel.  Both are requested back to back (two queues drain side by side), and the
                // records go straight to the bucket array (the LDS stage collects the level walk's records: one instance per wave has
                // nothing to collect).
                uint32_t nReplay = 0, nBuckets = 0;                 // wave totals
                uint32_t replayRank[FLAT_CHUNKS] = {}, bucketRank[FLAT_CHUNKS] = {}, chunksOfLeaf[FLAT_CHUNKS] = {};
#pragma unroll
                for (uint32_t c = 0; c < FLAT_CHUNKS; c++) if (c < chunks) {
                    const bool here = (reached[c] >> lane) & 1ull;
                    nNodes += here ? 1u : 0u;
                    const uint64_t rm = a.occlusion ? (reached[c] & hiddenM[c]) : 0ull;
                    replayRank[c] = nReplay + (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
                    nReplay += (uint32_t)__popcll(rm);
                    const bool emitLeaf = here && ((leafM[c] >> lane) & 1ull);
                    const uint32_t segCount = segFirstCountA[c] >> 16;
                    chunksOfLeaf[c] = emitLeaf ? (segCount + a.factor - 1u) / a.factor : 0u;
                    uint32_t incl = chunksOfLeaf[c];
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
                    bucketRank[c] = nBuckets + incl - chunksOfLeaf[c];
                    nBuckets += (uint32_t)__shfl((int)incl, 63);
                }
                uint32_t replayBase = 0, bucketBase = 0;
                if (lane == 0) {
                    if (nReplay != 0u) replayBase = atomicAdd(&a.counters[CNT_REPLAY_NODES], nReplay);
                    if (nBuckets != 0u) bucketBase = atomicAdd(&a.counters[a.bucketCounter], nBuckets);
                }
                replayBase = (uint32_t)__shfl((int)replayBase, 0); bucketBase = (uint32_t)__shfl((int)bucketBase, 0);
#pragma unroll
                for (uint32_t c = 0; c < FLAT_CHUNKS; c++) if (c < chunks) {
                    if (a.occlusion && ((reached[c] & hiddenM[c]) >> lane) & 1ull) {
                        const uint32_t slot = replayBase + replayRank[c];
                        if (slot < a.recordCapacity) a.replayNodes[slot] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (nodeIdA[c] & 0x3FFFFFFFu)};
                        else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                    }
                    // bucket records of `factor` meshlets per reached leaf that passed (computeCulling.hlsl:385-406)
                    const uint32_t segFirst = segFirstCountA[c] & 0xFFFFu, segCount = segFirstCountA[c] >> 16;
                    for (uint32_t k = 0; k < chunksOfLeaf[c]; k++) {
                        const uint32_t slot = bucketBase + bucketRank[c] + k;
                        if (slot >= a.recordCapacity) { atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u); continue; }
                        BucketRecord b;
                        b.instanceIndex = instIndex; b.groupIdPacked = ownerGroupA[c] & 0x7FFFFFFFu;
                        b.meshletIndexAndCount = (min(a.factor, segCount - k * a.factor) << 16) | ((segFirst + k * a.factor) & 0xFFFFu);
                        b.pageSlabDescriptorIndex = slabDescA[c]; b.pageSlabByteOffset = slabOffA[c];
                        b.firstBit = iw.bitBase + firstBitRelA[c] + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                        buckets[slot] = b;
                    }
                }
                HSTAMP(3);
#ifdef BRMI_TILE_STAMPS
                if (lane < 8u) { unsigned long long v = 0; for (int k = 0; k < 8; k++) if (lane == (uint32_t)k) v = hph[k]; atomicAdd(a.debugStamps + 48u + lane, v); }
#endif
                continue;
            }
        }
        const uint32_t mdIndex = sc.clodOffsets[instIndex].clodMeshMetadataIndex;
        // this launch handles the meshes whose widest BVH level fits its LDS frontier class, and the top of wider ones
        const uint32_t width = meshLevelWidth[mdIndex];
        if (width < widthLo || width > widthHi) continue;
        const bool spill = width > spillAbove;
        const brmi_clod_mesh_metadata md = sc.meshMetadata[mdIndex];
        const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
        const m4 model = load_m4(&obj->model[0][0]);
        const float scale = max_axis_scale(model);
        const f3 instC{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; const float instR = inst.boundingSphere[3];
        if (!REPLAY) {
            // K1 (PureComputeObjectCullCS)
            const f3 c = to_view_space(instC, model, view);
            const float r = instR * scale;
            const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
            const bool visible = !bad && !sphere_culled(a, cam, c, r);
            nTested++;
            if (!visible) continue;
            nVisible++;
            startNode = md.rootNode;
        }
        const bool skinned = (sc.perMesh[inst.perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) != 0;
        const uint32_t bitBase = a.instanceBitBase[instIndex];
        __syncthreads();
        if (lane == 0) { frontier[0][0] = startNode; counts[0] = 1u; counts[1] = 0u; }
        __syncthreads();
        for (uint32_t level = 0; level < 64u; level++) {
            const uint32_t cur = level & 1u, nxt = cur ^ 1u;
            const uint32_t n = min(counts[cur], HIER_CAP);
            if (n == 0u) break;
            if (spill && n > HIER_CAP / BRMI_BVH_MAX_CHILDREN) {      // the level after this one may not fit: the level kernels take over from here
                uint32_t baseSlot = 0;
                if (lane == 0) baseSlot = atomicAdd(&a.counters[CNT_FRONTIER0], n);
                baseSlot = (uint32_t)__shfl((int)baseSlot, 0);
                for (uint32_t k = lane; k < n; k += 64u) {
                    if (baseSlot + k < a.recordCapacity) spillOut[baseSlot + k] = NodeRecord{instIndex, (REPLAY ? 0x80000000u : 0u) | (1u << 30) | (frontier[cur][k] & 0x3FFFFFFFu)};
                    else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                }
                break;
            }
            for (uint32_t base = 0; base < n; base += 64u) {
                const bool have = base + lane < n;
                bool isInternal = false, emitLeaf = false, occluded = false;
                uint32_t nodeId = 0, childBase = 0, childCount = 0, segFirst = 0, segCount = 0, ownerGroup = 0, slabDesc = 0, slabOff = 0, firstBit = 0;
                if (have) {
                    nodeId = frontier[cur][base + lane];
                    nNodes++;
                    const brmi_lod_node node = sc.lodNodes[md.lodNodesBase + nodeId];
                    const f3 cullC = skinned ? instC : f3{node.cullCenterAndRadius[0], node.cullCenterAndRadius[1], node.cullCenterAndRadius[2]};
                    const float cullR = skinned ? instR : node.cullCenterAndRadius[3];
                    const f3 cVS = to_view_space(cullC, model, view);
                    const float rW = cullR * scale;
                    const bool culled = !REPLAY && sphere_culled(a, cam, cVS, rW);
                    if (!culled) {
                        if (node.isLeaf != BRMI_NODE_INTERNAL) {
                            const brmi_lod_group* g = sc.lodGroups + (md.groupsBase + node.ownerGroupId);
                            const f3 gc = xyz(mul_point(f3{g->centerAndRadius[0], g->centerAndRadius[1], g->centerAndRadius[2]}, model));
                            const float gr = g->centerAndRadius[3] * scale;
                            const float eod = projected_error(gc, gr, node.maxQuadricError, scale, camPos, zNear, ortho);
                            bool ok = eod >= threshold;
                            if (ok && refined_child_suppresses(sc, md.groupsBase, node.countMinusOne - 1u, node.countMinusOne != 0u, model, scale, camPos, zNear, threshold, ortho)) ok = false;
                            if (ok) {
                                const brmi_lod_segment seg = sc.lodSegments[md.segmentsBase + node.indexOrOffset];
                                const brmi_group_page_map_entry pe = sc.groupPageMap[md.pageMapBase + seg.pageIndex];
                                if (seg.meshletCount != 0u && pe.slabDescriptorIndex != 0u) {
                                    emitLeaf = true;
                                    segFirst = seg.firstMeshletInPage; segCount = seg.meshletCount; ownerGroup = node.ownerGroupId;
                                    slabDesc = pe.slabDescriptorIndex; slabOff = pe.slabByteOffset;
                                    firstBit = bitBase + a.segPrefix[md.segmentsBase + node.indexOrOffset];
                                }
                            }
                        } else {
                            const f3 lc = xyz(mul_point(f3{node.lodCenterAndRadius[0], node.lodCenterAndRadius[1], node.lodCenterAndRadius[2]}, model));
                            const float lr = node.lodCenterAndRadius[3] * scale;
                            const float nodeEod = projected_error(lc, lr, node.maxQuadricError, scale, camPos, zNear, ortho);
                            if (nodeEod >= threshold) {
                                if (a.occlusion && occlusion_test(a, cam, REPLAY, cullC, cullR, cVS, rW, obj)) occluded = !REPLAY;
                                else { isInternal = true; childBase = node.indexOrOffset; childCount = min(node.countMinusOne + 1u, BRMI_BVH_MAX_CHILDREN); }
                            }
                        }
                    }
                }
                if (!REPLAY && a.occlusion) {
                    const uint32_t slot = wave_append(&a.counters[CNT_REPLAY_NODES], occluded);
                    if (occluded) {
                        if (slot < a.recordCapacity) a.replayNodes[slot] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (nodeId & 0x3FFFFFFFu)};
                        else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                    }
                }
                {   // leaf: bucket records of `factor` meshlets (computeCulling.hlsl:385-406).  One reservation for the whole wave
                    // (an atomic with return per chunk would put ~2 us of latency on every chunk of this single-wave workgroup).
                    const uint32_t nChunks = emitLeaf ? (segCount + a.factor - 1u) / a.factor : 0u;
                    uint32_t incl = nChunks;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
                    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
                    if (total != 0u) {
                        if (staged + total > HIER_STAGE) flush();
                        if (total <= HIER_STAGE) {
                            const uint32_t baseSlot = staged + (incl - nChunks);
                            for (uint32_t k = 0; k < nChunks; k++) {
                                const uint32_t first = segFirst + k * a.factor;
                                const uint32_t cnt = min(a.factor, segCount - k * a.factor);
                                BucketRecord b;
                                b.instanceIndex = instIndex; b.groupIdPacked = (REPLAY ? 0x80000000u : 0u) | (ownerGroup & 0x7FFFFFFFu);
                                b.meshletIndexAndCount = (cnt << 16) | (first & 0xFFFFu);
                                b.pageSlabDescriptorIndex = slabDesc; b.pageSlabByteOffset = slabOff;
                                b.firstBit = firstBit + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                                stage[baseSlot + k] = b;
                            }
                            staged += total;
                        } else {
                            // more chunks in one step than the stage holds (very large segments): straight to the global array
                            uint32_t baseSlot = 0;
                            if (lane == 0) baseSlot = atomicAdd(&a.counters[a.bucketCounter], total);
                            baseSlot = (uint32_t)__shfl((int)baseSlot, 0) + (incl - nChunks);
                            for (uint32_t k = 0; k < nChunks; k++) {
                                const uint32_t slot = baseSlot + k;
                                if (slot < a.recordCapacity) {
                                    const uint32_t first = segFirst + k * a.factor;
                                    const uint32_t cnt = min(a.factor, segCount - k * a.factor);
                                    BucketRecord b;
                                    b.instanceIndex = instIndex; b.groupIdPacked = (REPLAY ? 0x80000000u : 0u) | (ownerGroup & 0x7FFFFFFFu);
                                    b.meshletIndexAndCount = (cnt << 16) | (first & 0xFFFFu);
                                    b.pageSlabDescriptorIndex = slabDesc; b.pageSlabByteOffset = slabOff;
                                    b.firstBit = firstBit + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                                    buckets[slot] = b;
                                } else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                            }
                        }
                    }
                }
                {   // internal: pre-filter the children, survivors go to the next level's frontier (computeCulling.hlsl:477-530).
                    // The children of all nodes of this step are dealt to the lanes (exclusive scan of the child counts), so their
                    // node records are fetched side by side instead of one dependent round trip per child index.
                    const uint32_t myChildren = isInternal ? childCount : 0u;
                    uint32_t incl = myChildren;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
                    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
                    if (total != 0u) {
                        childOff[lane] = incl - myChildren; childFirst[lane] = childBase;
                        if (lane == 63) childOff[64] = total;
                        __syncthreads();
                        for (uint32_t task = lane; task < ((total + 63u) & ~63u); task += 64u) {
                            bool emit = false;
                            uint32_t childId = 0;
                            if (task < total) {
                                uint32_t parent = 0;
#pragma unroll
                                for (uint32_t step = 32; step > 0; step >>= 1) if (childOff[parent + step] <= task) parent += step;
                                childId = childFirst[parent] + (task - childOff[parent]);
                                const brmi_lod_node* ch = sc.lodNodes + (md.lodNodesBase + childId);
                                const f3 cc = skinned ? instC : f3{ch->cullCenterAndRadius[0], ch->cullCenterAndRadius[1], ch->cullCenterAndRadius[2]};
                                const float cr = skinned ? instR : ch->cullCenterAndRadius[3];
                                const f3 ccVS = to_view_space(cc, model, view);
                                emit = REPLAY || !sphere_culled(a, cam, ccVS, cr * scale);
                                if (emit && ch->isLeaf == BRMI_NODE_INTERNAL) {
                                    const f3 wc = xyz(mul_point(f3{ch->lodCenterAndRadius[0], ch->lodCenterAndRadius[1], ch->lodCenterAndRadius[2]}, model));
                                    const float e = projected_error(wc, ch->lodCenterAndRadius[3] * scale, ch->maxQuadricError, scale, camPos, zNear, ortho);
                                    if (e < threshold) emit = false;
                                }
                            }
                            const uint32_t slot = wave_append(&counts[nxt], emit);
                            if (emit) {
                                if (slot < HIER_CAP) frontier[nxt][slot] = childId;
                                else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
            if (lane == 0) counts[cur] = 0u;
            __syncthreads();
        }
    }
    flush();
    // statistics: one atomic per wave and counter, on one of 64 stripes (every stripe has its own 128 B line: thousands of
    // same-line atomics serialise at ~90 per microsecond); brmi_read_counters adds the stripes up
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nNodes += (uint32_t)__shfl_xor((int)nNodes, o);
    if (lane == 0) {
        uint32_t* stripe = a.counters + CNT_STRIPES + (blockIdx.x & (CNT_STRIPE_COUNT - 1u)) * CNT_STRIPE_WORDS;
        if (nTested) atomicAdd(&stripe[0], nTested);
        if (nVisible) atomicAdd(&stripe[1], nVisible);
        if (nNodes) atomicAdd(&stripe[2], nNodes);
    }
}

// Flat evaluation of a hierarchy of 257 .. 8192 nodes (the dense workload's terrain-like meshes: 4,270 nodes, six levels): one 1024-thread
// workgroup per draw, a node per thread and chunk of 1024, the per-node verdicts and parent links in LDS, "reached" propagated there, one pair of
// reservations per draw.  The level walk spent a launch per level on such a mesh (k_traverse: ~10 us each, three per phase) behind a chain of
// LDS-walk steps; here every node of the hierarchy is fetched in at most eight rounds whatever the depth.  Same tests, same records.
constexpr uint32_t FLAT_WIDE_MAX = 8192;
__global__ void __launch_bounds__(1024) k_cull_flat_wide(CullArgs a, BucketRecord* buckets) {
    __shared__ uint8_t verdict[FLAT_WIDE_MAX];          // bit 0 passes as a child, 1 lets its children through, 2 hidden by the depth chain, 3 leaf that emits, 4 reached
    __shared__ uint16_t parentOf[FLAT_WIDE_MAX], recordsOf[FLAT_WIDE_MAX];
    __shared__ uint32_t waveSum[2][16], changed, bases[2];
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* lodCam = sc.cullingCameras + viewId;
    const bool ortho = cam->isOrtho != 0;
    const f3 camPos{lodCam->positionWorldSpace[0], lodCam->positionWorldSpace[1], lodCam->positionWorldSpace[2]};
    const float zNear = lodCam->zNear, threshold = lodCam->errorOverDistanceThreshold;
    const m4 view = load_m4(&cam->view[0][0]);
    uint32_t nTested = 0, nVisible = 0, nNodes = 0;
    for (uint32_t seed = blockIdx.x; seed < sc.activeDrawCount; seed += gridDim.x) {
        const uint32_t instIndex = sc.activeDraws[seed];
        const InstanceWalk iw = a.instanceWalk[instIndex];
        if (iw.flatCount <= 256u || iw.flatCount > FLAT_WIDE_MAX) continue;          // (block-uniform) the walk's, or not flat at all
        const brmi_per_mesh_instance inst = sc.perMeshInstance[instIndex];
        const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
        const m4 model = load_m4(&obj->model[0][0]);
        const m4 prevModel = a.occlusion ? load_m4(&obj->prevModel[0][0]) : model;
        const float scale = max_axis_scale(model);
        const f3 instC{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; const float instR = inst.boundingSphere[3];
        {   // K1 (PureComputeObjectCullCS)
            const f3 c = to_view_space(instC, model, view);
            const float r = instR * scale;
            const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
            if (t == 0) nTested++;
            if (bad || sphere_culled(a, cam, c, r)) continue;
            if (t == 0) nVisible++;
        }
        const bool skinned = iw.skinned != 0u;
        __syncthreads();                                    // the previous draw's LDS state has been read
        for (uint32_t node = t; node < iw.flatCount; node += 1024u) {
            const FlatNode fn = a.flatNodes[iw.flatBase + node];
            const bool internal = (fn.info & 1u);
            const f3 cullC = skinned ? instC : f3{fn.cull[0], fn.cull[1], fn.cull[2]};
            const float cullR = skinned ? instR : fn.cull[3];
            const f3 cVS = to_view_space(cullC, model, view);
            const float rW = cullR * scale;
            const bool inFrustum = !sphere_culled(a, cam, cVS, rW);
            bool pre = inFrustum, expand = false, hidden = false, leafOk = false;
            uint32_t records = 0;
            if (inFrustum && internal) {
                const f3 lc = xyz(mul_point(f3{fn.lod[0], fn.lod[1], fn.lod[2]}, model));
                const float e = projected_error(lc, fn.lod[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                pre = e >= threshold;
                if (pre) { hidden = a.occlusion && occlusion_test_prev(a, cam, cullC, cullR, prevModel); expand = !hidden; }
            } else if (inFrustum) {
                const FlatLeaf fl = a.flatLeaves[iw.flatBase + node];
                const f3 gc = xyz(mul_point(f3{fl.group[0], fl.group[1], fl.group[2]}, model));
                const float eod = projected_error(gc, fl.group[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                bool ok = eod >= threshold;
                if (ok && ((fn.info >> 1) & 1u)) {      // refined_child_suppresses
                    const f3 cc = xyz(mul_point(f3{fl.child[0], fl.child[1], fl.child[2]}, model));
                    const float ce = projected_error(cc, fl.child[3] * scale, fl.childParentError, scale, camPos, zNear, ortho);
                    if (!(ce < threshold)) ok = false;
                }
                if (ok && ((fn.info >> 2) & 1u)) {
                    leafOk = sc.groupPageMap[fn.pageMapIndex].slabDescriptorIndex != 0u;
                    records = leafOk ? ((fn.segFirstCount >> 16) + a.factor - 1u) / a.factor : 0u;
                }
            }
            verdict[node] = (uint8_t)((pre ? 1u : 0u) | (expand ? 2u : 0u) | (hidden ? 4u : 0u) | (leafOk ? 8u : 0u) | (node == 0u ? 16u : 0u));
            parentOf[node] = (uint16_t)(fn.info >> 8); recordsOf[node] = (uint16_t)records;
        }
        // reached: the root, or a node that passed as a child of a reached node that lets its children through (parents lie below their
        // children in the breadth-first order, so a sweep settles a level at least)
        for (;;) {
            __syncthreads();
            if (t == 0) changed = 0u;
            __syncthreads();
            for (uint32_t node = t; node < iw.flatCount; node += 1024u) {
                const uint32_t v = verdict[node];
                if (!(v & 16u) && (v & 1u)) { const uint32_t pv = verdict[parentOf[node]]; if ((pv & 18u) == 18u) { verdict[node] = (uint8_t)(v | 16u); changed = 1u; } }
            }
            __syncthreads();
            if (changed == 0u) break;
        }
        // per thread: its nodes' replay records and bucket records, then a block-wide exclusive scan of both
        uint32_t myReplay = 0, myBuckets = 0;
        for (uint32_t node = t; node < iw.flatCount; node += 1024u) {
            const uint32_t v = verdict[node];
            if (v & 16u) { nNodes++; if (a.occlusion && (v & 4u)) myReplay++; if (v & 8u) myBuckets += recordsOf[node]; }
        }
        uint32_t inclR = myReplay, inclB = myBuckets;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t r = (uint32_t)__shfl_up((int)inclR, o), b = (uint32_t)__shfl_up((int)inclB, o); if (lane >= (uint32_t)o) { inclR += r; inclB += b; } }
        if (lane == 63u) { waveSum[0][wave] = inclR; waveSum[1][wave] = inclB; }
        __syncthreads();
        uint32_t baseR = 0, baseB = 0, totalR = 0, totalB = 0;
        for (uint32_t w = 0; w < 16u; w++) { if (w < wave) { baseR += waveSum[0][w]; baseB += waveSum[1][w]; } totalR += waveSum[0][w]; totalB += waveSum[1][w]; }
        if (t == 0) {
            bases[0] = totalR ? atomicAdd(&a.counters[CNT_REPLAY_NODES], totalR) : 0u;
            bases[1] = totalB ? atomicAdd(&a.counters[a.bucketCounter], totalB) : 0u;
        }
        __syncthreads();
        uint32_t slotR = bases[0] + baseR + inclR - myReplay, slotB = bases[1] + baseB + inclB - myBuckets;
        for (uint32_t node = t; node < iw.flatCount; node += 1024u) {
            const uint32_t v = verdict[node];
            if (!(v & 16u)) continue;
            if (a.occlusion && (v & 4u)) {
                const FlatNode fn = a.flatNodes[iw.flatBase + node];
                if (slotR < a.recordCapacity) a.replayNodes[slotR] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (fn.nodeId & 0x3FFFFFFFu)};
                else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
                slotR++;
            }
            if (v & 8u) {
                const FlatNode fn = a.flatNodes[iw.flatBase + node];
                const brmi_group_page_map_entry pe = sc.groupPageMap[fn.pageMapIndex];
                const uint32_t segFirst = fn.segFirstCount & 0xFFFFu, segCount = fn.segFirstCount >> 16;
                for (uint32_t k = 0; k < recordsOf[node]; k++, slotB++) {
                    if (slotB >= a.recordCapacity) { atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u); continue; }
                    BucketRecord b;
                    b.instanceIndex = instIndex; b.groupIdPacked = fn.ownerGroup & 0x7FFFFFFFu;
                    b.meshletIndexAndCount = (min(a.factor, segCount - k * a.factor) << 16) | ((segFirst + k * a.factor) & 0xFFFFu);
                    b.pageSlabDescriptorIndex = pe.slabDescriptorIndex; b.pageSlabByteOffset = pe.slabByteOffset;
                    b.firstBit = iw.bitBase + fn.firstBitRel + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                    buckets[slotB] = b;
                }
            }
        }
    }
    // statistics: one atomic per wave and counter on a stripe of its own
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nNodes += (uint32_t)__shfl_xor((int)nNodes, o);
    if (lane == 0) {
        uint32_t* stripe = a.counters + CNT_STRIPES + ((blockIdx.x * 16u + wave) & (CNT_STRIPE_COUNT - 1u)) * CNT_STRIPE_WORDS;
        if (nTested) atomicAdd(&stripe[0], nTested);
        if (nVisible) atomicAdd(&stripe[1], nVisible);
        if (nNodes) atomicAdd(&stripe[2], nNodes);
    }
}

// Level-synchronous flat traversal for scenes of MANY draws (Zorah-class: 100 k instances of two 600-node hierarchies, 83 k of them in the frustum,
// seven nodes reached in each on average).  One wave per draw -- the LDS walk above, or the flat evaluation of every node -- keeps the chip busy with
// chains of dependent loads: 83 k instances x ~25 us of chain over the ~3 k waves that fit is 0.7 ms (measured 0.7 - 1.0 ms, and its one
// reservation per draw on one counter, served at ~90 per microsecond, costs as much again), and evaluating all 600 nodes of every instance is 3 ms.
// Here a LANE is a task: level 0 takes a draw (K1, then the root), every later level a (instance, flat position) record of the frontier the level
// before wrote; a node's children sit side by side in the breadth-first tables (FlatNode::children), so their pre-filter is eight independent
// loads.  Appends are aggregated per workgroup (three atomics per 256 tasks).  Same tests, same arithmetic, same records as the walk; the order of
// the bucket records differs, which the survivor ranking (a bit per (instance, segment, meshlet)) does not see.  flatMaxDepth launches.
template <bool FIRST>
__global__ void __launch_bounds__(256) k_cull_flat_level(CullArgs a, uint32_t level, const NodeRecord* in, NodeRecord* out, BucketRecord* buckets) {
    wave_prio<PRIO_CULL>();
    __shared__ uint32_t waveTot[3][4], bases[3];
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* lodCam = sc.cullingCameras + viewId;
    const bool ortho = cam->isOrtho != 0;
    const f3 camPos{lodCam->positionWorldSpace[0], lodCam->positionWorldSpace[1], lodCam->positionWorldSpace[2]};
    const float zNear = lodCam->zNear, threshold = lodCam->errorOverDistanceThreshold;
    const m4 view = load_m4(&cam->view[0][0]);
    const uint32_t count = FIRST ? sc.activeDrawCount : min(a.counters[CNT_FRONTIER0 + level], a.recordCapacity);
    uint32_t* nextCount = &a.counters[CNT_FRONTIER0 + level + 1u];
    const uint32_t rounded = (count + 255u) & ~255u;        // workgroup-uniform trip count (barriers inside)
    uint32_t nTested = 0, nVisible = 0, nNodes = 0;
    for (uint32_t idx = blockIdx.x * 256u + t; idx < rounded; idx += gridDim.x * 256u) {
        bool have = idx < count;
        uint32_t instIndex = 0, pos = 0;
        InstanceWalk iw{0u, 0u, 0u, 0u}; brmi_per_mesh_instance inst{};
        if (have) {
            if (FIRST) instIndex = sc.activeDraws[idx];
            else { const NodeRecord rec = in[idx]; instIndex = rec.instanceIndex; pos = rec.nodeIdPacked; }
            iw = a.instanceWalk[instIndex]; inst = sc.perMeshInstance[instIndex];
            if (iw.flatCount == 0u) have = false;         // (launch_cull takes this path only when every mesh has flat tables)
        }
        bool hidden = false, leafOk = false;
        uint32_t childMask = 0, firstChild = 0, nChunks = 0, slabDesc = 0, slabOff = 0;
        FlatNode fn{};
        if (have) {
            const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
            const m4 model = load_m4(&obj->model[0][0]);
            const float scale = max_axis_scale(model);
            const f3 instC{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}; const float instR = inst.boundingSphere[3];
            if (FIRST) {   // K1 (PureComputeObjectCullCS)
                const f3 c = to_view_space(instC, model, view);
                const float r = instR * scale;
                const bool bad = isnan(c.x) || isnan(c.y) || isnan(c.z) || isinf(c.x) || isinf(c.y) || isinf(c.z) || isnan(r) || isinf(r);
                nTested++;
                if (bad || sphere_culled(a, cam, c, r)) have = false; else nVisible++;
            }
            if (have) {
                nNodes++;
                fn = a.flatNodes[iw.flatBase + pos];
                const bool skinned = iw.skinned != 0u;
                const bool internal = (fn.info & 1u);
                const f3 cullC = skinned ? instC : f3{fn.cull[0], fn.cull[1], fn.cull[2]};
                const float cullR = skinned ? instR : fn.cull[3];
                const f3 cVS = to_view_space(cullC, model, view);
                const float rW = cullR * scale;
                const bool inFrustum = !sphere_culled(a, cam, cVS, rW);
                if (inFrustum && internal) {
                    const f3 lc = xyz(mul_point(f3{fn.lod[0], fn.lod[1], fn.lod[2]}, model));
                    const float e = projected_error(lc, fn.lod[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                    if (e >= threshold) {
                        hidden = a.occlusion && occlusion_test_prev(a, cam, cullC, cullR, load_m4(&obj->prevModel[0][0]));
                        if (!hidden) {
                            // the children that pass as children: in the frustum and, internal ones, above the error threshold (computeCulling.hlsl:477-530)
                            firstChild = fn.children & 0xFFFFu;
                            const uint32_t cc = min(fn.children >> 16, BRMI_BVH_MAX_CHILDREN);
#pragma unroll
                            for (uint32_t c = 0; c < BRMI_BVH_MAX_CHILDREN; c++) if (c < cc) {
                                const FlatNode* ch = a.flatNodes + (iw.flatBase + firstChild + c);
                                const float4 cs = *reinterpret_cast<const float4*>(ch->cull), ls = *reinterpret_cast<const float4*>(ch->lod);
                                const float chErr = ch->maxQuadricError; const uint32_t chInfo = ch->info;
                                const f3 ccVS = to_view_space(skinned ? instC : f3{cs.x, cs.y, cs.z}, model, view);
                                bool pre = !sphere_culled(a, cam, ccVS, (skinned ? instR : cs.w) * scale);
                                if (pre && (chInfo & 1u)) {
                                    const f3 wc = xyz(mul_point(f3{ls.x, ls.y, ls.z}, model));
                                    pre = projected_error(wc, ls.w * scale, chErr, scale, camPos, zNear, ortho) >= threshold;
                                }
                                childMask |= pre ? (1u << c) : 0u;
                            }
                        }
                    }
                } else if (inFrustum) {
                    const FlatLeaf fl = a.flatLeaves[iw.flatBase + pos];
                    const f3 gc = xyz(mul_point(f3{fl.group[0], fl.group[1], fl.group[2]}, model));
                    const float eod = projected_error(gc, fl.group[3] * scale, fn.maxQuadricError, scale, camPos, zNear, ortho);
                    bool ok = eod >= threshold;
                    if (ok && ((fn.info >> 1) & 1u)) {      // refined_child_suppresses
                        const f3 cc = xyz(mul_point(f3{fl.child[0], fl.child[1], fl.child[2]}, model));
                        const float ce = projected_error(cc, fl.child[3] * scale, fl.childParentError, scale, camPos, zNear, ortho);
                        if (!(ce < threshold)) ok = false;
                    }
                    if (ok && ((fn.info >> 2) & 1u)) {
                        const brmi_group_page_map_entry pe = sc.groupPageMap[fn.pageMapIndex];
                        slabDesc = pe.slabDescriptorIndex; slabOff = pe.slabByteOffset;
                        leafOk = slabDesc != 0u;
                        nChunks = leafOk ? ((fn.segFirstCount >> 16) + a.factor - 1u) / a.factor : 0u;
                    }
                }
            }
        }
        // one reservation per workgroup and output: frontier records, bucket records, replay nodes
        const uint32_t mine[3] = {(uint32_t)__popc(childMask), nChunks, hidden ? 1u : 0u};
        uint32_t incl[3] = {mine[0], mine[1], mine[2]};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 3; q++) { const uint32_t v = (uint32_t)__shfl_up((int)incl[q], o); if (lane >= (uint32_t)o) incl[q] += v; }
        }
        __syncthreads();                                 // the previous round's bases have been read
        if (lane == 63u) { waveTot[0][wave] = incl[0]; waveTot[1][wave] = incl[1]; waveTot[2][wave] = incl[2]; }
        __syncthreads();
        if (t < 3u) {
            const uint32_t total = waveTot[t][0] + waveTot[t][1] + waveTot[t][2] + waveTot[t][3];
            uint32_t* counter = t == 0u ? nextCount : (t == 1u ? &a.counters[a.bucketCounter] : &a.counters[CNT_REPLAY_NODES]);
            bases[t] = total ? atomicAdd(counter, total) : 0u;
        }
        __syncthreads();
        uint32_t slot[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { uint32_t wb = 0; for (uint32_t w = 0; w < 4u; w++) if (w < wave) wb += waveTot[q][w]; slot[q] = bases[q] + wb + incl[q] - mine[q]; }
        for (uint32_t m = childMask; m != 0u; m &= m - 1u, slot[0]++) {
            if (slot[0] < a.recordCapacity) out[slot[0]] = NodeRecord{instIndex, firstChild + (uint32_t)__ffs((int)m) - 1u};
            else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
        }
        if (nChunks != 0u) {
            const uint32_t segFirst = fn.segFirstCount & 0xFFFFu, segCount = fn.segFirstCount >> 16;
            for (uint32_t k = 0; k < nChunks; k++, slot[1]++) {
                if (slot[1] >= a.recordCapacity) { atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u); continue; }
                BucketRecord b;
                b.instanceIndex = instIndex; b.groupIdPacked = fn.ownerGroup & 0x7FFFFFFFu;
                b.meshletIndexAndCount = (min(a.factor, segCount - k * a.factor) << 16) | ((segFirst + k * a.factor) & 0xFFFFu);
                b.pageSlabDescriptorIndex = slabDesc; b.pageSlabByteOffset = slabOff;
                b.firstBit = iw.bitBase + fn.firstBitRel + k * a.factor; b.pad0 = 0; b.pad1 = 0;
                buckets[slot[1]] = b;
            }
        }
        if (hidden) {
            if (slot[2] < a.recordCapacity) a.replayNodes[slot[2]] = NodeRecord{instIndex, 0x80000000u | (1u << 30) | (fn.nodeId & 0x3FFFFFFFu)};
            else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
        }
    }
    // statistics: one atomic per wave and counter on a stripe of its own
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nNodes += (uint32_t)__shfl_xor((int)nNodes, o); nTested += (uint32_t)__shfl_xor((int)nTested, o); nVisible += (uint32_t)__shfl_xor((int)nVisible, o); }
    if (lane == 0) {
        uint32_t* stripe = a.counters + CNT_STRIPES + ((blockIdx.x * 4u + wave) & (CNT_STRIPE_COUNT - 1u)) * CNT_STRIPE_WORDS;
        if (nTested) atomicAdd(&stripe[0], nTested);
        if (nVisible) atomicAdd(&stripe[1], nVisible);
        if (nNodes) atomicAdd(&stripe[2], nNodes);
    }
}

// ComputeSkinnedMeshletBounds (workGraphCulling.hlsl:1405-1467): the meshlet sphere moved by every bone the meshlet lists,
// merged pairwise into one enclosing sphere
BRMI_DEV float4 skinned_meshlet_bounds(const brmi_scene_buffers& sc, const brmi_meshlet_descriptor* desc, const brmi_page_header* hdr, const uint8_t* page, uint32_t slot, float4 staticBounds) {
    const uint32_t boneCount = desc->boneCount;
    if (slot == 0xFFFFFFFFu || boneCount == 0u || sc.skinningMatrices == nullptr) return staticBounds;
    const uint32_t* boneList = reinterpret_cast<const uint32_t*>(page + hdr->boneIndexStreamOffset + desc->boneListOffset * 4u);
    const f3 c0{staticBounds.x, staticBounds.y, staticBounds.z};
    f3 mc{0.0f, 0.0f, 0.0f}; float mr = 0.0f; bool init = false;
    for (uint32_t b = 0; b < boneCount; b++) {
        const m4 m = load_bone_skin_matrix(sc.skinningMatrices, slot, boneList[b]);
        const f3 tc = xyz(mul_point(c0, m));
        const float tr = staticBounds.w * max_axis_scale(m);
        if (!init) { mc = tc; mr = tr; init = true; continue; }
        const f3 delta = tc - mc;
        const float dist = length3(delta);
        if (dist + tr <= mr) continue;
        if (dist + mr <= tr) { mc = tc; mr = tr; continue; }
        const float newRadius = 0.5f * (dist + mr + tr);
        const float t = (newRadius - mr) / max2(dist, 1e-12f);
        mc = mc + delta * t;
        mr = newRadius;
    }
    if (!init) return staticBounds;
    return make_float4(mc.x, mc.y, mc.z, mr * (1.0f + 1e-5f));
}

// K3: per-meshlet cull ---------------------------------------------------------------------------
// SIDE: workgroups behind the first `mainBlocks` run the second half of the light clustering (page prefix + fill; four clusters each)
struct LcRide { uint32_t mainBlocks; ClusterArgs lc; };
// SIDE == 2 (split frames, round 4): the workgroups behind the first `mainBlocks` clear the visibility keys instead.  As riders of the traversal's
// launch the clear's 8,192 and the light clustering's 3,456 single-wave workgroups inherit the walk's 150 VGPRs; beside another frame's shading waves
// (3 x 136 of a SIMD's 512 registers taken) each of them waited for a shading wave to retire -- k_cull_hierarchy 38 us alone, 97 us in flight
// (kernel trace), on the chain the next frame waits for.  This kernel's waves fit the gap (<= 104 registers), and the light clustering of a split
// frame runs on the shading stream (brmi_execute_split).
struct ClearRide { uint32_t mainBlocks; ulonglong2* vis2; uint64_t n2; uint32_t clearBlocks; };
template <int SIDE>
__global__ void __launch_bounds__(256) k_cull_clusters(CullArgs a, const BucketRecord* buckets, TempVisible* temp, uint32_t* bitmask, uint8_t* blockDirty,
                                                       typename std::conditional<SIDE == 1, LcRide, typename std::conditional<SIDE == 2, ClearRide, NoSide>::type>::type ride) {
    wave_prio<PRIO_CULL>();
    uint32_t mainBlocks = gridDim.x;
    if constexpr (SIDE == 1) {
        mainBlocks = ride.mainBlocks;
        if (blockIdx.x >= ride.mainBlocks) { lc_fill_block(ride.lc, blockIdx.x - ride.mainBlocks, threadIdx.x); return; }
    }
    if constexpr (SIDE == 2) {
        mainBlocks = ride.mainBlocks;
        if (blockIdx.x >= ride.mainBlocks) {
            const uint64_t stride = (uint64_t)ride.clearBlocks * 256u;
            for (uint64_t i = (uint64_t)(blockIdx.x - ride.mainBlocks) * 256u + threadIdx.x; i < ride.n2; i += stride) ride.vis2[i] = make_ulonglong2(BRMI_VIS_EMPTY, BRMI_VIS_EMPTY);
            return;
        }
    }
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t bucketCount = min(a.counters[a.bucketCounter], a.recordCapacity);
    if (a.feedback && blockIdx.x == 0u && threadIdx.x == 0u) __hip_atomic_store(a.feedback + (a.phase == 1u ? 2 : 4), bucketCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* lodCam = sc.cullingCameras + viewId;
    const bool ortho = cam->isOrtho != 0;
    const f3 camPos{lodCam->positionWorldSpace[0], lodCam->positionWorldSpace[1], lodCam->positionWorldSpace[2]};
    const float zNear = lodCam->zNear, threshold = lodCam->errorOverDistanceThreshold;
    const m4 view = load_m4(&cam->view[0][0]);
    uint32_t* tempCount = &a.counters[a.phase == 2 ? CNT_TEMP_VISIBLE2 : CNT_TEMP_VISIBLE];
    // one lane per (bucket, meshlet-in-bucket): `factor` lanes cooperate on a record
    const uint64_t totalLanes = (uint64_t)bucketCount * a.factor;
    const uint64_t rounded = (totalLanes + 255ull) & ~255ull;      // workgroup-uniform trip count (barriers inside)
    // (round 5: the survivors' and the occluded meshlets' slots are reserved once per workgroup -- a Zorah-class frame tests 850 k meshlets, and one
    // atomic with return per wave and list, 11 k on two lines, was a third of the kernel)
    __shared__ uint32_t waveSurv[4], waveOccl[4], blockSlots[2];
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < rounded; idx += (uint64_t)mainBlocks * blockDim.x) {
        bool survives = false, occluded = false, tested = false;
        uint4 packed = make_uint4(0, 0, 0, 0);
        uint32_t bit = 0;
        uint4 againLo = make_uint4(0u, 0u, 0u, 0u); uint2 againHi = make_uint2(0u, 0u);      // the replay record of an occluded meshlet (as plain words: a BucketRecord object here left an unused 36 B stack slot in the kernel's descriptor)
        if (idx < totalLanes) {
            const uint32_t bi = (uint32_t)(idx / a.factor), m = (uint32_t)(idx % a.factor);
            const BucketRecord b = buckets[bi];
            const uint32_t count = b.meshletIndexAndCount >> 16, first = b.meshletIndexAndCount & 0xFFFFu;
            if (m < count && b.pageSlabDescriptorIndex != 0u) {
                tested = true;
                const bool replay = (b.groupIdPacked >> 31) != 0;
                const uint32_t lm = first + m;
                const uint8_t* slab = sc.slabs[b.pageSlabDescriptorIndex];
                const brmi_page_header* hdr = reinterpret_cast<const brmi_page_header*>(slab + b.pageSlabByteOffset);
                if (lm < hdr->meshletCount) {
                    const brmi_meshlet_descriptor* desc = reinterpret_cast<const brmi_meshlet_descriptor*>(slab + b.pageSlabByteOffset + hdr->descriptorOffset + lm * 64u);
                    float4 bounds = *reinterpret_cast<const float4*>(desc->bounds);
                    const uint32_t triAndRefined = desc->triangleCountAndRefinedGroup;
                    const brmi_per_mesh_instance inst = sc.perMeshInstance[b.instanceIndex];
                    if ((sc.perMesh[inst.perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) != 0u)
                        bounds = skinned_meshlet_bounds(sc, desc, hdr, slab + b.pageSlabByteOffset, inst.skinningInstanceSlot, bounds);
                    const brmi_clod_mesh_metadata* md = sc.meshMetadata + sc.clodOffsets[b.instanceIndex].clodMeshMetadataIndex;
                    const m4 model = load_m4(&sc.perObject[inst.perObjectBufferIndex].model[0][0]);
                    const float scale = max_axis_scale(model);
                    const f3 cVS = to_view_space(f3{bounds.x, bounds.y, bounds.z}, model, view);
                    const float rW = bounds.w * scale;
                    survives = replay || !sphere_outside_frustum(cVS, rW, cam->clippingPlanes);
                    if (survives) {
                        const int refined = (int)(triAndRefined >> 16) - 1;
                        if (refined_child_suppresses(sc, md->groupsBase, (uint32_t)refined, refined >= 0, model, scale, camPos, zNear, threshold, ortho)) survives = false;
                    }
                    if (survives && a.bandActive) {
                        // tile-bounds test of the screen-tile split: conservative sphere vs the band's two planes
                        if (dot3(f3{a.bandTop[0], a.bandTop[1], a.bandTop[2]}, cVS) < -rW || dot3(f3{a.bandBottom[0], a.bandBottom[1], a.bandBottom[2]}, cVS) < -rW) survives = false;
                    }
                    if (survives && stripe_on(a.stripes) && stripe_rejects(a.stripes, cam, cVS, rW)) survives = false;      // no row of this GPU's: dropped, not replayed
                    if (survives && a.occlusion &&
                        occlusion_test(a, cam, replay, f3{bounds.x, bounds.y, bounds.z}, bounds.w, cVS, rW, sc.perObject + inst.perObjectBufferIndex)) {
                        survives = false;
                        if (!replay) {
                            occluded = true;
                            againLo = make_uint4(b.instanceIndex, 0x80000000u | (b.groupIdPacked & 0x7FFFFFFFu), (1u << 16) | (lm & 0xFFFFu), b.pageSlabDescriptorIndex);
                            againHi = make_uint2(b.pageSlabByteOffset, b.firstBit + m);
                        }
                    }
                    if (survives) {
                        packed = pack_visible_cluster(viewId, b.instanceIndex, lm, b.groupIdPacked & 0x7FFFFFFFu, b.pageSlabDescriptorIndex, b.pageSlabByteOffset);
                        bit = b.firstBit + m;
                    }
                }
            }
        }
        {   // statistics: one atomic per wave on one of 64 stripes
            const uint64_t tm = __ballot(tested);
            if (tm != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&a.counters[CNT_STRIPES + (blockIdx.x & (CNT_STRIPE_COUNT - 1u)) * CNT_STRIPE_WORDS + STRIPE_MESHLETS_TESTED], (uint32_t)__popcll(tm));
        }
        occluded = occluded && a.occlusion && a.phase == 1u;
        const uint64_t survM = __ballot(survives), occlM = __ballot(occluded);
        const uint32_t wv = threadIdx.x >> 6;
        __syncthreads();                                  // the previous round's slots have been read
        if ((threadIdx.x & 63u) == 0u) { waveSurv[wv] = (uint32_t)__popcll(survM); waveOccl[wv] = (uint32_t)__popcll(occlM); }
        __syncthreads();
        if (threadIdx.x < 2u) {
            const uint32_t* ws = threadIdx.x == 0u ? waveSurv : waveOccl;
            const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3];
            blockSlots[threadIdx.x] = total ? atomicAdd(threadIdx.x == 0u ? tempCount : &a.counters[CNT_REPLAY_MESHLETS], total) : 0u;
        }
        __syncthreads();
        uint32_t slot = blockSlots[0] + lane_rank(survM), rs = blockSlots[1] + lane_rank(occlM);
        for (uint32_t w = 0; w < wv; w++) { slot += waveSurv[w]; rs += waveOccl[w]; }
        if (occluded) {      // (only set in phase 1 of a frame with occlusion culling)
            if (rs < a.recordCapacity) { uint4* dst = reinterpret_cast<uint4*>(&a.replayBuckets[rs]); dst[0] = againLo; dst[1] = make_uint4(againHi.x, againHi.y, 0u, 0u); }
            else atomicAdd(&a.counters[CNT_DROPPED_RECORDS], 1u);
        }
        if (survives) {
            if (slot < a.visibleCapacity) {
                TempVisible t; t.packed = packed; t.bit = bit; t.pad0 = t.pad1 = t.pad2 = 0;
                temp[slot] = t;
                atomicOr(&bitmask[bit >> 5], 1u << (bit & 31u));
                blockDirty[bit >> 16] = 1;        // (the bit's 2048-word block of the ranking; every writer stores 1)
            } else atomicAdd(&a.counters[CNT_DROPPED_CLUSTERS], 1u);
        }
    }
}

// rank = exclusive popcount scan over the bitmask ----------------------------------------------
constexpr uint32_t SCAN_BLOCK_WORDS = 2048;   // words per workgroup (256 threads x 8)

// blockDirty: a block no survivor set a bit in sums to zero unread and needs no word prefixes (nobody asks for the rank of a bit that is not set): phase 2
// of a Zorah-class frame places a hundred clusters in a bitmask of 12.8 M words.
__global__ void __launch_bounds__(256) k_scan_reduce(const uint32_t* bitmask, uint32_t totalWords, uint32_t* blockSums, const uint8_t* blockDirty) {
    __shared__ uint32_t partial[4];
    if (blockDirty[blockIdx.x] == 0) { if (threadIdx.x == 0) blockSums[blockIdx.x] = 0u; return; }
    const uint32_t base = blockIdx.x * SCAN_BLOCK_WORDS;
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_BLOCK_WORDS; i += 256) { const uint32_t w = base + i; if (w < totalWords) s += __popc(bitmask[w]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o);
    if ((threadIdx.x & 63u) == 0) partial[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) blockSums[blockIdx.x] = partial[0] + partial[1] + partial[2] + partial[3];
}

// single workgroup: exclusive scan of blockSums in place; total -> counters[outIndex] (clamped to capacity)
__global__ void __launch_bounds__(1024) k_scan_blocks(uint32_t* blockSums, uint32_t nBlocks, uint32_t* counters, uint32_t outIndex, uint32_t capacity, uint32_t usedIndex, uint32_t* hostFeedback) {
    __shared__ uint32_t waveTotals[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nBlocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nBlocks ? blockSums[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((threadIdx.x & 63u) >= (uint32_t)o) incl += t; }
        if ((threadIdx.x & 63u) == 63u) waveTotals[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t waveBase = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) waveBase += waveTotals[w];
        const uint32_t c = carry;
        if (i < nBlocks) blockSums[i] = c + waveBase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + waveBase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t placed = min(carry, capacity - (usedIndex == 0xFFFFFFFFu ? 0u : min(counters[usedIndex], capacity)));
        counters[outIndex] = placed;
        if (hostFeedback) __hip_atomic_store(hostFeedback, placed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (phase 2: the host's hint for the next frames)
    }
}

__global__ void __launch_bounds__(256) k_scan_words(const uint32_t* bitmask, uint32_t totalWords, const uint32_t* blockSums, uint32_t* wordPrefix, const uint8_t* blockDirty) {
    __shared__ uint32_t waveTotals[4];
    __shared__ uint32_t carry;
    if (blockDirty[blockIdx.x] == 0) return;
    if (threadIdx.x == 0) carry = blockSums[blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * SCAN_BLOCK_WORDS;
    for (uint32_t chunk = 0; chunk < SCAN_BLOCK_WORDS; chunk += 256) {
        const uint32_t w = base + chunk + threadIdx.x;
        const uint32_t v = w < totalWords ? __popc(bitmask[w]) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((threadIdx.x & 63u) >= (uint32_t)o) incl += t; }
        if ((threadIdx.x & 63u) == 63u) waveTotals[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t waveBase = 0;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) waveBase += waveTotals[k];
        const uint32_t c = carry;
        if (w < totalWords) wordPrefix[w] = c + waveBase + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c + waveBase + incl;
        __syncthreads();
    }
}

// The three kernels above as ONE launch for bitmasks of up to 64 blocks (131 k words: every benchmark frame has 6-10): a block counts its 2048
// words, publishes the sum with the launch's epoch in one 64-bit agent-scope store, waits for the sums of the blocks before it (all <= 64
// blocks are resident at once: 256 CUs), and scans its words from there; the last block writes the total.  Two launches less per culling
// phase -- with another frame's shading pass filling the chip every small launch of the geometry stream waits 5-15 us for its slots.
constexpr uint32_t SCAN_CHAIN_BLOCKS = 64;
__global__ void __launch_bounds__(256) k_scan_chained(const uint32_t* bitmask, uint32_t totalWords, unsigned long long* agg, uint32_t epoch, uint32_t* wordPrefix,
                                                     uint32_t* counters, uint32_t outIndex, uint32_t capacity, uint32_t usedIndex, uint32_t* hostFeedback) {
    wave_prio<PRIO_SCAN>();
    __shared__ uint32_t waveTotals[4];
    __shared__ uint32_t blockPrefix, ticket;
    // the block's place in the chain is the order in which blocks START (a ticket), not blockIdx: a block only ever waits for blocks that are
    // running already, whatever order the dispatcher picks (agg[64] is the ticket word; the last block puts it back to zero)
    if (threadIdx.x == 0) ticket = (uint32_t)__hip_atomic_fetch_add(&agg[SCAN_CHAIN_BLOCKS], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t block = ticket;
    const uint32_t base = block * SCAN_BLOCK_WORDS + threadIdx.x * 8u;       // eight consecutive words per thread
    uint32_t pc[8], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { const uint32_t w = base + k; pc[k] = w < totalWords ? (uint32_t)__popc(bitmask[w]) : 0u; sum += pc[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((threadIdx.x & 63u) >= (uint32_t)o) incl += t; }
    if ((threadIdx.x & 63u) == 63u) waveTotals[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t waveBase = 0, blockSum = 0;
    for (uint32_t k = 0; k < 4u; k++) { if (k < (threadIdx.x >> 6)) waveBase += waveTotals[k]; blockSum += waveTotals[k]; }
    if (threadIdx.x == 0) __hip_atomic_store(&agg[block], ((unsigned long long)epoch << 32) | blockSum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64u) {
        uint32_t mine = 0;
        if (threadIdx.x < block) {
            unsigned long long v;
            do { v = __hip_atomic_load(&agg[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != epoch);
            mine = (uint32_t)v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o);
        if (threadIdx.x == 0) blockPrefix = mine;
    }
    __syncthreads();
    uint32_t run = blockPrefix + waveBase + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { const uint32_t w = base + k; if (w < totalWords) wordPrefix[w] = run; run += pc[k]; }
    if (block == gridDim.x - 1u && threadIdx.x == 0) {
        __hip_atomic_store(&agg[SCAN_CHAIN_BLOCKS], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every ticket of this launch has been taken
        const uint32_t placed = min(blockPrefix + blockSum, capacity - (usedIndex == 0xFFFFFFFFu ? 0u : min(counters[usedIndex], capacity)));
        counters[outIndex] = placed;
        if (hostFeedback) __hip_atomic_store(hostFeedback, placed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (phase 2: the host's hint for the next frames)
    }
}

// Places every survivor at its rank.  Also accumulates the vertex / triangle totals of the clusters that will be
// rasterised (statistics for the algorithmic-byte count), one atomic per wave: a per-cluster atomic on three
// shared words would serialise the whole rasteriser (~90 same-address atomics per microsecond).
// LOCAL_RANK (survivor bitmasks of <= LOCAL_RANK_WORDS words: every BASELINE-class scene): no rank kernel runs before this one; every
// workgroup that has survivors to place scans the popcounts itself into LDS (a few thousand L2-resident words), workgroup 0 publishes the
// total.  One ~5 us launch less per culling phase.  Larger bitmasks take the three scan launches (a variant with one LDS prefix per group
// of words up to 2^17 words was measured on the dense frame's 25 k words: 55 us per phase against 27).
#ifndef BRMI_LOCAL_RANK_WORDS
#define BRMI_LOCAL_RANK_WORDS 8192
#endif
constexpr uint32_t LOCAL_RANK_WORDS = BRMI_LOCAL_RANK_WORDS;
struct LocalRank { uint32_t totalWords, outIndex, usedIndex; uint32_t* hostFeedback; };
// Round 6: the lists of a frame that holds clusters back (brmi_raster.hip).  drawList == nullptr: every cluster of the visible list is rasterised in list order, as before.
struct DrawLists {
    uint32_t* drawList; HeldRecord* held; uint32_t countAll;      // countAll: phase 2 of such a frame -- every placed cluster is drawn and counts
    // the prediction's inputs (phase 1): the chain the reference's test just read, the per-meshlet boxes, the per-object constants, the chain's size
    // (box_behind_chain, brmi_internal.h: the same question the re-test asks -- of the PREVIOUS frame's chain with the previous frame's matrices.  The reference's own
    // test asks it of the meshlet's SPHERE with four texels of a coarser mip, and lets through twice the clusters that own a pixel: profiles/r06_experiments.md --
    // a box's near corner is the surface's, a sphere's near point half a cluster in front of it.)
    const MeshletBox* boxes; const uint32_t* pageBoxBase; const float* objConst; uint32_t maxTexels; BoxViewport vp; HzbDesc hzb;
};
template <bool LOCAL_RANK, bool HOLD>      // HOLD: phase 1 of a frame that holds clusters back (the prediction and the two lists; the plain instantiations keep their registers)
__global__ void __launch_bounds__(256) k_scatter_visible(const TempVisible* temp, uint32_t* counters, uint32_t tempCountIndex, const uint32_t* bitmask,
                                                        const uint32_t* wordPrefix, uint4* visible, uint32_t baseIndexCounter, uint32_t capacity, uint32_t visibleCapacity,
                                                        brmi_scene_buffers sc, ClusterSetup* setup, uint32_t resolveCapacity, ClusterUv* clusterUv, LocalRank lr, DrawLists dl) {
    wave_prio<PRIO_SCAN>();
    const uint8_t* const* slabs = sc.slabs;
    const uint32_t n = min(counters[tempCountIndex], visibleCapacity);
    const uint32_t base = baseIndexCounter == 0xFFFFFFFFu ? 0u : counters[baseIndexCounter];
    const uint32_t rounded = (n + 63u) & ~63u;
    __shared__ uint32_t prefixLds[LOCAL_RANK ? LOCAL_RANK_WORDS : 1];
    __shared__ uint32_t waveTotals[4];
    if (LOCAL_RANK) {
        if (blockIdx.x != 0u && blockIdx.x * blockDim.x >= rounded) return;       // nothing to place here (workgroup 0 always publishes the total)
        const uint32_t span = (lr.totalWords + 255u) / 256u;
        const uint32_t w0 = threadIdx.x * span, w1 = min(w0 + span, lr.totalWords);
        uint32_t sum = 0;
        for (uint32_t w = w0; w < w1; w++) sum += __popc(bitmask[w]);
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((threadIdx.x & 63u) >= (uint32_t)o) incl += t; }
        if ((threadIdx.x & 63u) == 63u) waveTotals[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t waveBase = 0, all = 0;
        for (uint32_t w = 0; w < 4; w++) { if (w < (threadIdx.x >> 6)) waveBase += waveTotals[w]; all += waveTotals[w]; }
        uint32_t run = waveBase + incl - sum;
        for (uint32_t w = w0; w < w1; w++) { prefixLds[w] = run; run += __popc(bitmask[w]); }
        if (blockIdx.x == 0u && threadIdx.x == 0u) {
            const uint32_t placed = min(all, capacity - (lr.usedIndex == 0xFFFFFFFFu ? 0u : min(counters[lr.usedIndex], capacity)));
            counters[lr.outIndex] = placed;
            if (lr.hostFeedback) __hip_atomic_store(lr.hostFeedback, placed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
    // (statistics double as the reservation of the clusters' tables in the resolve arena.  Round 5: ONE atomic per workgroup -- vertices in the low,
    // triangles in the high half of one 64-bit word -- instead of three per wave on one cache line: a Zorah-class frame places 490 k clusters, and
    // 23 k same-line atomics at ~90 per microsecond were 250 us, the whole kernel.  The count of placed clusters is one thread's sum.)
    __shared__ unsigned long long blockBase;
    __shared__ uint32_t waveV[4], waveT[4], waveD[4], waveH[4], blockDraw, blockHeld;
    unsigned long long drawnVT = 0ull;      // this thread's share of the vertex | triangle << 32 sums of the clusters on the draw list
    if (blockIdx.x == 0u && threadIdx.x == 0u) { const uint32_t room = base < capacity ? capacity - base : 0u; atomicAdd(&counters[CNT_RASTER_CLUSTERS], min(n, room)); if (resolveCapacity == 0u && n != 0u) atomicOr(&counters[CNT_RESOLVE_SPILL], 1u); }
    const uint32_t rounded256 = (n + 255u) & ~255u;      // workgroup-uniform trip count (barriers inside)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded256; i += gridDim.x * blockDim.x) {
        uint32_t verts = 0, tris = 0, placed = 0, dst = 0;
        TempVisible t{};
        const uint8_t* slab = nullptr; uint32_t pageOff = 0;
        const brmi_page_header* hdr = nullptr; const brmi_meshlet_descriptor* desc = nullptr;
        uint32_t held = 0u, boxIndex = 0xFFFFFFFFu, perObject = 0u;
        if (i < n) {
            t = temp[i];
            const uint32_t w = t.bit >> 5, b = t.bit & 31u;
            const uint32_t rank = (LOCAL_RANK ? prefixLds[w] : wordPrefix[w]) + __popc(bitmask[w] & ((1u << b) - 1u));
            dst = base + rank;
            if (dst < capacity) {
                visible[dst] = t.packed;
                slab = slabs[vc_slab(t.packed)];
                pageOff = vc_page_offset(t.packed);
                hdr = reinterpret_cast<const brmi_page_header*>(slab + pageOff);
                desc = reinterpret_cast<const brmi_meshlet_descriptor*>(slab + pageOff + hdr->descriptorOffset + vc_meshlet(t.packed) * 64u);
                verts = min((desc->bitsAndVertexCount >> 24) & 0xFFu, BRMI_MESHLET_MAX_VERTS);
                tris = min(desc->triangleCountAndRefinedGroup & 0xFFFFu, BRMI_MESHLET_MAX_TRIS);
                placed = 1;
                if (HOLD) {
                    // (round 6: the cluster is visible as far as the reference's tests go and has its place in the list; is it worth drawing FIRST?)
                    const brmi_per_mesh_instance inst = sc.perMeshInstance[vc_instance(t.packed)];
                    perObject = inst.perObjectBufferIndex;
                    const uint32_t boxBase = dl.pageBoxBase[(size_t)vc_slab(t.packed) * 1024u + (pageOff >> 18)];
                    if (boxBase != 0xFFFFFFFFu && (sc.perMesh[inst.perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) == 0u) {
                        boxIndex = boxBase + vc_meshlet(t.packed);
                        const MeshletBox bx = dl.boxes[boxIndex];
                        const float* oc = dl.objConst + (size_t)perObject * OBJ_CONST_FLOATS;
                        if (bx.valid && box_behind_chain(dl.hzb, bx, oc + 36, oc + 52, dl.vp, dl.maxTexels)) held = 1u;
                    }
                }
            }
        }
        uint32_t inclV = verts, inclT = tris;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inclV, o), tt = (uint32_t)__shfl_up((int)inclT, o);
            if ((threadIdx.x & 63u) >= (uint32_t)o) { inclV += v; inclT += tt; }
        }
        // the draw list and the held records: slots per workgroup, like the arena's
        const bool isHeld = HOLD && placed != 0u && held != 0u, isDraw = HOLD && placed != 0u && !isHeld;
        const uint64_t drawM = __ballot(isDraw), heldM = __ballot(isHeld);
        __syncthreads();                                  // the previous round's base and wave sums have been read
        if ((threadIdx.x & 63u) == 63u) { waveV[threadIdx.x >> 6] = inclV; waveT[threadIdx.x >> 6] = inclT; waveD[threadIdx.x >> 6] = (uint32_t)__popcll(drawM); waveH[threadIdx.x >> 6] = (uint32_t)__popcll(heldM); }
        __syncthreads();
        if (threadIdx.x == 0u) {
            const unsigned long long totV = (unsigned long long)waveV[0] + waveV[1] + waveV[2] + waveV[3], totT = (unsigned long long)waveT[0] + waveT[1] + waveT[2] + waveT[3];
            blockBase = (totV | totT) ? atomicAdd(reinterpret_cast<unsigned long long*>(&counters[CNT_SUM_VERTS_LO]), totV | (totT << 32)) : 0ull;
        }
        if (HOLD && threadIdx.x == 64u) { const uint32_t tot = waveD[0] + waveD[1] + waveD[2] + waveD[3]; blockDraw = tot ? atomicAdd(&counters[CNT_DRAW1], tot) : 0u; }
        if (HOLD && threadIdx.x == 128u) { const uint32_t tot = waveH[0] + waveH[1] + waveH[2] + waveH[3]; blockHeld = tot ? atomicAdd(&counters[CNT_HELD1], tot) : 0u; }
        __syncthreads();
        if (dl.countAll && placed) drawnVT += (unsigned long long)verts | ((unsigned long long)tris << 32);
        if (isDraw || isHeld) {
            uint32_t slot = (isDraw ? blockDraw : blockHeld) + lane_rank(isDraw ? drawM : heldM);
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) slot += isDraw ? waveD[w] : waveH[w];
            if (isDraw) { dl.drawList[slot] = dst; drawnVT += (unsigned long long)verts | ((unsigned long long)tris << 32); }
            else dl.held[slot] = HeldRecord{dst, boxIndex, perObject, verts | (tris << 16)};
        }
        unsigned long long baseV = (uint32_t)blockBase, baseT = blockBase >> 32;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) { baseV += waveV[w]; baseT += waveT[w]; }
        if (placed) {
            // resolve the cluster for the rasteriser and the G-buffer pass
            const uint32_t instanceIndex = vc_instance(t.packed);
            const brmi_per_mesh_instance inst = sc.perMeshInstance[instanceIndex];
            const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
            ClusterSetup cs;
            cs.posBase = slab + pageOff + hdr->positionBitstreamOffset + desc->positionBitOffset;
            cs.triBase = slab + pageOff + hdr->triangleStreamOffset + desc->triangleByteOffset;
            cs.nrmBase = slab + pageOff + hdr->normalArrayOffset + desc->vertexAttributeOffset * 4u;
            const brmi_per_mesh* pm = sc.perMesh + inst.perMeshBufferIndex;
            cs.counts = verts | (tris << 8) | ((hdr->compressedPositionQuantExp & 0xFFu) << 16) | (((obj->objectFlags & BRMI_OBJECT_FLAG_REVERSE_WINDING) != 0 ? 1u : 0u) << 24)
                      | ((pm->vertexFlags & BRMI_VERTEX_SKINNED) ? BRMI_CS_SKINNED : 0u) | ((hdr->attributeMask & BRMI_PAGE_ATTRIBUTE_JOINTS) ? BRMI_CS_JOINTS : 0u)
                      | ((hdr->attributeMask & BRMI_PAGE_ATTRIBUTE_WEIGHTS) ? BRMI_CS_WEIGHTS : 0u);
            if (clusterUv) {      // scenes with textured / alpha-tested materials: UV set 0 of the meshlet and the material's class
                const uint32_t mflags = sc.materials[pm->materialDataIndex].materialFlags;
                const bool layerTextures = openpbr_has_textures(sc.openpbrMaterials + sc.materials[pm->materialDataIndex].openPBRMaterialDataIndex);
                cs.counts |= ((mflags & BRMI_MATERIAL_ALPHA_TEST) ? BRMI_CS_ALPHA : 0u) | (((mflags & BRMI_MATERIAL_ANY_TEXTURE) || layerTextures) ? BRMI_CS_TEXTURED : 0u);
                ClusterUv cu{nullptr, nullptr, nullptr, 0, 0u};
                if (hdr->attributeMask & BRMI_PAGE_ATTRIBUTE_COLOR) { cs.counts |= BRMI_CS_COLOR; cu.color = slab + pageOff + hdr->colorArrayOffset + desc->vertexAttributeOffset * 4u; }
                if (hdr->uvSetCount != 0u) {
                    cu.desc = slab + pageOff + hdr->uvDescriptorOffset + (vc_meshlet(t.packed) * hdr->uvSetCount) * 32u;
                    cu.stream = slab + pageOff + *reinterpret_cast<const uint32_t*>(slab + pageOff + hdr->uvBitstreamDirectoryOffset);
                    cu.directory = (int32_t)hdr->uvBitstreamDirectoryOffset - (int32_t)(hdr->uvDescriptorOffset + (vc_meshlet(t.packed) * hdr->uvSetCount) * 32u);
                    cu.setCount = hdr->uvSetCount;
                }
                clusterUv[dst] = cu;
            }
            cs.jointDelta = (int32_t)(hdr->jointArrayOffset + desc->vertexAttributeOffset * 32u) - (int32_t)(hdr->normalArrayOffset + desc->vertexAttributeOffset * 4u);
            cs.weightDelta = (int32_t)(hdr->weightArrayOffset + desc->vertexAttributeOffset * 32u) - (int32_t)(hdr->normalArrayOffset + desc->vertexAttributeOffset * 4u);
            cs.perObjectIndex = inst.perObjectBufferIndex; cs.instanceIndex = instanceIndex; cs.viewId = vc_view(t.packed);
            cs.materialDataIndex = pm->materialDataIndex; cs.normalMatrixIndex = obj->normalMatrixBufferIndex;
            const unsigned long long v0 = baseV + (inclV - verts), t0 = baseT + (inclT - tris);
            const bool fits = v0 + verts <= resolveCapacity && t0 + tris <= resolveCapacity;
            cs.vertBase = fits ? (uint32_t)v0 : BRMI_ARENA_NONE; cs.triBase32 = fits ? (uint32_t)t0 : BRMI_ARENA_NONE;
            if (!fits && resolveCapacity != 0u) atomicOr(&counters[CNT_RESOLVE_SPILL], 1u);      // selects the G-buffer kernel variant (brmi_gbuffer); (capacity 0 = a frame without tables: flagged once, below -- half a million atomics on one word were 70 us of the 8K frame)
            setup[dst] = cs;
        }
    }
    if (HOLD || dl.countAll) {      // one atomic per wave and launch (the sums are statistics: brmi_algorithmic_bytes)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) drawnVT += (unsigned long long)__shfl_xor((long long)drawnVT, o);
        if ((threadIdx.x & 63u) == 0u && drawnVT != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[CNT_DRAWN_VT]), drawnVT);
    }
}

static inline uint32_t grid_for(uint64_t items, uint32_t block, uint32_t maxBlocks) {
    uint64_t g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > maxBlocks) g = maxBlocks;
    return (uint32_t)g;
}

// phase 2 starts from the replay buffers: the replayed meshlets become the first bucket records, the replayed nodes the
// level-0 frontier; per-level frontier counters start from zero
__global__ void k_seed_phase2(uint32_t* counters, uint32_t capacity) { seed_phase2(counters, capacity, threadIdx.x); }

int launch_cull(brmi_pass* p, uint32_t phase, hipStream_t s) {
    if (int rc = ensure_frame_constants(p, s)) return rc;
    if (phase != 1 && phase != 2) return fail(p, BRMI_ERR_INVALID, "brmi_cull: phase %u (1 or 2)", phase);
    if (phase == 2 && !p->cfg.enableOcclusionCulling) return fail(p, BRMI_ERR_STATE, "brmi_cull: phase 2 needs a pass created with enableOcclusionCulling");
    if (phase == 2 && !p->hzbValid) return fail(p, BRMI_ERR_STATE, "brmi_cull: phase 2 needs brmi_build_hzb on the phase-1 depth first");
    CullArgs a;
    a.sc = p->scene; a.counters = p->counters();
    a.instanceBitBase = p->wsPtr<uint32_t>(p->ws.instanceBitBase); a.segPrefix = p->wsPtr<uint32_t>(p->ws.segPrefix);
    a.flatNodes = p->wsPtr<FlatNode>(p->ws.flatNodes); a.flatLeaves = p->wsPtr<FlatLeaf>(p->ws.flatLeaves); a.instanceWalk = p->wsPtr<InstanceWalk>(p->ws.instanceWalk); a.debugStamps = p->wsPtr<unsigned long long>(p->ws.debugStamps);
    a.recordCapacity = p->cfg.maxTraversalRecords; a.visibleCapacity = p->cfg.maxVisibleClusters;
    uint32_t f = p->cfg.phase2ExpansionFactor; f = f < 1 ? 1 : (f > 64 ? 64 : f);
    { uint32_t n = 1; for (uint32_t c = 2; c <= 64; c <<= 1) if (c <= f) n = c; f = n; }
    a.factor = f; a.phase = phase; a.packedFlat = 0u; a.wideFlat = 0u;
    a.feedback = p->ensureFeedback() ? p->phase2FeedbackDev : nullptr;

    // the band test's two planes through the eye only bound a row band under a symmetric perspective projection: an orthographic or
    // off-centre camera keeps the frustum test alone (the rasteriser's row filter still confines the band; nothing is lost but the early cull)
    const bool symmetricPerspective = p->camHost.isOrtho == 0 && p->camHost.projection[2][0] == 0.0f && p->camHost.projection[2][1] == 0.0f &&
                                      p->camHost.projection[3][0] == 0.0f && p->camHost.projection[3][1] == 0.0f;
    a.bandActive = ((p->bandY0 != 0 || p->bandY1 != p->cfg.height) && symmetricPerspective) ? 1u : 0u;
    for (int k = 0; k < 3; k++) { a.bandTop[k] = p->bandPlaneTop[k]; a.bandBottom[k] = p->bandPlaneBottom[k]; }
    a.bandPlanes = reinterpret_cast<const float4*>(p->wsPtr<m4>(p->ws.frameConst) + 3);
    a.stripes = p->stripes;
    const brmi_pass* chain = p->chainOwner(phase);      // frames in flight: phase 1 reads the chain of the pass that rendered the frame before
    a.occlusion = (p->cfg.enableOcclusionCulling && chain->hzbValid && p->camHost.isOrtho == 0) ? 1u : 0u;
    a.replayNodes = p->wsPtr<NodeRecord>(p->ws.replayNodes); a.replayBuckets = p->wsPtr<BucketRecord>(p->ws.replayBuckets);
    a.hzb = chain->hzbDesc();
    if (phase == 1) {
        // Round 6: this frame holds clusters back when the pass can (brmi_set_scene), the previous frame's chain is there to predict from, and the frames before had
        // enough clusters for the two extra launches to pay (the count is a host-mapped word a frame or two old; either way the same image)
        const uint32_t lastVisible = p->phase2FeedbackHost ? reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[3] : 0u;
        const uint32_t lastPhase2 = p->phase2FeedbackHost ? reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[0] : 0xFFFFFFFFu;
        // (a still camera's small frame too -- but not a frame of a few hundred clusters: a rank's sky band of 195 clusters pays the three extra launches for nothing; holdFloor)
        p->holdThisFrame = p->holdEnabled && a.occlusion != 0u && (p->holdMinClusters == 0u || (lastVisible != 0xFFFFFFFFu && lastVisible >= p->holdMinClusters) ||
                                                                  (p->holdStillMax != 0u && lastPhase2 < p->holdStillMax && lastVisible != 0xFFFFFFFFu && lastVisible >= p->holdFloor));
    }
    a.frontier0Counter = phase == 1 ? (uint32_t)CNT_FRONTIER0 : (uint32_t)CNT_REPLAY_NODES;
    a.bucketCounter = CNT_BUCKETS;   // phase 2: seeded with the replayed meshlets, the bucket array is the replay buffer itself
    NodeRecord* fa = p->wsPtr<NodeRecord>(p->ws.frontierA); NodeRecord* fb = p->wsPtr<NodeRecord>(p->ws.frontierB);
    BucketRecord* buckets = phase == 1 ? p->wsPtr<BucketRecord>(p->ws.buckets) : a.replayBuckets;
    TempVisible* temp = p->wsPtr<TempVisible>(p->ws.tempVisible);
    uint32_t* bitmask = p->wsPtr<uint32_t>(phase == 1 ? p->ws.bitmask1 : p->ws.bitmask2);
    uint8_t* blockDirty = p->wsPtr<uint8_t>(p->ws.blockDirty) + (phase == 1 ? 0u : p->scanBlocks + 1u);
    uint32_t* wordPrefix = p->wsPtr<uint32_t>(p->ws.wordPrefix); uint32_t* blockSums = p->wsPtr<uint32_t>(p->ws.blockSums);

    const uint32_t maxBlocks = 1024;
    bool lightGridRides = false;      // this call's launches carry the light clustering (brmi_execute)
    bool flatLevels = false;          // phase 1 ran the level-synchronous flat traversal: nothing is left for the walk or the level kernels
    // one launch of k_cull_hierarchy for the meshes that fit its LDS frontier, the level kernels for the rest (or for everything: tests)
    const bool hierarchy = p->minLevelWidth <= HIER_CAP_MAX && !p->forceLevelKernels;
    const bool levelKernels = p->maxLevelWidth > p->spillWidth || p->forceLevelKernels;
    const uint32_t* meshWidth = p->wsPtr<uint32_t>(p->ws.meshLevelWidth);
    // meshes of both kinds: the one-launch walk also starts the wide ones and hands their frontiers to the level kernels (spill mode)
    const bool spillMode = hierarchy && levelKernels;
    a.meshLevelWidth = meshWidth; a.levelKernelsWidthLo = 0u;
    if (spillMode) a.frontier0Counter = CNT_FRONTIER0;
    const uint32_t widthAll = spillMode ? 0xFFFFFFFFu : 0u, spillAbove = spillMode ? p->spillWidth : 0xFFFFFFFFu;
    if (phase == 1) {
        if (!p->frameStateCleared) BRMI_HIP(p, hipMemsetAsync(p->counters(), 0, p->ws.frameClearBytes, s));      // counters + both survivor bitmasks
        p->frameStateCleared = false;
        // (the first ceil(draws / 8) waves of the walk take eight draws each: hierarchies of <= 8 nodes, the flat tables of brmi_set_scene)
        a.packedFlat = (hierarchy && !p->hostFlatNodes.empty() && p->packedFlat) ? 1u : 0u;
        // (not beside another frame's shading half: a 1024-thread workgroup with 40 KB of LDS waits long for a CU that can take it, and the
        // dense frame in flight went 0.795 -> 0.91 ms; alone the same frame's cull stage goes 0.180 -> 0.142 ms)
        a.wideFlat = (hierarchy && p->anyWideFlat && p->wideFlat && !p->splitFrame) ? 1u : 0u;
        // Scenes of very many draws (Zorah-class): the level-synchronous flat traversal, a lane per (instance, node) task, one launch per level of the
        // deepest hierarchy (k_cull_flat_level).  Its launches carry no riders: the visibility clear moves onto k_cull_clusters (ClearRide) and the
        // light clustering is launched by the frame where it finds none done.
        // (also, from a quarter of that count on, in a SPLIT frame of a scene whose hierarchies need the 24 KB-frontier variant of the walk: its single-wave workgroups of 150 registers
        // and 24 KB of LDS find few slots beside another frame's shading waves -- San-Miguel-class, 7,577 draws: 0.848 -> 0.823 ms in flight although the stage alone is 0.107 -> 0.119;
        // the dense and Bistro-class frames, with fewer draws, lose either way)
        const uint32_t draws = p->scene.activeDrawCount, minDraws = std::max(1u, p->flatLevelsMinDraws);
        flatLevels = hierarchy && p->allMeshesFlat && !p->forceLevelKernels && (draws >= minDraws || (p->splitFrame && p->maxLevelWidth > 256u && draws >= std::max(1u, minDraws / 4u)));
        if (flatLevels) {
            if (p->clearVisibilityWithTraversal && (p->bandPixelCount & 1ull) == 0ull) { p->clearVisibilityWithTraversal = false; p->clearVisibilityWithClusterCull = true; }
            else if (p->clearVisibilityWithTraversal) flatLevels = false;      // (an odd pixel count: the riding clear of the walk handles it)
        }
        if (flatLevels) {
            hipLaunchKernelGGL(k_cull_flat_level<true>, dim3(grid_for(p->scene.activeDrawCount, 256, 8192)), dim3(256), 0, s, a, 0u, (const NodeRecord*)nullptr, fb, buckets);
            // (frontier sizes live on the device; every level strides a fixed grid and a workgroup that finds none of its tasks leaves after one load)
            for (uint32_t level = 1; level < p->flatMaxDepth; level++)
                hipLaunchKernelGGL(k_cull_flat_level<false>, dim3(2048), dim3(256), 0, s, a, level, (level & 1u) ? fb : fa, (level & 1u) ? fa : fb, buckets);
            BRMI_LAUNCH_CHECK(p, "k_cull_flat_level");
        }
        const dim3 hgrid(std::min(std::max(1u, p->scene.activeDrawCount), 16384u) + (a.packedFlat ? (p->scene.activeDrawCount + 7u) / 8u : 0u));
        if (hierarchy && !flatLevels) {
            // ONE launch: the 6 KB-frontier variant when every mesh is narrow (<= 256 nodes per level), else the 24 KB variant for all meshes up
            // to 1024 (two launches, one per class, ran one after the other: San-Miguel-class cull 178 -> 140 us with one)
            const bool wide = p->maxLevelWidth > 256u;
            const uint32_t widthHi = spillMode ? widthAll : (wide ? HIER_CAP_MAX : 256u);
            if (p->clearVisibilityWithTraversal) {
                SideJobs sj{reinterpret_cast<ulonglong2*>(static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel), p->bandPixelCount >> 1, hgrid.x, p->clearRiderBlocks, cluster_args_of(p)};
                const dim3 grid(hgrid.x + sj.clearBlocks + p->numLightClusters);
                if (wide) hipLaunchKernelGGL((k_cull_hierarchy<false, 1024, BRMI_HIER_STAGE_WIDE, true>), grid, dim3(64), 0, s, a, buckets, meshWidth, 0u, widthHi, spillAbove, fa, sj);
                else hipLaunchKernelGGL((k_cull_hierarchy<false, 256, 128, true>), grid, dim3(64), 0, s, a, buckets, meshWidth, 0u, widthHi, spillAbove, fa, sj);
                p->clearVisibilityWithTraversal = false; lightGridRides = true;
            } else if (wide) hipLaunchKernelGGL((k_cull_hierarchy<false, 1024, BRMI_HIER_STAGE_WIDE>), hgrid, dim3(64), 0, s, a, buckets, meshWidth, 0u, widthHi, spillAbove, fa, NoSide{});
            else hipLaunchKernelGGL((k_cull_hierarchy<false, 256, 128>), hgrid, dim3(64), 0, s, a, buckets, meshWidth, 0u, widthHi, spillAbove, fa, NoSide{});
        }
        if (a.wideFlat && !flatLevels) hipLaunchKernelGGL(k_cull_flat_wide, dim3(std::min(std::max(1u, p->scene.activeDrawCount), 4096u)), dim3(1024), 0, s, a, buckets);
        if (levelKernels && !spillMode && !flatLevels) hipLaunchKernelGGL(k_cull_instances, dim3(grid_for(p->scene.activeDrawCount, 256, maxBlocks)), dim3(256), 0, s, a, fa);
        BRMI_LAUNCH_CHECK(p, "k_cull_instances");
    } else {
        // brmi_execute seeds in the tail of the depth-chain build that precedes this call (one launch less)
        if (!p->phase2Seeded) hipLaunchKernelGGL(k_seed_phase2, dim3(1), dim3(128), 0, s, p->counters(), a.recordCapacity);
        p->phase2Seeded = false;
        const NoSide none{};
        if (hierarchy && p->maxLevelWidth <= 256u) hipLaunchKernelGGL((k_cull_hierarchy<true, 256, 128>), dim3(2048), dim3(64), 0, s, a, buckets, meshWidth, 0u, 256u, spillAbove, fa, none);
        else if (hierarchy) hipLaunchKernelGGL((k_cull_hierarchy<true, 1024, BRMI_HIER_STAGE_WIDE>), dim3(2048), dim3(64), 0, s, a, buckets, meshWidth, 0u, spillMode ? widthAll : HIER_CAP_MAX, spillAbove, fa, none);
    }
    // frontier sizes are only known on the device: size the grids for the worst case that can matter
    const uint32_t travGrid = grid_for(std::min<uint64_t>(p->cfg.maxTraversalRecords, (uint64_t)p->scene.lodNodeCount * 4 + 4096), 256, maxBlocks);
    const uint32_t levelLaunches = spillMode ? std::min(p->maxLevels, std::max(1u, p->spillLevels)) : p->maxLevels;
    // (phase 1 of a scene whose every hierarchy is evaluated flat leaves nothing for the level kernels)
    const bool flatCoversPhase1 = phase == 1 && hierarchy && spillMode && p->allMeshesFlat && a.wideFlat != 0u && !p->forceLevelKernels;
    for (uint32_t level = 0; level < levelLaunches && levelKernels && !flatCoversPhase1 && !flatLevels; level++) {
        // without the walk in front (forced level kernels) phase 2 reads level 0 from the replay buffer; then ping-pong like phase 1 (level 0 writes fb)
        const NodeRecord* in = level == 0 ? ((phase == 1 || spillMode) ? fa : a.replayNodes) : ((level & 1u) ? fb : fa);
        hipLaunchKernelGGL(k_traverse, dim3(travGrid), dim3(256), 0, s, a, level, in, (level & 1u) ? fa : fb, buckets);
        BRMI_LAUNCH_CHECK(p, "k_traverse");
    }
    // grid-stride kernels that usually find little to do: a few hundred workgroups retire in ~3 us, a thousand in ~6
    // Frames of hundreds of thousands of records (Zorah-class: 180 k bucket records, 490 k survivors) need more than that: a lane's chain of dependent loads
    // (record -> page header -> descriptor -> instance -> object -> depth chain) three or four times over was most of both kernels.  The counts of the frames
    // before (host-mapped words the kernels write; read without waiting, any value is safe: the kernels stride their grids) size the launches.
    uint32_t smallGrid = phase == 1 ? 512u : 128u, scatterGrid = smallGrid;
    if (p->phase2FeedbackHost) {
        const uint32_t lastBuckets = reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[phase == 1 ? 2 : 4];
        const uint32_t lastVisible = reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[phase == 1 ? 3 : 0];      // (0xFFFFFFFF = unknown is clamped below)
        smallGrid = std::max(smallGrid, grid_for((uint64_t)std::min<uint32_t>(lastBuckets, p->cfg.maxTraversalRecords) * f, 256, 8192));
        scatterGrid = std::max(scatterGrid, grid_for(std::min<uint32_t>(lastVisible, p->cfg.maxVisibleClusters), 256, 8192));
    }
    if (lightGridRides) {
        hipLaunchKernelGGL(k_cull_clusters<1>, dim3(smallGrid + (p->numLightClusters + 3u) / 4u), dim3(256), 0, s, a, buckets, temp, bitmask, blockDirty, LcRide{smallGrid, cluster_args_of(p)});
        p->lightGridDone = true;
    } else if (phase == 1 && p->clearVisibilityWithClusterCull) {
        const uint32_t clearBlocks = 2048;
        hipLaunchKernelGGL(k_cull_clusters<2>, dim3(smallGrid + clearBlocks), dim3(256), 0, s, a, buckets, temp, bitmask, blockDirty,
                           ClearRide{smallGrid, reinterpret_cast<ulonglong2*>(static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel), p->bandPixelCount >> 1, clearBlocks});
        p->clearVisibilityWithClusterCull = false;
    } else hipLaunchKernelGGL(k_cull_clusters<0>, dim3(smallGrid), dim3(256), 0, s, a, buckets, temp, bitmask, blockDirty, NoSide{});
    BRMI_LAUNCH_CHECK(p, "k_cull_clusters");
    // phase 2 appends behind the phase-1 clusters: its capacity is what phase 1 left
    const uint32_t outIndex = phase == 1 ? CNT_VISIBLE : CNT_VISIBLE2, usedIndex = phase == 1 ? 0xFFFFFFFFu : (uint32_t)CNT_VISIBLE;
    // every workgroup of the scatter scanning the bitmask itself pays off while the bitmask is small (BASELINE-class scenes: ~1-4 k words);
    // from 8 k words on the three scan launches are faster (dense frame, 25 k words: 27 us against 55 us per phase)
    // phase 2: the ranking kernel also tells the host how many clusters it placed (launch_raster's hint for the frames that follow)
    if (phase == 2 && p->phase2DirectMax != 0u) (void)p->ensureFeedback();
    uint32_t* feedback = phase == 2 ? p->phase2FeedbackDev : (p->phase2FeedbackDev ? p->phase2FeedbackDev + 3 : nullptr);      // (phase 1: word 3, the sizes of the next frames' launches)
    const bool localRank = p->totalWords <= LOCAL_RANK_WORDS && !p->forceLevelKernels;
    if (localRank) {
        // ranked inside the scatter kernel
    } else {
        if (p->scanBlocks <= SCAN_CHAIN_BLOCKS && p->scanChained) {
            if (++p->scanEpoch == 0u) p->scanEpoch = 1u;
            hipLaunchKernelGGL(k_scan_chained, dim3(p->scanBlocks), dim3(256), 0, s, bitmask, p->totalWords, p->wsPtr<unsigned long long>(p->ws.scanAgg), p->scanEpoch, wordPrefix,
                               p->counters(), outIndex, p->cfg.maxVisibleClusters, usedIndex, feedback);
        } else {
            hipLaunchKernelGGL(k_scan_reduce, dim3(p->scanBlocks), dim3(256), 0, s, bitmask, p->totalWords, blockSums, blockDirty);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, blockSums, p->scanBlocks, p->counters(), outIndex, p->cfg.maxVisibleClusters, usedIndex, feedback);
            hipLaunchKernelGGL(k_scan_words, dim3(p->scanBlocks), dim3(256), 0, s, bitmask, p->totalWords, blockSums, wordPrefix, blockDirty);
        }
    }
    // (a frame whose G-buffer pass works without the per-cluster tables reserves nothing in the resolve arena: every cluster is marked "no tables")
    if (phase == 1) p->inlineResolve = resolve_inline_frame(p);
    const uint32_t arenaCapacity = p->inlineResolve ? 0u : p->resolveCapacity;
    auto scatter = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(scatterGrid), dim3(256), 0, s, temp, p->counters(), (uint32_t)(phase == 1 ? CNT_TEMP_VISIBLE : CNT_TEMP_VISIBLE2), bitmask, wordPrefix,
                           static_cast<uint4*>(p->res[BRMI_RES_VISIBLE_CLUSTERS]), phase == 1 ? 0xFFFFFFFFu : (uint32_t)CNT_VISIBLE, p->cfg.maxVisibleClusters, p->cfg.maxVisibleClusters, p->scene, p->wsPtr<ClusterSetup>(p->ws.clusterSetup), arenaCapacity,
                           (p->sceneHasTextures || p->sceneHasAlphaTest || p->sceneHasVertexColors) ? p->wsPtr<ClusterUv>(p->ws.clusterUv) : nullptr, LocalRank{p->totalWords, outIndex, usedIndex, feedback},
                           (phase == 1 && p->holdThisFrame) ? DrawLists{p->wsPtr<uint32_t>(p->ws.drawList), p->wsPtr<HeldRecord>(p->ws.heldRecords), 0u, p->wsPtr<MeshletBox>(p->ws.meshletBoxes), p->wsPtr<uint32_t>(p->ws.pageBoxBase),
                                                                      p->wsPtr<float>(p->ws.objConst), p->holdMaxTexels, BoxViewport{(float)p->cfg.width, (float)p->cfg.height, 0.0f, 0.0f, 0, (int)p->bandY0, (int)p->cfg.width - 1, (int)p->bandY1 - 1}, a.hzb}
                                                          : DrawLists{nullptr, nullptr, p->holdThisFrame ? 1u : 0u, nullptr, nullptr, nullptr, 0u, BoxViewport{}, HzbDesc{}});
    };
    const bool holdLists = phase == 1 && p->holdThisFrame;
    if (localRank) { if (holdLists) scatter(k_scatter_visible<true, true>); else scatter(k_scatter_visible<true, false>); }
    else { if (holdLists) scatter(k_scatter_visible<false, true>); else scatter(k_scatter_visible<false, false>); }
    BRMI_LAUNCH_CHECK(p, "compaction");
    return BRMI_OK;
}


// Round 6: the object-space box of every meshlet of every resident page (brmi_setup; MeshletBox in brmi_internal.h).  One workgroup per page, a wave per meshlet.
__global__ void __launch_bounds__(256) k_meshlet_boxes(const uint8_t* const* slabs, const PageRef* pages, MeshletBox* boxes) {
    const PageRef pr = pages[blockIdx.x];
    const uint8_t* page = slabs[pr.slab] + pr.byteOffset;
    const brmi_page_header* hdr = reinterpret_cast<const brmi_page_header*>(page);
    const uint32_t lane = threadIdx.x & 63u;
    const bool float3 = (hdr->compressedPositionQuantExp & 0xFFu) == BRMI_POSITION_FORMAT_FLOAT3;
    for (uint32_t m = threadIdx.x >> 6; m < pr.meshletCount; m += 4u) {
        const brmi_meshlet_descriptor* desc = reinterpret_cast<const brmi_meshlet_descriptor*>(page + hdr->descriptorOffset + m * 64u);
        const uint32_t verts = min((desc->bitsAndVertexCount >> 24) & 0xFFu, BRMI_MESHLET_MAX_VERTS);
        f3 lo{3.0e38f, 3.0e38f, 3.0e38f}, hi{-3.0e38f, -3.0e38f, -3.0e38f};
        bool finite = true;
        if (float3) for (uint32_t v = lane; v < verts; v += 64u) {
            const float* pp = reinterpret_cast<const float*>(page + hdr->positionBitstreamOffset + desc->positionBitOffset + v * 12u);
            const f3 q{pp[0], pp[1], pp[2]};
            finite = finite && fabsf(q.x) < 1.0e30f && fabsf(q.y) < 1.0e30f && fabsf(q.z) < 1.0e30f;      // (NaN fails the comparison)
            lo = min3v(lo, q); hi = max3v(hi, q);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo.x = min2(lo.x, __shfl_xor(lo.x, o)); lo.y = min2(lo.y, __shfl_xor(lo.y, o)); lo.z = min2(lo.z, __shfl_xor(lo.z, o));
            hi.x = max2(hi.x, __shfl_xor(hi.x, o)); hi.y = max2(hi.y, __shfl_xor(hi.y, o)); hi.z = max2(hi.z, __shfl_xor(hi.z, o));
        }
        finite = __all(finite);
        if (lane == 0u) boxes[pr.boxBase + m] = MeshletBox{{lo.x, lo.y, lo.z}, (float3 && verts != 0u && finite) ? 1u : 0u, {hi.x, hi.y, hi.z}, 0u};
    }
}

int launch_meshlet_boxes(brmi_pass* p, hipStream_t s) {
    if (p->hostPageRefs.empty()) return BRMI_OK;
    hipLaunchKernelGGL(k_meshlet_boxes, dim3((uint32_t)p->hostPageRefs.size()), dim3(256), 0, s, p->scene.slabs, p->wsPtr<PageRef>(p->ws.pageRefs), p->wsPtr<MeshletBox>(p->ws.meshletBoxes));
    BRMI_LAUNCH_CHECK(p, "k_meshlet_boxes");
    return BRMI_OK;
}

}  // namespace brmi
