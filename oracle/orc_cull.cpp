// orc_cull.cpp -- CPU restatement of the hierarchical culling chain (K1-K3).
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows:
//   K1  PureComputeObjectCullCS            BR/shaders/ClusterLOD/computeCulling.hlsl:103-190
//   K2  PureComputeTraverseFrontierCS      BR/shaders/ClusterLOD/computeCulling.hlsl:192-531
//       CLodPrepareRenderableLeaf          BR/shaders/ClusterLOD/workGraphCulling.hlsl:1711-1784
//       CLodRefinedChildSuppressesParent   BR/shaders/ClusterLOD/workGraphCulling.hlsl:1631-1700
//       ProjectedGeometricError            BR/shaders/ClusterLOD/workGraphCulling.hlsl:1522-1541
//   K3  ClusterCullBody                    BR/shaders/ClusterLOD/workGraphCulling.hlsl:2398-2760
//       OcclusionCullingPerspectiveTexture2D  BR/shaders/Include/occlusionCulling.hlsli:165-212
//       sphere_screen_extents              BR/shaders/Include/Misc/sphereScreenExtents.hlsli:14-31
// Static frame: every group is resident (CLodGroupIsResident == true), streaming requests,
// telemetry, VSM, voxel and Reyes branches are out of scope.
//
// The visible set is order-independent; the list is emitted in the canonical order
// (instance index, mesh-local segment index, meshlet offset in segment) so that the cluster index
// stored in the visibility key - and therefore equal-depth tie-breaking - is reproducible.
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

#include "orc_common.h"

namespace orc {

struct HzbView {   // linear-depth mip chain (mip 0 = full resolution), row-major per mip
    const float* data = nullptr; const uint64_t* mipOffsets = nullptr; uint32_t mipCount = 0; uint32_t width = 0, height = 0;
};

static float3 toViewSpace(float3 c, const mat4& model, const mat4& view) {
    float4 w = mulPoint(c, model);
    return xyz(mul(w, view));
}
static bool sphereOutsideFrustum(float3 c, float r, const float planes[6][4]) {
    for (int i = 0; i < 6; i++) {
        float d = dot(float3{planes[i][0], planes[i][1], planes[i][2]}, c) + planes[i][3];
        if (d < -r) return true;
    }
    return false;
}
static float projectedGeometricError(float3 worldCenter, float worldRadius, float errMesh, float errScale, float3 camPos, float zNear, bool ortho) {
    const float wsErr = errMesh * errScale;
    if (ortho) return wsErr;
    float dist = length(worldCenter - camPos);
    float denom = fmax2(dist - worldRadius, zNear);
    return wsErr / denom;
}
static bool refinedChildSuppressesParent(const brmi_scene_buffers& sc, uint32_t groupsBase, uint32_t childLocal, bool hasChild,
                                         const mat4& model, float scale, const brmi_culling_camera& lodCam, bool ortho) {
    if (!hasChild) return false;
    const brmi_lod_group& g = sc.lodGroups[groupsBase + childLocal];
    float3 c = xyz(mulPoint(float3{g.centerAndRadius[0], g.centerAndRadius[1], g.centerAndRadius[2]}, model));
    float r = g.centerAndRadius[3] * scale;
    float eod = projectedGeometricError(c, r, g.maxParentError, scale, float3{lodCam.positionWorldSpace[0], lodCam.positionWorldSpace[1], lodCam.positionWorldSpace[2]}, lodCam.zNear, ortho);
    if (eod < lodCam.errorOverDistanceThreshold) return false;
    return true;   // resident
}

// ceil(log2(x)) clamped to [0, maxMip], evaluated exactly on the float's bits.  The shader calls log2(), whose
// last-bit behaviour next to powers of two differs between GPUs and libm; the exact value is the stable definition.
static uint32_t ceilLog2Clamped(float x, uint32_t maxMip) {
    if (!(x > 1.0f)) return 0;                       // log2 <= 0 (and NaN / -inf) clamp to mip 0
    const uint32_t u = asuint(x);
    const int e = (int)((u >> 23) & 0xFFu) - 127;
    const uint32_t m = (uint32_t)e + ((u & 0x7FFFFFu) ? 1u : 0u);
    return m < maxMip ? m : maxMip;
}

// sphere_screen_extents + OcclusionCullingPerspectiveTexture2D (explicit-parameter overload): everything up to the four taps
struct OccTaps { float L, B, R, T; uint32_t mip, x0, y0, x1, y1, mw, mh; };
static OccTaps occlusionTaps(float viewW, float viewH, float mips, float sx, float sy, float p00, float p11, float3 centerVS, float radius) {
    OccTaps o;
    float3 p = centerVS; p.y = -p.y;
    float rad2 = radius * radius, d = p.z * radius;
    float hv = std::sqrt(p.x * p.x + p.z * p.z - rad2);
    float ha = p.x * hv, hb = p.x * radius, hc = p.z * hv;
    float L = (ha - d) * p00 / (hc + hb);
    float R = (ha + d) * p00 / (hc - hb);
    float vv = std::sqrt(p.y * p.y + p.z * p.z - rad2);
    float va = p.y * vv, vb = p.y * radius, vc = p.z * vv;
    float B = (va - d) * p11 / (vc + vb);
    float T = (va + d) * p11 / (vc - vb);
    L = -L; R = -R;
    o.L = L; o.B = B; o.R = R; o.T = T;
    // vUV = saturate(vLBRT.xwzy * (0.5,-0.5,0.5,-0.5) + 0.5)
    float u0 = saturate(L * 0.5f + 0.5f), v0 = saturate(T * -0.5f + 0.5f), u1 = saturate(R * 0.5f + 0.5f), v1 = saturate(B * -0.5f + 0.5f);
    float ax0 = u0 * viewW, ay0 = v0 * viewH, ax1 = u1 * viewW, ay1 = v1 * viewH;
    float ex = ax1 - ax0, ey = ay1 - ay0;
    o.mip = ceilLog2Clamped(fmax2(ex, ey), (uint32_t)mips - 1u);
    float pu0 = u0 * sx, pv0 = v0 * sy, pu1 = u1 * sx, pv1 = v1 * sy;
    const float ssx = fmax2(sx, 1e-6f), ssy = fmax2(sy, 1e-6f);
    uint32_t hzbW = (uint32_t)std::nearbyint(viewW / ssx), hzbH = (uint32_t)std::nearbyint(viewH / ssy);
    hzbW = hzbW < 1 ? 1 : hzbW; hzbH = hzbH < 1 ? 1 : hzbH;
    uint32_t mw = hzbW >> o.mip, mh = hzbH >> o.mip; mw = mw < 1 ? 1 : mw; mh = mh < 1 ? 1 : mh;
    auto px = [&](float u, uint32_t res) { uint32_t v = (uint32_t)std::floor(u * (float)res); return v > res - 1 ? res - 1 : v; };
    o.x0 = px(pu0, mw); o.y0 = px(pv0, mh); o.x1 = px(pu1, mw); o.y1 = px(pv1, mh); o.mw = mw; o.mh = mh;
    return o;
}
static bool occlusionCulled(const HzbView& hzb, const brmi_camera& cam, const mat4& proj, float3 centerVS, float sphereDepth, float radius) {
    const OccTaps o = occlusionTaps((float)cam.depthResX, (float)cam.depthResY, (float)cam.numDepthMips, cam.UVScaleToNextPowerOf2[0], cam.UVScaleToNextPowerOf2[1], proj.m[0][0], proj.m[1][1], centerVS, radius);
    if (o.mip >= hzb.mipCount) return false;
    const float* m = hzb.data + hzb.mipOffsets[o.mip];
    float d0 = m[(uint64_t)o.y0 * o.mw + o.x0], d1 = m[(uint64_t)o.y0 * o.mw + o.x1], d2 = m[(uint64_t)o.y1 * o.mw + o.x1], d3 = m[(uint64_t)o.y1 * o.mw + o.x0];
    float mx = fmax2(fmax2(d0, d1), fmax2(d2, d3));
    return mx < sphereDepth - radius;
}

// ComputeSkinnedMeshletBounds (workGraphCulling.hlsl:1405-1467): the meshlet sphere moved by every bone the meshlet lists,
// merged pairwise into one enclosing sphere
static void skinnedMeshletBounds(const brmi_scene_buffers& sc, const brmi_meshlet_descriptor& desc, const brmi_page_header& hdr, const uint8_t* slab, uint32_t pageOff,
                                 uint32_t slot, float3& center, float& radius) {
    if (slot == 0xFFFFFFFFu || desc.boneCount == 0u || sc.skinningMatrices == nullptr) return;
    const uint32_t boneListBase = pageOff + hdr.boneIndexStreamOffset + desc.boneListOffset * 4u;
    const float3 c0 = center; const float r0 = radius;
    float3 mc{0, 0, 0}; float mr = 0.0f; bool init = false;
    for (uint32_t b = 0; b < desc.boneCount; b++) {
        uint32_t joint; std::memcpy(&joint, slab + boneListBase + b * 4u, 4);
        const float* prod = sc.skinningMatrices + ((size_t)slot * 64u + joint) * 16u;
        mat4 m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.m[i][j] = prod[j * 4 + i];     // LoadBoneSkinMatrix: transpose(bone * invBind)
        const float3 tc = xyz(mulPoint(c0, m));
        const float tr = r0 * maxAxisScale(m);
        if (!init) { mc = tc; mr = tr; init = true; continue; }
        const float3 delta = tc - mc;
        const float dist = length(delta);
        if (dist + tr <= mr) continue;
        if (dist + mr <= tr) { mc = tc; mr = tr; continue; }
        const float newRadius = 0.5f * (dist + mr + tr);
        const float t = (newRadius - mr) / fmax2(dist, 1e-12f);
        mc = mc + delta * t;
        mr = newRadius;
    }
    if (!init) return;
    center = mc; radius = mr * (1.0f + 1e-5f);
}

struct VisKey { uint32_t inst, seg, off; brmi_visible_cluster packed; };

struct Bucket { uint32_t inst, group, seg, first, count, slab, pageOff, segFirst; bool replay; };
struct NodeRec { uint32_t inst, node; bool allowRefine, replay; };

}  // namespace orc

using namespace orc;

extern "C" {
// Pieces of the culling tests on their own, for tests that hold them against independent restatements (tests/test_oracle_cpu.py).
// in: viewW, viewH, mips, uvScale.x, uvScale.y, proj[0][0], proj[1][1], centre (view space, 3), radius   out: L, B, R, T, mip, x0, y0, x1, y1, mipW, mipH
void orc_occlusion_taps(const float* in, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++, in += 11, out += 11) {
        const OccTaps o = occlusionTaps(in[0], in[1], in[2], in[3], in[4], in[5], in[6], float3{in[7], in[8], in[9]}, in[10]);
        out[0] = o.L; out[1] = o.B; out[2] = o.R; out[3] = o.T; out[4] = (float)o.mip; out[5] = (float)o.x0; out[6] = (float)o.y0; out[7] = (float)o.x1; out[8] = (float)o.y1; out[9] = (float)o.mw; out[10] = (float)o.mh;
    }
}
// in: centre (3), radius, six planes (24)   out: 1 = outside
void orc_sphere_outside_frustum(const float* in, uint32_t* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++, in += 28) { float pl[6][4]; std::memcpy(pl, in + 4, sizeof(pl)); out[i] = sphereOutsideFrustum(float3{in[0], in[1], in[2]}, in[3], pl) ? 1u : 0u; }
}
// in: world centre (3), world radius, mesh-space error, scale, camera position (3), zNear, ortho (0 / 1)   out: error over distance
void orc_projected_error(const float* in, float* out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++, in += 11) out[i] = projectedGeometricError(float3{in[0], in[1], in[2]}, in[3], in[4], in[5], float3{in[6], in[7], in[8]}, in[9], in[10] != 0.0f);
}

typedef struct orc_cull_params {
    uint32_t phase;                 // 1 or 2
    uint32_t enableOcclusion;       // test against `hzb`
    uint32_t phase2ExpansionFactor;
    uint32_t capacity;              // visible cluster capacity
    const float* hzbData; const uint64_t* hzbMipOffsets; uint32_t hzbMipCount;
    // phase 1 -> phase 2 hand-over (replay buffers), owned by the caller
    uint32_t* replayNodes;   uint32_t replayNodeCapacity;   uint32_t* replayNodeCount;      // (inst,node) pairs
    uint32_t* replayMeshlets; uint32_t replayMeshletCapacity; uint32_t* replayMeshletCount; // (inst,seg,localMeshlet,group) quads
} orc_cull_params;

// Linear-depth mip chain used by the occlusion test: mip 0 = the depth map padded to the next power of two
// (padding = "empty", 0x7F7FFFFF, so it never occludes), each further mip the max of 2x2 texels
// (SpdReduce4 in BR/shaders/downsample.hlsl:108-112).  Returns the float count written.
uint64_t orc_build_hzb(const float* depth, uint32_t W, uint32_t H, float* out, uint64_t* mipOffsets, uint32_t* mipCountOut) {
    uint32_t pw = 1, ph = 1; while (pw < W) pw <<= 1; while (ph < H) ph <<= 1;
    uint64_t off = 0; uint32_t mip = 0; uint32_t w = pw, h = ph;
    const float empty = asfloat(BRMI_DEPTH_EMPTY_BITS);
    for (;;) {
        mipOffsets[mip] = off;
        float* dst = out + off;
        if (mip == 0) {
            for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) dst[(uint64_t)y * w + x] = (x < W && y < H) ? depth[(uint64_t)y * W + x] : empty;
        } else {
            const float* src = out + mipOffsets[mip - 1];
            uint32_t sw = pw >> (mip - 1), sh = ph >> (mip - 1); sw = sw < 1 ? 1 : sw; sh = sh < 1 ? 1 : sh;
            for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
                const uint32_t x0 = x * 2 < sw ? x * 2 : sw - 1, x1 = x * 2 + 1 < sw ? x * 2 + 1 : sw - 1, y0 = y * 2 < sh ? y * 2 : sh - 1, y1 = y * 2 + 1 < sh ? y * 2 + 1 : sh - 1;
                dst[(uint64_t)y * w + x] = fmax2(fmax2(src[(uint64_t)y0 * sw + x0], src[(uint64_t)y0 * sw + x1]), fmax2(src[(uint64_t)y1 * sw + x0], src[(uint64_t)y1 * sw + x1]));
            }
        }
        off += (uint64_t)w * h; mip++;
        if (w == 1 && h == 1) break;
        w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
    }
    *mipCountOut = mip;
    return off;
}

// Returns the number of visible clusters written (canonical order).  `scene` holds HOST pointers.
int orc_cull(const brmi_scene_buffers* scp, const orc_cull_params* prm, brmi_visible_cluster* out, uint32_t* outCount, brmi_counters* counters) {
    const brmi_scene_buffers& sc = *scp;
    const brmi_per_frame& pf = sc.perFrame[0];
    const uint32_t viewId = pf.mainCameraIndex;
    const brmi_camera& cam = sc.cameras[viewId];
    const brmi_culling_camera& lodCam = sc.cullingCameras[viewId];
    const bool ortho = cam.isOrtho != 0;
    const mat4& view = M(cam.view);
    HzbView hzb; hzb.data = prm->hzbData; hzb.mipOffsets = prm->hzbMipOffsets; hzb.mipCount = prm->hzbMipCount; hzb.width = cam.depthResX; hzb.height = cam.depthResY;
    const bool occl = prm->enableOcclusion && hzb.data != nullptr && !ortho;
    brmi_counters cnt{};
    std::vector<NodeRec> frontier, next;
    std::vector<Bucket> buckets;
    std::vector<VisKey> visible;
    uint32_t factor = prm->phase2ExpansionFactor; factor = factor < 1 ? 1 : (factor > 64 ? 64 : factor);
    { uint32_t n = 1; for (uint32_t c = 2; c <= 64; c <<= 1) if (c <= factor) n = c; factor = n; }   // PureComputeNormalizePhase2ExpansionFactor

    auto segOfLeaf = [&](const brmi_clod_mesh_metadata& md, const brmi_lod_node& n) -> const brmi_lod_segment& { return sc.lodSegments[md.segmentsBase + n.indexOrOffset]; };

    if (prm->phase == 1) {
        // K1: object cull, seed the root node of every surviving instance
        for (uint32_t d = 0; d < sc.activeDrawCount; d++) {
            const uint32_t ii = sc.activeDraws[d];
            cnt.instancesTested++;
            const brmi_per_mesh_instance& inst = sc.perMeshInstance[ii];
            const mat4& model = M(sc.perObject[inst.perObjectBufferIndex].model);
            float3 c = toViewSpace(float3{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]}, model, view);
            float r = inst.boundingSphere[3] * maxAxisScale(model);
            bool culled = false;
            if (std::isnan(c.x) || std::isnan(c.y) || std::isnan(c.z) || std::isinf(c.x) || std::isinf(c.y) || std::isinf(c.z) || std::isnan(r) || std::isinf(r)) culled = true;
            else culled = sphereOutsideFrustum(c, r, cam.clippingPlanes);
            if (culled) continue;
            cnt.instancesVisible++;
            const brmi_clod_mesh_metadata& md = sc.meshMetadata[sc.clodOffsets[ii].clodMeshMetadataIndex];
            frontier.push_back({ii, md.rootNode, true, false});
        }
    } else {
        // phase 2: seed from the replay buffers (SeedPureComputeReplayNodesCS / ...ClustersCS)
        for (uint32_t i = 0; i < *prm->replayNodeCount; i++) frontier.push_back({prm->replayNodes[2 * i], prm->replayNodes[2 * i + 1], true, true});
        for (uint32_t i = 0; i < *prm->replayMeshletCount; i++) {
            const uint32_t ii = prm->replayMeshlets[4 * i], segI = prm->replayMeshlets[4 * i + 1], lm = prm->replayMeshlets[4 * i + 2], grp = prm->replayMeshlets[4 * i + 3];
            const brmi_clod_mesh_metadata& md = sc.meshMetadata[sc.clodOffsets[ii].clodMeshMetadataIndex];
            const brmi_lod_segment& seg = sc.lodSegments[md.segmentsBase + segI];
            const brmi_group_page_map_entry& pe = sc.groupPageMap[md.pageMapBase + seg.pageIndex];
            buckets.push_back({ii, grp, segI, lm, 1, pe.slabDescriptorIndex, pe.slabByteOffset, seg.firstMeshletInPage, true});
        }
    }

    // K2: level-synchronous BFS
    uint32_t levels = 0;
    while (!frontier.empty() && levels < 64) {
        next.clear();
        for (const NodeRec& rec : frontier) {
            cnt.nodesVisited++;
            const brmi_per_mesh_instance& inst = sc.perMeshInstance[rec.inst];
            const brmi_clod_mesh_metadata& md = sc.meshMetadata[sc.clodOffsets[rec.inst].clodMeshMetadataIndex];
            const brmi_per_mesh& pm = sc.perMesh[inst.perMeshBufferIndex];
            const bool skinned = (pm.vertexFlags & BRMI_VERTEX_SKINNED) != 0;
            const brmi_per_object& obj = sc.perObject[inst.perObjectBufferIndex];
            const mat4& model = M(obj.model);
            const brmi_lod_node& node = sc.lodNodes[md.lodNodesBase + rec.node];
            const float scale = maxAxisScale(model);
            float3 cullC = skinned ? float3{inst.boundingSphere[0], inst.boundingSphere[1], inst.boundingSphere[2]} : float3{node.cullCenterAndRadius[0], node.cullCenterAndRadius[1], node.cullCenterAndRadius[2]};
            float cullR = skinned ? inst.boundingSphere[3] : node.cullCenterAndRadius[3];
            float3 cVS = toViewSpace(cullC, model, view);
            float rW = cullR * scale;
            if (!rec.replay && sphereOutsideFrustum(cVS, rW, cam.clippingPlanes)) continue;
            const float3 camPos{lodCam.positionWorldSpace[0], lodCam.positionWorldSpace[1], lodCam.positionWorldSpace[2]};
            if (node.isLeaf != BRMI_NODE_INTERNAL) {
                const brmi_lod_group& g = sc.lodGroups[md.groupsBase + node.ownerGroupId];
                float3 gc = xyz(mulPoint(float3{g.centerAndRadius[0], g.centerAndRadius[1], g.centerAndRadius[2]}, model));
                float gr = g.centerAndRadius[3] * scale;
                float eod = projectedGeometricError(gc, gr, node.maxQuadricError, scale, camPos, lodCam.zNear, ortho);
                if (!(rec.allowRefine && eod >= lodCam.errorOverDistanceThreshold)) continue;
                if (refinedChildSuppressesParent(sc, md.groupsBase, node.countMinusOne - 1u, node.countMinusOne != 0u, model, scale, lodCam, ortho)) continue;
                const brmi_lod_segment& seg = segOfLeaf(md, node);
                if (seg.meshletCount == 0) continue;
                const brmi_group_page_map_entry& pe = sc.groupPageMap[md.pageMapBase + seg.pageIndex];
                if (pe.slabDescriptorIndex == 0) continue;
                uint32_t base = seg.firstMeshletInPage, remaining = seg.meshletCount;
                while (remaining > 0) {
                    uint32_t chunk = remaining < factor ? remaining : factor;
                    buckets.push_back({rec.inst, node.ownerGroupId, node.indexOrOffset, base, chunk, pe.slabDescriptorIndex, pe.slabByteOffset, seg.firstMeshletInPage, rec.replay});
                    base += chunk; remaining -= chunk;
                }
                continue;
            }
            float3 lc = xyz(mulPoint(float3{node.lodCenterAndRadius[0], node.lodCenterAndRadius[1], node.lodCenterAndRadius[2]}, model));
            float lr = node.lodCenterAndRadius[3] * scale;
            float nodeEod = projectedGeometricError(lc, lr, node.maxQuadricError, scale, camPos, lodCam.zNear, ortho);
            if (!(rec.allowRefine && nodeEod >= lodCam.errorOverDistanceThreshold)) continue;
            if (occl) {
                bool oc;
                if (rec.replay) oc = occlusionCulled(hzb, cam, M(cam.projection), cVS, -cVS.z, rW);
                else {
                    const mat4& prevModel = M(obj.prevModel);
                    float3 pc = toViewSpace(cullC, prevModel, M(cam.prevView));
                    oc = occlusionCulled(hzb, cam, M(cam.prevUnjitteredProjection), pc, -pc.z, cullR * maxAxisScale(prevModel));
                }
                if (oc) {
                    if (!rec.replay && prm->replayNodes && *prm->replayNodeCount < prm->replayNodeCapacity) {
                        prm->replayNodes[2 * *prm->replayNodeCount] = rec.inst; prm->replayNodes[2 * *prm->replayNodeCount + 1] = rec.node; (*prm->replayNodeCount)++;
                    }
                    continue;
                }
            }
            const uint32_t childCount = (node.countMinusOne + 1u) < BRMI_BVH_MAX_CHILDREN ? (node.countMinusOne + 1u) : BRMI_BVH_MAX_CHILDREN;
            for (uint32_t k = 0; k < childCount; k++) {
                const uint32_t childId = node.indexOrOffset + k;
                const brmi_lod_node& ch = sc.lodNodes[md.lodNodesBase + childId];
                float3 cc = skinned ? cullC : float3{ch.cullCenterAndRadius[0], ch.cullCenterAndRadius[1], ch.cullCenterAndRadius[2]};
                float cr = skinned ? cullR : ch.cullCenterAndRadius[3];
                float3 ccVS = toViewSpace(cc, model, view);
                if (!rec.replay && sphereOutsideFrustum(ccVS, cr * scale, cam.clippingPlanes)) continue;
                if (ch.isLeaf == BRMI_NODE_INTERNAL) {
                    float3 wc = xyz(mulPoint(float3{ch.lodCenterAndRadius[0], ch.lodCenterAndRadius[1], ch.lodCenterAndRadius[2]}, model));
                    float e = projectedGeometricError(wc, ch.lodCenterAndRadius[3] * scale, ch.maxQuadricError, scale, camPos, lodCam.zNear, ortho);
                    if (e < lodCam.errorOverDistanceThreshold) continue;
                }
                next.push_back({rec.inst, childId, true, rec.replay});
            }
        }
        frontier.swap(next);
        levels++;
    }

    // K3: per-meshlet cull
    cnt.bucketRecords = (uint32_t)buckets.size();
    for (const Bucket& b : buckets) {
        if (b.slab == 0) continue;
        const brmi_per_mesh_instance& inst = sc.perMeshInstance[b.inst];
        const brmi_clod_mesh_metadata& md = sc.meshMetadata[sc.clodOffsets[b.inst].clodMeshMetadataIndex];
        const brmi_per_object& obj = sc.perObject[inst.perObjectBufferIndex];
        const mat4& model = M(obj.model);
        const float scale = maxAxisScale(model);
        const uint8_t* slab = sc.slabs[b.slab];
        const brmi_page_header* hdr = pageHeader(slab, b.pageOff);
        const uint32_t groupId = b.group;
        for (uint32_t m = 0; m < b.count; m++) {
            const uint32_t lm = b.first + m;
            cnt.meshletsTested++;
            if (lm >= hdr->meshletCount) continue;
            const brmi_meshlet_descriptor& desc = *meshletDesc(slab, b.pageOff, hdr->descriptorOffset, lm);
            float3 bc{desc.bounds[0], desc.bounds[1], desc.bounds[2]};
            float br = desc.bounds[3];
            if (sc.perMesh[inst.perMeshBufferIndex].vertexFlags & BRMI_VERTEX_SKINNED) skinnedMeshletBounds(sc, desc, *hdr, slab, b.pageOff, inst.skinningInstanceSlot, bc, br);
            float3 cVS = toViewSpace(bc, model, view);
            float rW = br * scale;
            bool survives = b.replay || !sphereOutsideFrustum(cVS, rW, cam.clippingPlanes);
            if (survives) {
                const int32_t refined = descRefinedGroup(desc);
                if (refinedChildSuppressesParent(sc, md.groupsBase, (uint32_t)refined, refined >= 0, model, scale, lodCam, ortho)) survives = false;
            }
            if (survives && occl) {
                bool oc;
                if (b.replay) oc = occlusionCulled(hzb, cam, M(cam.projection), cVS, -cVS.z, rW);
                else {
                    const mat4& prevModel = M(obj.prevModel);
                    float3 pc = toViewSpace(bc, prevModel, M(cam.prevView));
                    oc = occlusionCulled(hzb, cam, M(cam.prevUnjitteredProjection), pc, -pc.z, br * maxAxisScale(prevModel));
                }
                if (oc) {
                    if (!b.replay && prm->replayMeshlets && *prm->replayMeshletCount < prm->replayMeshletCapacity) {
                        uint32_t k = *prm->replayMeshletCount;
                        prm->replayMeshlets[4 * k] = b.inst; prm->replayMeshlets[4 * k + 1] = b.seg; prm->replayMeshlets[4 * k + 2] = lm; prm->replayMeshlets[4 * k + 3] = b.group; (*prm->replayMeshletCount)++;
                    }
                    survives = false;
                }
            }
            if (!survives) continue;
            visible.push_back({b.inst, b.seg, lm - b.segFirst, packVisibleCluster(viewId, b.inst, lm, groupId, b.slab, b.pageOff)});
        }
    }
    std::sort(visible.begin(), visible.end(), [](const VisKey& a, const VisKey& b) {
        if (a.inst != b.inst) return a.inst < b.inst;
        if (a.seg != b.seg) return a.seg < b.seg;
        return a.off < b.off;
    });
    uint32_t n = (uint32_t)visible.size();
    if (n > prm->capacity) { cnt.droppedClusters = n - prm->capacity; n = prm->capacity; }
    for (uint32_t i = 0; i < n; i++) out[i] = visible[i].packed;
    *outCount = n;
    if (prm->phase == 1) cnt.visibleClusters = n; else cnt.visibleClustersPhase2 = n;
    if (counters) *counters = cnt;
    return 0;
}

}  // extern "C"
