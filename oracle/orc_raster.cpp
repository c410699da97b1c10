// orc_raster.cpp -- CPU restatement of the compute software rasteriser (K5) and the depth copy (K6).
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows:
//   SWRasterCluster                 BR/shaders/ClusterLOD/softwareRaster.hlsl:290-612
//   SWDecodeTriangle                BR/shaders/ClusterLOD/softwareRaster.hlsl:60-89
//   SWRasterClipScanlineConstraint  BR/shaders/ClusterLOD/softwareRaster.hlsl:262-288
//   PackVisKey / UnpackVisKey       BR/shaders/Include/visibilityPacking.hlsli:11-37
//   PerViewPrimaryDepthCopyCS       BR/shaders/gbuffer.hlsl:114-161
//   SWAlphaTestFailed, per-pixel perspective-correct texcoord   BR/shaders/ClusterLOD/softwareRaster.hlsl:135-172,525-540,574-589
//     (CLOD_SW_RASTER_DYNAMIC_ALPHA_TEST: the test runs for materials flagged MATERIAL_ALPHA_TEST; rcp = 1/x, orc_common.h)
//
// Wave semantics: the reference picks its scan strategy with WaveActiveAnyTrue(rectWidth > 4)
// (softwareRaster.hlsl:502), so the result depends on which triangles share a wave.  CDNA is
// wave64-only: thread GI = triangle t of the 128-thread group, lanes [0,64) and [64,128) form
// the two waves, and only lanes that reached the vote (did not `continue`) take part in it.
// Every cluster is rasterised by these compute rules (no hardware path).
#include <climits>
#include <vector>

#include "orc_common.h"
#include "orc_texture.h"

namespace orc {

static inline int toInt(float f) {   // float -> int, saturating (GPU conversion semantics)
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return INT_MAX;
    if (f <= -2147483648.0f) return INT_MIN;
    return (int)f;
}

static inline void atomicMinU64(uint64_t* p, uint64_t v) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

static inline uint64_t packVisKey(float depth, uint32_t cluster, uint32_t tri) {
    uint64_t depthBits = asuint(depth) >> 1;
    return (depthBits << BRMI_VIS_META_BITS) | ((uint64_t)(cluster & 0x3FFFFFFu) << BRMI_VIS_TRI_BITS) | (uint64_t)(tri & 0x7Fu);
}

static void clipScanline(float value, float step, int& first, int& last, bool& has) {
    if (!has) return;
    if (step > 0.0f) { int c = toInt(std::ceil(-value / step)); first = first > c ? first : c; }
    else if (step < 0.0f) { int f = toInt(std::floor(value / -step)); last = last < f ? last : f; }
    else has = value >= 0.0f;
    has = has && first <= last;
}

struct TriLane {
    bool active;
    float d0, d1, d2;
    int minX, minY, maxX, maxY, rectWidth;
    float row_b0, row_b1, dx_b0, dx_b1, dy_b0, dy_b1, dx_b2;
    float invW0, invW1, invW2; float2 uv0, uv1, uv2;
};

// voteMode: how WaveActiveAnyTrue(rectWidth > 4) (softwareRaster.hlsl:502) is evaluated.  0 = over the 64 triangles of a wave64 (the
// restatement's choice); 1 = over 32-triangle groups, as wave32 hardware would (NVIDIA, RDNA); 2 = always true;
// 3 = always false.  Modes 1-3 exist to MEASURE how much of the image depends on that choice (tests/test_oracle_cpu.py).
void rasterCluster(const brmi_scene_buffers& sc, const brmi_visible_cluster& pc, uint32_t clusterIndex, uint64_t* vis, uint32_t visW, uint32_t visH,
                   uint32_t bandY0, uint32_t bandY1, int voteMode = 0) {
    const uint32_t viewID = vcViewID(pc), instanceID = vcInstanceID(pc), localMeshlet = vcLocalMeshlet(pc);
    const uint8_t* slab = sc.slabs[vcSlabDescriptor(pc)];
    const uint32_t pageOff = vcPageByteOffset(pc);
    const brmi_page_header& hdr = *pageHeader(slab, pageOff);
    const brmi_meshlet_descriptor& desc = *meshletDesc(slab, pageOff, hdr.descriptorOffset, localMeshlet);
    const uint32_t vertCount = descVertexCount(desc), triCount = descTriangleCount(desc);
    const brmi_per_mesh_instance& meshInst = sc.perMeshInstance[instanceID];
    const brmi_per_mesh& mesh = sc.perMesh[meshInst.perMeshBufferIndex];
    const brmi_per_object& obj = sc.perObject[meshInst.perObjectBufferIndex];
    const brmi_culling_camera& cam = sc.cullingCameras[viewID];
    const brmi_view_raster_info& ri = sc.viewRasterInfo[viewID];

    const float visWidth = (float)(ri.scissorMaxX - ri.scissorMinX), visHeight = (float)(ri.scissorMaxY - ri.scissorMinY);
    const float sMinXf = (float)ri.scissorMinX, sMinYf = (float)ri.scissorMinY;
    const uint32_t posBase = pageOff + hdr.positionBitstreamOffset;
    const mat4 mvp = mul(M(obj.model), M(cam.viewProjection));
    const float4 modelViewZ = mulCol(M(obj.model), float4{cam.viewZ[0], cam.viewZ[1], cam.viewZ[2], cam.viewZ[3]});
    const bool skinned = (mesh.vertexFlags & BRMI_VERTEX_SKINNED) != 0;

    float2 gsScreen[BRMI_MESHLET_MAX_VERTS]; float gsDepth[BRMI_MESHLET_MAX_VERTS];
    float gsInvW[BRMI_MESHLET_MAX_VERTS]; float2 gsTexcoord[BRMI_MESHLET_MAX_VERTS];
    const uint32_t materialDataIndex = mesh.materialDataIndex;
    const bool alphaTest = (sc.materials[materialDataIndex].materialFlags & BRMI_MATERIAL_ALPHA_TEST) != 0u;
    // softwareRaster.hlsl:525-540: texcoord at the pixel from the stepped barycentrics, then SWAlphaTestFailed
    auto pixelFails = [&](const TriLane& L, float b0, float b1, float b2) {
        if (!alphaTest) return false;
        const float pc0 = b0 * L.invW0, pc1 = b1 * L.invW1, pc2 = b2 * L.invW2;
        const float invSum = rcp(pc0 + pc1 + pc2);
        const float2 texcoord = (L.uv0 * pc0 + L.uv1 * pc1 + L.uv2 * pc2) * invSum;
        return alphaTestFailed(sc, texcoord, materialDataIndex);
    };
    for (uint32_t v = 0; v < vertCount && v < BRMI_MESHLET_MAX_VERTS; v++) {
        float3 lp = loadPosition(slab, hdr.compressedPositionQuantExp, posBase, desc.positionBitOffset, v);
        if (skinned && (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_JOINTS)) {
            uint32_t joints[8]; float weights[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            std::memcpy(joints, slab + pageOff + hdr.jointArrayOffset + (desc.vertexAttributeOffset + v) * 32u, 32);
            if (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_WEIGHTS) std::memcpy(weights, slab + pageOff + hdr.weightArrayOffset + (desc.vertexAttributeOffset + v) * 32u, 32);
            mat4 skin = buildSkinMatrix(sc, meshInst.skinningInstanceSlot, joints, weights);
            lp = xyz(mulPoint(lp, skin));
        }
        float4 lp4{lp.x, lp.y, lp.z, 1.0f};
        float4 clip = mul(lp4, mvp);
        float viewZ = dot(lp4, modelViewZ);
        float invW = 1.0f / clip.w;
        float ndcx = clip.x * invW, ndcy = clip.y * invW;
        gsScreen[v].x = (ndcx + 1.0f) * 0.5f * visWidth + sMinXf;
        gsScreen[v].y = (1.0f - ndcy) * 0.5f * visHeight + sMinYf;
        gsDepth[v] = -viewZ;
        gsInvW[v] = invW;
        gsTexcoord[v] = decodeCompressedUV(slab, pageOff, hdr, localMeshlet, 0u, v);
    }
    const bool reverseWinding = (obj.objectFlags & BRMI_OBJECT_FLAG_REVERSE_WINDING) != 0;
    const uint32_t triBase = pageOff + hdr.triangleStreamOffset;

    for (uint32_t waveBase = 0; waveBase < triCount; waveBase += 64) {
        TriLane lanes[64];
        bool any = false;
        for (uint32_t l = 0; l < 64; l++) {
            TriLane& L = lanes[l]; L.active = false;
            const uint32_t t = waveBase + l;
            if (t >= triCount) continue;
            uint32_t tri[3]; decodeTriangle(slab, triBase, desc.triangleByteOffset, t, tri);
            if (reverseWinding) { uint32_t tmp = tri[1]; tri[1] = tri[2]; tri[2] = tmp; }
            const float2 s0 = gsScreen[tri[0]], s1 = gsScreen[tri[1]], s2 = gsScreen[tri[2]];
            const float d0 = gsDepth[tri[0]], d1 = gsDepth[tri[1]], d2 = gsDepth[tri[2]];
            if (d0 <= 0.0f || d1 <= 0.0f || d2 <= 0.0f) continue;
            const float2 e01 = s1 - s0, e02 = s2 - s0;
            const float twiceArea = e01.x * e02.y - e01.y * e02.x;
            if (twiceArea >= 0.0f) continue;
            const float invTwiceArea = -1.0f / twiceArea;
            const float bbMinX = fmin2(fmin2(s0.x, s1.x), s2.x), bbMinY = fmin2(fmin2(s0.y, s1.y), s2.y);
            const float bbMaxX = fmax2(fmax2(s0.x, s1.x), s2.x), bbMaxY = fmax2(fmax2(s0.y, s1.y), s2.y);
            int minX = toInt(std::floor(bbMinX)), minY = toInt(std::floor(bbMinY)), maxX = toInt(std::floor(bbMaxX)), maxY = toInt(std::floor(bbMaxY));
            auto imax = [](int a, int b) { return a > b ? a : b; }; auto imin = [](int a, int b) { return a < b ? a : b; };
            minX = imax(minX, (int)ri.scissorMinX); minY = imax(minY, (int)ri.scissorMinY);
            maxX = imin(maxX, (int)ri.scissorMaxX - 1); maxY = imin(maxY, (int)ri.scissorMaxY - 1);
            minX = imax(minX, 0); minY = imax(minY, 0);
            maxX = imin(maxX, (int)visW - 1); maxY = imin(maxY, (int)visH - 1);
            if (minX > maxX || minY > maxY) continue;
            const float ox = (float)minX + 0.5f, oy = (float)minY + 0.5f;
            const float2 e12 = s2 - s1, e20 = s0 - s2;
            L.row_b0 = ((ox - s1.x) * e12.y - (oy - s1.y) * e12.x) * invTwiceArea;
            L.row_b1 = ((ox - s2.x) * e20.y - (oy - s2.y) * e20.x) * invTwiceArea;
            L.dx_b0 = e12.y * invTwiceArea; L.dx_b1 = e20.y * invTwiceArea;
            L.dy_b0 = -e12.x * invTwiceArea; L.dy_b1 = -e20.x * invTwiceArea;
            L.dx_b2 = -(L.dx_b0 + L.dx_b1);
            L.d0 = d0; L.d1 = d1; L.d2 = d2;
            L.invW0 = gsInvW[tri[0]]; L.invW1 = gsInvW[tri[1]]; L.invW2 = gsInvW[tri[2]];
            L.uv0 = gsTexcoord[tri[0]]; L.uv1 = gsTexcoord[tri[1]]; L.uv2 = gsTexcoord[tri[2]];
            L.minX = minX; L.minY = minY; L.maxX = maxX; L.maxY = maxY; L.rectWidth = maxX - minX + 1;
            L.active = true;
            any = any || (L.rectWidth > 4);
        }
        bool anyHalf[2] = {false, false};
        for (uint32_t l = 0; l < 64; l++) if (lanes[l].active && lanes[l].rectWidth > 4) anyHalf[l >> 5] = true;
        for (uint32_t l = 0; l < 64; l++) {
            const TriLane& L = lanes[l];
            if (!L.active) continue;
            // WaveActiveAnyTrue over the lanes still active
            const bool useScanlineRanges = voteMode == 0 ? any : voteMode == 1 ? anyHalf[l >> 5] : voteMode == 2;
            const uint32_t t = waveBase + l;
            float sb0 = L.row_b0, sb1 = L.row_b1;
            for (int py = L.minY; py <= L.maxY; py++) {
                const bool rowInBand = (uint32_t)py >= bandY0 && (uint32_t)py < bandY1;
                if (useScanlineRanges) {
                    const float sb2 = 1.0f - sb0 - sb1;
                    int first = 0, last = L.rectWidth - 1; bool has = true;
                    clipScanline(sb0, L.dx_b0, first, last, has);
                    clipScanline(sb1, L.dx_b1, first, last, has);
                    clipScanline(sb2, L.dx_b2, first, last, has);
                    if (has && rowInBand) {
                        float b0 = sb0 + (float)first * L.dx_b0, b1 = sb1 + (float)first * L.dx_b1;
                        for (int px = L.minX + first; px <= L.minX + last; px++) {
                            const float b2 = 1.0f - b0 - b1;
                            if (!pixelFails(L, b0, b1, b2)) {
                                const float depth = b0 * L.d0 + b1 * L.d1 + b2 * L.d2;
                                atomicMinU64(&vis[(uint64_t)py * visW + (uint32_t)px], packVisKey(depth, clusterIndex, t));
                            }
                            b0 += L.dx_b0; b1 += L.dx_b1;
                        }
                    }
                } else if (rowInBand) {
                    float b0 = sb0, b1 = sb1;
                    for (int px = L.minX; px <= L.maxX; px++) {
                        const float b2 = 1.0f - b0 - b1;
                        if (b0 >= 0.0f && b1 >= 0.0f && b2 >= 0.0f && !pixelFails(L, b0, b1, b2)) {
                            const float depth = b0 * L.d0 + b1 * L.d1 + b2 * L.d2;
                            atomicMinU64(&vis[(uint64_t)py * visW + (uint32_t)px], packVisKey(depth, clusterIndex, t));
                        }
                        b0 += L.dx_b0; b1 += L.dx_b1;
                    }
                }
                sb0 += L.dy_b0; sb1 += L.dy_b1;
            }
        }
    }
}

}  // namespace orc

using namespace orc;

extern "C" {

// Rasterise clusters[first .. first+count) into the linear u64 visibility image (W x H).
// `threads` > 1 runs clusters in parallel (OpenMP); the result is identical (min is commutative).
int orc_raster(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, uint32_t first, uint32_t count,
               uint64_t* vis, uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1, int threads) {
    if (bandY1 == 0) { bandY0 = 0; bandY1 = H; }
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < (int64_t)count; i++) rasterCluster(*sc, clusters[first + i], first + (uint32_t)i, vis, W, H, bandY0, bandY1);
    return 0;
}

// the same with the wave vote evaluated another way (rasterCluster: voteMode) -- measurement only
int orc_raster_vote(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, uint32_t first, uint32_t count,
                    uint64_t* vis, uint32_t W, uint32_t H, int voteMode, int threads) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < (int64_t)count; i++) rasterCluster(*sc, clusters[first + i], first + (uint32_t)i, vis, W, H, 0, H, voteMode);
    return 0;
}

// clusters[indices[i]] for i < n, each under its OWN index of the list (the key's cluster index): what a pass that draws part of the visible list must not have needed
int orc_raster_subset(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, const uint32_t* indices, uint32_t n, uint64_t* vis, uint32_t W, uint32_t H, int threads) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < (int64_t)n; i++) rasterCluster(*sc, clusters[indices[i]], indices[i], vis, W, H, 0, H);
    return 0;
}

int orc_clear_visibility(uint64_t* vis, uint64_t pixels) { for (uint64_t i = 0; i < pixels; i++) vis[i] = BRMI_VIS_EMPTY; return 0; }

// PerViewPrimaryDepthCopyCS: linear depth (0x7F7FFFFF where empty)
int orc_depth_copy(const uint64_t* vis, float* depth, uint64_t pixels, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < (int64_t)pixels; i++) {
        const uint64_t k = vis[i];
        depth[i] = (k == BRMI_VIS_EMPTY) ? asfloat(BRMI_DEPTH_EMPTY_BITS) : asfloat(((uint32_t)(k >> BRMI_VIS_META_BITS)) << 1);
    }
    return 0;
}

}  // extern "C"
