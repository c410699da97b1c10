// orc_cone.cpp -- what the rasteriser's triangle setup does to a visible cluster, as a verdict per cluster.
// TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// The product library keeps a per-meshlet normal cone (library-internal side table, not part of the reference's CLodMeshletDescriptor:
// BR/include/Mesh/ClusterLODShaderTypes.h:49-75 carries a bounding sphere only) and does not rasterise a visible cluster whose every triangle
// the reference's setup would cull as back-facing (softwareRaster.hlsl:456: `twiceArea >= 0` after the OBJECT_FLAG_REVERSE_WINDING swap of :431).
// The checker's side of that: for every visible cluster
//   bit 0  the fp32 restatement of the setup (orc_raster.cpp, softwareRaster.hlsl:416-470) leaves NO triangle active -- what a skipped cluster must satisfy;
//   bit 1  every triangle faces away from the eye in exact (float64) object-space arithmetic: the most ANY normal-based test could reject;
//   bit 2  a float64 normal cone made here (axis = mean unit normal, tightened; apices behind / in front of every triangle plane) rejects it with `margin`.
#include <algorithm>
#include <vector>

#include "orc_common.h"

namespace orc {

struct d3 { double x, y, z; };
static inline d3 sub(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline d3 crs(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline double dt(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// The homogeneous object-space point every ray of the projection passes through: the null vector of the (x, y, w) columns of the object-to-clip
// matrix (row-vector convention).  det[clip xyw of p0, p1, p2] = n . (E.xyz - p0 E.w), n = (p1 - p0) x (p2 - p0).
static void eyeOfMvp(const mat4& mvp, double E[4]) {
    double c[4][3];
    for (int r = 0; r < 4; r++) { c[r][0] = mvp.m[r][0]; c[r][1] = mvp.m[r][1]; c[r][2] = mvp.m[r][3]; }
    for (int i = 0; i < 4; i++) {
        int r[3], k = 0;
        for (int j = 0; j < 4; j++) if (j != i) r[k++] = j;
        const double det = c[r[0]][0] * (c[r[1]][1] * c[r[2]][2] - c[r[1]][2] * c[r[2]][1]) - c[r[0]][1] * (c[r[1]][0] * c[r[2]][2] - c[r[1]][2] * c[r[2]][0]) +
                           c[r[0]][2] * (c[r[1]][0] * c[r[2]][1] - c[r[1]][1] * c[r[2]][0]);
        E[i] = (i & 1) ? -det : det;
    }
}

}  // namespace orc

using namespace orc;

extern "C" {

int orc_cluster_facing(const brmi_scene_buffers* scp, const brmi_visible_cluster* clusters, uint32_t count, uint32_t visW, uint32_t visH, float margin, uint8_t* flags, int threads) {
    const brmi_scene_buffers& sc = *scp;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
    for (int64_t ci = 0; ci < (int64_t)count; ci++) {
        const brmi_visible_cluster& pc = clusters[ci];
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
        const bool skinned = (mesh.vertexFlags & BRMI_VERTEX_SKINNED) != 0 && (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_JOINTS);
        const bool reverseWinding = (obj.objectFlags & BRMI_OBJECT_FLAG_REVERSE_WINDING) != 0;
        float2 scr[BRMI_MESHLET_MAX_VERTS]; float dep[BRMI_MESHLET_MAX_VERTS]; d3 pos[BRMI_MESHLET_MAX_VERTS];
        for (uint32_t v = 0; v < vertCount && v < BRMI_MESHLET_MAX_VERTS; v++) {
            float3 lp = loadPosition(slab, hdr.compressedPositionQuantExp, posBase, desc.positionBitOffset, v);
            pos[v] = {lp.x, lp.y, lp.z};
            if (skinned) {
                uint32_t joints[8]; float weights[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                std::memcpy(joints, slab + pageOff + hdr.jointArrayOffset + (desc.vertexAttributeOffset + v) * 32u, 32);
                if (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_WEIGHTS) std::memcpy(weights, slab + pageOff + hdr.weightArrayOffset + (desc.vertexAttributeOffset + v) * 32u, 32);
                lp = xyz(mulPoint(lp, buildSkinMatrix(sc, meshInst.skinningInstanceSlot, joints, weights)));
            }
            const float4 lp4{lp.x, lp.y, lp.z, 1.0f};
            const float4 clip = mul(lp4, mvp);
            const float invW = 1.0f / clip.w;
            const float ndcx = clip.x * invW, ndcy = clip.y * invW;
            scr[v].x = (ndcx + 1.0f) * 0.5f * visWidth + sMinXf;
            scr[v].y = (1.0f - ndcy) * 0.5f * visHeight + sMinYf;
            dep[v] = -dot(lp4, modelViewZ);
        }
        double E[4]; eyeOfMvp(mvp, E);
        double sgn = reverseWinding ? -1.0 : 1.0;
        if (E[3] < 0.0) { for (double& e : E) e = -e; sgn = -sgn; }
        const uint32_t triBase = pageOff + hdr.triangleStreamOffset;
        bool noneActive = true, allBack = !skinned;
        d3 axisSum{0, 0, 0};
        std::vector<d3> nrm; nrm.reserve(triCount);
        std::vector<uint32_t> tv0; tv0.reserve(triCount);
        for (uint32_t t = 0; t < triCount; t++) {
            uint32_t tri[3]; decodeTriangle(slab, triBase, desc.triangleByteOffset, t, tri);
            {   // exact facing, stored winding
                const d3 n = crs(sub(pos[tri[1]], pos[tri[0]]), sub(pos[tri[2]], pos[tri[0]]));
                const double len = std::sqrt(dt(n, n));
                if (len > 0.0) { const d3 u{n.x / len, n.y / len, n.z / len}; nrm.push_back(u); tv0.push_back(tri[0]); axisSum = {axisSum.x + u.x, axisSum.y + u.y, axisSum.z + u.z}; }
                const d3 toEye{E[0] - pos[tri[0]].x * E[3], E[1] - pos[tri[0]].y * E[3], E[2] - pos[tri[0]].z * E[3]};
                if (sgn * dt(n, toEye) > 0.0) allBack = false;
            }
            if (reverseWinding) { const uint32_t tmp = tri[1]; tri[1] = tri[2]; tri[2] = tmp; }
            const float2 s0 = scr[tri[0]], s1 = scr[tri[1]], s2 = scr[tri[2]];
            if (dep[tri[0]] <= 0.0f || dep[tri[1]] <= 0.0f || dep[tri[2]] <= 0.0f) continue;
            const float2 e01 = s1 - s0, e02 = s2 - s0;
            const float twiceArea = e01.x * e02.y - e01.y * e02.x;
            if (twiceArea >= 0.0f) continue;
            const float bbMinX = fmin2(fmin2(s0.x, s1.x), s2.x), bbMinY = fmin2(fmin2(s0.y, s1.y), s2.y);
            const float bbMaxX = fmax2(fmax2(s0.x, s1.x), s2.x), bbMaxY = fmax2(fmax2(s0.y, s1.y), s2.y);
            auto toI = [](float f) { return f >= 2147483648.0f ? INT32_MAX : (f <= -2147483648.0f ? INT32_MIN : (int)f); };
            int minX = toI(std::floor(bbMinX)), minY = toI(std::floor(bbMinY)), maxX = toI(std::floor(bbMaxX)), maxY = toI(std::floor(bbMaxY));
            minX = std::max(std::max(minX, (int)ri.scissorMinX), 0); minY = std::max(std::max(minY, (int)ri.scissorMinY), 0);
            maxX = std::min(std::min(maxX, (int)ri.scissorMaxX - 1), (int)visW - 1); maxY = std::min(std::min(maxY, (int)ri.scissorMaxY - 1), (int)visH - 1);
            if (minX > maxX || minY > maxY) continue;
            noneActive = false;
        }
        // the float64 cone
        bool coneReject = false;
        if (!skinned && !nrm.empty()) {
            double len = std::sqrt(dt(axisSum, axisSum));
            if (len > 1e-12) {
                d3 A{axisSum.x / len, axisSum.y / len, axisSum.z / len};
                auto minDot = [&](const d3& a, size_t* arg) { double m = 2.0; for (size_t k = 0; k < nrm.size(); k++) { const double d = dt(nrm[k], a); if (d < m) { m = d; if (arg) *arg = k; } } return m; };
                // tighten: move the axis towards the worst normal while that raises the minimum
                double best = minDot(A, nullptr);
                for (int it = 0; it < 32; it++) {
                    size_t w = 0; minDot(A, &w);
                    const double step = 0.5 / (1.0 + it * 0.5);
                    d3 B{A.x + (nrm[w].x - A.x) * step * 0.5, A.y + (nrm[w].y - A.y) * step * 0.5, A.z + (nrm[w].z - A.z) * step * 0.5};
                    const double bl = std::sqrt(dt(B, B)); if (bl < 1e-12) break;
                    B = {B.x / bl, B.y / bl, B.z / bl};
                    const double mb = minDot(B, nullptr);
                    if (mb > best) { best = mb; A = B; }
                }
                if (best > 0.0) {
                    const double sinA = std::sqrt(std::max(0.0, 1.0 - best * best));
                    const d3 c{desc.bounds[0], desc.bounds[1], desc.bounds[2]};
                    double tBack = -1e300, tFront = -1e300;      // apices c - A tBack (behind every plane), c + A tFront (in front of every plane)
                    for (size_t k = 0; k < nrm.size(); k++) {
                        const double num = dt(nrm[k], sub(c, pos[tv0[k]])), den = dt(nrm[k], A);
                        tBack = std::max(tBack, num / den); tFront = std::max(tFront, -num / den);
                    }
                    const d3 X = sgn > 0.0 ? d3{c.x - A.x * tBack, c.y - A.y * tBack, c.z - A.z * tBack} : d3{c.x + A.x * tFront, c.y + A.y * tFront, c.z + A.z * tFront};
                    d3 v{X.x * E[3] - E[0], X.y * E[3] - E[1], X.z * E[3] - E[2]};      // eye -> apex
                    if (sgn < 0.0) v = {-v.x, -v.y, -v.z};
                    const double vl = std::sqrt(dt(v, v));
                    const d3 toC{c.x * E[3] - E[0], c.y * E[3] - E[1], c.z * E[3] - E[2]};
                    const bool eyeInside = dt(toC, toC) <= (double)desc.bounds[3] * desc.bounds[3] * E[3] * E[3] * 1.02;
                    coneReject = !eyeInside && vl > 0.0 && dt(v, A) >= (sinA + (double)margin) * vl;
                }
            }
        }
        flags[ci] = (noneActive ? 1u : 0u) | (allBack ? 2u : 0u) | (coneReject ? 4u : 0u);
    }
    return 0;
}

}  // extern "C"

// Experiment (profiles/r06_experiments.md): how many clusters of the visible list a finer, conservative occlusion test against a depth chain could keep
// away from the rasteriser.  Per cluster, flags: bit 0 = rejected with the cluster's ACTUAL screen box and nearest vertex depth (the bound of any bounding
// volume), bit 1 = with the object-space AABB of its vertices (8 corners projected), bit 2 = with the descriptor's sphere (box of the sphere's extents,
// depth - radius).  Every texel of the box is read at the finest mip where the box is at most maxTexels x maxTexels texels.
namespace orc {
static bool boxOccluded(const float* hzb, const uint64_t* mipOffsets, uint32_t mipCount, uint32_t W, uint32_t H, int x0, int y0, int x1, int y1, float nearDepth, int maxTexels) {
    if (x1 < 0 || y1 < 0 || x0 >= (int)W || y0 >= (int)H) return true;      // off screen: draws nothing
    x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, (int)W - 1); y1 = std::min(y1, (int)H - 1);
    uint32_t pw = 1, ph = 1; while (pw < W) pw <<= 1; while (ph < H) ph <<= 1;
    uint32_t mip = 0;
    while (mip + 1 < mipCount && (((x1 >> mip) - (x0 >> mip) + 1) > maxTexels || ((y1 >> mip) - (y0 >> mip) + 1) > maxTexels)) mip++;
    const uint32_t mw = std::max(1u, pw >> mip);
    const float* m = hzb + mipOffsets[mip];
    for (int y = y0 >> mip; y <= (y1 >> mip); y++) for (int x = x0 >> mip; x <= (x1 >> mip); x++) if (!(m[(uint64_t)y * mw + x] < nearDepth)) return false;
    return true;
}
}
extern "C" int orc_cluster_occlusion_stats(const brmi_scene_buffers* scp, const brmi_visible_cluster* clusters, uint32_t count, uint32_t W, uint32_t H,
                                           const float* hzb, const uint64_t* mipOffsets, uint32_t mipCount, int maxTexels, uint8_t* flags, int threads) {
    const brmi_scene_buffers& sc = *scp;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
    for (int64_t ci = 0; ci < (int64_t)count; ci++) {
        const brmi_visible_cluster& pc = clusters[ci];
        const uint32_t viewID = vcViewID(pc), instanceID = vcInstanceID(pc), localMeshlet = vcLocalMeshlet(pc);
        const uint8_t* slab = sc.slabs[vcSlabDescriptor(pc)];
        const uint32_t pageOff = vcPageByteOffset(pc);
        const brmi_page_header& hdr = *pageHeader(slab, pageOff);
        const brmi_meshlet_descriptor& desc = *meshletDesc(slab, pageOff, hdr.descriptorOffset, localMeshlet);
        const uint32_t vertCount = descVertexCount(desc);
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
        const bool skinned = (mesh.vertexFlags & BRMI_VERTEX_SKINNED) != 0 && (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_JOINTS);
        uint8_t f = 0;
        if (!skinned) {
            auto project = [&](float3 lp, float& sx, float& sy, float& d) {
                const float4 lp4{lp.x, lp.y, lp.z, 1.0f};
                const float4 clip = mul(lp4, mvp);
                const float invW = 1.0f / clip.w;
                sx = (clip.x * invW + 1.0f) * 0.5f * visWidth + sMinXf; sy = (1.0f - clip.y * invW) * 0.5f * visHeight + sMinYf; d = -dot(lp4, modelViewZ);
            };
            float bx0 = 1e30f, by0 = 1e30f, bx1 = -1e30f, by1 = -1e30f, dmin = 1e30f;
            float3 lo{1e30f, 1e30f, 1e30f}, hi{-1e30f, -1e30f, -1e30f};
            for (uint32_t v = 0; v < vertCount && v < BRMI_MESHLET_MAX_VERTS; v++) {
                const float3 lp = loadPosition(slab, hdr.compressedPositionQuantExp, posBase, desc.positionBitOffset, v);
                lo = fmin3v(lo, lp); hi = fmax3v(hi, lp);
                float sx, sy, d; project(lp, sx, sy, d);
                bx0 = fmin2(bx0, sx); by0 = fmin2(by0, sy); bx1 = fmax2(bx1, sx); by1 = fmax2(by1, sy); dmin = fmin2(dmin, d);
            }
            if (dmin > 0.0f && boxOccluded(hzb, mipOffsets, mipCount, W, H, (int)std::floor(bx0), (int)std::floor(by0), (int)std::floor(bx1), (int)std::floor(by1), dmin * (1.0f - 1e-5f), maxTexels)) f |= 1;
            {
                float ax0 = 1e30f, ay0 = 1e30f, ax1 = -1e30f, ay1 = -1e30f, admin = 1e30f;
                for (int k = 0; k < 8; k++) {
                    float sx, sy, d; project(float3{(k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z}, sx, sy, d);
                    ax0 = fmin2(ax0, sx); ay0 = fmin2(ay0, sy); ax1 = fmax2(ax1, sx); ay1 = fmax2(ay1, sy); admin = fmin2(admin, d);
                }
                if (admin > 0.0f && boxOccluded(hzb, mipOffsets, mipCount, W, H, (int)std::floor(ax0), (int)std::floor(ay0), (int)std::floor(ax1), (int)std::floor(ay1), admin * (1.0f - 1e-5f), maxTexels)) f |= 2;
            }
            {   // sphere: conservative box from the centre's projection and radius / depth
                const float3 c{desc.bounds[0], desc.bounds[1], desc.bounds[2]};
                const float r = desc.bounds[3] * maxAxisScale(M(obj.model));
                float sx, sy, d; project(c, sx, sy, d);
                if (d - r > 0.0f) {
                    // radius in pixels, bounded above: r * focal / (d - r)
                    const float fx = std::fabs(M(cam.viewProjection).m[0][0]) + std::fabs(M(cam.viewProjection).m[1][0]) + std::fabs(M(cam.viewProjection).m[2][0]);      // crude: |row sums| >= focal
                    (void)fx;
                    const float focalY = 0.5f * visHeight * cam.projY, focalX = 0.5f * visWidth * cam.projX;
                    const float rx = r * focalX / (d - r) * 1.05f, ry = r * focalY / (d - r) * 1.05f;
                    if (boxOccluded(hzb, mipOffsets, mipCount, W, H, (int)std::floor(sx - rx), (int)std::floor(sy - ry), (int)std::floor(sx + rx), (int)std::floor(sy + ry), (d - r) * (1.0f - 1e-5f), maxTexels)) f |= 4;
                }
            }
        }
        flags[ci] = f;
    }
    return 0;
}
