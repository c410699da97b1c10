// brmi_resolve.hip -- G-buffer reconstruction from the visibility buffer (K7 + K8) for gfx950.
//
// Computes what EvaluateGBufferOptimized -> ResolveClodCommonSampleFromVisKeyWithFace does
// (BR/shaders/gbuffer.hlsl:4-35, BR/shaders/Include/clodResolveCommon.hlsli:1414-1721): unpack the
// key, re-fetch the triangle's three vertices, re-project, analytic perspective-correct
// barycentrics at the pixel centre (CalcFullBary, clodResolveCommon.hlsli:104-143), interpolate
// position and normal, evaluate the constant-factor material, write the seven G-buffer surfaces.
// MI355X-first differences from the reference's schedule:
//   * The reference bins pixels per material permutation (histogram -> scan -> pixel list ->
//     per-material ExecuteIndirect, VisUtil.hlsl:37-243) only to run one specialised PSO per bin.
//     One kernel handles every constant-factor material, so the four extra full-screen passes over
//     the 8 B/px surface disappear; the linear-depth write of K6 is fused in too.
//   * one lane per pixel in tile order: a wave64 is exactly one 8x8 tile, so every surface read
//     or written by the wave is one contiguous 256 B..1 KB segment.
#include "brmi_device.h"
#include "brmi_internal.h"

namespace brmi {

struct GBufferArgs {
    brmi_scene_buffers sc;
    const uint4* clusters;
    const uint32_t* counters;
    const unsigned long long* vis;
    float4* normals; uint32_t* albedo; unsigned long long* coat; unsigned long long* emissive; unsigned long long* fuzz;
    uint32_t* metallicRoughness; uint32_t* motion; float* depth;
    uint32_t W, H, tilesX, bandY0, bandY1; uint64_t firstPixel, pixelCount;
    uint32_t clusterCapacity;
    const m4* frameConst; const float* objConst;
};

struct Bary { f3 lambda; };

// CalcFullBary: only lambda is consumed when no texture / normal map is bound
BRMI_DEV f3 calc_bary_lambda(f4 pt0, f4 pt1, f4 pt2, float ndcX, float ndcY) {
    const f3 invW{rcpf(pt0.w), rcpf(pt1.w), rcpf(pt2.w)};
    const float n0x = pt0.x * invW.x, n0y = pt0.y * invW.x, n1x = pt1.x * invW.y, n1y = pt1.y * invW.y, n2x = pt2.x * invW.z, n2y = pt2.y * invW.z;
    const float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
    const float invDet = rcpf(ax * by - ay * bx);
    const f3 ddx = f3{n1y - n2y, n2y - n0y, n0y - n1y} * invDet * invW;
    const f3 ddy = f3{n2x - n1x, n0x - n2x, n1x - n0x} * invDet * invW;
    const float ddxSum = dot3(ddx, f3{1.0f, 1.0f, 1.0f});
    const float ddySum = dot3(ddy, f3{1.0f, 1.0f, 1.0f});
    const float dx = ndcX - n0x, dy = ndcY - n0y;
    const float interpInvW = invW.x + dx * ddxSum + dy * ddySum;
    const float interpW = rcpf(interpInvW);
    f3 l;
    l.x = interpW * (invW.x + dx * ddx.x + dy * ddy.x);
    l.y = interpW * (0.0f + dx * ddx.y + dy * ddy.y);
    l.z = interpW * (0.0f + dx * ddx.z + dy * ddy.z);
    return l;
}

BRMI_DEV f3 oct_decode_normal(uint32_t packed) {
    const int sp = (int)packed;
    const int x = (int)((uint32_t)sp << 16) >> 16, y = sp >> 16;
    const float ex = max2(-1.0f, (float)x / 32767.0f), ey = max2(-1.0f, (float)y / 32767.0f);
    f3 v{ex, ey, 1.0f - fabsf(ex) - fabsf(ey)};
    if (v.z < 0.0f) {
        const float fx = (1.0f - fabsf(v.y)) * (v.x >= 0.0f ? 1.0f : -1.0f);
        const float fy = (1.0f - fabsf(v.x)) * (v.y >= 0.0f ? 1.0f : -1.0f);
        v.x = fx; v.y = fy;
    }
    return normalize3(v);
}

__global__ void __launch_bounds__(256) k_gbuffer(GBufferArgs a) {
    const brmi_scene_buffers& sc = a.sc;
    const brmi_per_frame* pf = sc.perFrame;
    const brmi_camera* cam = sc.cameras + pf->mainCameraIndex;
    const uint32_t clusterCount = min(a.counters[CNT_VISIBLE] + a.counters[CNT_VISIBLE2], a.clusterCapacity);
    // view-projection products are frame constants; every lane derives them the way the shader does
    const m4 unjVP = uni_m4(a.frameConst[1]), prevVP = uni_m4(a.frameConst[2]);
    (void)cam;
    const float winX = uni((float)pf->screenResX), winY = uni((float)pf->screenResY);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.pixelCount; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = a.firstPixel + j;
        const uint32_t tile = (uint32_t)(i >> 6), within = (uint32_t)(i & 63u);
        const uint32_t px = (tile % a.tilesX) * 8u + (within >> 3), py = (tile / a.tilesX) * 8u + (within & 7u);
        if (px >= a.W || py >= a.H || py < a.bandY0 || py >= a.bandY1) continue;
        const unsigned long long key = a.vis[i];
        if (a.depth) a.depth[i] = (key == BRMI_VIS_EMPTY) ? as_f32(BRMI_DEPTH_EMPTY_BITS) : as_f32(((uint32_t)(key >> BRMI_VIS_META_BITS)) << 1);
        if (key == BRMI_VIS_EMPTY) continue;
        const uint32_t triId = (uint32_t)(key & 0x7Full);
        const uint32_t clusterIndex = (uint32_t)((key >> BRMI_VIS_TRI_BITS) & 0x3FFFFFFull);
        if (clusterIndex >= clusterCount) continue;
        const uint4 pc = a.clusters[clusterIndex];
        const uint32_t instanceID = vc_instance(pc), localMeshlet = vc_meshlet(pc);
        const brmi_per_mesh_instance inst = sc.perMeshInstance[instanceID];
        const brmi_per_mesh* mesh = sc.perMesh + inst.perMeshBufferIndex;
        const uint8_t* slab = sc.slabs[vc_slab(pc)];
        const uint32_t pageOff = vc_page_offset(pc);
        const brmi_page_header* hdr = reinterpret_cast<const brmi_page_header*>(slab + pageOff);
        const brmi_meshlet_descriptor* desc = reinterpret_cast<const brmi_meshlet_descriptor*>(slab + pageOff + hdr->descriptorOffset + localMeshlet * 64u);
        if (triId >= (desc->triangleCountAndRefinedGroup & 0xFFFFu)) continue;
        const uint8_t* tb = slab + pageOff + hdr->triangleStreamOffset + desc->triangleByteOffset + triId * 3u;
        const uint32_t ti[3] = {tb[0], tb[1], tb[2]};
        const uint8_t* posBase = slab + pageOff + hdr->positionBitstreamOffset + desc->positionBitOffset;
        const uint8_t* nrmBase = slab + pageOff + hdr->normalArrayOffset + desc->vertexAttributeOffset * 4u;
        f3 p[3], n[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (hdr->compressedPositionQuantExp == BRMI_POSITION_FORMAT_FLOAT3) {
                const float* pp = reinterpret_cast<const float*>(posBase + ti[k] * 12u);
                p[k] = f3{pp[0], pp[1], pp[2]};
            } else p[k] = f3{0.0f, 0.0f, 0.0f};
            n[k] = oct_decode_normal(*reinterpret_cast<const uint32_t*>(nrmBase + ti[k] * 4u));
        }
        const brmi_per_object* obj = sc.perObject + inst.perObjectBufferIndex;
        const brmi_material_info* mat = sc.materials + mesh->materialDataIndex;
        const m4 model = load_m4(&obj->model[0][0]);
        const m4 objectToClip = load_m4(a.objConst + (size_t)inst.perObjectBufferIndex * 36u + 16u);
        const f4 clip0 = mul_point(p[0], objectToClip), clip1 = mul_point(p[1], objectToClip), clip2 = mul_point(p[2], objectToClip);
        const float uvx = ((float)px + 0.5f) / winX, uvy = ((float)py + 0.5f) / winY;
        const float ndcX = uvx * 2.0f - 1.0f, ndcY = (1.0f - uvy) * 2.0f - 1.0f;
        const f3 l = calc_bary_lambda(clip0, clip1, clip2, ndcX, ndcY);
        const f3 posOS{dot3(f3{p[0].x, p[1].x, p[2].x}, l), dot3(f3{p[0].y, p[1].y, p[2].y}, l), dot3(f3{p[0].z, p[1].z, p[2].z}, l)};
        const f3 worldPosition = xyz(mul_point(posOS, model));
        const f3 normalOS = normalize3(f3{dot3(f3{n[0].x, n[1].x, n[2].x}, l), dot3(f3{n[0].y, n[1].y, n[2].y}, l), dot3(f3{n[0].z, n[1].z, n[2].z}, l)});
        const m4 normalMatrix = load_m4(sc.normalMatrices + (size_t)obj->normalMatrixBufferIndex * 16u);
        const f3 worldNormal = normalize3(mul_v3m3(normalOS, normalMatrix));

        // constant-factor material (SampleMaterialEvalFromUvCache without texture permutations)
        const f3 baseColor = f3{mat->baseColorFactor[0], mat->baseColorFactor[1], mat->baseColorFactor[2]} * f3{1.0f, 1.0f, 1.0f};
        const float metallic = mat->metallicFactor, roughness = mat->roughnessFactor, ao = 1.0f;
        const f3 emissiveIn{mat->emissiveFactor[0], mat->emissiveFactor[1], mat->emissiveFactor[2]};
        const uint32_t opIndex = mat->openPBRMaterialDataIndex;
        const brmi_openpbr_material_info* op = sc.openpbrMaterials + opIndex;
        const f3 canonicalEmissive = f3{op->emissionColor[0], op->emissionColor[1], op->emissionColor[2]} * op->emissionLuminance;
        const f3 coatColor = sat3(f3{op->coatColor[0], op->coatColor[1], op->coatColor[2]});
        const float coatWeight = sat(op->coatWeight), coatRoughness = sat(op->coatRoughness);
        const f3 fuzzColor = sat3(f3{op->fuzzColor[0], op->fuzzColor[1], op->fuzzColor[2]});
        const float fuzzWeight = sat(op->fuzzWeight), fuzzRoughness = sat(op->fuzzRoughness);
        const f3 emissive = dot3(emissiveIn, emissiveIn) > 0.0f ? emissiveIn : canonicalEmissive;

        // ComputeClodMotionVector
        const f4 clipCur = mul_point(worldPosition, unjVP);
        const f3 prevWorld = xyz(mul_point(posOS, load_m4(&obj->prevModel[0][0])));
        const f4 clipPrev = mul_point(prevWorld, prevVP);
        const float mvx = clipCur.x / clipCur.w - clipPrev.x / clipPrev.w, mvy = clipCur.y / clipCur.w - clipPrev.y / clipPrev.w;

        a.normals[i] = make_float4(worldNormal.x, worldNormal.y, worldNormal.z, (float)opIndex);
        a.albedo[i] = pack_unorm4(baseColor.x, baseColor.y, baseColor.z, ao);
        a.coat[i] = pack_half4(coatColor.x, coatColor.y, coatColor.z, coatWeight);
        a.emissive[i] = pack_half4(emissive.x, emissive.y, emissive.z, 0.0f);
        a.fuzz[i] = pack_half4(fuzzColor.x, fuzzColor.y, fuzzColor.z, fuzzRoughness);
        a.metallicRoughness[i] = pack_unorm4(metallic, roughness, coatRoughness, fuzzWeight);
        a.motion[i] = f32_to_f16_bits(mvx) | (f32_to_f16_bits(mvy) << 16);
    }
}

int launch_gbuffer(brmi_pass* p, hipStream_t s) {
    GBufferArgs a;
    a.sc = p->scene; a.clusters = static_cast<const uint4*>(p->res[BRMI_RES_VISIBLE_CLUSTERS]); a.counters = p->counters();
    a.vis = static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]);
    a.normals = static_cast<float4*>(p->res[BRMI_RES_GBUF_NORMALS]); a.albedo = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_ALBEDO]);
    a.coat = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_COAT]); a.emissive = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_EMISSIVE]);
    a.fuzz = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_FUZZ]); a.metallicRoughness = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_METALLIC_ROUGHNESS]);
    a.motion = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_MOTION_VECTORS]); a.depth = static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]);
    a.W = p->cfg.width; a.H = p->cfg.height; a.tilesX = p->tilesX; a.bandY0 = p->bandY0; a.bandY1 = p->bandY1; a.firstPixel = p->bandFirstPixel; a.pixelCount = p->bandPixelCount;
    a.clusterCapacity = p->cfg.maxVisibleClusters;
    a.frameConst = p->wsPtr<m4>(p->ws.frameConst); a.objConst = p->wsPtr<float>(p->ws.objConst);
    hipLaunchKernelGGL(k_gbuffer, dim3(4096), dim3(256), 0, s, a);
    BRMI_LAUNCH_CHECK(p, "k_gbuffer");
    return BRMI_OK;
}

}  // namespace brmi
