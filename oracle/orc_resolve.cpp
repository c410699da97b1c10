// orc_resolve.cpp -- CPU restatement of the G-buffer reconstruction from the visibility buffer (K8).
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows:
//   EvaluateGBufferOptimized                    BR/shaders/gbuffer.hlsl:4-35
//   ResolveClodCommonSampleFromVisKeyWithFace   BR/shaders/Include/clodResolveCommon.hlsli:1414-1721
//   LoadMeshletResolveData_Wave                 BR/shaders/Include/clodResolveCommon.hlsli:716-809
//   DecodeTriangleCompact                       BR/shaders/Include/clodResolveCommon.hlsli:1347-1378
//   CalcFullBary / InterpolateWithDeriv         BR/shaders/Include/clodResolveCommon.hlsli:104-161
//   UnpackSnorm16x2 / OctDecodeNormal           BR/shaders/Include/clodResolveCommon.hlsli:627-655
//   ComputeClodMotionVector                     BR/shaders/Include/clodResolveCommon.hlsli:1380-1389
//   SampleMaterialEvalFromUvCache               BR/shaders/Include/utilities.hlsli:1850-2075
//   BuildClodMaterialUvData / AppendClodMaterialUvSample   BR/shaders/Include/clodResolveCommon.hlsli:271-430
//   cotangent_frame_from_derivs / BuildMaterialTBN          BR/shaders/Include/utilities.hlsli:323-336,1278-1287
//   ResolveCanonicalOpenPBRSurface              BR/shaders/Include/utilities.hlsli:136-161
//   getContactRefinementParallaxCoordsAndHeight BR/shaders/Include/parallax.hlsli:39-120 (call: utilities.hlsli:1869-1897)
// Scope: triangle clusters (no Reyes / voxel).  Vertex colours (CLOD_PAGE_ATTRIBUTE_COLOR) tint the base colour.  Material texture slots: base colour, opacity, metallic,
// roughness, normal map, AO, emissive and the six OpenPBR coat / fuzz slots, each through the software sampler of orc_texture.h;
// contact-refinement parallax (MATERIAL_PARALLAX) moves the texcoord of every slot that shares the height map's UV set; no texture
// streaming feedback.  A slot's UV set index below MATERIAL_MAX_UNIQUE_UV_SETS (8) is decoded
// as that set (a set the page does not carry decodes to (0, 0)); any other index uses set 0.
// G-buffer formats: BR/include/Render/RenderGraphBuildHelper.h:41-139, BR/src/Renderer.cpp:1618.
#include "orc_common.h"
#include "orc_texture.h"

namespace orc {

struct Bary { float3 lambda, ddx, ddy; };

static Bary calcFullBary(float4 pt0, float4 pt1, float4 pt2, float2 pixelNdc, float2 winSize) {
    Bary r;
    const float3 invW{rcp(pt0.w), rcp(pt1.w), rcp(pt2.w)};
    const float2 ndc0{pt0.x * invW.x, pt0.y * invW.x}, ndc1{pt1.x * invW.y, pt1.y * invW.y}, ndc2{pt2.x * invW.z, pt2.y * invW.z};
    const float2 a = ndc2 - ndc1, b = ndc0 - ndc1;
    const float invDet = rcp(a.x * b.y - a.y * b.x);
    r.ddx = float3{ndc1.y - ndc2.y, ndc2.y - ndc0.y, ndc0.y - ndc1.y} * invDet * invW;
    r.ddy = float3{ndc2.x - ndc1.x, ndc0.x - ndc2.x, ndc1.x - ndc0.x} * invDet * invW;
    float ddxSum = dot(r.ddx, float3{1, 1, 1});
    float ddySum = dot(r.ddy, float3{1, 1, 1});
    const float2 delta = pixelNdc - ndc0;
    const float interpInvW = invW.x + delta.x * ddxSum + delta.y * ddySum;
    const float interpW = rcp(interpInvW);
    r.lambda.x = interpW * (invW.x + delta.x * r.ddx.x + delta.y * r.ddy.x);
    r.lambda.y = interpW * (0.0f + delta.x * r.ddx.y + delta.y * r.ddy.y);
    r.lambda.z = interpW * (0.0f + delta.x * r.ddx.z + delta.y * r.ddy.z);
    r.ddx = r.ddx * (2.0f / winSize.x);
    r.ddy = r.ddy * (2.0f / winSize.y);
    ddxSum *= (2.0f / winSize.x);
    ddySum *= (2.0f / winSize.y);
    r.ddy = r.ddy * -1.0f;
    ddySum *= -1.0f;
    const float interpW_ddx = 1.0f / (interpInvW + ddxSum);
    const float interpW_ddy = 1.0f / (interpInvW + ddySum);
    r.ddx = interpW_ddx * (r.lambda * interpInvW + r.ddx) - r.lambda;
    r.ddy = interpW_ddy * (r.lambda * interpInvW + r.ddy) - r.lambda;
    return r;
}
static inline float interp(const Bary& b, float v0, float v1, float v2) { return dot(float3{v0, v1, v2}, b.lambda); }
// InterpolateWithDeriv: value, d/dx, d/dy
static inline float3 interpDeriv(const Bary& b, float v0, float v1, float v2) { const float3 m{v0, v1, v2}; return {dot(m, b.lambda), dot(m, b.ddx), dot(m, b.ddy)}; }
struct UvSample { float2 uv, dUVdx, dUVdy; };
static inline float swizzle(float4 v, uint32_t idx) { return idx == 0 ? v.x : idx == 1 ? v.y : idx == 2 ? v.z : v.w; }

static float3 octDecodeNormal(uint32_t packed) {
    const int32_t sp = (int32_t)packed;
    const int32_t x = (int32_t)((uint32_t)sp << 16) >> 16, y = sp >> 16;
    const float ex = fmax2(-1.0f, (float)x / 32767.0f), ey = fmax2(-1.0f, (float)y / 32767.0f);
    float3 v{ex, ey, 1.0f - std::fabs(ex) - std::fabs(ey)};
    if (v.z < 0.0f) {
        const float fx = (1.0f - std::fabs(v.y)) * (v.x >= 0.0f ? 1.0f : -1.0f);
        const float fy = (1.0f - std::fabs(v.x)) * (v.y >= 0.0f ? 1.0f : -1.0f);
        v.x = fx; v.y = fy;
    }
    return normalize(v);
}

// getContactRefinementParallaxCoordsAndHeight (parallax.hlsli:46-120): 16 coarse steps along the tangent-space view ray, on the first
// hit one refinement pass with the step divided by the steps left, then a secant between the last two points.  T, B, N are the rows
// of the cotangent frame.  The HLSL reads p1 / p2 / parallaxAmount uninitialised when the ray never dips below the height field or
// the secant is degenerate; here both start as zero (p1 = p2 = 0 gives parallaxAmount = 0), which is what DXC's undef lowers to.
static inline float wrap1(float x) { const float y = x + 1.0f; return y - std::floor(y); }      // WrapFloat2: frac(input + 1.0)
static float2 parallaxCoords(const brmi_scene_buffers& sc, uint32_t heightMapIndex, uint32_t heightSamplerIndex, float3 T, float3 B, float3 N,
                             float2 uv, float3 viewDirWS, float heightmapScale, float2 dUVdx, float2 dUVdy) {
    uv.y = 1.0f - uv.y;
    const float3 viewDir = normalize(float3{dot(T, viewDirWS), dot(B, viewDirWS), dot(N, viewDirWS)});     // mul(TBN, viewDir)
    const float maxHeight = heightmapScale, minHeight = maxHeight * 0.5f;
    int numSteps = 16;
    const float viewCorrection = (-viewDir.z) + 2.0f;
    float stepSize = 1.0f / ((float)numSteps + 1.0f);
    float2 stepOffset{viewDir.x * maxHeight * stepSize, viewDir.y * maxHeight * stepSize};
    float2 lastOffset{wrap1(viewDir.x * minHeight + uv.x), wrap1(viewDir.y * minHeight + uv.y)};
    float lastRayDepth = 1.0f, lastHeight = 1.0f;
    float2 p1{0.0f, 0.0f}, p2{0.0f, 0.0f};
    bool refine = false;
    while (numSteps > 0) {
        const float2 candidateOffset{wrap1(lastOffset.x - stepOffset.x), wrap1(lastOffset.y - stepOffset.y)};
        const float currentRayDepth = lastRayDepth - stepSize;
        const float currentHeight = viewCorrection * sampleGrad(sc, heightMapIndex, heightSamplerIndex, candidateOffset, dUVdx, dUVdy).x;      // Texture2D<float>
        if (currentHeight > currentRayDepth) {
            p1 = float2{currentRayDepth, currentHeight};
            p2 = float2{lastRayDepth, lastHeight};
            if (refine) { lastHeight = currentHeight; break; }
            refine = true;
            lastRayDepth = p2.x;
            stepSize /= (float)numSteps;
            stepOffset.x /= (float)numSteps; stepOffset.y /= (float)numSteps;
            continue;
        }
        lastOffset = candidateOffset;
        lastRayDepth = currentRayDepth;
        lastHeight = currentHeight;
        numSteps -= 1;
    }
    const float diff1 = p1.x - p1.y, diff2 = p2.x - p2.y;
    const float denominator = diff2 - diff1;
    float parallaxAmount = 0.0f;
    if (denominator != 0.0f) parallaxAmount = (p1.x * diff2 - p2.x * diff1) / denominator;
    const float offset = ((1.0f - parallaxAmount) * -maxHeight) + minHeight;
    return float2{viewDir.x * offset + uv.x, viewDir.y * offset + uv.y};
}

struct GBufferOut {
    float* normals; uint32_t* albedo; uint64_t* coat; uint64_t* emissive; uint64_t* fuzz; uint32_t* metallicRoughness; uint32_t* motion;
    // forward shading (shaders.hlsl:221-229, GetFragmentInfoDirect utilities.hlsli:2791-2807): the material inputs BEFORE the G-buffer's
    // UNORM8 / fp16 quantisation and the interpolated world position, 24 floats per pixel: base colour rgb + ao | metallic, roughness, coat
    // roughness, fuzz weight | coat colour rgb + weight | emissive rgb + 0 | fuzz colour rgb + roughness | world position xyz + 1
    float* forwardInputs = nullptr;
};

static bool resolvePixel(const brmi_scene_buffers& sc, const brmi_visible_cluster* clusters, uint32_t clusterCount, uint64_t key,
                         uint32_t px, uint32_t py, uint64_t idx, const GBufferOut& o) {
    if (key == BRMI_VIS_EMPTY) return false;
    const brmi_per_frame& pf = sc.perFrame[0];
    const brmi_camera& cam = sc.cameras[pf.mainCameraIndex];
    const uint32_t triId = (uint32_t)(key & 0x7Fu);
    const uint32_t clusterIndex = (uint32_t)((key >> BRMI_VIS_TRI_BITS) & 0x3FFFFFFu);
    if (clusterIndex >= clusterCount) return false;
    const brmi_visible_cluster& pc = clusters[clusterIndex];
    const uint32_t instanceID = vcInstanceID(pc), localMeshlet = vcLocalMeshlet(pc);
    const brmi_per_mesh_instance& inst = sc.perMeshInstance[instanceID];
    const brmi_per_mesh& mesh = sc.perMesh[inst.perMeshBufferIndex];
    const uint8_t* slab = sc.slabs[vcSlabDescriptor(pc)];
    const uint32_t pageOff = vcPageByteOffset(pc);
    const brmi_page_header& hdr = *pageHeader(slab, pageOff);
    const brmi_meshlet_descriptor& desc = *meshletDesc(slab, pageOff, hdr.descriptorOffset, localMeshlet);
    if (triId >= descTriangleCount(desc)) return false;
    uint32_t tri[3]; decodeTriangle(slab, pageOff + hdr.triangleStreamOffset, desc.triangleByteOffset, triId, tri);
    const uint32_t posBase = pageOff + hdr.positionBitstreamOffset;
    float3 p[3], n[3];
    const bool skinned = (mesh.vertexFlags & BRMI_VERTEX_SKINNED) != 0;
    for (int k = 0; k < 3; k++) {
        p[k] = loadPosition(slab, hdr.compressedPositionQuantExp, posBase, desc.positionBitOffset, tri[k]);
        n[k] = octDecodeNormal(load32(slab, pageOff + hdr.normalArrayOffset + (desc.vertexAttributeOffset + tri[k]) * 4u));
        if (skinned) {   // ApplyClodSkinning
            uint32_t joints[8] = {0, 0, 0, 0, 0, 0, 0, 0}; float weights[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_JOINTS) std::memcpy(joints, slab + pageOff + hdr.jointArrayOffset + (desc.vertexAttributeOffset + tri[k]) * 32u, 32);
            if (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_WEIGHTS) std::memcpy(weights, slab + pageOff + hdr.weightArrayOffset + (desc.vertexAttributeOffset + tri[k]) * 32u, 32);
            const mat4 skin = buildSkinMatrix(sc, inst.skinningInstanceSlot, joints, weights);
            p[k] = xyz(mulPoint(p[k], skin));
            n[k] = mul3(n[k], skin);
        }
    }
    const brmi_per_object& obj = sc.perObject[inst.perObjectBufferIndex];
    const brmi_material_info& mat = sc.materials[mesh.materialDataIndex];

    const mat4 viewProj = mul(M(cam.view), M(cam.projection));
    const mat4 objectToClip = mul(M(obj.model), viewProj);
    const float4 clip0 = mulPoint(p[0], objectToClip), clip1 = mulPoint(p[1], objectToClip), clip2 = mulPoint(p[2], objectToClip);
    const float2 winSize{(float)pf.screenResX, (float)pf.screenResY};
    const float2 pixelUv{((float)px + 0.5f) / winSize.x, ((float)py + 0.5f) / winSize.y};
    const float2 pixelNdc{pixelUv.x * 2.0f - 1.0f, (1.0f - pixelUv.y) * 2.0f - 1.0f};
    const Bary bary = calcFullBary(clip0, clip1, clip2, pixelNdc, winSize);

    const float3 posOS{interp(bary, p[0].x, p[1].x, p[2].x), interp(bary, p[0].y, p[1].y, p[2].y), interp(bary, p[0].z, p[1].z, p[2].z)};
    const float3 worldPosition = xyz(mulPoint(posOS, M(obj.model)));
    const float3 normalOS = normalize(float3{interp(bary, n[0].x, n[1].x, n[2].x), interp(bary, n[0].y, n[1].y, n[2].y), interp(bary, n[0].z, n[1].z, n[2].z)});
    const mat4& normalMatrix = *reinterpret_cast<const mat4*>(sc.normalMatrices + (size_t)obj.normalMatrixBufferIndex * 16u);
    const float3 worldNormal = normalize(mul3(normalOS, normalMatrix));

    // SampleMaterialEvalFromUvCache: factors x the texture slots the material enables
    const uint32_t flags = mat.materialFlags;
    auto uvOf = [&](uint32_t uvSetIndex) {       // AppendClodMaterialUvSample
        const uint32_t set = uvSetIndex < 8u ? uvSetIndex : 0u;
        const float2 a = decodeCompressedUV(slab, pageOff, hdr, localMeshlet, set, tri[0]), b = decodeCompressedUV(slab, pageOff, hdr, localMeshlet, set, tri[1]),
                     c = decodeCompressedUV(slab, pageOff, hdr, localMeshlet, set, tri[2]);
        const float3 iu = interpDeriv(bary, a.x, b.x, c.x), iv = interpDeriv(bary, a.y, b.y, c.y);
        return UvSample{{iu.x, iv.x}, {iu.y, iv.y}, {iu.z, iv.z}};
    };
    // the cotangent frame of the normal map / parallax (BuildMaterialUvBindings: the normal slot's UVs, else the height slot's)
    float3 Tn{}, Bn{};
    auto resolvedSet = [](uint32_t uvSetIndex) { return uvSetIndex < 8u ? uvSetIndex : 0u; };
    if (flags & (BRMI_MATERIAL_NORMAL_MAP | BRMI_MATERIAL_PARALLAX)) {
        // dpdx / dpdy of the object-space position through the model's 3x3 (clodResolveCommon.hlsli:1607-1624)
        const float3 ipx = interpDeriv(bary, p[0].x, p[1].x, p[2].x), ipy = interpDeriv(bary, p[0].y, p[1].y, p[2].y), ipz = interpDeriv(bary, p[0].z, p[1].z, p[2].z);
        const float3 dpdx = mul3(float3{ipx.y, ipy.y, ipz.y}, M(obj.model)), dpdy = mul3(float3{ipx.z, ipy.z, ipz.z}, M(obj.model));
        const UvSample u = uvOf((flags & BRMI_MATERIAL_NORMAL_MAP) ? mat.normalUvSetIndex : mat.heightUvSetIndex);
        // cotangent_frame_from_derivs
        const float3 dp2perp = cross(dpdy, worldNormal), dp1perp = cross(worldNormal, dpdx);
        const float3 T = dp2perp * u.dUVdx.x + dp1perp * u.dUVdy.x, B = dp2perp * u.dUVdx.y + dp1perp * u.dUVdy.y;
        const float invmax = rsqrt(fmax2(dot(T, T), dot(B, B)));
        Tn = T * invmax; Bn = B * invmax;
    }
    // PSO_PARALLAX (utilities.hlsli:1869-1897): every slot whose UV cache entry is the height map's samples at the displaced texcoord
    bool hasParallaxUv = false; float2 parallaxUv{};
    if (flags & BRMI_MATERIAL_PARALLAX) {
        const UvSample h = uvOf(mat.heightUvSetIndex);
        const float3 camPos{cam.positionWorldSpace[0], cam.positionWorldSpace[1], cam.positionWorldSpace[2]};
        parallaxUv = parallaxCoords(sc, mat.heightMapIndex, mat.heightSamplerIndex, Tn, Bn, worldNormal, h.uv, normalize(camPos - worldPosition), mat.heightMapScale, h.dUVdx, h.dUVdy);
        hasParallaxUv = true;
    }
    auto uvOfSlot = [&](uint32_t uvSetIndex) {      // ResolveMaterialUvSample: the parallax result keeps the height map's gradients
        UvSample u = uvOf(uvSetIndex);
        if (hasParallaxUv && resolvedSet(uvSetIndex) == resolvedSet(mat.heightUvSetIndex)) u.uv = parallaxUv;
        return u;
    };
    auto sample = [&](uint32_t textureIndex, uint32_t samplerIndex, uint32_t uvSetIndex) {
        const UvSample u = uvOfSlot(uvSetIndex);
        return sampleGrad(sc, textureIndex, samplerIndex, u.uv, u.dUVdx, u.dUVdy);
    };
    // DecodeCompressedColor + the vertexColor interpolation (clodResolveCommon.hlsli:657-667,1521-1531,1641-1648)
    float3 vertexColor{1.0f, 1.0f, 1.0f};
    if (hdr.attributeMask & BRMI_PAGE_ATTRIBUTE_COLOR) {
        float3 c[3];
        for (int k = 0; k < 3; k++) {
            const uint32_t packed = load32(slab, pageOff + hdr.colorArrayOffset + (desc.vertexAttributeOffset + tri[k]) * 4u);
            c[k] = float3{(float)(packed & 0xFFu) / 255.0f, (float)((packed >> 8) & 0xFFu) / 255.0f, (float)((packed >> 16) & 0xFFu) / 255.0f};
        }
        vertexColor = float3{interp(bary, c[0].x, c[1].x, c[2].x), interp(bary, c[0].y, c[1].y, c[2].y), interp(bary, c[0].z, c[1].z, c[2].z)};
    }
    float4 baseColor4{mat.baseColorFactor[0], mat.baseColorFactor[1], mat.baseColorFactor[2], mat.baseColorFactor[3]};
    if (flags & BRMI_MATERIAL_BASE_COLOR_TEXTURE) {
        const float4 t = sample(mat.baseColorTextureIndex, mat.baseColorSamplerIndex, mat.baseColorUvSetIndex);
        baseColor4 = float4{baseColor4.x * t.x, baseColor4.y * t.y, baseColor4.z * t.z, baseColor4.w * t.w};
    }
    if (flags & BRMI_MATERIAL_OPACITY_TEXTURE) baseColor4.w *= sample(mat.opacityTextureIndex, mat.opacitySamplerIndex, mat.opacityUvSetIndex).w;
    float metallic = mat.metallicFactor, roughness = mat.roughnessFactor;
    if (flags & BRMI_MATERIAL_METALLIC_TEXTURE) metallic = swizzle(sample(mat.metallicTextureIndex, mat.metallicSamplerIndex, mat.metallicUvSetIndex), mat.metallicChannel) * mat.metallicFactor;
    if (flags & BRMI_MATERIAL_ROUGHNESS_TEXTURE) roughness = swizzle(sample(mat.roughnessTextureIndex, mat.roughnessSamplerIndex, mat.roughnessUvSetIndex), mat.roughnessChannel) * mat.roughnessFactor;
    float3 normalWS = worldNormal;
    if (flags & BRMI_MATERIAL_NORMAL_MAP) {
        const UvSample u = uvOfSlot(mat.normalUvSetIndex);
        const float4 t = sampleGrad(sc, mat.normalTextureIndex, mat.normalSamplerIndex, u.uv, u.dUVdx, u.dUVdy);
        float3 tn = normalize(float3{t.x, t.y, t.z} * 2.0f - float3{1.0f, 1.0f, 1.0f});
        if (flags & BRMI_MATERIAL_NEGATE_NORMALS) tn = -tn;
        if (flags & BRMI_MATERIAL_INVERT_NORMAL_GREEN) tn.y = -tn.y;
        // mul(tangentSpaceNormal, float3x3(T, B, N)): rows of the frame
        normalWS = normalize(float3{(tn.x * Tn.x + tn.y * Bn.x) + tn.z * worldNormal.x, (tn.x * Tn.y + tn.y * Bn.y) + tn.z * worldNormal.y, (tn.x * Tn.z + tn.y * Bn.z) + tn.z * worldNormal.z});
    }
    float ao = 1.0f;
    if (flags & BRMI_MATERIAL_AO_TEXTURE) ao = swizzle(sample(mat.aoMapIndex, mat.aoSamplerIndex, mat.aoUvSetIndex), mat.aoChannel);
    float3 emissiveIn{mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]};
    if (flags & BRMI_MATERIAL_EMISSIVE_TEXTURE) {
        const float4 t = sample(mat.emissiveTextureIndex, mat.emissiveSamplerIndex, mat.emissiveUvSetIndex);
        emissiveIn = float3{swizzle(t, mat.emissiveChannels[0]), swizzle(t, mat.emissiveChannels[1]), swizzle(t, mat.emissiveChannels[2])} * emissiveIn;
    }
    const float3 baseColor = float3{baseColor4.x, baseColor4.y, baseColor4.z} * vertexColor;
    const brmi_openpbr_material_info& op = sc.openpbrMaterials[mat.openPBRMaterialDataIndex];
    const float3 canonicalEmissive = float3{op.emissionColor[0], op.emissionColor[1], op.emissionColor[2]} * op.emissionLuminance;
    float3 coatColor = saturate(float3{op.coatColor[0], op.coatColor[1], op.coatColor[2]});
    float coatWeight = saturate(op.coatWeight), coatRoughness = saturate(op.coatRoughness);
    float3 fuzzColor = saturate(float3{op.fuzzColor[0], op.fuzzColor[1], op.fuzzColor[2]});
    float fuzzWeight = saturate(op.fuzzWeight), fuzzRoughness = saturate(op.fuzzRoughness);
    {   // ApplyOpenPBRTextureSampling (utilities.hlsli:720-846): the six coat / fuzz slots, bound when texture AND sampler index are valid
        const uint32_t* tb = op.textureBindings;      // layout: include/brmi_types.h
        auto bound = [&](int k) { return tb[2 * k] != 0xFFFFFFFFu && tb[2 * k + 1] != 0xFFFFFFFFu; };
        auto fetch = [&](int k) { return sample(tb[2 * k], tb[2 * k + 1], tb[26 + k]); };
        if (bound(0)) { const float4 t = fetch(0); coatColor = coatColor * float3{swizzle(t, tb[12]), swizzle(t, tb[13]), swizzle(t, tb[14])}; }
        if (bound(1)) coatWeight *= swizzle(fetch(1), tb[16]);
        if (bound(2)) coatRoughness *= swizzle(fetch(2), tb[17]);
        if (bound(3)) { const float4 t = fetch(3); fuzzColor = fuzzColor * float3{swizzle(t, tb[19]), swizzle(t, tb[20]), swizzle(t, tb[21])}; }
        if (bound(4)) fuzzWeight *= swizzle(fetch(4), tb[23]);
        if (bound(5)) fuzzRoughness *= swizzle(fetch(5), tb[24]);
        coatColor = saturate(coatColor); coatWeight = saturate(coatWeight); coatRoughness = saturate(coatRoughness);
        fuzzColor = saturate(fuzzColor); fuzzWeight = saturate(fuzzWeight); fuzzRoughness = saturate(fuzzRoughness);
    }
    const float3 emissive = dot(emissiveIn, emissiveIn) > 0.0f ? emissiveIn : canonicalEmissive;

    // ComputeClodMotionVector
    const mat4 unjVP = mul(M(cam.view), M(cam.unjitteredProjection));
    const mat4 prevVP = mul(M(cam.prevView), M(cam.prevUnjitteredProjection));
    const float4 clipCur = mulPoint(worldPosition, unjVP);
    const float3 prevWorld = xyz(mulPoint(posOS, M(obj.prevModel)));
    const float4 clipPrev = mulPoint(prevWorld, prevVP);
    const float2 mv{clipCur.x / clipCur.w - clipPrev.x / clipPrev.w, clipCur.y / clipCur.w - clipPrev.y / clipPrev.w};

    o.normals[idx * 4 + 0] = normalWS.x; o.normals[idx * 4 + 1] = normalWS.y; o.normals[idx * 4 + 2] = normalWS.z;
    o.normals[idx * 4 + 3] = (float)mat.openPBRMaterialDataIndex;
    o.albedo[idx] = pack_unorm4(baseColor.x, baseColor.y, baseColor.z, ao);
    o.coat[idx] = pack_half4(coatColor.x, coatColor.y, coatColor.z, coatWeight);
    o.emissive[idx] = pack_half4(emissive.x, emissive.y, emissive.z, 0.0f);
    o.fuzz[idx] = pack_half4(fuzzColor.x, fuzzColor.y, fuzzColor.z, fuzzRoughness);
    o.metallicRoughness[idx] = pack_unorm4(metallic, roughness, coatRoughness, fuzzWeight);
    o.motion[idx] = (uint32_t)f32_to_f16(mv.x) | ((uint32_t)f32_to_f16(mv.y) << 16);
    if (o.forwardInputs) {
        float* r = o.forwardInputs + idx * 24u;
        const float v[24] = {baseColor.x, baseColor.y, baseColor.z, ao, metallic, roughness, coatRoughness, fuzzWeight, coatColor.x, coatColor.y, coatColor.z, coatWeight,
                             emissive.x, emissive.y, emissive.z, 0.0f, fuzzColor.x, fuzzColor.y, fuzzColor.z, fuzzRoughness, worldPosition.x, worldPosition.y, worldPosition.z, 1.0f};
        for (int k = 0; k < 24; k++) r[k] = v[k];
    }
    return true;
}

}  // namespace orc

using namespace orc;

extern "C" {

// CalcFullBary + InterpolateWithDeriv on caller-supplied clip-space corners (test hook, tests/test_oracle_cpu.py).  in: 19 floats per sample -- three
// float4 corners, the pixel's NDC (2), the window size (2), three corner values; out: 12 floats -- lambda, ddx, ddy, then value / d/dx / d/dy of the corner values.
int orc_calc_full_bary(const float* in, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* p = in + i * 19;
        const Bary b = calcFullBary(float4{p[0], p[1], p[2], p[3]}, float4{p[4], p[5], p[6], p[7]}, float4{p[8], p[9], p[10], p[11]}, float2{p[12], p[13]}, float2{p[14], p[15]});
        const float3 v = interpDeriv(b, p[16], p[17], p[18]);
        float* o = out + i * 12;
        o[0] = b.lambda.x; o[1] = b.lambda.y; o[2] = b.lambda.z; o[3] = b.ddx.x; o[4] = b.ddx.y; o[5] = b.ddx.z; o[6] = b.ddy.x; o[7] = b.ddy.y; o[8] = b.ddy.z;
        o[9] = v.x; o[10] = v.y; o[11] = v.z;
    }
    return 0;
}

// parallaxCoords on its own, one call per sample (tests): frame rows T / B / N, texcoord, world-space view direction, gradients
int orc_parallax_coords(const brmi_scene_buffers* sc, uint32_t heightMapIndex, uint32_t heightSamplerIndex, float heightmapScale, const float* T, const float* B, const float* N,
                        const float* uv, const float* viewDir, const float* dUVdx, const float* dUVdy, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float2 r = parallaxCoords(*sc, heightMapIndex, heightSamplerIndex, float3{T[i * 3], T[i * 3 + 1], T[i * 3 + 2]}, float3{B[i * 3], B[i * 3 + 1], B[i * 3 + 2]},
                                        float3{N[i * 3], N[i * 3 + 1], N[i * 3 + 2]}, float2{uv[i * 2], uv[i * 2 + 1]}, float3{viewDir[i * 3], viewDir[i * 3 + 1], viewDir[i * 3 + 2]},
                                        heightmapScale, float2{dUVdx[i * 2], dUVdx[i * 2 + 1]}, float2{dUVdy[i * 2], dUVdy[i * 2 + 1]});
        out[i * 2] = r.x; out[i * 2 + 1] = r.y;
    }
    return 0;
}

// All images are linear W x H.  Pixels without geometry are left untouched (the reference does not
// write them); callers pass zero-initialised buffers.
int orc_gbuffer_forward(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, uint32_t clusterCount, const uint64_t* vis,
                        uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1,
                        float* normals, uint32_t* albedo, uint64_t* coat, uint64_t* emissive, uint64_t* fuzz, uint32_t* metallicRoughness, uint32_t* motion,
                        float* forwardInputs /* may be null: 24 floats per pixel, see GBufferOut */, int threads);
int orc_gbuffer(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, uint32_t clusterCount, const uint64_t* vis,
                uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1,
                float* normals, uint32_t* albedo, uint64_t* coat, uint64_t* emissive, uint64_t* fuzz, uint32_t* metallicRoughness, uint32_t* motion,
                int threads) {
    return orc_gbuffer_forward(sc, clusters, clusterCount, vis, W, H, bandY0, bandY1, normals, albedo, coat, emissive, fuzz, metallicRoughness, motion, nullptr, threads);
}
int orc_gbuffer_forward(const brmi_scene_buffers* sc, const brmi_visible_cluster* clusters, uint32_t clusterCount, const uint64_t* vis,
                        uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1,
                        float* normals, uint32_t* albedo, uint64_t* coat, uint64_t* emissive, uint64_t* fuzz, uint32_t* metallicRoughness, uint32_t* motion,
                        float* forwardInputs, int threads) {
    GBufferOut o{normals, albedo, coat, emissive, fuzz, metallicRoughness, motion};
    o.forwardInputs = forwardInputs;
    if (bandY1 == 0) { bandY0 = 0; bandY1 = H; }
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads > 0 ? threads : 1)
    for (int64_t y = bandY0; y < (int64_t)bandY1; y++)
        for (uint32_t x = 0; x < W; x++) {
            const uint64_t idx = (uint64_t)y * W + x;
            resolvePixel(*sc, clusters, clusterCount, vis[idx], x, (uint32_t)y, idx, o);
        }
    return 0;
}

}  // extern "C"
