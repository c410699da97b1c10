// brmi_frame.hip -- per-frame constants of the whole path in ONE launch.
//
// The reference's shaders rebuild these values in every thread; they only depend on the camera, the objects, the materials and
// the lights, so they are evaluated once per brmi_update with the shaders' operation order and read back by the stages.  They
// used to be five small kernels in three stages; a kernel that does a few hundred threads of work still costs ~5 us on MI355X,
// so the five jobs share one launch, each taking a range of workgroups:
//   object constants      frameConst[0..2], objConst[o] = {model * cullCam.viewProjection, model * (view * projection), model * viewZ}
//   material words        the five packed G-buffer words of every constant-factor material
//   material constants    per-material part of PopulateFragmentInfoFromOpenPBR
//   shade tables          uv / cluster tile per column and row, first depth of every cluster slice
//   light spheres         view-space bounding sphere of every active light
#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_texture.h"
#include "brmi_shade_math.h"

namespace brmi {

// Frame / object constants ------------------------------------------------------------------------
// The reference shaders rebuild these matrix products in every thread (softwareRaster.hlsl:336-337,
// clodResolveCommon.hlsli:1531-1535,1697-1702).  They only depend on the camera and the object, so they
// are evaluated once per frame / once per object with the same operation order and read back later.
//   frameConst[0] = mul(view, projection)           frameConst[1] = mul(view, unjitteredProjection)
//   frameConst[2] = mul(prevView, prevUnjitteredProjection)
//   objConst[o]   = { mul(model, cullCam.viewProjection), mul(model, frameConst[0]), mul(model, cullCam.viewZ) }
BRMI_DEV void job_object_constants(const brmi_scene_buffers& sc, m4* frameConst, float* objConst, FrameSnapshot* snap, const float* bandPlanes, uint32_t o) {
    const uint32_t viewId = sc.perFrame->mainCameraIndex;
    if (o < 64u) {      // the job's first workgroup: the camera and the per-frame record the shading half will read (FrameSnapshot)
        const uint32_t* pfSrc = reinterpret_cast<const uint32_t*>(sc.perFrame); uint32_t* pfDst = reinterpret_cast<uint32_t*>(&snap->perFrame);
        for (uint32_t i = o; i < sizeof(brmi_per_frame) / 4u; i += 64u) pfDst[i] = (i == offsetof(brmi_per_frame, mainCameraIndex) / 4u) ? 0u : pfSrc[i];
        const uint32_t* cSrc = reinterpret_cast<const uint32_t*>(sc.cameras + viewId); uint32_t* cDst = reinterpret_cast<uint32_t*>(&snap->camera);
        for (uint32_t i = o; i < sizeof(brmi_camera) / 4u; i += 64u) cDst[i] = cSrc[i];
    }
    const brmi_camera* cam = sc.cameras + viewId;
    const brmi_culling_camera* cc = sc.cullingCameras + viewId;
    const m4 viewM = load_m4(&cam->view[0][0]);
    const m4 viewProj = mul_mm(viewM, load_m4(&cam->projection[0][0]));
    if (o == 0) {
        frameConst[0] = viewProj;
        frameConst[1] = mul_mm(viewM, load_m4(&cam->unjitteredProjection[0][0]));
        frameConst[2] = mul_mm(load_m4(&cam->prevView[0][0]), load_m4(&cam->prevUnjitteredProjection[0][0]));
        // the band's two view-space planes (brmi_update) where the traversal kernels read them like the camera's clipping planes: as kernel arguments they cost
        // those kernels seven scalar registers each held across the whole walk
        float* bp = &frameConst[3].m[0][0];
        for (int k = 0; k < 8; k++) bp[k] = bandPlanes[k];
    }
    if (o >= sc.perObjectCount) return;
    const m4 model = load_m4(&sc.perObject[o].model[0][0]);
    const m4 mvp = mul_mm(model, load_m4(&cc->viewProjection[0][0]));
    const m4 otc = mul_mm(model, viewProj);
    const f4 mvz = mul_mcol(model, f4{cc->viewZ[0], cc->viewZ[1], cc->viewZ[2], cc->viewZ[3]});
    float* d = objConst + (size_t)o * OBJ_CONST_FLOATS;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { d[i * 4 + j] = mvp.m[i][j]; d[16 + i * 4 + j] = otc.m[i][j]; }
    d[32] = mvz.x; d[33] = mvz.y; d[34] = mvz.z; d[35] = mvz.w;
    // round 6: where the object was in the PREVIOUS frame's view (the draw list's prediction looks a cluster's box up in the previous frame's depth
    // chain, as the reference's occlusion test does with the sphere: prevModel, prevView, prevUnjitteredProjection -- occlusionCulling.hlsli:165-212's caller).
    // Only a prediction reads these: no arithmetic contract.
    const m4 prevModel = load_m4(&sc.perObject[o].prevModel[0][0]);
    const m4 prevView = load_m4(&cam->prevView[0][0]);
    const m4 pmvp = mul_mm(prevModel, mul_mm(prevView, load_m4(&cam->prevUnjitteredProjection[0][0])));
    const f4 pmvz = mul_mcol(prevModel, f4{prevView.m[0][2], prevView.m[1][2], prevView.m[2][2], prevView.m[3][2]});
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) d[36 + i * 4 + j] = pmvp.m[i][j];
    d[52] = pmvz.x; d[53] = pmvz.y; d[54] = pmvz.z; d[55] = pmvz.w;
}

// The constant-factor material of SampleMaterialEvalFromUvCache (no texture permutations) only depends on the material
// record: its five packed G-buffer words are evaluated once per material per frame instead of once per pixel.
BRMI_DEV void job_material_words(const brmi_scene_buffers& sc, MaterialWords* out, AlphaMaterial* alphaMats, uint32_t i) {
    if (i >= sc.materialCount) return;
    const brmi_material_info* mat = sc.materials + i;
    // the rasteriser's alpha test: texture bindings and SampleLevel(.., 0)'s level choice, resolved once per material
    if (alphaMats && (mat->materialFlags & BRMI_MATERIAL_ALPHA_TEST)) alphaMats[i] = load_alpha_material(mat, sc.textures, sc.textureCount, sc.samplers, sc.samplerCount);
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
    MaterialWords w;
    w.albedo = pack_unorm4(baseColor.x, baseColor.y, baseColor.z, ao);
    w.metallicRoughness = pack_unorm4(metallic, roughness, coatRoughness, fuzzWeight);
    w.coat = pack_half4(coatColor.x, coatColor.y, coatColor.z, coatWeight);
    w.emissive = pack_half4(emissive.x, emissive.y, emissive.z, 0.0f);
    w.fuzz = pack_half4(fuzzColor.x, fuzzColor.y, fuzzColor.z, fuzzRoughness);
    w.opIndexF = (float)opIndex; w.pad = openpbr_has_textures(op) ? 1u : 0u;        // bit 0: the OpenPBR record binds coat / fuzz textures
    out[i] = w;
}

// Do all materials of the scene store the same coat / fuzz word?  (One wave; the words are those of job_material_words, and a record that
// binds coat / fuzz textures makes its plane per-pixel.)  The G-buffer kernels skip a plane's stores while it holds that word everywhere.
BRMI_DEV void job_layer_uniform(const brmi_scene_buffers& sc, LayerUniform* out, uint32_t lane) {
    unsigned long long coat0 = 0ull, fuzz0 = 0ull;
    bool coatSame = true, fuzzSame = true;
    auto words = [&](uint32_t i, unsigned long long& c, unsigned long long& f, bool& tex) {
        const brmi_openpbr_material_info* op = sc.openpbrMaterials + sc.materials[i].openPBRMaterialDataIndex;
        const f3 coatColor = sat3(f3{op->coatColor[0], op->coatColor[1], op->coatColor[2]});
        const f3 fuzzColor = sat3(f3{op->fuzzColor[0], op->fuzzColor[1], op->fuzzColor[2]});
        c = pack_half4(coatColor.x, coatColor.y, coatColor.z, sat(op->coatWeight));
        f = pack_half4(fuzzColor.x, fuzzColor.y, fuzzColor.z, sat(op->fuzzRoughness));
        tex = openpbr_has_textures(op);
    };
    bool tex0 = false;
    if (sc.materialCount != 0u) words(0u, coat0, fuzz0, tex0);
    coatSame = fuzzSame = sc.materialCount != 0u && !tex0;
    for (uint32_t i = lane; i < sc.materialCount; i += 64u) {
        unsigned long long c, f; bool tex;
        words(i, c, f, tex);
        coatSame = coatSame && !tex && c == coat0;
        fuzzSame = fuzzSame && !tex && f == fuzz0;
    }
    coatSame = __all(coatSame); fuzzSame = __all(fuzzSame);
    if (lane == 0u) { out->coatWord = coat0; out->fuzzWord = fuzz0; out->coatUniform = coatSame ? 1u : 0u; out->fuzzUniform = fuzzSame ? 1u : 0u; }
}

// After brmi_setup (new plane memory): a uniform plane is filled with its word once; `filled` is what lets the G-buffer kernels skip it.
__global__ void __launch_bounds__(256) k_fill_layer_planes(LayerUniform* L, unsigned long long* coat, unsigned long long* fuzz, uint64_t n) {
    const bool doCoat = L->coatUniform != 0u, doFuzz = L->fuzzUniform != 0u;
    const unsigned long long cw = L->coatWord, fw = L->fuzzWord;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (doCoat) coat[i] = cw;
        if (doFuzz) fuzz[i] = fw;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { L->coatFilled = doCoat ? 1u : 0u; L->fuzzFilled = doFuzz ? 1u : 0u; L->coatFilledWord = cw; L->fuzzFilledWord = fw; }
}

// view-space bounding spheres of the active lights, once per frame (testSphereAABB's transform, lightCulling.hlsl:15-21)
// ... and the shading pass's record of the light (brmi_light.hip, ShadeLightLanes): what getLightParametersForFragment reads, 64 B, indexed
// (four float4) by the position in the active-light list.  normalize(dirWorldSpace) of a spot light is per light, not per pixel (lighting.hlsli:640).
BRMI_DEV void job_light_spheres(const brmi_scene_buffers& sc, float4* lightVS, uint32_t* lightMeta, float4* shadeLights, uint32_t i) {
    const brmi_per_frame* pf = sc.perFrame;
    if (i >= pf->numLights) return;
    const m4 view = load_m4(&sc.cameras[pf->mainCameraIndex].view[0][0]);
    const uint32_t li = sc.activeLightIndices[i];
    const brmi_light_info* l = sc.lights + li;
    const f3 c = xyz(mul_point(f3{l->boundingSphere[0], l->boundingSphere[1], l->boundingSphere[2]}, view));
    lightVS[i] = make_float4(c.x, c.y, c.z, l->boundingSphere[3]);
    lightMeta[i] = (l->type & 3u) | (li << 2);
    float4* r = shadeLights + (size_t)i * 4u;
    const f3 dir{l->dirWorldSpace[0], l->dirWorldSpace[1], l->dirWorldSpace[2]};
    if (l->type == BRMI_LIGHT_DIRECTIONAL) r[0] = make_float4(-dir.x, -dir.y, -dir.z, -1.0f);     // lightToFrag; a negative range marks the type
    else r[0] = make_float4(l->posWorldSpace[0], l->posWorldSpace[1], l->posWorldSpace[2], max2(l->maxRange, 0.0f));
    // dist > maxRange (lighting.hlsli:614-617, dist = the correctly rounded sqrt of the squared distance) as a test on the SQUARED distance: sqrt is monotone, so
    // the pixels the light reaches are exactly those with d2 <= T, T = the largest float whose root is <= maxRange -- found here once per light (R * R is within
    // an ulp or two of it).  Round 5: the pixel needs neither a conservative bound nor the exact root for the cut; a negative range rejects every pixel.
    float T = -1.0f;
    if (!(l->maxRange < 0.0f)) {
        const float R = max2(l->maxRange, 0.0f);
        T = R * R;
        for (int i = 0; i < 4 && sqrtf(T) > R; i++) T = as_f32(as_u32(T) - 1u);
        for (int i = 0; i < 4; i++) { const float n = as_f32(as_u32(T) + 1u); if (sqrtf(n) <= R) T = n; else break; }
    }
    r[1] = make_float4(l->attenuation[0], l->attenuation[1], l->attenuation[2], T);
    r[2] = make_float4(l->color[0] * l->color[3], l->color[1] * l->color[3], l->color[2] * l->color[3], l->innerConeAngle);
    const bool spotL = l->type == BRMI_LIGHT_SPOT;
    const f3 sd = spotL ? normalize3(dir) : f3{0.0f, 0.0f, 0.0f};
    r[3] = make_float4(sd.x, sd.y, sd.z, spotL ? max2(l->outerConeAngle, -1.0f) : -2.0f);
}

// Per-material part of PopulateFragmentInfoFromOpenPBR (utilities.hlsli:2590-2637): depends only on the
// OpenPBR material record, so it is evaluated once per material per frame instead of once per pixel.
BRMI_DEV MatConst material_constants_of(const brmi_openpbr_material_info* op) {
    MatConst m;
    m.baseWeight = sat(op->baseWeight); m.specularWeight = sat(op->specularWeight);
    m.specR = sat(op->specularColor[0]); m.specG = sat(op->specularColor[1]); m.specB = sat(op->specularColor[2]);
    const float unscaledF0 = ior_to_f0(op->specularIor);
    const float scaledF0 = min2(unscaledF0 * sat(m.specularWeight), 0.9999f);
    const float safeF0 = min2(sat(scaledF0), 0.9999f);
    const float sq = sqrtf(safeF0);
    m.weightedSpecularIor = (1.0f + sq) / max2(1.0f - sq, 1.0e-4f);
    m.dielF0Scalar = ior_to_f0(m.weightedSpecularIor);
    m.coatF0Scalar = ior_to_f0(op->coatIor);
    m.coatIor = op->coatIor; m.coatDarkening = sat(op->coatDarkening); m.baseDiffuseRoughness = sat(op->baseDiffuseRoughness);
    {   // the pixel's own expressions (brmi_shade.h, PopulateFragmentInfoFromOpenPBR + EvaluateOpenPBRBaseLayerDirect), evaluated once per material
        const f3 F0 = satq3(f3{m.specR, m.specG, m.specB} * m.dielF0Scalar);
        const float tmp = 50.0f * 0.33f;
        m.dielF0[0] = F0.x; m.dielF0[1] = F0.y; m.dielF0[2] = F0.z; m.f90Diel = satq(dot3(F0, f3{tmp, tmp, tmp}));
    }
    {   // OpenPBRDiffuseEON, the part that only depends on the material's diffuse roughness (IBL.hlsli:94-131)
        const float rough = m.baseDiffuseRoughness;
        const float A = qrcp(1.0f + fon_a() * rough);
        const float g[4] = {0.0571085289f, 0.491881867f, -0.332181442f, 0.0714429953f};
        m.fonA = A;
        for (int k = 0; k < 4; k++) m.fonK[k] = A * rough * g[k];
        m.eonSingleScale = (1.0f / PI_F) * A;
        m.eonAvgE = A * (1.0f + fon_b() * rough);
        m.eonOneMinusAvgE = 1.0f - m.eonAvgE;
        m.eonInvDen = qrcp(max2(1.0e-4f, 1.0f - m.eonAvgE));
    }
    return m;
}
BRMI_DEV void job_material_constants(const brmi_scene_buffers& sc, MatConst* out, uint32_t i) {
    if (i >= sc.openpbrMaterialCount) return;
    out[i] = material_constants_of(sc.openpbrMaterials + i);
}
// (material, roughness code) -> the light-independent table part of make_pixel_ctx (brmi_shade_math.h): the same functions the shader
// called per pixel, evaluated once per pair
BRMI_DEV void job_shade_material_table(const brmi_scene_buffers& sc, const float* lutF, ShadeRows* rows, ShadeAverages* avgs, GgxQuad* quads, uint32_t i) {
    const uint32_t m = i >> 8, code = i & 255u;
    const Luts L{lutF, lutF + 32768, lutF + 32768 + 1024, lutF + 32768 + 2048, sc.lutFuzzLTC, lutF + 32768 + 2048 + 32};
    if (m == 0u) { const float pc = clampf(L.unorm8[code], BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f); quads[code] = ggx_quad_of(pc * pc); }   // f.roughness / f.coatRoughness of the code
    if (m >= sc.openpbrMaterialCount) return;
    const MatConst mc = material_constants_of(sc.openpbrMaterials + m);
    const float prc = clampf(L.unorm8[code], BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
    const float alpha = sat(prc * prc), ior = max2(mc.weightedSpecularIor, 1.0f);        // BaseState::specularAlpha / weightedSpecularIor
    build_shade_rows(L, ior, alpha, rows[i], avgs[i]);
}

// Per-frame tables of the shading pass.  Everything here is what the shader computes per pixel from px, py or view depth
// alone, evaluated once per column / row / slice with the shader's own (correctly rounded) arithmetic:
//   uvx[px] = (px + 0.5) / resX     tileX[px] = (uint)(px / (resX / gx))       (lighting.hlsli cluster lookup)
//   uvy[py], tileY[py] likewise
//   sliceStart[s] = smallest view depth whose cluster slice is >= s (the slice formula is monotone in depth), s = 1..gz;
//   sliceStart[0] = 0, sliceStart[gz + 1] = +inf.
// The slice starts are evaluated on the host (brmi_update: they depend on the camera's depth range and the grid alone, a bisection of ~30
// dependent fp64 logarithms per slice that was this kernel's critical path) and travel as kernel arguments.
BRMI_DEV void job_shade_tables(const brmi_scene_buffers& sc, ShadeTables t, uint32_t W, uint32_t H, const StripeMap& stripes, const float* sliceStart, uint32_t i) {
    const brmi_per_frame* pf = sc.perFrame;
    const float resX = (float)pf->screenResX, resY = (float)pf->screenResY;
    const uint32_t gx = pf->lightClusterGridSizeX, gy = pf->lightClusterGridSizeY, gz = pf->lightClusterGridSizeZ;
    const float tsx = resX / (float)gx, tsy = resY / (float)gy;
    if (i < W) t.x[i] = AxisEntry{((float)i + 0.5f) / resX, (uint32_t)((float)i / tsx)};
    // (interleaved partition: the table is indexed by the surface row, its entries are those of the frame row behind it)
    if (i < H) { const float fy = (float)stripe_rrow(stripes, i); t.y[i] = AxisEntry{(fy + 0.5f) / resY, (uint32_t)(fy / tsy)}; }
    if (i <= gz + 1u) t.sliceStart[i] = sliceStart[i];
}

struct FrameJobs {
    brmi_scene_buffers sc;
    m4* frameConst; FrameSnapshot* snapshot; float* objConst; MaterialWords* matWords; MatConst* matConst; ShadeTables tables; float4* lightVS; uint32_t* lightMeta; float4* shadeLights;
    AlphaMaterial* alphaMats; LayerUniform* layer;
    const float* lutF; ShadeRows* shadeRows; ShadeAverages* shadeAvgs; GgxQuad* ggxQuads;
    uint32_t W, H; StripeMap stripes;
    float bandPlanes[8];         // top plane xyz, 0, bottom plane xyz, 0 (brmi_update)
    float sliceStart[64];        // first view depth of every light-cluster slice (brmi_update), [0] = 0, [gz + 1] = +inf
    uint4* frameState; uint64_t frameState16;      // brmi_execute: the culling pass's counters + survivor bitmasks, zeroed here (job 7)
    uint32_t firstBlock[8];      // block ranges of the seven jobs
};

__global__ void __launch_bounds__(64) k_frame_constants(FrameJobs j) {
    wave_prio<PRIO_SCAN>();
    const uint32_t b = blockIdx.x;
    if (b < j.firstBlock[1]) job_object_constants(j.sc, j.frameConst, j.objConst, j.snapshot, j.bandPlanes, (b - j.firstBlock[0]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[2]) job_material_words(j.sc, j.matWords, j.alphaMats, (b - j.firstBlock[1]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[3]) job_material_constants(j.sc, j.matConst, (b - j.firstBlock[2]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[4]) job_shade_tables(j.sc, j.tables, j.W, j.H, j.stripes, j.sliceStart, (b - j.firstBlock[3]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[5]) job_light_spheres(j.sc, j.lightVS, j.lightMeta, j.shadeLights, (b - j.firstBlock[4]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[6]) job_shade_material_table(j.sc, j.lutF, j.shadeRows, j.shadeAvgs, j.ggxQuads, (b - j.firstBlock[5]) * 64u + threadIdx.x);
    else if (b < j.firstBlock[7]) { for (uint64_t i = (uint64_t)(b - j.firstBlock[6]) * 64u + threadIdx.x; i < j.frameState16; i += (uint64_t)(j.firstBlock[7] - j.firstBlock[6]) * 64u) j.frameState[i] = make_uint4(0u, 0u, 0u, 0u); }
    else job_layer_uniform(j.sc, j.layer, threadIdx.x);
}

ShadeTables shade_tables_of(const brmi_pass* p) {
    uint32_t* tb = p->wsPtr<uint32_t>(p->ws.shadeTables);
    const uint32_t W = p->cfg.width, H = p->cfg.height;
    return ShadeTables{reinterpret_cast<AxisEntry*>(tb), reinterpret_cast<AxisEntry*>(tb + 2 * W), reinterpret_cast<float*>(tb + 2 * W + 2 * H)};
}

brmi_scene_buffers shading_scene_of(const brmi_pass* p) {
    brmi_scene_buffers sc = p->scene;
    FrameSnapshot* snap = p->wsPtr<FrameSnapshot>(p->ws.frameSnapshot);
    sc.perFrame = &snap->perFrame; sc.cameras = &snap->camera; sc.cameraCount = 1u;
    return sc;
}

// Launches the constants kernel if brmi_update / brmi_set_scene happened since the last time (every stage calls this first).
int ensure_frame_constants(brmi_pass* p, hipStream_t s) {
    if (p->constantsSerial == p->updateSerial) return BRMI_OK;
    if (p->pfHost.numLights > p->scene.lightCount) return fail(p, BRMI_ERR_INVALID, "perFrame.numLights (%u) exceeds the light buffer (%u)", p->pfHost.numLights, p->scene.lightCount);
    FrameJobs j;
    j.sc = p->scene; j.frameConst = p->wsPtr<m4>(p->ws.frameConst); j.snapshot = p->wsPtr<FrameSnapshot>(p->ws.frameSnapshot); j.objConst = p->wsPtr<float>(p->ws.objConst);
    j.matWords = p->wsPtr<MaterialWords>(p->ws.matWords); j.matConst = p->wsPtr<MatConst>(p->ws.matConst); j.tables = shade_tables_of(p);
    j.lightVS = p->wsPtr<float4>(p->ws.lightVS); j.lightMeta = p->wsPtr<uint32_t>(p->ws.lightMeta); j.shadeLights = p->wsPtr<float4>(p->ws.shadeLights);
    j.alphaMats = p->sceneHasAlphaTest ? p->wsPtr<AlphaMaterial>(p->ws.alphaMats) : nullptr;
    j.layer = p->wsPtr<LayerUniform>(p->ws.layerUniform);
    j.W = p->cfg.width; j.H = p->cfg.height; j.stripes = p->stripes;
    for (int k = 0; k < 3; k++) { j.bandPlanes[k] = p->bandPlaneTop[k]; j.bandPlanes[4 + k] = p->bandPlaneBottom[k]; }
    j.bandPlanes[3] = j.bandPlanes[7] = 0.0f;
    for (uint32_t k = 0; k < 64; k++) j.sliceStart[k] = k < p->sliceStartHost.size() ? p->sliceStartHost[k] : 0.0f;
    auto blocks = [](uint32_t n) { return (std::max(1u, n) + 63u) / 64u; };
    j.lutF = p->wsPtr<float>(p->ws.lutF); j.shadeRows = p->wsPtr<ShadeRows>(p->ws.shadeRows); j.shadeAvgs = p->wsPtr<ShadeAverages>(p->ws.shadeAvgs); j.ggxQuads = p->wsPtr<GgxQuad>(p->ws.ggxQuads);
    const uint32_t counts[6] = {blocks(p->scene.perObjectCount), blocks(p->scene.materialCount), blocks(p->scene.openpbrMaterialCount),
                                blocks(std::max(std::max(j.W, j.H), 64u)), blocks(p->pfHost.numLights),
                                // the (material, roughness) table only depends on the OpenPBR records and the lookup tables: built with the first
                                // frame after brmi_setup (OpenPBR records edited on the device later: brmi_set_scene + brmi_setup again)
                                p->constantsSerial == 0 ? blocks(std::max(1u, p->scene.openpbrMaterialCount) * 256u) : 0u};
    j.firstBlock[0] = 0;
    for (int k = 0; k < 6; k++) j.firstBlock[k + 1] = j.firstBlock[k] + counts[k];
    // brmi_execute without a clear launch of its own (the visibility clear rides on the traversal kernel): the frame state is zeroed here
    j.frameState = p->wsPtr<uint4>(p->ws.counters); j.frameState16 = p->clearFrameStateWithConstants ? p->ws.frameClearBytes / 16 : 0ull;
    j.firstBlock[7] = j.firstBlock[6] + (uint32_t)std::min<uint64_t>((j.frameState16 + 511u) / 512u, 4096u);
    hipLaunchKernelGGL(k_frame_constants, dim3(j.firstBlock[7] + 1u), dim3(64), 0, s, j);       // + the layer-uniformity job's block
    BRMI_LAUNCH_CHECK(p, "k_frame_constants");
    if (p->layerPlanesDirty) {
        // first constants after brmi_setup: planes whose word is the same for every material are filled once (the G-buffer kernels then skip them)
        if (p->cfg.keepUniformLayerPlanes) {
            hipLaunchKernelGGL(k_fill_layer_planes, dim3(2048), dim3(256), 0, s, j.layer, static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_COAT]), static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_FUZZ]),
                               std::min(p->resBytes[BRMI_RES_GBUF_COAT], p->resBytes[BRMI_RES_GBUF_FUZZ]) / 8u);
            BRMI_LAUNCH_CHECK(p, "k_fill_layer_planes");
            // once per brmi_setup: the host picks the G-buffer kernel's instantiation from the answer
            LayerUniform lu;
            BRMI_HIP(p, hipMemcpyAsync(&lu, j.layer, sizeof(lu), hipMemcpyDeviceToHost, s));
            BRMI_HIP(p, hipStreamSynchronize(s));
            p->layerPlanesUniform = lu.coatUniform != 0u && lu.fuzzUniform != 0u && lu.coatFilled != 0u && lu.fuzzFilled != 0u;
        } else p->layerPlanesUniform = false;
        p->layerPlanesDirty = false;
    }
    if (p->clearFrameStateWithConstants) p->frameStateCleared = true;
    p->constantsSerial = p->updateSerial;
    return BRMI_OK;
}

}  // namespace brmi
