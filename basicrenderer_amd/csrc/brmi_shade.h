// brmi_shade.h -- the deferred shading pass (K11) as device code: k_shade<> (brmi_light.hip) and the fused G-buffer + shading kernel
// (brmi_resolve.hip) are built from it.  Reference: BR/shaders/deferred.hlsl:11-106, BR/shaders/Include/lighting.hlsli:81-196,391-661,
// BR/shaders/Include/IBL.hlsli:94-672, BR/shaders/Include/PBR.hlsli:8-190.
#ifndef BRMI_SHADE_H
#define BRMI_SHADE_H
#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_shade_math.h"

namespace brmi {

// =================================== K11 =======================================================
struct BaseState {
    f3 weightedBaseColor, diffuseColor; float baseDiffuseRoughness, specularAlpha, weightedSpecularIor;
    f3 dielectricSpecularF0; float dielectricSpecularWeight; f3 metalSpecularF0, metalAverageFresnel, metalMultipleScatterScale; float metalSpecularWeight;
};
struct CoatState { f3 tint; float presence, ior, roughness; f3 extraBaseLayerScale; f3 t0, lt0; float eta; };   // t0 = sqrt(tint), lt0 = log2(t0), eta = 1 / ior: light-independent parts of coat_passage
struct FuzzState { float roughness; f3 tint; float presence; f3 t, b, n, viewDirLocal; float viewReflected; float sa, ca; f3 ltcView; };   // sa, ca, ltcView: light-independent parts of fuzz_sheen
struct Frag {
    f3 posWS, normalWS, viewWS, albedo, diffuseColor, emissive, dielectricSpecularF0, metalSpecularF0, metalAverageFresnel, coatColor, coatF0, fuzzColor;
    float NdotV, roughness, baseDiffuseRoughness, specularAlpha, weightedSpecularIor, dielectricSpecularWeight, metalSpecularWeight, coatWeight, coatIor, coatDarkening, coatRoughness, fuzzWeight, fuzzRoughness;
};

// MakeOpenPBRBaseLayerState saturates its inputs; every one of them arrives saturated here (shade_pixel builds the Frag from UNORM codes, clamped
// roughness and MatConst values that material_constants_of saturated; the IOR is >= 1 by its formula), and saturating twice is the identity -- round 5
// dropped the second pass (17 instructions per pixel).
BRMI_DEV BaseState make_base_state(const Frag& f) {
    BaseState s;
    s.weightedBaseColor = f.albedo; s.diffuseColor = f.diffuseColor; s.baseDiffuseRoughness = f.baseDiffuseRoughness;
    s.specularAlpha = f.specularAlpha; s.weightedSpecularIor = f.weightedSpecularIor;
    s.dielectricSpecularF0 = f.dielectricSpecularF0; s.dielectricSpecularWeight = f.dielectricSpecularWeight;
    s.metalAverageFresnel = f.metalAverageFresnel; s.metalSpecularF0 = f.metalSpecularF0; s.metalSpecularWeight = f.metalSpecularWeight;
    s.metalMultipleScatterScale = s.metalSpecularWeight * s.metalAverageFresnel * s.metalAverageFresnel;
    return s;
}
BRMI_DEV CoatState make_coat_state(const BaseState& b, const Frag& f) {
    CoatState s;
    s.tint = satq3(f.coatColor); s.presence = satq(f.coatWeight); s.ior = max2(f.coatIor, 1.0f); s.roughness = satq(f.coatRoughness);
    // OpenPBRComputeCoatExtraBaseLayerScale
    const float safeIor = max2(s.ior, 1.0f);
    const float K_s = average_fresnel(safeIor);
    const float K_r = 1.0f - (1.0f - K_s) / max2(safeIor * safeIor, 1.0e-4f);
    const float ds = average_fresnel(b.weightedSpecularIor);
    const float specBase = satq(b.dielectricSpecularWeight * ds + (1.0f - b.dielectricSpecularWeight));
    const float effRough = lerpf(1.0f, sqrtf(satq(b.specularAlpha)), specBase);
    const float K = lerpf(K_s, K_r, effRough);
    const f3 fromMetal = b.metalSpecularWeight * b.metalAverageFresnel;
    const f3 fromDiel = b.dielectricSpecularWeight * lerp3(b.weightedBaseColor, f3{1.0f, 1.0f, 1.0f}, ds);
    const f3 E_b = satq3(fromMetal + fromDiel);
    const f3 Delta = f3{1.0f - K, 1.0f - K, 1.0f - K} / max3v(f3{1.0f, 1.0f, 1.0f} - E_b * K, f3{1.0e-4f, 1.0e-4f, 1.0e-4f});
    const float mod = satq(s.presence) * satq(f.coatDarkening);
    s.extraBaseLayerScale = lerp3(f3{1.0f, 1.0f, 1.0f}, satq3(Delta), mod);
    s.t0 = f3{sqrtf(s.tint.x), sqrtf(s.tint.y), sqrtf(s.tint.z)}; s.eta = rcpf(s.ior);
    s.lt0 = f3{__builtin_amdgcn_logf(s.t0.x), __builtin_amdgcn_logf(s.t0.y), __builtin_amdgcn_logf(s.t0.z)};
    return s;
}
BRMI_DEV f3 coat_passage(const CoatState& s, float NdotX) {
    const float c = satq(NdotX);
    if (c <= 0.0f || min2(s.tint.x, min2(s.tint.y, s.tint.z)) >= 1.0f) return f3{1.0f, 1.0f, 1.0f};
    const float eta = s.eta;
    const float rc = sqrtf(max2(0.0f, 1.0f - (1.0f - c * c) / max2(eta * eta, 1.0e-4f)));
    const float ds = rcpf(max2(rc, 1.0e-4f));
    // pow(x, y) = exp2(y * log2 x), as DXC lowers it (tolerance-level, like the Schlick power); log2(t0) is hoisted with t0
    const f3 tr{__builtin_amdgcn_exp2f(ds * s.lt0.x), __builtin_amdgcn_exp2f(ds * s.lt0.y), __builtin_amdgcn_exp2f(ds * s.lt0.z)};
    return lerp3(f3{1.0f, 1.0f, 1.0f}, tr, s.presence);
}
BRMI_DEV float coat_reflected(const Luts& L, const CoatState& s, float NdotX) {
    const float si = max2(s.ior, 1.0e-4f), sa = satq(s.roughness), sc = satq(NdotX);
    const float refl = (sa <= 0.0f) ? fresnel_dielectric(si, sc) : 1.0f - lut_od_e(L, si, sa, sc);
    return satq(s.presence * refl);
}
BRMI_DEV f3 coat_scale_incoming(const Luts& L, const CoatState& s, float NdotV) {
    const float rp = coat_reflected(L, s, NdotV);
    return coat_passage(s, NdotV) * f3{1.0f - rp, 1.0f - rp, 1.0f - rp} * s.extraBaseLayerScale;
}
BRMI_DEV f3 coat_scale_outgoing(const Luts& L, const CoatState& s, float NdotL) {
    const float rp = coat_reflected(L, s, NdotL);
    return coat_passage(s, NdotL) * f3{1.0f - rp, 1.0f - rp, 1.0f - rp};
}
// coat_reflected / coat_scale_outgoing with the coat's (ior, roughness) table rows prepared once per pixel
BRMI_DEV OdPrep prep_coat_od(const Luts& L, const CoatState& s) { return prep_od_e(L, max2(s.ior, 1.0e-4f), satq(s.roughness)); }
BRMI_DEV float coat_reflected_prepared(const Luts& L, const CoatState& s, const OdPrep& od, float NdotX) {
    const float si = max2(s.ior, 1.0e-4f), sa = satq(s.roughness), sc = satq(NdotX);
    const float refl = (sa <= 0.0f) ? fresnel_dielectric(si, sc) : 1.0f - sample_od_e(L, od, sc);
    return satq(s.presence * refl);
}
BRMI_DEV f3 coat_scale_outgoing_prepared(const Luts& L, const CoatState& s, const OdPrep& od, float NdotL) {
    const float rp = coat_reflected_prepared(L, s, od, NdotL);
    return coat_passage(s, NdotL) * f3{1.0f - rp, 1.0f - rp, 1.0f - rp};
}

BRMI_DEV float fuzz_dir_reflectance(const Luts& L, float r, float c) { return satq(lut_fuzz_ltc(L, r, c).z); }
BRMI_DEV float fuzz_incoming_reflected(const Luts& L, float w, float r, float NdotV) { return satq(satq(w) * fuzz_dir_reflectance(L, r, NdotV)); }
BRMI_DEV f3 to_local(const FuzzState& s, f3 d) { return f3{dot3(d, s.t), dot3(d, s.b), dot3(d, s.n)}; }
BRMI_DEV FuzzState make_fuzz_state(const Luts& L, const Frag& f) {
    FuzzState s;
    s.roughness = satq(f.fuzzRoughness); s.tint = satq3(f.fuzzColor); s.presence = satq(f.fuzzWeight);
    s.n = normalize3(f.normalWS);
    const f3 v = normalize3(f.viewWS);
    const f3 pv = v - s.n * dot3(v, s.n);
    if (dot3(pv, pv) > 1.0e-6f) s.t = normalize3(pv);
    else { const f3 helper = fabsf(s.n.z) < 0.999f ? f3{0.0f, 0.0f, 1.0f} : f3{0.0f, 1.0f, 0.0f}; s.t = normalize3(cross3(helper, s.n)); }
    s.b = cross3(s.n, s.t);
    s.viewDirLocal = to_local(s, v);
    s.viewReflected = fuzz_incoming_reflected(L, s.presence, s.roughness, s.viewDirLocal.z);
    float phi = atan2f(s.viewDirLocal.y, s.viewDirLocal.x);
    if (phi < 0.0f) phi += 2.0f * PI_F;
    const float ang = -phi;
    s.sa = sinf(ang); s.ca = cosf(ang);
    s.ltcView = lut_fuzz_ltc(L, s.roughness, s.viewDirLocal.z);
    return s;
}
// `ll` = the light direction in the fuzz frame (to_local(s, normalize(lightDir)), which the caller has already)
BRMI_DEV f3 fuzz_sheen(const FuzzState& s, f3 ll) {
    if (s.viewDirLocal.z <= 0.0f || ll.z <= 0.0f) return f3{0.0f, 0.0f, 0.0f};
    const float sa = s.sa, ca = s.ca;
    const f3 axis{0.0f, 0.0f, 1.0f};
    const f3 ls = ll * ca + axis * dot3(ll, axis) * (1.0f - ca) + sa * cross3(axis, ll);
    const f3 ltc = s.ltcView;
    const float aInv = ltc.x, bInv = ltc.y;
    f3 wo{aInv * ls.x + bInv * ls.z, aInv * ls.y, ls.z};
    const float len = length3(wo);
    float e = 0.0f;
    if (len > 0.0f) {
        wo = wo / len;
        const float det = aInv * aInv;
        const float jac = det / max2(len * len * len, 1.0e-6f);
        e = satq(wo.z) * (1.0f / PI_F) * jac;
    }
    return s.presence * ltc.z * s.tint * e;
}
BRMI_DEV float fon_dir_albedo(float mu, float roughness) {
    const float m = satq(mu), mc = 1.0f - m;
    const float g1 = 0.0571085289f, g2 = 0.491881867f, g3 = -0.332181442f, g4 = 0.0714429953f;
    const float gOverPi = mc * (g1 + mc * (g2 + mc * (g3 + mc * g4)));
    return qdiv(1.0f + roughness * gOverPi, 1.0f + fon_a() * roughness);
}
BRMI_DEV f3 diffuse_eon(f3 albedo, float rough, float NdotV, float NdotL, float VdotL) {
    const float muIn = satq(NdotV), muOut = satq(NdotL);
    const float s = VdotL - muIn * muOut;
    const float sOverT = s > 0.0f ? s / max2(max2(muIn, muOut), 1.0e-4f) : s;
    const float A = 1.0f / (1.0f + fon_a() * rough);
    const f3 single = albedo * (1.0f / PI_F) * A * (1.0f + rough * sOverT);
    const float EOut = fon_dir_albedo(muOut, rough), EIn = fon_dir_albedo(muIn, rough);
    const float avgE = A * (1.0f + fon_b() * rough);
    const f3 msAlbedo = (albedo * albedo) * avgE / max3v(f3{1.0f, 1.0f, 1.0f} - albedo * (1.0f - avgE), f3{1.0e-4f, 1.0e-4f, 1.0e-4f});
    const float k = max2(1.0e-4f, 1.0f - EOut) * max2(1.0e-4f, 1.0f - EIn) / max2(1.0e-4f, 1.0f - avgE);
    const f3 multi = (msAlbedo * (1.0f / PI_F)) * f3{k, k, k};
    return single + multi;
}

// The shading pass's record of a light (k_frame_constants), four float4 indexed by the position in the active-light list.
//   [0] point / spot: world position, maxRange       directional: lightToFrag (= -direction), -1
//   [1] attenuation polynomial, conservative upper bound of maxRange^2
//   [2] colour x intensity, cos(inner)        [3] spot: normalize(direction), cos(outer); any other light: 0, -2
struct ShadeLightRecord { float r0[4], r1[4], r2[4], r3[4]; };   // the same record as plain words (scalar loads)
#ifndef BRMI_SHADE_METAL_STASH
#define BRMI_SHADE_METAL_STASH 1      // the stand-alone variant of k_shade<0> parks the metal lobe's inputs in LDS
#endif

struct ShadeArgs {
    ShadeTables tables;
    const brmi_per_frame* perFrame; const brmi_camera* cameras; uint32_t openpbrMaterialCount; const float* lutFuzzLTC;
    const float* depth; const float4* normals; const uint32_t* albedo; const unsigned long long* coat; const unsigned long long* emissive;
    const unsigned long long* fuzz; const uint32_t* metallicRoughness;
    const float4* shadeLights; const uint2* clusterList; const uint32_t* listEntries; const float4* listRecords;
    unsigned long long* hdr;
    uint32_t W, H, tilesX, bandY0, bandY1; uint64_t firstPixel, pixelCount;
    uint32_t enablePunctual, clustered;
    const float* lutF;   // expanded tables: odE[32768] odAvg[1024] imE[1024] imAvg[32] unorm8[256]
    const MatConst* matConst;
    uint32_t sceneHasCoat;                  // 0: no OpenPBR record has a coat weight > 0, the coat plane's weight is 0 everywhere
    const ShadeRows* shadeRows; const ShadeAverages* shadeAvgs;     // (OpenPBR material, roughness code) -> folded table rows and averages (k_frame_constants)
    const GgxQuad* ggxQuads;                                        // roughness code -> the GGX albedo fit as quadratics in N.V
    uint32_t* counters; uint32_t* deferred;   // pixels (band-relative tiled index) left to the general kernel
    // The deferred pixels go to 64 striped lists (tile t appends to stripe (t / 64) % 64, so no stripe can exceed its share): one
    // list with one counter would take an atomic with return per tile on a single address (~90 per microsecond on MI355X;
    // 130 k tiles = 1.4 ms when most pixels carry coat or fuzz).
    uint32_t deferredWord, nextDeferredWord;   // word inside a stripe: this call's list length / the next call's (cleared here)
    uint32_t stripeCapacity;
};

BRMI_DEV float half_at(unsigned long long v, int k) { return f16_bits_to_f32((uint32_t)(v >> (16 * k)) & 0xFFFFu); }
BRMI_DEV float bcast(float v, uint32_t lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)lane)); }

// Frame constants of the shading pass (uniform; evaluated by every lane like the shader does).
struct ShadeFrame {
    Luts L; uint32_t gx, gy, gz, nearSlices, numLights;
    float zNear, zFar, zSplit, resX, resY, tsx, tsy, logStart, logEnd, om5, om6;
    float nearScale, farScale, log2Near;     // fast estimate of the cluster slice (corrected against the exact slice starts)
};
BRMI_DEV ShadeFrame make_shade_frame(const ShadeArgs& a) {
    const brmi_per_frame* pf = a.perFrame;
    const brmi_camera* cam = a.cameras + pf->mainCameraIndex;
    ShadeFrame k;
    k.L = Luts{a.lutF, a.lutF + 32768, a.lutF + 32768 + 1024, a.lutF + 32768 + 2048, a.lutFuzzLTC, a.lutF + 32768 + 2048 + 32};
    k.gx = pf->lightClusterGridSizeX; k.gy = pf->lightClusterGridSizeY; k.gz = pf->lightClusterGridSizeZ;
    k.zNear = uni(cam->zNear); k.zFar = uni(cam->zFar); k.zSplit = uni(pf->clusterZSplitDepth);
    k.resX = uni((float)pf->screenResX); k.resY = uni((float)pf->screenResY);
    k.tsx = k.resX / (float)k.gx; k.tsy = k.resY / (float)k.gy;
    k.tsx = uni(k.tsx); k.tsy = uni(k.tsy);
    k.logStart = uni(logf(k.zSplit / k.zNear)); k.logEnd = uni(logf(k.zFar / k.zNear));      // only feed the slice ESTIMATE below (the exact slice starts come from the table)
    k.nearSlices = pf->nearClusterCount; k.numLights = pf->numLights;
    const float om = 1.0f - 1.0f / 7.0f;
    k.om5 = uni(powf(om, 5.0f)); k.om6 = uni(powf(om, 6.0f));
    k.nearScale = uni((float)k.nearSlices / (k.zSplit - k.zNear));
    k.farScale = uni((float)(k.gz - k.nearSlices) / (k.logEnd - k.logStart));
    k.log2Near = uni(log2f(k.zNear));
    return k;
}

// the G-buffer words of one pixel, as stored
struct RawPixel { float d; float4 ns; uint32_t al, mr; unsigned long long cs, es, fs; AxisEntry ax, ay; };   // + the column / row entries of the shading tables
BRMI_DEV RawPixel load_raw_pixel(const ShadeArgs& a, uint64_t i, uint32_t px, uint32_t py) {
    RawPixel r;
    r.ax = a.tables.x[px]; r.ay = a.tables.y[py];
    r.d = a.depth[i]; r.ns = a.normals[i]; r.al = a.albedo[i]; r.mr = a.metallicRoughness[i]; r.cs = a.coat[i]; r.es = a.emissive[i]; r.fs = a.fuzz[i];
    return r;
}
BRMI_DEV RawPixel empty_raw_pixel() { RawPixel r{}; r.d = as_f32(BRMI_DEPTH_EMPTY_BITS); return r; }

// what the specialised kernel keeps in flight for the next tile: coat / fuzz words reduced to the coat weight
// `tileBase` (first pixel of the wave's tile) is wave-uniform: plane base + tile offset is scalar, the lane index is the only vector part
#ifndef BRMI_SHADE_PREFETCH_AXIS
#define BRMI_SHADE_PREFETCH_AXIS 1
#endif
BRMI_DEV RawPixel load_raw_pixel_plain(const ShadeArgs& a, uint64_t tileBase, uint32_t lane, uint32_t px, uint32_t py) {
    RawPixel r;
    if (BRMI_SHADE_PREFETCH_AXIS) { r.ax = a.tables.x[px]; r.ay = a.tables.y[py]; }
    else { r.ax.tile = px; r.ay.tile = py; }
    // read once: streaming loads
    const float* dp = a.depth + tileBase; const float4* np = a.normals + tileBase; const uint32_t* ap = a.albedo + tileBase; const uint32_t* mp = a.metallicRoughness + tileBase;
    const unsigned long long* ep = a.emissive + tileBase; const uint16_t* cp = reinterpret_cast<const uint16_t*>(a.coat + tileBase);
    r.d = __builtin_nontemporal_load(&dp[lane]);
    r.ns = make_float4(__builtin_nontemporal_load(&np[lane].x), __builtin_nontemporal_load(&np[lane].y), __builtin_nontemporal_load(&np[lane].z), __builtin_nontemporal_load(&np[lane].w));
    r.al = __builtin_nontemporal_load(&ap[lane]); r.mr = __builtin_nontemporal_load(&mp[lane]); r.es = __builtin_nontemporal_load(&ep[lane]);
    // the coat plane is only looked at for its weight, and only when some material of the scene has a coat at all (brmi_set_scene)
    r.cs = a.sceneHasCoat ? (unsigned long long)__builtin_nontemporal_load(&cp[lane * 4u + 3u]) << 48 : 0ull; r.fs = 0ull;   // coat weight only (a plain pixel has no other coat / fuzz input)
    return r;
}

// Per-pixel part of calculateLightContributionPBR / EvaluateOpenPBRBaseLayerDirect.  The reference
// re-derives all of this for every light; nothing here depends on the light, so it is evaluated once
// per pixel with the same operations in the same order (bit-identical operands for the light loop).
struct PixelCtx {
    BaseState base;
    float NoV;
    CoatState coat; FuzzState fuzz;
    f3 coatIn, coatComp;
    float cachedView, mView, invMAvg;
    f3 dielComp;
    float f90Diel, f90Metal;
    f3 eonSinglePre, eonMsPre; float eonEInOverDen, fonA, fonK0, fonK1, fonK2, fonK3;
    const float2* odRow; const float2* imRow;   // folded rows of the pixel's (material, roughness code): {value, step to the next} pairs
    OdPrep coatOd;              // GENERAL: rows of the coat's (ior, roughness)
};

// fon_dir_albedo with the material's diffuse roughness folded into the coefficients (MatConst)
BRMI_DEV float fon_dir_albedo_folded(float mu, float fonA, float k0, float k1, float k2, float k3) {
    BRMI_FP_FAST
    const float mc = 1.0f - satq(mu);
    return fonA + mc * (k0 + mc * (k1 + mc * (k2 + mc * k3)));
}
template <int MODE>
BRMI_DEV PixelCtx make_pixel_ctx(const Luts& L, const Frag& f, const ShadeRows* rows, const ShadeAverages avg, const MatConst& mc, const GgxQuad* quads, uint32_t roughCode, uint32_t coatCode) {
    PixelCtx c;
    c.base = make_base_state(f);
    c.NoV = satq(dot3(f.normalWS, f.viewWS));
    // MODE bit 0: the pixel has a coat, bit 1: it has fuzz.  A layer that is absent has factors of exactly 1 / 0, so the
    // variants without it skip its terms without changing a bit of the result.
    if (MODE & 1) {
        c.coat = make_coat_state(c.base, f);
        c.coatIn = coat_scale_incoming(L, c.coat, c.NoV);
        c.coatOd = prep_coat_od(L, c.coat);
        c.coatComp = ggx_energy_compensation_q(quads[coatCode], c.NoV, f.coatF0);
    }
    if (MODE & 2) c.fuzz = make_fuzz_state(L, f);
    const BaseState& b = c.base;
    c.odRow = rows->od; c.imRow = rows->im;
    const float viewComp = sample_folded_row(c.odRow, c.NoV);
    c.cachedView = max2(0.0f, viewComp * avg.invAvgComp);
    c.mView = sample_folded_row(c.imRow, c.NoV);
    c.invMAvg = avg.invMAvgClamped;
    c.dielComp = ggx_energy_compensation_q(quads[roughCode], c.NoV, b.dielectricSpecularF0) * b.dielectricSpecularWeight;      // (the lobe's weight folded in: light_contribution)
    const float tmp = 50.0f * 0.33f;
    c.f90Diel = mc.f90Diel;
    c.f90Metal = satq(dot3(b.metalSpecularF0, f3{tmp, tmp, tmp}));
    {   // OpenPBRDiffuseEON, view-only factors; what only depends on the material's diffuse roughness comes from MatConst
        BRMI_FP_FAST
        c.fonA = mc.fonA; c.fonK0 = mc.fonK[0]; c.fonK1 = mc.fonK[1]; c.fonK2 = mc.fonK[2]; c.fonK3 = mc.fonK[3];
        c.eonSinglePre = b.diffuseColor * mc.eonSingleScale;
        const float EIn = fon_dir_albedo_folded(c.NoV, c.fonA, c.fonK0, c.fonK1, c.fonK2, c.fonK3);
        const f3 msAlbedo = qdiv3((b.diffuseColor * b.diffuseColor) * mc.eonAvgE, max3v(f3{1.0f, 1.0f, 1.0f} - b.diffuseColor * mc.eonOneMinusAvgE, f3{1.0e-4f, 1.0e-4f, 1.0e-4f}));
        c.eonMsPre = msAlbedo * (1.0f / PI_F);
        c.eonEInOverDen = max2(1.0e-4f, 1.0f - EIn) * mc.eonInvDen;
    }
    return c;
}

// `h` = normalize(L + V), NoH, LoH come from the caller (correctly rounded, contraction off: 1 - NoH^2 amplifies their error at low
// roughness); everything in here holds the HDR tolerance and may fuse.
// a * s + b per component as fused multiply-adds.  The f3 operators of brmi_device.h are compiled without contraction (the direction vectors need
// them that way), and a `clang fp contract(fast)` block does not reach into them: every `b + a * s` written with the operators inside the BRDF
// algebra was a multiply and an add (round 5: ~25 instructions per light evaluation).
BRMI_DEV f3 fma3(f3 a, float s, f3 b) { return f3{__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)}; }
BRMI_DEV f3 fma3v(f3 a, f3 s, f3 b) { return f3{__builtin_fmaf(a.x, s.x, b.x), __builtin_fmaf(a.y, s.y, b.y), __builtin_fmaf(a.z, s.z, b.z)}; }
// ---- round 5: direction vectors within an ulp instead of correctly rounded, where the surface is rough enough not to notice.
// N, V, L, H are kept correctly rounded because the GGX term amplifies their error: D ~ 1 / (1 - NoH^2 + (NoH a)^2)^2, so an error d of NoH moves D by
// ~4 d / a^2.  With the hardware reciprocal square root and one Newton step (and fused dot products) d ~ 3e-7; at alpha >= 0.1 (perceptual roughness >= 0.32)
// that is <= 1.2e-4 of D, a quarter of an fp16 ulp (4.9e-4 .. 9.8e-4) at the very peak of a highlight -- inside the north star's 1-ulp bar by construction, for
// every term of the sum.  A tile takes this path when ALL its live pixels are that rough (wave-uniform branch); smoother tiles keep the exact forms.
// What stays exact on both paths: N and V (the bent normal cancels to ~1e-4 where the stored normal faces away, and N.V, N.L scale terms that vanish with them),
// and L wherever |N.L| < SHADE_FAST_MIN_NOL.  Fast: L elsewhere, H, and the dot products N.H, L.H, V.L -- two normalisations and three dot products per light evaluation.
#ifndef BRMI_SHADE_FAST_DIRECTIONS
#define BRMI_SHADE_FAST_DIRECTIONS 1
#endif
struct FastDirections { static constexpr bool value = true; };
struct ExactDirections { static constexpr bool value = false; };
constexpr float SHADE_FAST_ALPHA = 0.1f, SHADE_FAST_MIN_NOL = 4.0e-3f;
BRMI_DEV float dot3f(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
BRMI_DEV float rsq_nr(float x) { const float r = __builtin_amdgcn_rsqf(x); return r * __builtin_fmaf((-0.5f * x) * r, r, 1.5f); }      // one Newton step on v_rsq_f32
BRMI_DEV f3 normalize3_fast(f3 a) { return a * rsq_nr(dot3f(a, a)); }
template <int MODE, int STASH>
BRMI_DEV f3 light_contribution(const Luts& L, const Frag& f, const PixelCtx& c, const float* stash, f3 lightToFrag, float NoL, float NoH, float LoH, float VdotL, float D, f3 lightColorIntensity, float attenuation, float spotAtt, f3 acc) {
    BRMI_FP_FAST
    const BaseState& base = c.base;
    const float NoV = c.NoV;
    // diffuse: EON x dielectric energy compensation
    const float lightComp = sample_folded_row(c.odRow, NoL);
    const float diffuseEnergyComp = max2(0.0f, c.cachedView * lightComp);
    f3 diffuse;
    {
        const float rough = base.baseDiffuseRoughness;
        const float muIn = NoV, muOut = NoL;                  // (both saturated by the caller)
        const float sv = VdotL - muIn * muOut;
        const float sOverT = sv > 0.0f ? qdiv(sv, max2(max2(muIn, muOut), 1.0e-4f)) : sv;
        const float singleScale = 1.0f + rough * sOverT;
        // STASH > 9: the five coefficients of the diffuse albedo fit wait in LDS too (read once per light evaluation)
        const float EOut = STASH > 9 ? fon_dir_albedo_folded(muOut, stash[12 * 256], stash[13 * 256], stash[14 * 256], stash[15 * 256], stash[16 * 256])
                                     : fon_dir_albedo_folded(muOut, c.fonA, c.fonK0, c.fonK1, c.fonK2, c.fonK3);
        const float k = max2(1.0e-4f, 1.0f - EOut) * c.eonEInOverDen;
        diffuse = fma3(c.eonMsPre, k, c.eonSinglePre * singleScale) * diffuseEnergyComp;
    }
    // specular: one D*V for both lobes (same roughness), one Schlick power
    const float DV = D * v_smith_ggx(base.specularAlpha, NoV, NoL);
    const float pw = __builtin_amdgcn_exp2f(5.0f * __builtin_amdgcn_logf(1.0f - LoH));   // pow(1 - LoH, 5) = exp2(5 log2 x), as DXC lowers it
    const f3 Fd = fma3(f3{c.f90Diel, c.f90Diel, c.f90Diel} - base.dielectricSpecularF0, pw, base.dielectricSpecularF0);
    // diffuse + dielectric lobe in one go: c.dielComp carries the dielectric weight (make_pixel_ctx)
    f3 brdf = fma3v(Fd * DV, c.dielComp, diffuse);
    // A dielectric pixel (metal weight 0) has metalSpec = 0 * (finite) = +-0, and dielSpec + (+-0) = dielSpec up to the sign of a zero
    // that the sums into `lighting` (which starts at +0) cannot carry: the metal lobe -- table fetch, Fresnel, multiple-scatter term --
    // is skipped for it.  A non-finite D*V keeps the full expression (0 * inf is NaN, not 0).
    if (!(base.metalSpecularWeight == 0.0f && fabsf(DV) <= 3.4028234e38f)) {
        // STASH: the lobe's per-pixel inputs wait in the lane's LDS slots (shade_pixel parked them) -- ten registers that only metallic pixels
        // read.  For the variant that has the chip to itself: beside another frame's k_raster_bins (35 KB of LDS per workgroup) the 9 KB per
        // workgroup cost more residency than the registers buy (frame in flight 0.571 -> 0.623 ms).
        const f3 mF0 = STASH ? f3{stash[0 * 256], stash[1 * 256], stash[2 * 256]} : base.metalSpecularF0;
        const f3 mMs = STASH ? f3{stash[3 * 256], stash[4 * 256], stash[5 * 256]} : base.metalMultipleScatterScale;
        const float f90Metal = STASH > 6 ? stash[6 * 256] : c.f90Metal, mView = STASH > 6 ? stash[7 * 256] : c.mView, invMAvg = STASH > 6 ? stash[8 * 256] : c.invMAvg;
        const float2* imRow = STASH ? c.odRow + 32 : c.imRow;                    // ShadeRows: od[32], im[32]
        const f3 Fm = fma3(f3{f90Metal, f90Metal, f90Metal} - mF0, pw, mF0);
        const float mLight = sample_folded_row(imRow, NoL);
        const float mTab = (mView * mLight) * invMAvg;
        const float mScale = min2(mTab, qrcp(max2(NoL, 1.0e-4f))) * (1.0f / PI_F);
        brdf = fma3(fma3(mMs, mScale, Fm * DV), base.metalSpecularWeight, brdf);
    }
    if (MODE != 0) {
        float fuzzScale = 1.0f;
        f3 fuzzFr{0.0f, 0.0f, 0.0f};
        if (MODE & 2) {
            const f3 llocal = to_local(c.fuzz, normalize3(lightToFrag));
            const float fuzzOut = (llocal.z <= 0.0f) ? 0.0f : satq(c.fuzz.presence * fuzz_dir_reflectance(L, c.fuzz.roughness, llocal.z));
            fuzzScale = (1.0f - c.fuzz.viewReflected) * (1.0f - fuzzOut);
            fuzzFr = fuzz_sheen(c.fuzz, llocal);
        }
        f3 baseScale{1.0f, 1.0f, 1.0f}, coatFr{0.0f, 0.0f, 0.0f};
        if (MODE & 1) {
            baseScale = c.coatIn * coat_scale_outgoing_prepared(L, c.coat, c.coatOd, NoL);
            if (c.coat.presence > 0.0f) {
                coatFr = specular_lobe(f.coatRoughness, f.coatF0, NoV, NoL, NoH, LoH);
                coatFr = coatFr * (c.coatComp * c.coat.presence);
            }
        }
        const f3 baseAtt = f3{fuzzScale, fuzzScale, fuzzScale} * baseScale;
        brdf = brdf * baseAtt + coatFr * f3{fuzzScale, fuzzScale, fuzzScale} + fuzzFr;
    }
    return fma3v(brdf, lightColorIntensity * (attenuation * spotAtt * NoL), acc);      // the pixel's sum so far + this light
}

// One pixel per lane of DeferredCSMain (deferred.hlsl:11-106).  Every lane of the wave runs through here together -- `live` lanes
// shade, the others (no geometry, outside the band, another material class) only lend a hand where the wave works as a team:
//   * the lights of a cluster are STAGED one per lane (a lane loads the 64 B record of one light of the list: all records of a
//     cluster arrive in one memory round trip, whatever the list length) and then broadcast light by light with v_readlane, so the
//     light loop has no memory access of its own (the per-light chain page -> index -> record was five dependent scalar loads);
//     the first cluster's records are requested as soon as the pixel's cluster is known, before the bulk of the per-pixel work;
//   * the broadcast is two-stage: position and range first, then a conservative reject on squared distance and facing (no lane of
//     the tile can receive anything from the light: skip it before the correctly rounded sqrt / divide and the rest of the record).
// MODE = class of pixel this instantiation shades (0 plain, 1 coat, 2 fuzz, 3 both: a layer that is absent has factors of exactly
// 1 / 0, so the plain variant needs half the registers -- the same idea as the reference's per-material-permutation pixel lists).
// Returns, for a live lane whose class is not MODE, that class (MODE 0 defers such pixels); 0 otherwise.
template <int MODE, int STASH = 0>      // STASH: floats of the metal lobe's inputs parked in LDS (0, 6 or 9)
BRMI_DEV uint32_t shade_pixel(const ShadeArgs& a, const ShadeFrame& k, const float* sliceStart, const float* unorm8, const float4* camK, const RawPixel& raw, bool live, uint64_t tileBase, uint32_t within) {
    const Luts& L = k.L;
    const uint32_t gx = k.gx, gy = k.gy, gz = k.gz, nearSlices = k.nearSlices;
    __shared__ float metalStash[STASH ? STASH : 1][STASH ? 256 : 1];                     // (blockDim.x == 256 in every kernel built from this)
    const float* stash = &metalStash[0][STASH ? threadIdx.x : 0u];
    live = live && as_u32(raw.d) != BRMI_DEPTH_EMPTY_BITS;
    Frag f; PixelCtx ctx; f3 posWS, posVS;        // only read by lanes that stay `live` (no initialiser: nothing to materialise for the others)
    uint32_t opaqueZero = 0u;
    asm volatile("" : "+v"(opaqueZero));            // the LDS reads of the camera constants stay inside this call (hoisted out of the tile loop they pin 27 registers)
    uint32_t ci = 0xFFFFFFFFu, cls = 0u;
    const uint32_t al = raw.al, mr = raw.mr;
    const unsigned long long cs = raw.cs, es = raw.es, fs = raw.fs;
    // ---- class of the pixel, its position and its light cluster: what the light staging waits for
    if (live) {
        f.coatWeight = satq(half_at(cs, 3)); f.fuzzWeight = satq(unorm8[mr >> 24]);
        cls = (f.coatWeight != 0.0f ? 1u : 0u) | (f.fuzzWeight != 0.0f ? 2u : 0u);
        if (cls != (uint32_t)MODE) live = false;             // MODE 0 defers it; the layered variants only see their own class
        else {
            cls = 0u;
            // the camera matrices live in LDS (wave-uniform reads, broadcast): as scalars they took 27 SGPRs for the whole kernel and pushed
            // others into VGPR lanes; here they are ordinary operands for a few dozen instructions
            m4 invProj, viewInv;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 p4 = camK[r + opaqueZero], v4 = camK[4 + r + opaqueZero];
                invProj.m[r][0] = p4.x; invProj.m[r][1] = p4.y; invProj.m[r][2] = p4.z; invProj.m[r][3] = p4.w;
                viewInv.m[r][0] = v4.x; viewInv.m[r][1] = v4.y; viewInv.m[r][2] = v4.z; viewInv.m[r][3] = v4.w;
            }
            AxisEntry ax = raw.ax, ay = raw.ay;
            if (MODE == 0 && !BRMI_SHADE_PREFETCH_AXIS) { ax = a.tables.x[raw.ax.tile]; ay = a.tables.y[raw.ay.tile]; }
            float uvx = ax.uv, uvy = ay.uv;
            uvy = 1.0f - uvy;
            const f4 clipPos{uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f, 1.0f, 1.0f};
            const f4 viewPosH = mul_vm(clipPos, invProj);
            posVS = xyz(viewPosH) * raw.d;
            posWS = xyz(mul_point(posVS, viewInv));
            if (a.clustered) {
                const float z = fabsf(posVS.z);
                // slice: a hardware-log estimate (within one slice of the shader's formula), corrected against the exact first
                // depth of that slice and of the next one -- the value of the formula without its two divisions and logf
                const float est = z < k.zSplit ? (z - k.zNear) * k.nearScale
                                               : (float)nearSlices + ((__builtin_amdgcn_logf(z) - k.log2Near) * 0.69314718f - k.logStart) * k.farScale;
                int e = (int)min2(max2(est, 0.0f), (float)gz);
                const float lo = sliceStart[e], hi = sliceStart[e + 1];
                e += (z >= hi) ? 1 : 0; e -= (z < lo) ? 1 : 0;
                ci = (uint32_t)((float)ax.tile + (float)ay.tile * (float)gx + (float)(uint32_t)e * (float)gx * (float)gy);
                if (ci >= gx * gy * gz) ci = 0xFFFFFFFFu;
            } else ci = 0u;
        }
    }
    uint64_t pending = a.enablePunctual ? __ballot(live && ci != 0xFFFFFFFFu) : 0ull;
    uint32_t uci = 0u, listBase = 0u, listCount = 0u;
    // every live pixel of the tile at least SHADE_FAST_ALPHA rough (plain pixels only: a coat has a roughness of its own): the tile's direction vectors take the fast forms
    bool fastTile = false;
    if (MODE == 0 && BRMI_SHADE_FAST_DIRECTIONS) {
        const float prq = clampf(unorm8[(mr >> 8) & 0xFFu], BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
        fastTile = !__any(live && prq * prq < SHADE_FAST_ALPHA);
    }
    // ---- GetFragmentInfoScreenSpace + PopulateFragmentInfoFromOpenPBR + the light-independent part of the BRDF
    if (live) {
        const float4 cp = camK[8 + opaqueZero];
        const f3 toEye = f3{cp.x, cp.y, cp.z} - posWS;
        // (V stays correctly rounded on both paths: the bent normal below is nrm + k V, which cancels to a vector of length ~1e-4 where the stored normal points
        // away from the eye, and an ulp of V is then a part in a thousand of N -- measured: 205 fp16 ulps on such pixels with a fast V)
        const f3 viewDir = normalize3_q(toEye);
        f.posWS = posWS; f.viewWS = viewDir;
        const float4 ns = raw.ns;
        const f3 nrm{ns.x, ns.y, ns.z};
        // code / 255 from the LDS copy of the table: seven reads per pixel that do not go through the vector-memory path
        const f3 baseColor{unorm8[al & 0xFFu], unorm8[(al >> 8) & 0xFFu], unorm8[(al >> 16) & 0xFFu]};
        const float metal = unorm8[mr & 0xFFu], pr = unorm8[(mr >> 8) & 0xFFu], coatR = unorm8[(mr >> 16) & 0xFFu];
        const float prc = clampf(pr, BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
        f.roughness = prc * prc;
        const float NdotVraw = dot3(nrm, viewDir);
        f.normalWS = normalize3_q(nrm + max2(0.0f, -NdotVraw + BRMI_MIN_N_DOT_V) * viewDir);      // (N too: N.L and N.V carry its absolute error to terms that vanish with them)
        f.NdotV = max2(BRMI_MIN_N_DOT_V, NdotVraw);
        uint32_t opIndex = (uint32_t)(ns.w + 0.5f);
        if (opIndex >= a.openpbrMaterialCount) opIndex = 0;
#ifdef BRMI_ABLATE_MATCONST
        const MatConst mc = a.matConst[0];      // (experiment: one wave-uniform record instead of a gather per pixel; wrong image for mixed scenes)
#else
        const MatConst mc = a.matConst[opIndex];
#endif
        const float baseWeight = mc.baseWeight, specularWeight = mc.specularWeight;
        const f3 specularColor{mc.specR, mc.specG, mc.specB};
        const f3 weightedBaseColor = satq3(baseColor * baseWeight);
        f.dielectricSpecularF0 = f3{mc.dielF0[0], mc.dielF0[1], mc.dielF0[2]};      // sat(specularColor * dielF0Scalar), per material (k_frame_constants)
        const float coatPR = clampf(coatR, BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
        f.dielectricSpecularWeight = satq(1.0f - metal);
        f.metalSpecularWeight = satq(metal * specularWeight);
        f.metalSpecularF0 = f3{0.0f, 0.0f, 0.0f}; f.metalAverageFresnel = f3{0.0f, 0.0f, 0.0f};
        // every use of the metal lobe's inputs is multiplied by the metal weight: a dielectric pixel (weight exactly 0) skips them
        if (f.metalSpecularWeight != 0.0f) {
            f.metalSpecularF0 = satq3(weightedBaseColor * specularColor);
            const f3 safeF0 = satq3(weightedBaseColor), wmF0 = f3{1.0f, 1.0f, 1.0f} - safeF0;
            const float cosMax = 1.0f / 7.0f;
            const f3 wmF0b = f3{1.0f, 1.0f, 1.0f} - satq3(safeF0), wmTint = f3{1.0f, 1.0f, 1.0f} - satq3(specularColor);
            const f3 num = (satq3(safeF0) + wmF0b * k.om5) * wmTint;
            const float den = cosMax * k.om6;
            const f3 b = num * qrcp(max2(den, 1.0e-6f));
            f.metalAverageFresnel = satq3(safeF0 + wmF0 * (1.0f / 21.0f) - b * (1.0f / 126.0f));
        }
        f.albedo = weightedBaseColor;
        f.emissive = f3{half_at(es, 0), half_at(es, 1), half_at(es, 2)};
        f.coatColor = satq3(f3{half_at(cs, 0), half_at(cs, 1), half_at(cs, 2)});
        f.coatRoughness = coatPR * coatPR;
        f.coatF0 = satq3(f.coatColor * mc.coatF0Scalar);
        f.coatIor = mc.coatIor; f.coatDarkening = mc.coatDarkening;
        f.fuzzColor = satq3(f3{half_at(fs, 0), half_at(fs, 1), half_at(fs, 2)}); f.fuzzRoughness = satq(half_at(fs, 3));
        f.baseDiffuseRoughness = mc.baseDiffuseRoughness;
        f.specularAlpha = f.roughness; f.weightedSpecularIor = mc.weightedSpecularIor;
        f.diffuseColor = weightedBaseColor * (1.0f - metal);
        const uint32_t entry = opIndex * 256u + ((mr >> 8) & 0xFFu);
        ctx = make_pixel_ctx<MODE>(L, f, a.shadeRows + entry, a.shadeAvgs[entry], mc, a.ggxQuads, (mr >> 8) & 0xFFu, (mr >> 16) & 0xFFu);
        if (STASH && f.metalSpecularWeight != 0.0f) {
            metalStash[0][threadIdx.x] = ctx.base.metalSpecularF0.x; metalStash[1][threadIdx.x] = ctx.base.metalSpecularF0.y; metalStash[2][threadIdx.x] = ctx.base.metalSpecularF0.z;
            metalStash[3][threadIdx.x] = ctx.base.metalMultipleScatterScale.x; metalStash[4][threadIdx.x] = ctx.base.metalMultipleScatterScale.y; metalStash[5][threadIdx.x] = ctx.base.metalMultipleScatterScale.z;
            if (STASH > 6) { metalStash[6 % (STASH ? STASH : 1)][threadIdx.x] = ctx.f90Metal; metalStash[7 % (STASH ? STASH : 1)][threadIdx.x] = ctx.mView; metalStash[8 % (STASH ? STASH : 1)][threadIdx.x] = ctx.invMAvg; }
        }
        if (STASH > 9) {      // the emissive term (read once, after the lights) and the diffuse fit's coefficients (once per light evaluation): eight registers
            constexpr int S = STASH ? STASH : 1;
            metalStash[9 % S][threadIdx.x] = f.emissive.x; metalStash[10 % S][threadIdx.x] = f.emissive.y; metalStash[11 % S][threadIdx.x] = f.emissive.z;
            metalStash[12 % S][threadIdx.x] = ctx.fonA; metalStash[13 % S][threadIdx.x] = ctx.fonK0; metalStash[14 % S][threadIdx.x] = ctx.fonK1; metalStash[15 % S][threadIdx.x] = ctx.fonK2; metalStash[16 % S][threadIdx.x] = ctx.fonK3;
        }
    }
    f3 lighting{0.0f, 0.0f, 0.0f};
    // Waterfall over the distinct clusters of the wave (an 8x8 tile usually sits in one).  The loop and the staging run with every lane
    // of the wave; only the light loop proper is restricted to the lanes of the cluster.  The lane set comes from a ballot and `uci`
    // depends on the loop-carried mask, so neither can be replaced by the per-lane `ci`.
    while (pending != 0ull) {
        uci = (uint32_t)__builtin_amdgcn_readlane((int)ci, (int)((uint32_t)__ffsll((unsigned long long)pending) - 1u));
        listBase = 0u; listCount = k.numLights;
        if (a.clustered) { const auto* cl = kconst(reinterpret_cast<const uint32_t*>(a.clusterList)) + 2u * (size_t)uci; listBase = cl[0]; listCount = cl[1]; }
        const uint64_t same = __ballot(live && ci == uci);
        pending &= ~same;
        const bool mine = (same >> lane_id()) & 1ull;
        // The records of a cluster's lights lie in list order (k_lc_fill), so light q of the list is ONE wave-uniform 64 B record: a scalar
        // load brings it into SGPRs -- no staging registers (16 VGPRs) and no v_readlane per field.
#ifdef BRMI_ABLATE_LIGHTREC
        const float4* recBase = a.shadeLights;      // (experiment: every tile reads the head of ONE table -- scalar-cache hits; wrong image)
#else
        const float4* recBase = a.clustered ? a.listRecords + (size_t)listBase * 4u : a.shadeLights;
#endif
        // (two instantiations of the loop, chosen per tile: FAST = the direction vectors within an ulp, see SHADE_FAST_ALPHA)
        auto light_loop = [&](auto fastTag) {
            constexpr bool FAST = decltype(fastTag)::value;
            for (uint32_t q = 0; q < listCount; q++) {
                // one s_load_dwordx16 (as sixteen separate words the compiler fetched the record in four dependent pieces)
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                const f32x16 w = *reinterpret_cast<const __attribute__((address_space(4))) f32x16*>(kconst(recBase + (size_t)q * 4u));
                const ShadeLightRecord lr{{w[0], w[1], w[2], w[3]}, {w[4], w[5], w[6], w[7]}, {w[8], w[9], w[10], w[11]}, {w[12], w[13], w[14], w[15]}};
                const f3 lp{lr.r0[0], lr.r0[1], lr.r0[2]};
                f3 lightToFrag; float att = 1.0f, spot = 1.0f;
                if (lr.r0[3] < 0.0f) lightToFrag = lp;           // directional
                else {
                    // lighting.hlsli:614-617's `dist > maxRange` as d2 > T (k_frame_constants: the largest squared distance whose correctly rounded root is <= maxRange)
                    const f3 toL = lp - posWS;
                    const float d2 = dot3(toL, toL);
                    if (d2 > lr.r1[3]) continue;
                    float dist;
                    if (FAST) {
                        const float inv = rsq_nr(d2); lightToFrag = toL * inv; dist = d2 * inv;
                        // a light at grazing incidence: the term is proportional to N.L, whose ABSOLUTE error (~1.5e-7 with L an ulp off) is then a large part of it --
                        // 17 fp16 ulps on pixels that one such light alone leaves at 1e-4 (measured) -- so those lanes take the correctly rounded L after all
                        if (fabsf(dot3(f.normalWS, lightToFrag)) < SHADE_FAST_MIN_NOL) lightToFrag = normalize3_len(toL, d2, dist);
                    } else {
                        if (dot3(f.normalWS, toL) < -1.0e-5f * qsqrt(d2)) continue;      // facing away by more than any rounding of normalize() can undo: dropped before the exact root
                        lightToFrag = normalize3_len(toL, d2, dist);
                    }
                    att = qrcp((lr.r1[0] + lr.r1[1] * dist + lr.r1[2] * dist * dist) + 0.0001f);
                }
                const float NoL = satq(dot3(f.normalWS, lightToFrag));
                if (NoL == 0.0f) continue;
                const float outer = lr.r3[3];
                if (outer > -1.5f) {                                    // spot light
                    const f3 sd{lr.r3[0], lr.r3[1], lr.r3[2]};
                    const float inner = lr.r2[3];
                    const float cc = dot3(sd, normalize3_q(-lightToFrag));
                    if (!(cc > outer)) continue;
                    if (cc < inner) { const float t = satq((cc - outer) / (inner - outer)); spot = t * t * (3.0f - 2.0f * t); }
                }
                const f3 hv = lightToFrag + f.viewWS;
                const f3 h = FAST ? normalize3_fast(hv) : normalize3_q(hv);
                const float NoH = satq(FAST ? dot3f(f.normalWS, h) : dot3(f.normalWS, h)), LoH = satq(FAST ? dot3f(lightToFrag, h) : dot3(lightToFrag, h));
                const float VdotL = FAST ? dot3f(f.viewWS, lightToFrag) : dot3(f.viewWS, lightToFrag);
                const float D = d_ggx(ctx.base.specularAlpha, NoH);
                const f3 col{lr.r2[0], lr.r2[1], lr.r2[2]};
                lighting = light_contribution<MODE, STASH>(L, f, ctx, stash, lightToFrag, NoL, NoH, LoH, VdotL, D, col, att, spot, lighting);
            }
        };
        if (mine) { if (fastTile) light_loop(FastDirections{}); else light_loop(ExactDirections{}); }
    }
    if (live) {
        // EvaluateOpenPBREmissive
        if (MODE == 0) lighting = lighting + (STASH > 9 ? f3{stash[9 * 256], stash[10 * 256], stash[11 * 256]} : f.emissive);
        else {
            const float fuzzBase = (MODE & 2) ? 1.0f - fuzz_incoming_reflected(L, f.fuzzWeight, f.fuzzRoughness, f.NdotV) : 1.0f;
            const f3 coatT = (MODE & 1) ? coat_scale_incoming(L, ctx.coat, f.NdotV) : f3{1.0f, 1.0f, 1.0f};
            lighting = lighting + f.emissive * f3{fuzzBase, fuzzBase, fuzzBase} * coatT;
        }
        __builtin_nontemporal_store((unsigned long long)pack_half4(lighting.x, lighting.y, lighting.z, 1.0f), &(a.hdr + tileBase)[within]);
    }
    return cls;
}

// the specialised kernel must keep 3 waves per SIMD (<= 168 VGPRs); the general one is rare and may use the whole file

// what k_shade<> and the fused kernel keep in LDS: slice starts, code / 255 table, camera rows (blockDim.x == 256)
BRMI_DEV void shade_stage_lds(const ShadeArgs& a, const ShadeFrame& k, float* sliceStart, float* unormT, float4* camK) {
    if (threadIdx.x < 9u) {
        const brmi_camera* cam = a.cameras + a.perFrame->mainCameraIndex;
        camK[threadIdx.x] = threadIdx.x < 4u ? *reinterpret_cast<const float4*>(&cam->projectionInverse[threadIdx.x][0])
                          : threadIdx.x < 8u ? *reinterpret_cast<const float4*>(&cam->viewInverse[threadIdx.x - 4u][0]) : *reinterpret_cast<const float4*>(&cam->positionWorldSpace[0]);
    }
    unormT[threadIdx.x] = k.L.unorm8[threadIdx.x];
    if (threadIdx.x < 64) sliceStart[threadIdx.x] = threadIdx.x <= k.gz + 1u ? a.tables.sliceStart[threadIdx.x] : __uint_as_float(0x7F800000u);
    __syncthreads();
}
// layered pixels of tile t (relative to the band's first tile) go to the per-class lists the k_shade<1|2|3> launches walk (cls: 0 = shaded here)
BRMI_DEV void shade_defer(const ShadeArgs& a, uint32_t t, uint32_t cls, uint32_t lane) {
    const uint32_t stripe = (t >> 6) & (CNT_STRIPE_COUNT - 1u);     // wave-uniform; runs of 64 neighbouring tiles share a stripe (locality of the list)
    if (__any(cls != 0u)) {
        // one list per class (coat, fuzz, both) so that every layered variant walks a dense list
#pragma unroll
        for (uint32_t c = 1; c <= 3; c++) {
            const uint32_t slot = wave_append(&a.counters[CNT_STRIPES + stripe * CNT_STRIPE_WORDS + a.deferredWord + (c - 1u)], cls == c);
            if (cls == c) {
                if (slot < a.stripeCapacity) a.deferred[((size_t)(c - 1u) * CNT_STRIPE_COUNT + stripe) * a.stripeCapacity + slot] = (t << 6) | lane;
                else atomicAdd(&a.counters[CNT_DEFERRED_DROPPED], 1u);          // cannot happen by construction (a stripe holds its share of the band); counted like every other drop
            }
        }
    }
}

ShadeArgs shade_args_of(brmi_pass* p);      // host (brmi_light.hip): the arguments of this frame's shading calls

}  // namespace brmi
#endif
