// orc_light.cpp -- CPU restatement of light clustering (K9, K10) and clustered OpenPBR shading (K11).
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows:
//   clustering.hlsl:CSMain                   BR/shaders/clustering.hlsl:31-107
//   lightCulling.hlsl:CSMain                 BR/shaders/lightCulling.hlsl:40-126
//   DeferredCSMain                           BR/shaders/deferred.hlsl:11-106
//   GetFragmentInfoScreenSpace               BR/shaders/Include/utilities.hlsli:2639-2709
//   PopulateFragmentInfoFromOpenPBR          BR/shaders/Include/utilities.hlsli:2590-2637
//   lightFragment / ComputeClusterID         BR/shaders/Include/lighting.hlsli:166-196,391-661
//   getLightParametersForFragment            BR/shaders/Include/lighting.hlsli:81-113
//   calculateLightContributionPBR            BR/shaders/Include/lighting.hlsli:116-164
//   OpenPBR layers, EON diffuse, LUT lookups BR/shaders/Include/IBL.hlsli:94-672
//   GGX lobe, Schlick, energy compensation   BR/shaders/Include/PBR.hlsli:8-190
// Configuration restated: PSO_CLUSTERED_LIGHTING on, PSO_IMAGE_BASED_LIGHTING off, shadows off,
// GTAO off, punctual lights on (SURVEY.md section 7 hard part 4).
//
// Third-party arithmetic absent from the reference checkout: adobe/openpbr-bsdf (empty submodule,
// .gitmodules:19-21, no pinned SHA).  Its 32x32(x32) R16_UNORM energy tables and the 32x32 LTC
// table are INJECTED: oracle and kernels read the same caller-provided tables, bilinear filtering
// is done in software in fp32 (texel-centre convention of a linear-clamp sampler).
//
// One deliberate relocation: the 25 slice plane depths of the cluster grid use log()/exp()
// (clustering.hlsl:77-90), whose last-bit results differ between GPU and CPU math libraries and
// would make AABBs - hence light lists - irreproducible.  They are computed once on the host with
// logf/expf (orc_cluster_planes) and consumed by both sides.  The per-pixel slice lookup of ComputeClusterID
// (lighting.hlsli:166-196) evaluates its log() as the correctly rounded fp32 logarithm for the same reason.
#include <cmath>
#include <vector>

#include "orc_common.h"

namespace orc {

static const float PI = 3.1415926538f;
static const float MEDIUMP_FLT_MAX = 65504.0f;

struct Luts {
    const uint16_t* odE;   // [32][32][32] ior, alpha, cos
    const uint16_t* odAvg; // [32][32] ior, alpha
    const uint16_t* imE;   // [32][32] alpha, cos
    const uint16_t* imAvg; // [32] alpha
    const float* ltc;      // [32][32][4] rough, cos
};

static inline float texelU16(const uint16_t* t, uint32_t i) { return (float)t[i] / 65535.0f; }
// bilinear SampleLevel(linearClamp, uv, 0) on a W x H single-channel R16_UNORM table
static float sampleU16(const uint16_t* t, uint32_t W, uint32_t H, float u, float v) {
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0f = std::floor(x), y0f = std::floor(y);
    const float fx = x - x0f, fy = y - y0f;
    auto cl = [](float f, uint32_t n) { int i = (int)f; if (i < 0) i = 0; if (i > (int)n - 1) i = (int)n - 1; return (uint32_t)i; };
    const uint32_t x0 = cl(x0f, W), x1 = cl(x0f + 1.0f, W), y0 = cl(y0f, H), y1 = cl(y0f + 1.0f, H);
    const float a = lerp(texelU16(t, y0 * W + x0), texelU16(t, y0 * W + x1), fx);
    const float b = lerp(texelU16(t, y1 * W + x0), texelU16(t, y1 * W + x1), fx);
    return lerp(a, b, fy);
}
static float3 sampleLTC(const float* t, float u, float v) {
    const uint32_t W = 32, H = 32;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0f = std::floor(x), y0f = std::floor(y);
    const float fx = x - x0f, fy = y - y0f;
    auto cl = [](float f, uint32_t n) { int i = (int)f; if (i < 0) i = 0; if (i > (int)n - 1) i = (int)n - 1; return (uint32_t)i; };
    const uint32_t x0 = cl(x0f, W), x1 = cl(x0f + 1.0f, W), y0 = cl(y0f, H), y1 = cl(y0f + 1.0f, H);
    auto T = [&](uint32_t yy, uint32_t xx) { const float* p = t + ((size_t)yy * W + xx) * 4; return float3{p[0], p[1], p[2]}; };
    return lerp(lerp(T(y0, x0), T(y0, x1), fx), lerp(T(y1, x0), T(y1, x1), fx), fy);
}

// ---- PBR.hlsli ------------------------------------------------------------------------------
static float3 ggxDirAlbedoAnalytic(float NdotV, float alpha, float3 F0, float3 F90) {
    const float x = NdotV, y = alpha, x2 = x * x, y2 = y * y;
    const float c0[4] = {0.1003f, 0.9345f, 1.0f, 1.0f}, c1[4] = {-0.6303f, -2.323f, -1.765f, 0.2281f}, c2[4] = {9.748f, 2.229f, 8.263f, 15.94f},
                c3[4] = {-2.038f, -3.748f, 11.53f, -55.83f}, c4[4] = {29.34f, 1.424f, 28.96f, 13.08f}, c5[4] = {-8.245f, -0.7684f, -7.507f, 41.26f},
                c6[4] = {-26.44f, 1.436f, -36.11f, 54.9f}, c7[4] = {19.99f, 0.2913f, 15.86f, 300.2f}, c8[4] = {-5.448f, 0.6286f, 33.37f, -285.1f};
    float r[4];
    for (int i = 0; i < 4; i++)
        r[i] = c0[i] + c1[i] * x + c2[i] * y + c3[i] * x * y + c4[i] * x2 + c5[i] * y2 + c6[i] * x2 * y + c7[i] * x * y2 + c8[i] * x2 * y2;
    const float A = clampf(r[0] / r[2], 0.0f, 1.0f), B = clampf(r[1] / r[3], 0.0f, 1.0f);
    return F0 * A + F90 * B;
}
static float3 ggxEnergyCompensation(float NdotV, float alpha, float3 Fss) {
    const float Ess = ggxDirAlbedoAnalytic(NdotV, alpha, float3{1, 1, 1}, float3{1, 1, 1}).x;
    return float3{1.0f, 1.0f, 1.0f} + Fss * (1.0f - Ess) / Ess;
}
static float3 F_Schlick(float3 f0, float f90, float VoH) {
    const float p = std::pow(1.0f - VoH, 5.0f);
    return f0 + (float3{f90, f90, f90} - f0) * p;
}
static float V_SmithGGXCorrelated(float roughness, float NoV, float NoL) {
    const float a2 = roughness * roughness;
    const float lambdaV = NoL * std::sqrt((NoV - a2 * NoV) * NoV + a2);
    const float lambdaL = NoV * std::sqrt((NoL - a2 * NoL) * NoL + a2);
    const float v = 0.5f / (lambdaV + lambdaL);
    return fmin2(v, MEDIUMP_FLT_MAX);
}
static float D_GGX(float roughness, float NoH) {
    const float oneMinusNoHSquared = 1.0f - NoH * NoH;
    const float a = NoH * roughness;
    const float k = roughness / (oneMinusNoHSquared + a * a);
    const float d = k * k * (1.0f / PI);
    return fmin2(d, MEDIUMP_FLT_MAX);
}
static float3 specularLobe(float roughness, float3 f0, float NoV, float NoL, float NoH, float LoH) {
    const float D = D_GGX(roughness, NoH);
    const float V = V_SmithGGXCorrelated(roughness, NoV, NoL);
    const float tmp = 50.0f * 0.33f;
    const float f90 = saturate(dot(f0, float3{tmp, tmp, tmp}));
    const float3 F = F_Schlick(f0, f90, LoH);
    return (D * V) * F;
}
static inline float Fd_Lambert() { return 1.0f / PI; }

// ---- IBL.hlsli: OpenPBR ---------------------------------------------------------------------
static const float TABLE_SIZE = 32.0f, TABLE_SIZE_M1 = 31.0f, IOR_MAX = 2.5f, INV_IOR_MAX = 1.0f / 2.5f;
static const float FON_A = 0.5f - 2.0f / (3.0f * PI);
static const float FON_B = 2.0f / 3.0f - 28.0f / (15.0f * PI);

static float iorToF0(float ior) { const float s = fmax2(ior, 1.0f); const float f = (s - 1.0f) / (s + 1.0f); return f * f; }
static float iorToExactIndex(float ior) {
    const float safeIor = fmax2(ior, 1.0e-4f);
    const float half = 0.5f * TABLE_SIZE, halfM1 = half - 1.0f, inv = 1.0f / (IOR_MAX - 1.0f);
    if (safeIor < 1.0f) { const float invIor = 1.0f / safeIor; const float fr = (invIor - 1.0f) * inv; return halfM1 - fr * halfM1; }
    const float fr = (safeIor - 1.0f) * inv;
    return half + fr * halfM1;
}
static float alphaToExactIndex(float alpha) { return std::sqrt(saturate(alpha)) * TABLE_SIZE_M1; }
static float cosToExactIndex(float c) { return saturate(c) * TABLE_SIZE_M1; }
static float clampIndex(float e) { return clampf(e, 0.0f, TABLE_SIZE_M1); }
static float remapIndex(float e) { const float inv = 1.0f / TABLE_SIZE; const float mn = 0.5f * inv, mx = 1.0f - mn; return clampf(mn + e * inv, mn, mx); }
static float extrapolateBeyondIorMax(float tableValue, float ior) {
    if (ior > IOR_MAX || ior < INV_IOR_MAX) {
        const float f0Max = iorToF0(IOR_MAX);
        const float invRange = 1.0f / (1.0f - f0Max);
        const float f0 = iorToF0(fmax2(ior, 1.0e-4f));
        const float progress = (f0 - f0Max) * invRange;
        return (1.0f - progress) * tableValue;
    }
    return tableValue;
}
static float fresnelDielectric(float eta, float cosI) {
    const float c = saturate(cosI);
    if (std::fabs(eta - 1.0f) <= 1.0e-6f) return 0.0f;
    const float s2 = fmax2(0.0f, 1.0f - c * c);
    const float st2 = s2 / fmax2(eta * eta, 1.0e-6f);
    if (st2 >= 1.0f) return 1.0f;
    const float ct = std::sqrt(fmax2(0.0f, 1.0f - st2));
    const float eci = eta * c, ect = eta * ct;
    const float rs = (c - ect) / fmax2(c + ect, 1.0e-6f);
    const float rp = (ct - eci) / fmax2(ct + eci, 1.0e-6f);
    return 0.5f * (rs * rs + rp * rp);
}
static float lookUpOdAvg(const Luts& L, float ior, float alpha) {
    const float ei = clampIndex(iorToExactIndex(ior)), ea = clampIndex(alphaToExactIndex(alpha));
    return extrapolateBeyondIorMax(sampleU16(L.odAvg, 32, 32, remapIndex(ea), remapIndex(ei)), ior);
}
static float lookUpOdE(const Luts& L, float ior, float alpha, float cosT) {
    const float ei = clampIndex(iorToExactIndex(ior)), ea = clampIndex(alphaToExactIndex(alpha)), ec = clampIndex(cosToExactIndex(cosT));
    const int s0 = (int)std::floor(ei);
    const int s1 = (s0 + 1) < 31 ? (s0 + 1) : 31;
    const float st = ei - (float)s0;
    const float u = remapIndex(ec), v = remapIndex(ea);
    const float v0 = sampleU16(L.odE + (size_t)s0 * 1024, 32, 32, u, v), v1 = sampleU16(L.odE + (size_t)s1 * 1024, 32, 32, u, v);
    return extrapolateBeyondIorMax(lerp(v0, v1, st), ior);
}
static float lookUpImE(const Luts& L, float alpha, float cosT) {
    const float ea = clampIndex(alphaToExactIndex(alpha)), ec = clampIndex(cosToExactIndex(cosT));
    return sampleU16(L.imE, 32, 32, remapIndex(ec), remapIndex(ea));
}
static float lookUpImAvg(const Luts& L, float alpha) {
    const float ea = clampIndex(alphaToExactIndex(alpha));
    return sampleU16(L.imAvg, 32, 1, remapIndex(ea), 0.5f);
}
static float3 lookUpFuzzLTC(const Luts& L, float roughness, float cosT) {
    const float u = saturate(cosT) * (31.0f / 32.0f) + 0.5f / 32.0f, v = saturate(roughness) * (31.0f / 32.0f) + 0.5f / 32.0f;
    return sampleLTC(L.ltc, u, v);
}
static float averageFresnel(float eta) {
    const float s = fmax2(eta, 1.0e-4f);
    if (s > 1.0f) return (s - 1.0f) / (4.08567f + 1.00071f * s);
    const float s2 = s * s;
    return 0.997118f + 0.1014f * s - 0.965241f * s2 - 0.130607f * s2 * s;
}

struct BaseState {
    float3 weightedBaseColor, diffuseColor; float baseDiffuseRoughness, specularAlpha, weightedSpecularIor;
    float3 dielectricSpecularF0; float dielectricSpecularWeight; float3 metalSpecularF0, metalAverageFresnel, metalMultipleScatterScale; float metalSpecularWeight;
};
struct CoatState { float3 tint; float presence, ior, roughness; float3 extraBaseLayerScale; };
struct FuzzState { float roughness; float3 tint; float presence; float3 t, b, n; float3 viewDirLocal; float viewReflected; };

static BaseState makeBaseState(float3 weightedBaseColor, float3 diffuseColor, float baseDiffuseRoughness, float specularAlpha, float weightedSpecularIor,
                               float3 dielectricF0, float dielectricW, float3 metalAvgF, float3 metalF0, float metalW) {
    BaseState s;
    s.weightedBaseColor = saturate(weightedBaseColor); s.diffuseColor = diffuseColor; s.baseDiffuseRoughness = saturate(baseDiffuseRoughness);
    s.specularAlpha = saturate(specularAlpha); s.weightedSpecularIor = fmax2(weightedSpecularIor, 1.0f);
    s.dielectricSpecularF0 = saturate(dielectricF0); s.dielectricSpecularWeight = saturate(dielectricW);
    s.metalAverageFresnel = saturate(metalAvgF); s.metalSpecularF0 = saturate(metalF0); s.metalSpecularWeight = saturate(metalW);
    s.metalMultipleScatterScale = s.metalSpecularWeight * s.metalAverageFresnel * s.metalAverageFresnel;
    return s;
}
static float3 estimateOpaqueBaseAlbedo(const BaseState& s) {
    const float ds = averageFresnel(s.weightedSpecularIor);
    const float3 fromMetal = s.metalSpecularWeight * s.metalAverageFresnel;
    const float3 fromDiel = s.dielectricSpecularWeight * lerp(s.weightedBaseColor, float3{1, 1, 1}, ds);
    return saturate(fromMetal + fromDiel);
}
static float3 coatExtraBaseLayerScale(const BaseState& b, float coatWeight, float coatIor, float coatDarkening) {
    const float safeIor = fmax2(coatIor, 1.0f);
    const float K_s = averageFresnel(safeIor);
    const float K_r = 1.0f - (1.0f - K_s) / fmax2(safeIor * safeIor, 1.0e-4f);
    const float ds = averageFresnel(b.weightedSpecularIor);
    const float specBase = saturate(b.dielectricSpecularWeight * ds + (1.0f - b.dielectricSpecularWeight));
    const float effRough = lerp(1.0f, std::sqrt(saturate(b.specularAlpha)), specBase);
    const float K = lerp(K_s, K_r, effRough);
    const float3 E_b = estimateOpaqueBaseAlbedo(b);
    const float3 Delta = float3{1.0f - K, 1.0f - K, 1.0f - K} / fmax3v(float3{1, 1, 1} - E_b * K, float3{1.0e-4f, 1.0e-4f, 1.0e-4f});
    const float mod = saturate(coatWeight) * saturate(coatDarkening);
    return lerp(float3{1, 1, 1}, saturate(Delta), mod);
}
static CoatState makeCoatState(const BaseState& b, float3 coatColor, float coatWeight, float coatIor, float coatRoughness, float coatDarkening) {
    CoatState s;
    s.tint = saturate(coatColor); s.presence = saturate(coatWeight); s.ior = fmax2(coatIor, 1.0f); s.roughness = saturate(coatRoughness);
    s.extraBaseLayerScale = coatExtraBaseLayerScale(b, s.presence, s.ior, coatDarkening);
    return s;
}
static float3 coatPassageColorMultiplier(const CoatState& s, float NdotX) {
    const float c = saturate(NdotX);
    if (c <= 0.0f || fmin2(s.tint.x, fmin2(s.tint.y, s.tint.z)) >= 1.0f) return float3{1, 1, 1};
    const float3 t0{std::sqrt(s.tint.x), std::sqrt(s.tint.y), std::sqrt(s.tint.z)};
    const float eta = rcp(s.ior);
    const float rc = std::sqrt(fmax2(0.0f, 1.0f - (1.0f - c * c) / fmax2(eta * eta, 1.0e-4f)));
    const float ds = rcp(fmax2(rc, 1.0e-4f));
    const float3 tr{std::pow(t0.x, ds), std::pow(t0.y, ds), std::pow(t0.z, ds)};
    return lerp(float3{1, 1, 1}, tr, s.presence);
}
static float dielectricEnergyReflected(const Luts& L, float ior, float alpha, float cosT) {
    const float si = fmax2(ior, 1.0e-4f), sa = saturate(alpha), sc = saturate(cosT);
    if (sa <= 0.0f) return fresnelDielectric(si, sc);
    return 1.0f - lookUpOdE(L, si, sa, sc);
}
static float coatReflectedProportion(const Luts& L, const CoatState& s, float NdotX) { return saturate(s.presence * dielectricEnergyReflected(L, s.ior, s.roughness, NdotX)); }
static float3 coatScaleIncoming(const Luts& L, const CoatState& s, float NdotV) {
    const float rp = coatReflectedProportion(L, s, NdotV);
    return coatPassageColorMultiplier(s, NdotV) * float3{1.0f - rp, 1.0f - rp, 1.0f - rp} * s.extraBaseLayerScale;
}
static float3 coatScaleOutgoing(const Luts& L, const CoatState& s, float NdotL) {
    const float rp = coatReflectedProportion(L, s, NdotL);
    return coatPassageColorMultiplier(s, NdotL) * float3{1.0f - rp, 1.0f - rp, 1.0f - rp};
}

static float fuzzDirectionalReflectance(const Luts& L, float r, float c) { return saturate(lookUpFuzzLTC(L, r, c).z); }
static float fuzzIncomingReflected(const Luts& L, float w, float r, float NdotV) { return saturate(saturate(w) * fuzzDirectionalReflectance(L, r, NdotV)); }
static float3 worldToLocal(const FuzzState& s, float3 d) { return {dot(d, s.t), dot(d, s.b), dot(d, s.n)}; }
static FuzzState makeFuzzState(const Luts& L, float3 normal, float3 viewDir, float3 fuzzColor, float fuzzWeight, float fuzzRoughness) {
    FuzzState s;
    s.roughness = saturate(fuzzRoughness); s.tint = saturate(fuzzColor); s.presence = saturate(fuzzWeight);
    s.n = normalize(normal);
    const float3 v = normalize(viewDir);
    const float3 pv = v - s.n * dot(v, s.n);
    if (dot(pv, pv) > 1.0e-6f) s.t = normalize(pv);
    else { const float3 helper = std::fabs(s.n.z) < 0.999f ? float3{0, 0, 1} : float3{0, 1, 0}; s.t = normalize(cross(helper, s.n)); }
    s.b = cross(s.n, s.t);
    s.viewDirLocal = worldToLocal(s, v);
    s.viewReflected = fuzzIncomingReflected(L, s.presence, s.roughness, s.viewDirLocal.z);
    return s;
}
static float fuzzProportionReflected(const Luts& L, const FuzzState& s, float3 dl) { if (dl.z <= 0.0f) return 0.0f; return saturate(s.presence * fuzzDirectionalReflectance(L, s.roughness, dl.z)); }
static float fuzzBaseLayerScaleComplete(const Luts& L, const FuzzState& s, float3 lightDir) {
    return (1.0f - s.viewReflected) * (1.0f - fuzzProportionReflected(L, s, worldToLocal(s, normalize(lightDir))));
}
static float3 fuzzSheenBRDF(const Luts& L, const FuzzState& s, float3 lightDir) {
    const float3 ll = worldToLocal(s, normalize(lightDir));
    if (s.viewDirLocal.z <= 0.0f || ll.z <= 0.0f) return float3{0, 0, 0};
    float phi = std::atan2(s.viewDirLocal.y, s.viewDirLocal.x);
    if (phi < 0.0f) phi += 2.0f * PI;
    const float ang = -phi, sa = std::sin(ang), ca = std::cos(ang);
    const float3 axis{0, 0, 1};
    const float3 ls = ll * ca + axis * dot(ll, axis) * (1.0f - ca) + sa * cross(axis, ll);
    const float3 ltc = lookUpFuzzLTC(L, s.roughness, s.viewDirLocal.z);
    const float aInv = ltc.x, bInv = ltc.y;
    float3 wo{aInv * ls.x + bInv * ls.z, aInv * ls.y, ls.z};
    const float len = length(wo);
    float e = 0.0f;
    if (len > 0.0f) {
        wo = wo / len;
        const float det = aInv * aInv;
        const float jac = det / fmax2(len * len * len, 1.0e-6f);
        e = saturate(wo.z) * (1.0f / PI) * jac;
    }
    return s.presence * ltc.z * s.tint * e;
}

static float directionalAlbedoFON(float mu, float roughness) {
    const float m = saturate(mu), mc = 1.0f - m;
    const float g1 = 0.0571085289f, g2 = 0.491881867f, g3 = -0.332181442f, g4 = 0.0714429953f;
    const float gOverPi = mc * (g1 + mc * (g2 + mc * (g3 + mc * g4)));
    return (1.0f + roughness * gOverPi) / (1.0f + FON_A * roughness);
}
static float3 diffuseEON(float3 albedo, float rough, float NdotV, float NdotL, float VdotL) {
    const float muIn = saturate(NdotV), muOut = saturate(NdotL);
    const float s = VdotL - muIn * muOut;
    const float sOverT = s > 0.0f ? s / fmax2(fmax2(muIn, muOut), 1.0e-4f) : s;
    const float A = 1.0f / (1.0f + FON_A * rough);
    const float3 single = albedo * Fd_Lambert() * A * (1.0f + rough * sOverT);
    const float EOut = directionalAlbedoFON(muOut, rough), EIn = directionalAlbedoFON(muIn, rough);
    const float avgE = A * (1.0f + FON_B * rough);
    const float3 msAlbedo = (albedo * albedo) * avgE / fmax3v(float3{1, 1, 1} - albedo * (1.0f - avgE), float3{1.0e-4f, 1.0e-4f, 1.0e-4f});
    const float k = fmax2(1.0e-4f, 1.0f - EOut) * fmax2(1.0e-4f, 1.0f - EIn) / fmax2(1.0e-4f, 1.0f - avgE);
    const float3 multi = (msAlbedo * Fd_Lambert()) * float3{k, k, k};
    return single + multi;
}

struct Frag {   // the FragmentInfo fields the punctual-light path reads
    float3 posWS, posVS, normalWS, viewWS, albedo, diffuseColor, emissive, dielectricSpecularF0, metalSpecularF0, metalAverageFresnel, coatColor, coatF0, fuzzColor;
    float NdotV, roughness, baseDiffuseRoughness, specularAlpha, weightedSpecularIor, dielectricSpecularWeight, metalSpecularWeight, coatWeight, coatIor, coatDarkening, coatRoughness, fuzzWeight, fuzzRoughness;
};

static float3 lightContribution(const Luts& L, const Frag& f, float3 lightToFrag, float3 lightColor, float intensity, float attenuation, float spotAtt) {
    const float NoV = saturate(dot(f.normalWS, f.viewWS));
    const float NoL = saturate(dot(f.normalWS, lightToFrag));
    const BaseState base = makeBaseState(f.albedo, f.diffuseColor, f.baseDiffuseRoughness, f.specularAlpha, f.weightedSpecularIor, f.dielectricSpecularF0, f.dielectricSpecularWeight,
                                         f.metalAverageFresnel, f.metalSpecularF0, f.metalSpecularWeight);
    const CoatState coat = makeCoatState(base, f.coatColor, f.coatWeight, f.coatIor, f.coatRoughness, f.coatDarkening);
    const FuzzState fuzz = makeFuzzState(L, f.normalWS, f.viewWS, f.fuzzColor, f.fuzzWeight, f.fuzzRoughness);
    // EvaluateOpenPBRBaseLayerDirect
    float3 diffuse, specular;
    {
        const float NdotV = saturate(dot(f.normalWS, f.viewWS)), NdotL = saturate(dot(f.normalWS, lightToFrag));
        const float3 h = normalize(lightToFrag + f.viewWS);
        const float NdotH = saturate(dot(f.normalWS, h)), LdotH = saturate(dot(lightToFrag, h));
        const float VdotL = dot(f.viewWS, lightToFrag);
        const float viewComp = lookUpOdE(L, base.weightedSpecularIor, base.specularAlpha, saturate(NdotV));
        const float avgComp = lookUpOdAvg(L, base.weightedSpecularIor, base.specularAlpha);
        const float cachedView = fmax2(0.0f, viewComp / fmax2(avgComp, 1.0e-12f));
        const float lightComp = lookUpOdE(L, base.weightedSpecularIor, base.specularAlpha, saturate(NdotL));
        const float diffuseEnergyComp = fmax2(0.0f, cachedView * lightComp);
        diffuse = diffuseEON(base.diffuseColor, base.baseDiffuseRoughness, NdotV, NdotL, VdotL) * diffuseEnergyComp;
        const float mView = lookUpImE(L, base.specularAlpha, NdotV), mLight = lookUpImE(L, base.specularAlpha, NdotL), mAvg = lookUpImAvg(L, base.specularAlpha);
        const float mTab = mView * mLight / fmax2(mAvg, 1.0e-12f);
        const float mScale = fmin2(mTab, rcp(fmax2(NdotL, 1.0e-4f))) * Fd_Lambert();
        const float3 dielSpec = base.dielectricSpecularWeight * specularLobe(base.specularAlpha, base.dielectricSpecularF0, NdotV, NdotL, NdotH, LdotH) *
                                ggxEnergyCompensation(NdotV, base.specularAlpha, base.dielectricSpecularF0);
        const float3 metalSpec = base.metalSpecularWeight * (specularLobe(base.specularAlpha, base.metalSpecularF0, NdotV, NdotL, NdotH, LdotH) + base.metalMultipleScatterScale * mScale);
        specular = dielSpec + metalSpec;
    }
    const float fuzzScale = fuzzBaseLayerScaleComplete(L, fuzz, lightToFrag);
    const float3 baseScale = coatScaleIncoming(L, coat, NoV) * coatScaleOutgoing(L, coat, NoL);
    float3 coatFr{0, 0, 0};
    if (coat.presence > 0.0f) {
        const float3 h = normalize(lightToFrag + f.viewWS);
        const float NoH = saturate(dot(f.normalWS, h)), LoH = saturate(dot(lightToFrag, h));
        coatFr = specularLobe(f.coatRoughness, f.coatF0, NoV, NoL, NoH, LoH);
        coatFr = coatFr * (ggxEnergyCompensation(NoV, f.coatRoughness, f.coatF0) * coat.presence);
    }
    const float3 fuzzFr = fuzzSheenBRDF(L, fuzz, lightToFrag);
    const float3 baseAtt = float3{fuzzScale, fuzzScale, fuzzScale} * baseScale;
    const float3 brdf = (diffuse + specular) * baseAtt + coatFr * float3{fuzzScale, fuzzScale, fuzzScale} + fuzzFr;
    return brdf * lightColor * intensity * attenuation * spotAtt * NoL;
}

static float smoothstepf(float a, float b, float x) { const float t = saturate((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }

}  // namespace orc

using namespace orc;

extern "C" {

// calculateLightContributionPBR on caller-supplied surface / light values, one call per sample: the hook a test uses to hold the per-light term against
// an independently written float64 restatement of the HLSL (tests/test_oracle_cpu.py).  in: 50 floats per sample -- normal, view, lightToFrag, albedo,
// diffuseColor, dielectricSpecularF0, metalSpecularF0, metalAverageFresnel, coatColor, coatF0, fuzzColor (3 each), baseDiffuseRoughness, specularAlpha,
// weightedSpecularIor, dielectricSpecularWeight, metalSpecularWeight, coatWeight, coatIor, coatDarkening, coatRoughness, fuzzWeight, fuzzRoughness,
// lightColor (3), intensity, attenuation, spotAttenuation.
int orc_light_contribution(const brmi_scene_buffers* scp, const float* in, uint64_t n, float* out) {
    const brmi_scene_buffers& sc = *scp;
    const Luts L{sc.lutOpaqueDielectricEnergyComplement, sc.lutOpaqueDielectricAvgEnergyComplement, sc.lutIdealMetalEnergyComplement, sc.lutIdealMetalAvgEnergyComplement, sc.lutFuzzLTC};
    for (uint64_t i = 0; i < n; i++) {
        const float* p = in + i * 50;
        auto v3 = [&](int k) { return float3{p[k], p[k + 1], p[k + 2]}; };
        Frag f{};
        f.normalWS = v3(0); f.viewWS = v3(3); const float3 l = v3(6);
        f.albedo = v3(9); f.diffuseColor = v3(12); f.dielectricSpecularF0 = v3(15); f.metalSpecularF0 = v3(18); f.metalAverageFresnel = v3(21);
        f.coatColor = v3(24); f.coatF0 = v3(27); f.fuzzColor = v3(30);
        f.baseDiffuseRoughness = p[33]; f.specularAlpha = p[34]; f.weightedSpecularIor = p[35]; f.dielectricSpecularWeight = p[36]; f.metalSpecularWeight = p[37];
        f.coatWeight = p[38]; f.coatIor = p[39]; f.coatDarkening = p[40]; f.coatRoughness = p[41]; f.fuzzWeight = p[42]; f.fuzzRoughness = p[43];
        const float3 r = lightContribution(L, f, l, v3(44), p[47], p[48], p[49]);
        out[i * 3] = r.x; out[i * 3 + 1] = r.y; out[i * 3 + 2] = r.z;
    }
    return 0;
}

// ComputeClusterID's slice of a view depth (lighting.hlsli:166-196).  log() = the correctly rounded fp32 logarithm (through double on both
// sides): a pixel whose depth sits within an ulp of a slice boundary must land in the same slice on CPU and GPU, and their float logf
// differ in the last bit.  (tests/test_oracle_cpu.py holds it against an arbitrary-precision restatement.)
uint32_t clusterSlice(float z, float zNear, float zFar, float zSplit, uint32_t nearSlices, uint32_t gz) {
    if (z < zSplit) { const float t = (z - zNear) / (zSplit - zNear); return t > 0.0f ? (uint32_t)(t * (float)nearSlices) : 0u; }
    auto logCR = [](float x) { return (float)std::log((double)x); };
    const float logStart = logCR(zSplit / zNear), logEnd = logCR(zFar / zNear), logZ = logCR(z / zNear);
    const float u = (logZ - logStart) / (logEnd - logStart);
    return nearSlices + (u > 0.0f ? (uint32_t)(u * (float)(gz - nearSlices)) : 0u);
}

// Slice plane depths (view space, negative): planes[s] = near plane of slice s, planes[s+1] = far plane.
// clustering.hlsl:66-90.  Written with logf/expf on the host; `planes` has gridZ+1 pairs? No: one array
// of 2*gridZ floats (near, far per slice) because the two expressions are not bit-identical.
int orc_cluster_planes(float zNear, float zFar, uint32_t gridZ, uint32_t nearSlices, float zSplit, float* planesNearFar) {
    for (uint32_t sliceZ = 0; sliceZ < gridZ; sliceZ++) {
        float pn, pf;
        if (sliceZ < nearSlices) {
            const float sliceSize = (zSplit - zNear) / (float)nearSlices;
            pn = -(zNear + (float)sliceZ * sliceSize);
            pf = -(zNear + (float)(sliceZ + 1) * sliceSize);
        } else {
            const float logStart = std::log(zSplit / zNear), logEnd = std::log(zFar / zNear);
            const float t0 = (float)(sliceZ - nearSlices) / (float)(gridZ - nearSlices);
            const float t1 = (float)(sliceZ + 1 - nearSlices) / (float)(gridZ - nearSlices);
            pn = -zNear * std::exp(logStart + t0 * (logEnd - logStart));
            pf = -zNear * std::exp(logStart + t1 * (logEnd - logStart));
        }
        planesNearFar[2 * sliceZ] = pn; planesNearFar[2 * sliceZ + 1] = pf;
    }
    return 0;
}

// K9 + K10.  Clusters are processed in index order (one valid serialisation of the page allocator).
int orc_light_cluster(const brmi_scene_buffers* scp, const float* planesNearFar, brmi_light_cluster* clusters, brmi_light_page* pages, uint32_t poolSize, uint32_t* pagesUsed) {
    const brmi_scene_buffers& sc = *scp;
    const brmi_per_frame& pf = sc.perFrame[0];
    const brmi_camera& cam = sc.cameras[pf.mainCameraIndex];
    const uint32_t gx = pf.lightClusterGridSizeX, gy = pf.lightClusterGridSizeY, gz = pf.lightClusterGridSizeZ;
    const float W = (float)pf.screenResX, H = (float)pf.screenResY;
    const mat4& invProj = M(cam.projectionInverse);
    auto screenToView = [&](float sx, float sy, float sz) {
        const float4 ndc{2.0f * sx / W - 1.0f, 2.0f * (H - sy - 1.0f) / H - 1.0f, sz, 1.0f};
        float4 v = mul(ndc, invProj);
        return float3{v.x / v.w, v.y / v.w, v.z / v.w};
    };
    auto lineZ = [](float3 end, float zDistance) { const float t = zDistance / end.z; return float3{t * end.x, t * end.y, t * end.z}; };
    const float tsx = W / (float)gx, tsy = H / (float)gy;
    for (uint32_t z = 0; z < gz; z++) for (uint32_t y = 0; y < gy; y++) for (uint32_t x = 0; x < gx; x++) {
        const uint32_t idx = x + y * gx + z * gx * gy;
        const float3 minTile = screenToView((float)x * tsx, (float)y * tsy, 1.0f);
        const float3 maxTile = screenToView(((float)x + 1.0f) * tsx, ((float)y + 1.0f) * tsy, 1.0f);
        const float pn = planesNearFar[2 * z], pfar = planesNearFar[2 * z + 1];
        const float3 p0 = lineZ(minTile, pn), p1 = lineZ(maxTile, pn), p2 = lineZ(minTile, pfar), p3 = lineZ(maxTile, pfar);
        const float3 mn = fmin3v(fmin3v(p0, p1), fmin3v(p2, p3)), mx = fmax3v(fmax3v(p0, p1), fmax3v(p2, p3));
        brmi_light_cluster& c = clusters[idx];
        c.minPoint[0] = mn.x; c.minPoint[1] = mn.y; c.minPoint[2] = mn.z; c.minPoint[3] = 0.0f;
        c.maxPoint[0] = mx.x; c.maxPoint[1] = mx.y; c.maxPoint[2] = mx.z; c.maxPoint[3] = 0.0f;
        c.numLights = 0; c.ptrFirstPage = BRMI_LIGHT_PAGE_NULL; c.pad[0] = c.pad[1] = 0;
    }
    uint32_t counter = 0;
    auto alloc = [&]() { uint32_t i = counter++; return i >= poolSize ? BRMI_LIGHT_PAGE_NULL : i; };
    const uint32_t total = gx * gy * gz, lightCount = pf.numLights;
    for (uint32_t idx = 0; idx < total; idx++) {
        brmi_light_cluster& c = clusters[idx];
        uint32_t page = alloc();
        c.numLights = 0; c.ptrFirstPage = page;
        if (page == BRMI_LIGHT_PAGE_NULL) continue;
        pages[page].ptrNextPage = BRMI_LIGHT_PAGE_NULL;
        uint32_t inPage = 0;
        const float3 mn{c.minPoint[0], c.minPoint[1], c.minPoint[2]}, mx{c.maxPoint[0], c.maxPoint[1], c.maxPoint[2]};
        for (uint32_t i = 0; i < lightCount; i++) {
            if (inPage >= BRMI_LIGHTS_PER_PAGE) {
                pages[page].numLightsInPage = BRMI_LIGHTS_PER_PAGE;
                const uint32_t old = page;
                page = alloc();
                if (page == BRMI_LIGHT_PAGE_NULL) break;
                pages[page].ptrNextPage = old;
                c.ptrFirstPage = page;
                inPage = 0;
            }
            const uint32_t li = sc.activeLightIndices[i];
            const brmi_light_info& l = sc.lights[li];
            bool add = false;
            if (l.type == BRMI_LIGHT_POINT || l.type == BRMI_LIGHT_SPOT) {
                const float3 center = xyz(mulPoint(float3{l.boundingSphere[0], l.boundingSphere[1], l.boundingSphere[2]}, M(cam.view)));
                const float3 closest = fmax3v(mn, fmin3v(center, mx));
                const float3 d = closest - center;
                add = dot(d, d) <= l.boundingSphere[3] * l.boundingSphere[3];
            } else if (l.type == BRMI_LIGHT_DIRECTIONAL) add = true;
            if (add) { pages[page].lightIndices[inPage] = li; inPage++; c.numLights++; }
        }
        if (page != BRMI_LIGHT_PAGE_NULL) pages[page].numLightsInPage = inPage;
    }
    if (pagesUsed) *pagesUsed = counter < poolSize ? counter : poolSize;
    return 0;
}

// K11.  Inputs are the linear G-buffer images written by orc_gbuffer + orc_depth_copy.
// orc_shade_forward: the FORWARD variant of the same lighting (BASELINE.json configs[0], "forward PBR"; shaders.hlsl:221-229 PSMain ->
// GetFragmentInfoDirect, utilities.hlsli:2791-2807): `forwardInputs` (orc_gbuffer_forward) carries the material inputs before the G-buffer's
// quantisation and the interpolated world position, which replace the decoded G-buffer words and the position reconstructed from depth; the
// normal is the same fp32 value either way.  Null = the deferred path.
int orc_shade_forward(const brmi_scene_buffers* scp, uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1, const float* depth,
              const float* normals, const uint32_t* albedo, const uint64_t* coat, const uint64_t* emissive, const uint64_t* fuzz, const uint32_t* metallicRoughness,
              const brmi_light_cluster* clusters, const brmi_light_page* pages, uint32_t poolSize,
              uint32_t enablePunctual, uint32_t clusteredLighting, uint64_t* hdr, const float* forwardInputs, int threads);
int orc_shade(const brmi_scene_buffers* scp, uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1, const float* depth,
              const float* normals, const uint32_t* albedo, const uint64_t* coat, const uint64_t* emissive, const uint64_t* fuzz, const uint32_t* metallicRoughness,
              const brmi_light_cluster* clusters, const brmi_light_page* pages, uint32_t poolSize,
              uint32_t enablePunctual, uint32_t clusteredLighting, uint64_t* hdr, int threads) {
    return orc_shade_forward(scp, W, H, bandY0, bandY1, depth, normals, albedo, coat, emissive, fuzz, metallicRoughness, clusters, pages, poolSize, enablePunctual, clusteredLighting, hdr, nullptr, threads);
}
int orc_shade_forward(const brmi_scene_buffers* scp, uint32_t W, uint32_t H, uint32_t bandY0, uint32_t bandY1, const float* depth,
              const float* normals, const uint32_t* albedo, const uint64_t* coat, const uint64_t* emissive, const uint64_t* fuzz, const uint32_t* metallicRoughness,
              const brmi_light_cluster* clusters, const brmi_light_page* pages, uint32_t poolSize,
              uint32_t enablePunctual, uint32_t clusteredLighting, uint64_t* hdr, const float* forwardInputs, int threads) {
    const brmi_scene_buffers& sc = *scp;
    const brmi_per_frame& pf = sc.perFrame[0];
    const brmi_camera& cam = sc.cameras[pf.mainCameraIndex];
    const Luts L{sc.lutOpaqueDielectricEnergyComplement, sc.lutOpaqueDielectricAvgEnergyComplement, sc.lutIdealMetalEnergyComplement, sc.lutIdealMetalAvgEnergyComplement, sc.lutFuzzLTC};
    if (bandY1 == 0) { bandY0 = 0; bandY1 = H; }
    const uint32_t gx = pf.lightClusterGridSizeX, gy = pf.lightClusterGridSizeY, gz = pf.lightClusterGridSizeZ;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
    for (int64_t py = bandY0; py < (int64_t)bandY1; py++) {
        for (uint32_t px = 0; px < W; px++) {
            const uint64_t idx = (uint64_t)py * W + px;
            const float d = depth[idx];
            if (asuint(d) == BRMI_DEPTH_EMPTY_BITS) continue;
            float uvx = ((float)px + 0.5f) / (float)pf.screenResX, uvy = ((float)py + 0.5f) / (float)pf.screenResY;
            uvy = 1.0f - uvy;
            const float linearZ = d;
            const float4 clipPos{uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f, 1.0f, 1.0f};
            const float4 viewPosH = mul(clipPos, M(cam.projectionInverse));
            float3 posVS = xyz(viewPosH) * linearZ;
            float3 posWS = xyz(mulPoint(posVS, M(cam.viewInverse)));
            const float* fwd = forwardInputs ? forwardInputs + idx * 24u : nullptr;
            if (fwd) { posWS = float3{fwd[20], fwd[21], fwd[22]}; posVS = xyz(mulPoint(posWS, M(cam.view))); }      // input.positionWorldSpace / positionViewSpace of the forward PSInput
            const float3 viewDir = normalize(float3{cam.positionWorldSpace[0], cam.positionWorldSpace[1], cam.positionWorldSpace[2]} - posWS);

            // GetFragmentInfoScreenSpace
            Frag f;
            f.posWS = posWS; f.posVS = posVS; f.viewWS = viewDir;
            const float3 nrm{normals[idx * 4], normals[idx * 4 + 1], normals[idx * 4 + 2]};
            const float nw = normals[idx * 4 + 3];
            const uint32_t al = albedo[idx], mr = metallicRoughness[idx];
            const float3 baseColor = fwd ? float3{fwd[0], fwd[1], fwd[2]} : float3{unorm8_to_float(al), unorm8_to_float(al >> 8), unorm8_to_float(al >> 16)};
            auto H4 = [](uint64_t v, int k) { return f16_to_f32((uint16_t)(v >> (16 * k))); };
            const uint64_t cs = coat[idx], es = emissive[idx], fs = fuzz[idx];
            // the three fp16 planes: decoded words (deferred) or the unquantised inputs (forward)
            const float3 emissiveIn = fwd ? float3{fwd[12], fwd[13], fwd[14]} : float3{H4(es, 0), H4(es, 1), H4(es, 2)};
            const float3 coatColorIn = fwd ? float3{fwd[8], fwd[9], fwd[10]} : float3{H4(cs, 0), H4(cs, 1), H4(cs, 2)};
            const float coatWeightIn = fwd ? fwd[11] : H4(cs, 3);
            const float3 fuzzColorIn = fwd ? float3{fwd[16], fwd[17], fwd[18]} : float3{H4(fs, 0), H4(fs, 1), H4(fs, 2)};
            const float fuzzRoughnessIn = fwd ? fwd[19] : H4(fs, 3);
            const float metal = fwd ? fwd[4] : unorm8_to_float(mr), pr = fwd ? fwd[5] : unorm8_to_float(mr >> 8), coatR = fwd ? fwd[6] : unorm8_to_float(mr >> 16), fuzzW = fwd ? fwd[7] : unorm8_to_float(mr >> 24);
            const float prc = clampf(pr, BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
            f.roughness = prc * prc;
            float NdotV = dot(nrm, viewDir);
            f.normalWS = normalize(nrm + fmax2(0.0f, -NdotV + BRMI_MIN_N_DOT_V) * viewDir);
            f.NdotV = fmax2(BRMI_MIN_N_DOT_V, NdotV);
            // PopulateFragmentInfoFromOpenPBR
            const uint32_t opIndex = (uint32_t)(nw + 0.5f);
            const brmi_openpbr_material_info& op = sc.openpbrMaterials[opIndex < sc.openpbrMaterialCount ? opIndex : 0];
            const float baseWeight = saturate(op.baseWeight), specularWeight = saturate(op.specularWeight);
            const float3 specularColor = saturate(float3{op.specularColor[0], op.specularColor[1], op.specularColor[2]});
            const float3 weightedBaseColor = saturate(baseColor * baseWeight);
            float weightedSpecularIor;
            {   // OpenPBRApplySpecularWeightToIor
                const float unscaledF0 = iorToF0(op.specularIor);
                const float scaledF0 = fmin2(unscaledF0 * saturate(specularWeight), 0.9999f);
                const float safeF0 = fmin2(saturate(scaledF0), 0.9999f);
                const float sq = std::sqrt(safeF0);
                weightedSpecularIor = (1.0f + sq) / fmax2(1.0f - sq, 1.0e-4f);
            }
            const float dielF0Scalar = iorToF0(weightedSpecularIor);
            const float3 dielF0 = saturate(specularColor * dielF0Scalar);
            const float coatPR = clampf(coatR, BRMI_MIN_PERCEPTUAL_ROUGHNESS, 1.0f);
            const float coatF0Scalar = iorToF0(op.coatIor);
            f.dielectricSpecularWeight = saturate(1.0f - metal);
            f.metalSpecularWeight = saturate(metal * specularWeight);
            f.metalSpecularF0 = saturate(weightedBaseColor * specularColor);
            {   // OpenPBRMetalAverageFresnelWithF82Tint(weightedBaseColor, specularColor)
                const float3 safeF0 = saturate(weightedBaseColor), wmF0 = float3{1, 1, 1} - safeF0;
                const float cosMax = 1.0f / 7.0f, om = 1.0f - cosMax;
                const float om5 = std::pow(om, 5.0f), om6 = std::pow(om, 6.0f);
                const float3 wmF0b = float3{1, 1, 1} - saturate(safeF0), wmTint = float3{1, 1, 1} - saturate(specularColor);
                const float3 num = (saturate(safeF0) + wmF0b * om5) * wmTint;
                const float den = cosMax * om6;
                const float3 b = num / fmax2(den, 1.0e-6f);
                f.metalAverageFresnel = saturate(safeF0 + wmF0 * (1.0f / 21.0f) - b * (1.0f / 126.0f));
            }
            f.albedo = weightedBaseColor;
            f.emissive = emissiveIn;
            f.coatWeight = saturate(coatWeightIn);
            f.coatColor = saturate(coatColorIn);
            f.coatRoughness = coatPR * coatPR;
            f.coatF0 = saturate(f.coatColor * coatF0Scalar);
            f.coatIor = op.coatIor; f.coatDarkening = saturate(op.coatDarkening);
            f.fuzzWeight = saturate(fuzzW); f.fuzzColor = saturate(fuzzColorIn); f.fuzzRoughness = saturate(fuzzRoughnessIn);
            f.baseDiffuseRoughness = saturate(op.baseDiffuseRoughness);
            f.specularAlpha = f.roughness; f.weightedSpecularIor = weightedSpecularIor;
            f.dielectricSpecularF0 = dielF0;
            f.diffuseColor = weightedBaseColor * (1.0f - metal);

            float3 lighting{0, 0, 0};
            if (enablePunctual) {
                auto shadeLight = [&](uint32_t lightIndex) {
                    const brmi_light_info& l = sc.lights[lightIndex];
                    // getLightParametersForFragment
                    float3 lightToFrag; float att, dist = 0.0f, spot = 1.0f;
                    const float3 lpos{l.posWorldSpace[0], l.posWorldSpace[1], l.posWorldSpace[2]};
                    if (l.type == BRMI_LIGHT_DIRECTIONAL) { lightToFrag = -float3{l.dirWorldSpace[0], l.dirWorldSpace[1], l.dirWorldSpace[2]}; att = 1.0f; }
                    else {
                        lightToFrag = normalize(lpos - posWS);
                        dist = length(lpos - posWS);
                        att = 1.0f / ((l.attenuation[0] + l.attenuation[1] * dist + l.attenuation[2] * dist * dist) + 0.0001f);
                    }
                    if (l.type == BRMI_LIGHT_SPOT) {
                        const float3 ld{l.dirWorldSpace[0], l.dirWorldSpace[1], l.dirWorldSpace[2]};
                        const float c = dot(normalize(ld), normalize(-lightToFrag));
                        if (c > l.outerConeAngle) spot = (c < l.innerConeAngle) ? smoothstepf(l.outerConeAngle, l.innerConeAngle, c) : 1.0f; else spot = 0.0f;
                    }
                    if (l.type != BRMI_LIGHT_DIRECTIONAL && dist > l.maxRange) return;
                    const float3 c = lightContribution(L, f, lightToFrag, float3{l.color[0], l.color[1], l.color[2]}, l.color[3], att, spot);
                    lighting = lighting + (1.0f - 0.0f) * c;
                };
                if (clusteredLighting) {
                    // ComputeClusterID
                    const float tsx = (float)pf.screenResX / (float)gx, tsy = (float)pf.screenResY / (float)gy;
                    const uint32_t tx = (uint32_t)((float)px / tsx), ty = (uint32_t)((float)py / tsy);
                    const float z = std::fabs(posVS.z);
                    const uint32_t sliceZ = clusterSlice(z, cam.zNear, cam.zFar, pf.clusterZSplitDepth, pf.nearClusterCount, gz);
                    const uint32_t ci = (uint32_t)((float)tx + (float)ty * (float)gx + (float)sliceZ * (float)gx * (float)gy);
                    if (ci < gx * gy * gz) {
                        const brmi_light_cluster& cl = clusters[ci];
                        const uint32_t count = cl.numLights;
                        uint32_t page = cl.ptrFirstPage, remaining = count;
                        const uint32_t maxPages = ((count + BRMI_LIGHTS_PER_PAGE - 1u) / BRMI_LIGHTS_PER_PAGE) > 1u ? ((count + BRMI_LIGHTS_PER_PAGE - 1u) / BRMI_LIGHTS_PER_PAGE) : 1u;
                        uint32_t visited = 0;
                        while (page != BRMI_LIGHT_PAGE_NULL && page < poolSize && remaining > 0 && visited < maxPages) {
                            const brmi_light_page& pg = pages[page];
                            uint32_t n = pg.numLightsInPage < BRMI_LIGHTS_PER_PAGE ? pg.numLightsInPage : BRMI_LIGHTS_PER_PAGE;
                            n = n < remaining ? n : remaining;
                            if (n == 0) break;
                            for (uint32_t i = 0; i < n; i++) shadeLight(sc.activeLightIndices[pg.lightIndices[i]]);
                            remaining -= n; page = pg.ptrNextPage; visited++;
                        }
                    }
                } else {
                    for (uint32_t i = 0; i < pf.numLights; i++) shadeLight(sc.activeLightIndices[i]);
                }
            }
            {   // emissive through coat + fuzz (EvaluateOpenPBREmissive)
                const BaseState base = makeBaseState(f.albedo, f.diffuseColor, f.baseDiffuseRoughness, f.specularAlpha, f.weightedSpecularIor, f.dielectricSpecularF0, f.dielectricSpecularWeight,
                                                     f.metalAverageFresnel, f.metalSpecularF0, f.metalSpecularWeight);
                const CoatState coatS = makeCoatState(base, f.coatColor, f.coatWeight, f.coatIor, f.coatRoughness, f.coatDarkening);
                const float fuzzBase = 1.0f - fuzzIncomingReflected(L, f.fuzzWeight, f.fuzzRoughness, f.NdotV);
                const float3 coatT = coatScaleIncoming(L, coatS, f.NdotV);
                lighting = lighting + f.emissive * float3{fuzzBase, fuzzBase, fuzzBase} * coatT;
            }
            hdr[idx] = pack_half4(lighting.x, lighting.y, lighting.z, 1.0f);
        }
    }
    return 0;
}

}  // extern "C"
