// brmi_shade_math.h -- GGX / OpenPBR lookup-table arithmetic shared by the shading kernel and the per-frame table builder.
// Reference: BR/shaders/Include/PBR.hlsli:8-190, BR/shaders/Include/IBL.hlsli:94-672 (see brmi_light.hip for the per-function map).
#ifndef BRMI_SHADE_MATH_H
#define BRMI_SHADE_MATH_H

#include "brmi_device.h"

namespace brmi {

constexpr float PI_F = 3.1415926538f;
constexpr float MEDIUMP_MAX = 65504.0f;

// The R16_UNORM tables are expanded once (k_expand_luts: texel / 65535.0f, the UNORM decode) so the
// per-sample cost is four loads, not four correctly rounded divisions.  unorm8[] is the same for /255.
struct Luts { const float* odE; const float* odAvg; const float* imE; const float* imAvg; const float* ltc; const float* unorm8; };

BRMI_DEV float texel_u16(const float* t, uint32_t i) { return t[i]; }
BRMI_DEV uint32_t clamp_texel(float f, uint32_t n) { int i = (int)f; i = i < 0 ? 0 : i; i = i > (int)n - 1 ? (int)n - 1 : i; return (uint32_t)i; }
BRMI_DEV float sample_u16(const float* t, uint32_t W, uint32_t H, float u, float v) {
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const uint32_t x0 = clamp_texel(x0f, W), x1 = clamp_texel(x0f + 1.0f, W), y0 = clamp_texel(y0f, H), y1 = clamp_texel(y0f + 1.0f, H);
    const float r0 = lerpf(texel_u16(t, y0 * W + x0), texel_u16(t, y0 * W + x1), fx);
    const float r1 = lerpf(texel_u16(t, y1 * W + x0), texel_u16(t, y1 * W + x1), fx);
    return lerpf(r0, r1, fy);
}
BRMI_DEV f3 sample_ltc(const float* t, float u, float v) {
    const float x = u * 32.0f - 0.5f, y = v * 32.0f - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const uint32_t x0 = clamp_texel(x0f, 32), x1 = clamp_texel(x0f + 1.0f, 32), y0 = clamp_texel(y0f, 32), y1 = clamp_texel(y0f + 1.0f, 32);
    auto T = [&](uint32_t yy, uint32_t xx) { const float4 q = *reinterpret_cast<const float4*>(t + ((size_t)yy * 32u + xx) * 4u); return f3{q.x, q.y, q.z}; };
    return lerp3(lerp3(T(y0, x0), T(y0, x1), fx), lerp3(T(y1, x0), T(y1, x1), fx), fy);
}

// K11 is checked to 1 fp16 ULP, not bit for bit: outside the direction vectors (N, V, L, H stay on IEEE
// division / sqrt, 1 - NoH^2 amplifies their error) the BRDF algebra uses the hardware reciprocal and
// square root (1 ULP, what HLSL `/`, rcp and sqrt compile to on a GPU) instead of the ~11-instruction
// correctly rounded expansions.
// saturate as one v_med3_f32 (the shading pass only: for finite x the value of min(max(x, 0), 1); the sign of a zero result is not pinned)
BRMI_DEV float satq(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
BRMI_DEV f3 satq3(f3 v) { return f3{satq(v.x), satq(v.y), satq(v.z)}; }
BRMI_DEV float qdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
BRMI_DEV float qrcp(float b) { return __builtin_amdgcn_rcpf(b); }
BRMI_DEV float qsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
BRMI_DEV f3 qdiv3(f3 a, f3 b) { return f3{qdiv(a.x, b.x), qdiv(a.y, b.y), qdiv(a.z, b.z)}; }

// ---- PBR.hlsli
// BRMI_FP_FAST: a block of BRDF algebra whose result only has to hold the HDR tolerance (1 fp16 ULP) may contract a * b + c into an
// FMA (one rounding instead of two).  Never used on N, V, L, H or on 1 - NoH^2 (d_ggx), whose error the highlight amplifies.
#define BRMI_FP_FAST _Pragma("clang fp contract(fast)")
BRMI_DEV void ggx_dir_albedo_AB(float NdotV, float alpha, float& A, float& B) {
    BRMI_FP_FAST
    const float x = NdotV, y = alpha, x2 = x * x, y2 = y * y;
    const float c0[4] = {0.1003f, 0.9345f, 1.0f, 1.0f}, c1[4] = {-0.6303f, -2.323f, -1.765f, 0.2281f}, c2[4] = {9.748f, 2.229f, 8.263f, 15.94f},
                c3[4] = {-2.038f, -3.748f, 11.53f, -55.83f}, c4[4] = {29.34f, 1.424f, 28.96f, 13.08f}, c5[4] = {-8.245f, -0.7684f, -7.507f, 41.26f},
                c6[4] = {-26.44f, 1.436f, -36.11f, 54.9f}, c7[4] = {19.99f, 0.2913f, 15.86f, 300.2f}, c8[4] = {-5.448f, 0.6286f, 33.37f, -285.1f};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        r[i] = c0[i] + c1[i] * x + c2[i] * y + c3[i] * x * y + c4[i] * x2 + c5[i] * y2 + c6[i] * x2 * y + c7[i] * x * y2 + c8[i] * x2 * y2;
    A = clampf(qdiv(r[0], r[2]), 0.0f, 1.0f); B = clampf(qdiv(r[1], r[3]), 0.0f, 1.0f);
}
// For a fixed alpha the fit's four polynomials are quadratics in x = N.V: r[i](x) = q0[i] + x (q1[i] + x q2[i]).  alpha only depends on
// the 8-bit roughness code of the G-buffer, so the 12 coefficients are tabulated per code (k_frame_constants) and a pixel evaluates
// 8 FMAs instead of the 36-term form (tolerance-level: the same polynomial, associated differently).
struct GgxQuad { float q0[4], q1[4], q2[4]; };
static_assert(sizeof(GgxQuad) == 48, "three float4");
BRMI_DEV GgxQuad ggx_quad_of(float alpha) {
    const float y = alpha, y2 = y * y;
    const float c0[4] = {0.1003f, 0.9345f, 1.0f, 1.0f}, c1[4] = {-0.6303f, -2.323f, -1.765f, 0.2281f}, c2[4] = {9.748f, 2.229f, 8.263f, 15.94f},
                c3[4] = {-2.038f, -3.748f, 11.53f, -55.83f}, c4[4] = {29.34f, 1.424f, 28.96f, 13.08f}, c5[4] = {-8.245f, -0.7684f, -7.507f, 41.26f},
                c6[4] = {-26.44f, 1.436f, -36.11f, 54.9f}, c7[4] = {19.99f, 0.2913f, 15.86f, 300.2f}, c8[4] = {-5.448f, 0.6286f, 33.37f, -285.1f};
    GgxQuad g;
    for (int i = 0; i < 4; i++) { g.q0[i] = c0[i] + c2[i] * y + c5[i] * y2; g.q1[i] = c1[i] + c3[i] * y + c7[i] * y2; g.q2[i] = c4[i] + c6[i] * y + c8[i] * y2; }
    return g;
}
BRMI_DEV f3 ggx_energy_compensation_q(const GgxQuad& g, float x, f3 Fss) {
    BRMI_FP_FAST
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = g.q0[i] + x * (g.q1[i] + x * g.q2[i]);
    const float A = clampf(qdiv(r[0], r[2]), 0.0f, 1.0f), B = clampf(qdiv(r[1], r[3]), 0.0f, 1.0f);
    const float Ess = A + B, k = (1.0f - Ess) * qrcp(Ess);
    return f3{__builtin_fmaf(Fss.x, k, 1.0f), __builtin_fmaf(Fss.y, k, 1.0f), __builtin_fmaf(Fss.z, k, 1.0f)};
}
BRMI_DEV f3 ggx_energy_compensation(float NdotV, float alpha, f3 Fss) {
    BRMI_FP_FAST
    float A, B; ggx_dir_albedo_AB(NdotV, alpha, A, B);
    const float Ess = (f3{1.0f, 1.0f, 1.0f} * A + f3{1.0f, 1.0f, 1.0f} * B).x;
    return f3{1.0f, 1.0f, 1.0f} + Fss * (1.0f - Ess) * qrcp(Ess);
}
BRMI_DEV f3 f_schlick(f3 f0, float f90, float VoH) {
    const float pw = powf(1.0f - VoH, 5.0f);
    return f0 + (f3{f90, f90, f90} - f0) * pw;
}
BRMI_DEV float v_smith_ggx(float roughness, float NoV, float NoL) {
    BRMI_FP_FAST
    const float a2 = roughness * roughness;
    const float lambdaV = NoL * qsqrt((NoV - a2 * NoV) * NoV + a2);
    const float lambdaL = NoV * qsqrt((NoL - a2 * NoL) * NoL + a2);
    return min2(qdiv(0.5f, lambdaV + lambdaL), MEDIUMP_MAX);
}
BRMI_DEV float d_ggx(float roughness, float NoH) {
    const float oneMinus = 1.0f - NoH * NoH;
    const float aa = NoH * roughness;
    const float k = qdiv(roughness, oneMinus + aa * aa);
    return min2(k * k * (1.0f / PI_F), MEDIUMP_MAX);
}
BRMI_DEV f3 specular_lobe(float roughness, f3 f0, float NoV, float NoL, float NoH, float LoH) {
    const float D = d_ggx(roughness, NoH), V = v_smith_ggx(roughness, NoV, NoL);
    const float tmp = 50.0f * 0.33f;
    const float f90 = sat(dot3(f0, f3{tmp, tmp, tmp}));
    return (D * V) * f_schlick(f0, f90, LoH);
}

// ---- IBL.hlsli (OpenPBR)
constexpr float TBL = 32.0f, TBL_M1 = 31.0f, IOR_MAX = 2.5f, INV_IOR_MAX = 1.0f / 2.5f;
BRMI_DEV float fon_a() { return 0.5f - 2.0f / (3.0f * PI_F); }
BRMI_DEV float fon_b() { return 2.0f / 3.0f - 28.0f / (15.0f * PI_F); }
BRMI_DEV float ior_to_index(float ior) {
    const float safeIor = max2(ior, 1.0e-4f);
    const float half = 0.5f * TBL, halfM1 = half - 1.0f, inv = 1.0f / (IOR_MAX - 1.0f);
    if (safeIor < 1.0f) { const float invIor = 1.0f / safeIor; const float fr = (invIor - 1.0f) * inv; return halfM1 - fr * halfM1; }
    const float fr = (safeIor - 1.0f) * inv;
    return half + fr * halfM1;
}
BRMI_DEV float alpha_to_index(float alpha) { return qsqrt(sat(alpha)) * TBL_M1; }
BRMI_DEV float cos_to_index(float c) { return sat(c) * TBL_M1; }
BRMI_DEV float clamp_index(float e) { return clampf(e, 0.0f, TBL_M1); }
BRMI_DEV float remap_index(float e) { const float inv = 1.0f / TBL; const float mn = 0.5f * inv, mx = 1.0f - mn; return clampf(mn + e * inv, mn, mx); }
BRMI_DEV float extrapolate_ior(float tableValue, float ior) {
    if (ior > IOR_MAX || ior < INV_IOR_MAX) {
        const float f0Max = ior_to_f0(IOR_MAX);
        const float invRange = 1.0f / (1.0f - f0Max);
        const float f0 = ior_to_f0(max2(ior, 1.0e-4f));
        const float progress = (f0 - f0Max) * invRange;
        return (1.0f - progress) * tableValue;
    }
    return tableValue;
}
BRMI_DEV float fresnel_dielectric(float eta, float cosI) {
    const float c = sat(cosI);
    if (fabsf(eta - 1.0f) <= 1.0e-6f) return 0.0f;
    const float s2 = max2(0.0f, 1.0f - c * c);
    const float st2 = s2 / max2(eta * eta, 1.0e-6f);
    if (st2 >= 1.0f) return 1.0f;
    const float ct = sqrtf(max2(0.0f, 1.0f - st2));
    const float eci = eta * c, ect = eta * ct;
    const float rs = (c - ect) / max2(c + ect, 1.0e-6f);
    const float rp = (ct - eci) / max2(ct + eci, 1.0e-6f);
    return 0.5f * (rs * rs + rp * rp);
}
BRMI_DEV float lut_od_avg(const Luts& L, float ior, float alpha) {
    const float ei = clamp_index(ior_to_index(ior)), ea = clamp_index(alpha_to_index(alpha));
    return extrapolate_ior(sample_u16(L.odAvg, 32, 32, remap_index(ea), remap_index(ei)), ior);
}
BRMI_DEV float lut_od_e(const Luts& L, float ior, float alpha, float cosT) {
    const float ei = clamp_index(ior_to_index(ior)), ea = clamp_index(alpha_to_index(alpha)), ec = clamp_index(cos_to_index(cosT));
    const int s0 = (int)floorf(ei);
    const int s1 = (s0 + 1) < 31 ? (s0 + 1) : 31;
    const float st = ei - (float)s0;
    const float u = remap_index(ec), v = remap_index(ea);
    const float v0 = sample_u16(L.odE + (size_t)s0 * 1024u, 32, 32, u, v), v1 = sample_u16(L.odE + (size_t)s1 * 1024u, 32, 32, u, v);
    return extrapolate_ior(lerpf(v0, v1, st), ior);
}
BRMI_DEV float lut_im_e(const Luts& L, float alpha, float cosT) {
    const float ea = clamp_index(alpha_to_index(alpha)), ec = clamp_index(cos_to_index(cosT));
    return sample_u16(L.imE, 32, 32, remap_index(ec), remap_index(ea));
}
BRMI_DEV float lut_im_avg(const Luts& L, float alpha) {
    const float ea = clamp_index(alpha_to_index(alpha));
    return sample_u16(L.imAvg, 32, 1, remap_index(ea), 0.5f);
}
BRMI_DEV f3 lut_fuzz_ltc(const Luts& L, float roughness, float cosT) {
    const float u = sat(cosT) * (31.0f / 32.0f) + 0.5f / 32.0f, v = sat(roughness) * (31.0f / 32.0f) + 0.5f / 32.0f;
    return sample_ltc(L.ltc, u, v);
}
// Bilinear fetch with the row (v) part prepared once: identical arithmetic to sample_u16, split in two.
// Rows are 32-bit offsets into the expanded-table buffer (one SGPR base + VGPR offset per load) rather than 64-bit pointers.
struct LutRows { uint32_t r0, r1; float fy; };
BRMI_DEV LutRows prep_rows(uint32_t tableOffset, uint32_t H, float v) {
    const float y = v * (float)H - 0.5f;
    const float y0f = floorf(y);
    return LutRows{tableOffset + clamp_texel(y0f, H) * 32u, tableOffset + clamp_texel(y0f + 1.0f, H) * 32u, y - y0f};
}
BRMI_DEV float sample_rows(const float* lut, const LutRows& r, float u) {
    const float x = u * 32.0f - 0.5f;
    const float x0f = floorf(x);
    const float fx = x - x0f;
    const uint32_t x0 = clamp_texel(x0f, 32), x1 = clamp_texel(x0f + 1.0f, 32);
    return lerpf(lerpf(lut[r.r0 + x0], lut[r.r0 + x1], fx), lerpf(lut[r.r1 + x0], lut[r.r1 + x1], fx), r.fy);
}
// lut_od_e with (ior, alpha) prepared
struct OdPrep { LutRows s0, s1; float st, ior; };
BRMI_DEV OdPrep prep_od_e(const Luts& L, float ior, float alpha) {
    const float ei = clamp_index(ior_to_index(ior)), ea = clamp_index(alpha_to_index(alpha));
    const int s0 = (int)floorf(ei);
    const int s1 = (s0 + 1) < 31 ? (s0 + 1) : 31;
    const float v = remap_index(ea);
    return OdPrep{prep_rows((uint32_t)s0 * 1024u, 32, v), prep_rows((uint32_t)s1 * 1024u, 32, v), ei - (float)s0, ior};      // odE starts the buffer
}
BRMI_DEV float sample_od_e(const Luts& L, const OdPrep& p, float cosT) {
    const float u = remap_index(clamp_index(cos_to_index(cosT)));
    return extrapolate_ior(lerpf(sample_rows(L.odE, p.s0, u), sample_rows(L.odE, p.s1, u), p.st), p.ior);
}
BRMI_DEV LutRows prep_im_e(const Luts& L, float alpha) { return prep_rows((uint32_t)(L.imE - L.odE), 32, remap_index(clamp_index(alpha_to_index(alpha)))); }
BRMI_DEV float sample_im_e(const Luts& L, const LutRows& r, float cosT) { return sample_rows(L.odE, r, remap_index(clamp_index(cos_to_index(cosT)))); }
BRMI_DEV float average_fresnel(float eta) {
    const float s = max2(eta, 1.0e-4f);
    if (s > 1.0f) return (s - 1.0f) / (4.08567f + 1.00071f * s);
    const float s2 = s * s;
    return 0.997118f + 0.1014f * s - 0.965241f * s2 - 0.130607f * s2 * s;
}


// What the shading pass needs from (OpenPBR material, 8-bit perceptual roughness code) alone.  For such a pair the opaque-dielectric
// energy complement is a function of the cosine only: lut_od_e's bilinear fetch in each of the two IOR slices, the lerp between the
// slices and the IOR extrapolation are all linear in the texels, so they are folded ONCE per pair into a row of 32 values
//     odRow[x] = extrapolate_ior(lerp(lerp(s0.row0[x], s0.row1[x], fy), lerp(s1.row0[x], s1.row1[x], fy), st), ior)
// and a sample is one lerp between two neighbours of that row (2 loads + ~12 instructions instead of 8 loads + ~45); likewise the
// ideal-metal complement (imRow, a function of the roughness code alone).  The column index and weight are computed exactly as
// sample_rows computes them; what changes is the order in which the (multilinear) interpolation is associated, i.e. fp32 rounding in
// the last bits -- tolerance-level, like the hardware rcp in the BRDF algebra.  Built by k_frame_constants with the first frame after
// brmi_setup.
// Round 5: a row entry is the PAIR {value, next value - value}: one 8 B fetch and one fused multiply-add per sample, and the column is the cosine's
// index itself -- remap_index() followed by sample_rows()'s `u * 32 - 0.5` is the identity on [0, 31] up to one rounding of 2^-25, which moves the
// sample by ~1e-7 of a table step (tolerance-level; the old form cost 25 VALU instructions per sample, this one 8).
struct ShadeRows { float2 od[32]; float2 im[32]; };                  // 512 B per (material, code)
struct ShadeAverages { float invAvgComp, invMAvgClamped; };        // 1 / max(lut_od_avg, 1e-12), 1 / max(lut_im_avg, 1e-12) of the pair (hardware reciprocal, as the pixel took it)
static_assert(sizeof(ShadeRows) == 512 && sizeof(ShadeAverages) == 8, "table layouts");
BRMI_DEV void build_shade_rows(const Luts& L, float ior, float alpha, ShadeRows& r, ShadeAverages& a) {
    const OdPrep od = prep_od_e(L, ior, alpha);
    const LutRows im = prep_im_e(L, alpha);
    float odv[32], imv[32];
    for (uint32_t x = 0; x < 32u; x++) {
        const float v0 = lerpf(L.odE[od.s0.r0 + x], L.odE[od.s0.r1 + x], od.s0.fy), v1 = lerpf(L.odE[od.s1.r0 + x], L.odE[od.s1.r1 + x], od.s1.fy);
        odv[x] = extrapolate_ior(lerpf(v0, v1, od.st), ior);
        imv[x] = lerpf(L.odE[im.r0 + x], L.odE[im.r1 + x], im.fy);
    }
    for (uint32_t x = 0; x < 32u; x++) {
        const uint32_t x1 = x < 31u ? x + 1u : 31u;
        r.od[x] = make_float2(odv[x], odv[x1] - odv[x]); r.im[x] = make_float2(imv[x], imv[x1] - imv[x]);
    }
    a.invAvgComp = qrcp(max2(lut_od_avg(L, ior, alpha), 1.0e-12f)); a.invMAvgClamped = qrcp(max2(lut_im_avg(L, alpha), 1.0e-12f));
}
// one sample of a folded row at cos(theta) (any finite value: saturated here)
BRMI_DEV float sample_folded_row(const float2* row, float cosT) {
    BRMI_FP_FAST
#ifdef BRMI_ABLATE_ROWS
    return cosT * 0.5f;      // (experiment: the table gathers gone; wrong image)
#endif
    const float x = satq(cosT) * TBL_M1;
    const float x0f = floorf(x);
    const float2 p = row[(uint32_t)x0f];
    return __builtin_fmaf(x - x0f, p.y, p.x);
}

}  // namespace brmi
#endif
