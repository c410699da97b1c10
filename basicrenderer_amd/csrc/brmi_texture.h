// brmi_texture.h -- UV streams, the software sampler and the raster alpha test for gfx950.
//
// Reference: ReadPackedBits32 / SWDecodeCompressedUV (BR/shaders/ClusterLOD/softwareRaster.hlsl:30-44,174-215),
// LoadMeshletUvDescriptor (BR/shaders/Include/clodPageAccess.hlsli:66-87), SWAlphaTestFailed (softwareRaster.hlsl:135-172),
// Sample2DGrad (BR/shaders/Include/utilities.hlsli:395-402).
// CDNA has no texture-filtering path for this use (HIP texture objects go through the TA/TD units with their own, unspecified,
// fixed-point weights): SampleLevel / SampleGrad are evaluated in shader arithmetic, as the Direct3D 11.3 functional spec
// describes an isotropic sampler, every step in IEEE fp32 (DESIGN.md "software sampler"):
//   texel = RGBA8 code / 255 (rgb of an _SRGB format through the injected 256-entry decode table, before filtering);
//   addressing on integer texel coordinates (wrap, mirror, clamp); point = floor(u * w); linear = the 2x2 footprint around
//   u * w - 0.5 blended a + t * (b - a) along x then y; SampleGrad's lod = 0.5 * log2(max(|ddx * size|^2, |ddy * size|^2)) with
//   log2 = exponent + degree-5 mantissa polynomial; lod biased, clamped to the sampler's range and the mip chain; lod <= 0 takes
//   the mag filter; mip filter point = nearest level, linear = two levels blended by the fraction.  Anisotropy is not reproduced.
#ifndef BRMI_TEXTURE_H
#define BRMI_TEXTURE_H

#include "brmi_device.h"

namespace brmi {

// where UV set 0 of a visible cluster lives (written by the compaction kernel next to ClusterSetup); desc == nullptr: the page has no UV set
struct ClusterUv { const uint8_t* desc; const uint8_t* stream; };

BRMI_DEV uint32_t read_packed_bits32(const uint8_t* stream, uint32_t startBit, uint32_t bitCount) {
    if (bitCount == 0u) return 0u;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(stream);
    const uint32_t wordIndex = startBit >> 5, bitOffset = startBit & 31u;
    uint32_t packed = w[wordIndex] >> bitOffset;
    if (bitOffset + bitCount > 32u) packed |= w[wordIndex + 1u] << (32u - bitOffset);
    const uint32_t mask = bitCount >= 32u ? 0xFFFFFFFFu : ((1u << bitCount) - 1u);
    return packed & mask;
}

BRMI_DEV f2 decode_uv(const ClusterUv& cu, uint32_t vertex) {
    if (cu.desc == nullptr) return {0.0f, 0.0f};
    const uint4 d0 = *reinterpret_cast<const uint4*>(cu.desc);
    const uint2 d1 = *reinterpret_cast<const uint2*>(cu.desc + 16);
    const uint32_t bitsU = d1.y & 0xFFu, bitsV = (d1.y >> 8) & 0xFFu;
    uint32_t cursor = d0.x + vertex * (bitsU + bitsV);
    const uint32_t eu = read_packed_bits32(cu.stream, cursor, bitsU);
    cursor += bitsU;
    const uint32_t ev = read_packed_bits32(cu.stream, cursor, bitsV);
    return {as_f32(d0.y) + (float)eu * as_f32(d0.w), as_f32(d0.z) + (float)ev * as_f32(d1.x)};
}

BRMI_DEV int floor_to_int(float f) { return to_int_sat(floorf(f)); }
BRMI_DEV int address_texel(int i, int n, uint32_t mode) {
    if (mode == BRMI_ADDRESS_CLAMP) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    if (mode == BRMI_ADDRESS_MIRROR) {
        const int p = 2 * n;
        int t = i % p; if (t < 0) t += p;
        return t < n ? t : p - 1 - t;
    }
    int t = i % n; if (t < 0) t += n;
    return t;
}
BRMI_DEV f4 lerp4(f4 a, f4 b, float t) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)}; }

struct TexView { const uint32_t* texels; uint32_t width, height, mipCount; bool srgb; const uint32_t* mipOffset; const float* srgbToLinear; };
BRMI_DEV f4 fetch_texel(const TexView& tx, uint32_t level, int x, int y) {
    const uint32_t w = tx.width >> level ? tx.width >> level : 1u;
    const uint32_t c = tx.texels[(size_t)tx.mipOffset[level] + (size_t)y * w + (size_t)x];
    f4 r;
    if (tx.srgb) { r.x = tx.srgbToLinear[c & 0xFFu]; r.y = tx.srgbToLinear[(c >> 8) & 0xFFu]; r.z = tx.srgbToLinear[(c >> 16) & 0xFFu]; }
    else { r.x = (float)(c & 0xFFu) / 255.0f; r.y = (float)((c >> 8) & 0xFFu) / 255.0f; r.z = (float)((c >> 16) & 0xFFu) / 255.0f; }
    r.w = (float)(c >> 24) / 255.0f;
    return r;
}

BRMI_DEV f4 sample_level_filtered(const TexView& tx, const brmi_sampler_desc& sm, uint32_t level, f2 uv, uint32_t filter) {
    const int w = (int)(tx.width >> level ? tx.width >> level : 1u), h = (int)(tx.height >> level ? tx.height >> level : 1u);
    if (filter == BRMI_FILTER_POINT)
        return fetch_texel(tx, level, address_texel(floor_to_int(uv.x * (float)w), w, sm.addressU), address_texel(floor_to_int(uv.y * (float)h), h, sm.addressV));
    const float fx = uv.x * (float)w - 0.5f, fy = uv.y * (float)h - 0.5f;
    const float tx_ = fx - floorf(fx), ty_ = fy - floorf(fy);
    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
    const int xa = address_texel(x0, w, sm.addressU), xb = address_texel(x0 == 0x7FFFFFFF ? x0 : x0 + 1, w, sm.addressU);
    const int ya = address_texel(y0, h, sm.addressV), yb = address_texel(y0 == 0x7FFFFFFF ? y0 : y0 + 1, h, sm.addressV);
    const f4 t00 = fetch_texel(tx, level, xa, ya), t10 = fetch_texel(tx, level, xb, ya), t01 = fetch_texel(tx, level, xa, yb), t11 = fetch_texel(tx, level, xb, yb);
    return lerp4(lerp4(t00, t10, tx_), lerp4(t01, t11, tx_), ty_);
}

// Texture2D::SampleLevel.  An unbound slot reads as opaque white.
BRMI_DEV f4 sample_level(const brmi_scene_buffers& sc, uint32_t textureIndex, uint32_t samplerIndex, f2 uv, float lodIn) {
    if (textureIndex >= sc.textureCount || samplerIndex >= sc.samplerCount) return {1.0f, 1.0f, 1.0f, 1.0f};
    const brmi_texture_desc* td = sc.textures + textureIndex;
    const brmi_sampler_desc sm = sc.samplers[samplerIndex];
    const TexView tx{reinterpret_cast<const uint32_t*>(td->texels), td->width, td->height, td->mipCount, td->format == BRMI_TEXTURE_FORMAT_RGBA8_UNORM_SRGB, td->mipOffset, sc.srgbToLinear};
    float lod = min2(max2(lodIn + sm.mipLodBias, sm.minLod), sm.maxLod);
    lod = min2(max2(lod, 0.0f), (float)(tx.mipCount - 1u));
    const uint32_t filter = lod <= 0.0f ? sm.magFilter : sm.minFilter;
    if (sm.mipFilter == BRMI_FILTER_POINT) {
        uint32_t level = (uint32_t)floor_to_int(lod + 0.5f);
        if (level > tx.mipCount - 1u) level = tx.mipCount - 1u;
        return sample_level_filtered(tx, sm, level, uv, filter);
    }
    const uint32_t l0 = (uint32_t)floor_to_int(lod);
    const float frac = lod - floorf(lod);
    const f4 a = sample_level_filtered(tx, sm, l0, uv, filter);
    if (frac == 0.0f) return a;
    const uint32_t l1 = l0 + 1u > tx.mipCount - 1u ? tx.mipCount - 1u : l0 + 1u;
    return lerp4(a, sample_level_filtered(tx, sm, l1, uv, filter), frac);
}

BRMI_DEV float log2_poly(float x) {
    const uint32_t b = as_u32(x);
    const int e = (int)((b >> 23) & 0xFFu) - 127;
    const float t = as_f32((b & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
    const float p = t * (1.442609190940857f + t * (-0.7168022990226746f + t * (0.44070422649383545f + t * (-0.2247820496559143f + t * 0.05827096104621887f))));
    return (float)e + p;
}
// Texture2D::SampleGrad
BRMI_DEV f4 sample_grad(const brmi_scene_buffers& sc, uint32_t textureIndex, uint32_t samplerIndex, f2 uv, f2 dUVdx, f2 dUVdy) {
    if (textureIndex >= sc.textureCount || samplerIndex >= sc.samplerCount) return {1.0f, 1.0f, 1.0f, 1.0f};
    const float W = (float)sc.textures[textureIndex].width, H = (float)sc.textures[textureIndex].height;
    const float dxx = dUVdx.x * W, dxy = dUVdx.y * H, dyx = dUVdy.x * W, dyy = dUVdy.y * H;
    const float rho2 = max2(dxx * dxx + dxy * dxy, dyx * dyx + dyy * dyy);
    float lod;
    if (!(rho2 >= 1.17549435e-38f)) lod = -127.0f;
    else if (rho2 > 3.0e38f) lod = 128.0f;
    else lod = 0.5f * log2_poly(rho2);
    return sample_level(sc, textureIndex, samplerIndex, uv, lod);
}

// SWAlphaTestFailed (CLOD_SW_RASTER_DYNAMIC_ALPHA_TEST); the caller has already checked MATERIAL_ALPHA_TEST
struct AlphaMaterial { uint32_t flags, baseTex, baseSamp, opTex, opSamp; float alphaFactor, cutoff; };
BRMI_DEV AlphaMaterial load_alpha_material(const brmi_scene_buffers& sc, uint32_t materialDataIndex) {
    const brmi_material_info* m = sc.materials + materialDataIndex;
    return {m->materialFlags, m->baseColorTextureIndex, m->baseColorSamplerIndex, m->opacityTextureIndex, m->opacitySamplerIndex, m->baseColorFactor[3], m->alphaCutoff};
}
BRMI_DEV bool alpha_test_failed(const brmi_scene_buffers& sc, const AlphaMaterial& m, f2 uv) {
    float alpha = m.alphaFactor;
    if (m.flags & BRMI_MATERIAL_BASE_COLOR_TEXTURE) alpha *= sample_level(sc, m.baseTex, m.baseSamp, uv, 0.0f).w;
    if (m.flags & BRMI_MATERIAL_OPACITY_TEXTURE) alpha *= sample_level(sc, m.opTex, m.opSamp, uv, 0.0f).w;
    return alpha < m.cutoff;
}
// the texcoord of a pixel from the stepped barycentrics (softwareRaster.hlsl:526-531)
struct AlphaTri { float invW0, invW1, invW2; f2 uv0, uv1, uv2; };
BRMI_DEV f2 pixel_texcoord(const AlphaTri& t, float b0, float b1, float b2) {
    const float pc0 = b0 * t.invW0, pc1 = b1 * t.invW1, pc2 = b2 * t.invW2;
    const float invSum = 1.0f / (pc0 + pc1 + pc2);
    return {((t.uv0.x * pc0 + t.uv1.x * pc1) + t.uv2.x * pc2) * invSum, ((t.uv0.y * pc0 + t.uv1.y * pc1) + t.uv2.y * pc2) * invSum};
}

}  // namespace brmi
#endif
