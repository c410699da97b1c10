// orc_texture.h -- CPU restatement of the UV streams, the alpha test and the material texture fetches.
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows:
//   ReadPackedBits32 / SWDecodeCompressedUV       BR/shaders/ClusterLOD/softwareRaster.hlsl:30-44,174-215
//   LoadMeshletUvDescriptor / LoadPageUvBitstreamOffset   BR/shaders/Include/clodPageAccess.hlsli:66-93
//   SWAlphaTestFailed                             BR/shaders/ClusterLOD/softwareRaster.hlsl:135-172
//   Sample2DGrad / SampleMaterialTexture2DGrad    BR/shaders/Include/utilities.hlsli:395-402,472-502
//
// The fixed-function sampler behind SampleLevel / SampleGrad is not in the reference's sources (it is the GPU's).  It is
// restated here -- and in brmi_device.h, by specification -- as the Direct3D 11.3 functional spec describes an isotropic
// sampler (7.18.7-7.18.11), with every step in IEEE fp32 so that CPU and GPU agree bit for bit:
//   * texel (x, y) of level l: RGBA8 at texels + (mipOffset[l] + y * w_l + x) * 4, w_l = max(1, width >> l); a channel is
//     code / 255.0f; rgb of an _SRGB format goes through srgbToLinear[code] BEFORE filtering;
//   * addressing on integer texel coordinates: wrap = mod n, mirror = reflect with period 2n, clamp = [0, n - 1];
//   * point: texel (floor(u * w_l), floor(v * h_l));  linear: f = u * w_l - 0.5, x0 = floor(f), t = f - x0, the four texels
//     (x0, x0 + 1) x (y0, y0 + 1) blended a + t * (b - a), first along x, then along y;
//   * LOD of SampleGrad: rho^2 = max(|ddx * (W, H)|^2, |ddy * (W, H)|^2), lod = 0.5 * log2(rho^2) with log2 = exponent +
//     a degree-5 polynomial of the mantissa (|error| < 7e-5; hardware keeps 8 fractional LOD bits);
//   * lod = clamp(lod + mipLodBias, minLod, maxLod), then to [0, mipCount - 1]; lod <= 0 selects magFilter, else minFilter;
//     mipFilter point: level = floor(lod + 0.5); linear: levels floor(lod) and + 1 blended by the fraction;
//   * anisotropic filtering (glTF materials ask for 16x, GlTFLoader.cpp:865) is implementation-defined and not reproduced.
#ifndef ORC_TEXTURE_H
#define ORC_TEXTURE_H

#include <climits>

#include "orc_common.h"

namespace orc {

// ReadPackedBits32 (softwareRaster.hlsl:30-44); `byteBase` is where bit 0 lives
inline uint32_t readPackedBits32(const uint8_t* slab, uint32_t startBit, uint32_t bitCount) {
    if (bitCount == 0u) return 0u;
    const uint32_t wordIndex = startBit >> 5, bitOffset = startBit & 31u;
    uint32_t packed = load32(slab, wordIndex * 4u) >> bitOffset;
    if (bitOffset + bitCount > 32u) packed |= load32(slab, (wordIndex + 1u) * 4u) << (32u - bitOffset);
    const uint32_t mask = bitCount >= 32u ? 0xFFFFFFFFu : ((1u << bitCount) - 1u);
    return packed & mask;
}

// SWDecodeCompressedUV (softwareRaster.hlsl:174-215) == DecodeCompressedUV (clodResolveCommon.hlsli:226-264)
inline float2 decodeCompressedUV(const uint8_t* slab, uint32_t pageOff, const brmi_page_header& hdr, uint32_t localMeshlet, uint32_t uvSetIndex, uint32_t vertex) {
    if (uvSetIndex >= hdr.uvSetCount) return {0.0f, 0.0f};
    brmi_meshlet_uv_descriptor d;
    std::memcpy(&d, slab + pageOff + hdr.uvDescriptorOffset + (localMeshlet * hdr.uvSetCount + uvSetIndex) * 32u, 32);
    const uint32_t streamBase = pageOff + load32(slab, pageOff + hdr.uvBitstreamDirectoryOffset + uvSetIndex * 4u);
    const uint32_t bitsU = d.uvBits & 0xFFu, bitsV = (d.uvBits >> 8) & 0xFFu;
    uint32_t cursor = streamBase * 8u + d.uvBitOffset + vertex * (bitsU + bitsV);
    const uint32_t eu = readPackedBits32(slab, cursor, bitsU);
    cursor += bitsU;
    const uint32_t ev = readPackedBits32(slab, cursor, bitsV);
    return {d.uvMinU + (float)eu * d.uvScaleU, d.uvMinV + (float)ev * d.uvScaleV};
}

// ---- software sampler ----------------------------------------------------------------------------------------------
inline int floorToInt(float f) {
    const float fl = std::floor(f);
    if (!(fl == fl)) return 0;
    if (fl >= 2147483648.0f) return INT_MAX;
    if (fl <= -2147483648.0f) return INT_MIN;
    return (int)fl;
}
inline int addressTexel(int i, int n, uint32_t mode) {
    if (mode == BRMI_ADDRESS_CLAMP) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    if (mode == BRMI_ADDRESS_MIRROR) {
        const int p = 2 * n;
        int t = i % p; if (t < 0) t += p;
        return t < n ? t : p - 1 - t;
    }
    int t = i % n; if (t < 0) t += n;
    return t;
}
inline float4 fetchTexel(const brmi_scene_buffers& sc, const brmi_texture_desc& tx, uint32_t level, int x, int y) {
    const uint32_t w = tx.width >> level ? tx.width >> level : 1u;
    const uint8_t* t = tx.texels + ((size_t)tx.mipOffset[level] + (size_t)y * w + (size_t)x) * 4u;
    float4 r;
    if (tx.format == BRMI_TEXTURE_FORMAT_RGBA8_UNORM_SRGB) { r.x = sc.srgbToLinear[t[0]]; r.y = sc.srgbToLinear[t[1]]; r.z = sc.srgbToLinear[t[2]]; }
    else { r.x = (float)t[0] / 255.0f; r.y = (float)t[1] / 255.0f; r.z = (float)t[2] / 255.0f; }
    r.w = (float)t[3] / 255.0f;
    return r;
}
inline float4 lerp4(float4 a, float4 b, float t) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)}; }

inline float4 sampleLevelFiltered(const brmi_scene_buffers& sc, const brmi_texture_desc& tx, const brmi_sampler_desc& sm, uint32_t level, float2 uv, uint32_t filter) {
    const int w = (int)(tx.width >> level ? tx.width >> level : 1u), h = (int)(tx.height >> level ? tx.height >> level : 1u);
    if (filter == BRMI_FILTER_POINT) {
        const int x = addressTexel(floorToInt(uv.x * (float)w), w, sm.addressU), y = addressTexel(floorToInt(uv.y * (float)h), h, sm.addressV);
        return fetchTexel(sc, tx, level, x, y);
    }
    const float fx = uv.x * (float)w - 0.5f, fy = uv.y * (float)h - 0.5f;
    const float flx = std::floor(fx), fly = std::floor(fy);
    const float tx_ = fx - flx, ty_ = fy - fly;
    const int x0 = floorToInt(fx), y0 = floorToInt(fy);
    const int xa = addressTexel(x0, w, sm.addressU), xb = addressTexel(x0 == INT_MAX ? x0 : x0 + 1, w, sm.addressU);
    const int ya = addressTexel(y0, h, sm.addressV), yb = addressTexel(y0 == INT_MAX ? y0 : y0 + 1, h, sm.addressV);
    const float4 top = lerp4(fetchTexel(sc, tx, level, xa, ya), fetchTexel(sc, tx, level, xb, ya), tx_);
    const float4 bot = lerp4(fetchTexel(sc, tx, level, xa, yb), fetchTexel(sc, tx, level, xb, yb), tx_);
    return lerp4(top, bot, ty_);
}

// Texture2D::SampleLevel
inline float4 sampleLevel(const brmi_scene_buffers& sc, uint32_t textureIndex, uint32_t samplerIndex, float2 uv, float lodIn) {
    if (textureIndex >= sc.textureCount || samplerIndex >= sc.samplerCount) return {1.0f, 1.0f, 1.0f, 1.0f};   // unbound slot
    const brmi_texture_desc& tx = sc.textures[textureIndex];
    const brmi_sampler_desc& sm = sc.samplers[samplerIndex];
    float lod = fmin2(fmax2(lodIn + sm.mipLodBias, sm.minLod), sm.maxLod);
    lod = fmin2(fmax2(lod, 0.0f), (float)(tx.mipCount - 1u));
    const uint32_t filter = lod <= 0.0f ? sm.magFilter : sm.minFilter;
    if (sm.mipFilter == BRMI_FILTER_POINT) {
        uint32_t level = (uint32_t)floorToInt(lod + 0.5f);
        if (level > tx.mipCount - 1u) level = tx.mipCount - 1u;
        return sampleLevelFiltered(sc, tx, sm, level, uv, filter);
    }
    const float fl = std::floor(lod);
    const uint32_t l0 = (uint32_t)floorToInt(lod);
    const float frac = lod - fl;
    const float4 a = sampleLevelFiltered(sc, tx, sm, l0, uv, filter);
    if (frac == 0.0f) return a;                               // a + 0 * (b - a)
    const uint32_t l1 = l0 + 1u > tx.mipCount - 1u ? tx.mipCount - 1u : l0 + 1u;
    return lerp4(a, sampleLevelFiltered(sc, tx, sm, l1, uv, filter), frac);
}

// log2 of a positive float: exponent + polynomial of the mantissa (plain multiplies and adds, reproducible everywhere)
inline float log2Poly(float x) {
    const uint32_t b = asuint(x);
    const int e = (int)((b >> 23) & 0xFFu) - 127;
    const float t = asfloat((b & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
    const float p = t * (1.442609190940857f + t * (-0.7168022990226746f + t * (0.44070422649383545f + t * (-0.2247820496559143f + t * 0.05827096104621887f))));
    return (float)e + p;
}
// Texture2D::SampleGrad
inline float4 sampleGrad(const brmi_scene_buffers& sc, uint32_t textureIndex, uint32_t samplerIndex, float2 uv, float2 dUVdx, float2 dUVdy) {
    if (textureIndex >= sc.textureCount || samplerIndex >= sc.samplerCount) return {1.0f, 1.0f, 1.0f, 1.0f};
    const brmi_texture_desc& tx = sc.textures[textureIndex];
    const float W = (float)tx.width, H = (float)tx.height;
    const float2 dx{dUVdx.x * W, dUVdx.y * H}, dy{dUVdy.x * W, dUVdy.y * H};
    const float rho2 = fmax2(dot(dx, dx), dot(dy, dy));
    float lod;
    if (!(rho2 >= 1.17549435e-38f)) lod = -127.0f;          // zero, denormal or NaN footprint: the finest level
    else if (rho2 > 3.0e38f) lod = 128.0f;
    else lod = 0.5f * log2Poly(rho2);
    return sampleLevel(sc, textureIndex, samplerIndex, uv, lod);
}

// SWAlphaTestFailed with CLOD_SW_RASTER_DYNAMIC_ALPHA_TEST (softwareRaster.hlsl:135-172)
inline bool alphaTestFailed(const brmi_scene_buffers& sc, float2 uv, uint32_t materialDataIndex) {
    const brmi_material_info& m = sc.materials[materialDataIndex];
    if ((m.materialFlags & BRMI_MATERIAL_ALPHA_TEST) == 0u) return false;
    float alpha = m.baseColorFactor[3];
    if (m.materialFlags & BRMI_MATERIAL_BASE_COLOR_TEXTURE) alpha *= sampleLevel(sc, m.baseColorTextureIndex, m.baseColorSamplerIndex, uv, 0.0f).w;
    if (m.materialFlags & BRMI_MATERIAL_OPACITY_TEXTURE) alpha *= sampleLevel(sc, m.opacityTextureIndex, m.opacitySamplerIndex, uv, 0.0f).w;
    return alpha < m.alphaCutoff;
}

}  // namespace orc
#endif
