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

// where the UV sets and the vertex colours of a visible cluster live (written by the compaction kernel next to ClusterSetup); desc == nullptr: the page has no UV set.
// desc / stream address set 0 (the set of the rasteriser's alpha test and of nearly every material); the others are reached through the page's bitstream
// directory: `directory` = its offset from desc, `setCount` = CLodPageHeader::uvSetCount.
struct ClusterUv { const uint8_t* desc; const uint8_t* stream; const uint8_t* color; int32_t directory; uint32_t setCount; };      // color: the meshlet's RGBA8 vertex colours (nullptr: none)

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

// DecodeCompressedUV for any set (clodResolveCommon.hlsli:612-655): descriptors are [meshlet][set], one bitstream per set behind the directory; a set the
// page does not carry reads (0, 0)
BRMI_DEV f2 decode_uv_set(const ClusterUv& cu, uint32_t set, uint32_t vertex) {
    if (set == 0u) return decode_uv(cu, vertex);
    if (cu.desc == nullptr || set >= cu.setCount) return {0.0f, 0.0f};
    const uint32_t* dir = reinterpret_cast<const uint32_t*>(cu.desc + cu.directory);
    ClusterUv s = cu;
    s.desc = cu.desc + set * 32u;
    s.stream = cu.stream - dir[0] + dir[set];
    return decode_uv(s, vertex);
}

BRMI_DEV int floor_to_int(float f) { return to_int_sat(floorf(f)); }
// n > 0.  Power-of-two sizes (nearly every texture) wrap / mirror with a mask; the result is the same as the general modulo.
BRMI_DEV int address_texel(int i, int n, uint32_t mode) {
    if (mode == BRMI_ADDRESS_CLAMP) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    const bool pow2 = (n & (n - 1)) == 0;
    if (mode == BRMI_ADDRESS_MIRROR) {
        const int p = 2 * n;
        int t;
        if (pow2) t = i & (p - 1); else { t = i % p; if (t < 0) t += p; }
        return t < n ? t : p - 1 - t;
    }
    if (pow2) return i & (n - 1);
    int t = i % n; if (t < 0) t += n;
    return t;
}
BRMI_DEV f4 lerp4(f4 a, f4 b, float t) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)}; }

// code -> float tables: unorm[c] = c / 255.0f, srgb[c] = the injected sRGB decode.  Kernels stage both in LDS (512 floats):
// a texel costs four table reads instead of four correctly rounded divisions / three global gathers.
struct TexelTables { const float* t; };        // [0, 256) unorm, [256, 512) sRGB decode
BRMI_DEV void stage_texel_tables(float* lds512, const float* srgbToLinear, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < 256u; i += nthreads) { lds512[i] = (float)i / 255.0f; lds512[256u + i] = srgbToLinear ? srgbToLinear[i] : 0.0f; }
}

// one texture slot as the sampler sees it: descriptor and sampler state, fetched once per material (scalar loads when the
// material index is wave-uniform: pass the tables through kconst())
struct TexBinding { const uint32_t* texels; const uint32_t* mipOffset; uint32_t width, height, mipCount; bool srgb, bound; brmi_sampler_desc sm;
                    bool waveUniform; };      // the descriptors came through the constant address space (scalar loads): every field is the same in all lanes
template <typename T> struct is_kconst_ptr { static constexpr bool value = false; };
template <typename T> struct is_kconst_ptr<const __attribute__((address_space(4))) T*> { static constexpr bool value = true; };
template <typename TexPtr, typename SampPtr>
BRMI_DEV TexBinding bind_texture(TexPtr textures, uint32_t textureCount, SampPtr samplers, uint32_t samplerCount, uint32_t textureIndex, uint32_t samplerIndex) {
    TexBinding b{};
    b.waveUniform = is_kconst_ptr<TexPtr>::value && is_kconst_ptr<SampPtr>::value;
    b.bound = textureIndex < textureCount && samplerIndex < samplerCount;
    if (!b.bound) return b;
    const auto* td = textures + textureIndex;
    const auto* sd = samplers + samplerIndex;
    b.texels = reinterpret_cast<const uint32_t*>(td->texels); b.mipOffset = (const uint32_t*)td->mipOffset;
    b.width = td->width; b.height = td->height; b.mipCount = td->mipCount; b.srgb = td->format == BRMI_TEXTURE_FORMAT_RGBA8_UNORM_SRGB;
    b.sm.addressU = sd->addressU; b.sm.addressV = sd->addressV; b.sm.minFilter = sd->minFilter; b.sm.magFilter = sd->magFilter; b.sm.mipFilter = sd->mipFilter;
    b.sm.mipLodBias = sd->mipLodBias; b.sm.minLod = sd->minLod; b.sm.maxLod = sd->maxLod;
    return b;
}
BRMI_DEV bool same_binding(const TexBinding& a, const TexBinding& b) {
    return a.bound && b.bound && a.texels == b.texels && a.sm.addressU == b.sm.addressU && a.sm.addressV == b.sm.addressV && a.sm.minFilter == b.sm.minFilter && a.sm.magFilter == b.sm.magFilter &&
           a.sm.mipFilter == b.sm.mipFilter && a.sm.mipLodBias == b.sm.mipLodBias && a.sm.minLod == b.sm.minLod && a.sm.maxLod == b.sm.maxLod;
}

// texel memory is HBM: say so, a pointer read out of a descriptor is otherwise loaded through the flat path
typedef const __attribute__((address_space(1))) uint32_t* GlobalTexels;
BRMI_DEV GlobalTexels as_global(const uint32_t* p) { return (GlobalTexels)p; }
// The 2x2 (or single-texel) footprint of one level: addresses and weights first, the four texel words requested together.
struct Footprint { uint32_t c00, c10, c01, c11; float tx, ty; };
BRMI_DEV Footprint fetch_footprint(const TexBinding& tx, uint32_t levelOffset, uint32_t level, f2 uv, uint32_t filter) {
    const int w = (int)(tx.width >> level ? tx.width >> level : 1u), h = (int)(tx.height >> level ? tx.height >> level : 1u);
    GlobalTexels base = as_global(tx.texels + levelOffset);
    Footprint f;
    if (filter == BRMI_FILTER_POINT) {
        const int x = address_texel(floor_to_int(uv.x * (float)w), w, tx.sm.addressU), y = address_texel(floor_to_int(uv.y * (float)h), h, tx.sm.addressV);
        f.c00 = f.c10 = f.c01 = f.c11 = base[(size_t)y * (size_t)w + (size_t)x]; f.tx = 0.0f; f.ty = 0.0f;      // a + 0 * (a - a) = a
        return f;
    }
    const float fx = uv.x * (float)w - 0.5f, fy = uv.y * (float)h - 0.5f;
    f.tx = fx - floorf(fx); f.ty = fy - floorf(fy);
    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
    const int xa = address_texel(x0, w, tx.sm.addressU), xb = address_texel(x0 == 0x7FFFFFFF ? x0 : x0 + 1, w, tx.sm.addressU);
    const int ya = address_texel(y0, h, tx.sm.addressV), yb = address_texel(y0 == 0x7FFFFFFF ? y0 : y0 + 1, h, tx.sm.addressV);
    f.c00 = base[(size_t)ya * (size_t)w + (size_t)xa]; f.c10 = base[(size_t)ya * (size_t)w + (size_t)xb];
    f.c01 = base[(size_t)yb * (size_t)w + (size_t)xa]; f.c11 = base[(size_t)yb * (size_t)w + (size_t)xb];
    return f;
}
BRMI_DEV f4 decode_texel(const TexelTables& tb, uint32_t c, bool srgb) {
    const uint32_t o = srgb ? 256u : 0u;      // an offset, not a second pointer: the reads stay LDS reads
    return {tb.t[o + (c & 0xFFu)], tb.t[o + ((c >> 8) & 0xFFu)], tb.t[o + ((c >> 16) & 0xFFu)], tb.t[c >> 24]};
}
BRMI_DEV f4 filter_footprint(const TexelTables& tb, const Footprint& f, bool srgb) {
    return lerp4(lerp4(decode_texel(tb, f.c00, srgb), decode_texel(tb, f.c10, srgb), f.tx), lerp4(decode_texel(tb, f.c01, srgb), decode_texel(tb, f.c11, srgb), f.tx), f.ty);
}

// Texture2D::SampleLevel.  An unbound slot reads as opaque white.  Both mip offsets are requested together, then both footprints:
// two memory round trips per sample instead of four.  (When no lane of the wave blends levels the second footprint is skipped.)
// The part that depends on the level of detail alone -- clamps, filter choice, the two mip levels and their offsets -- is a LevelSetup:
// a caller that samples one texture many times at one LOD (the parallax march: 32 fetches with the pixel's gradients) prepares it once.
struct LevelSetup { uint32_t l0, l1, off0, off1, filter; float frac; };
BRMI_DEV LevelSetup prepare_level(const TexBinding& tx, float lodIn) {
    LevelSetup s;
    float lod = min2(max2(lodIn + tx.sm.mipLodBias, tx.sm.minLod), tx.sm.maxLod);
    lod = min2(max2(lod, 0.0f), (float)(tx.mipCount - 1u));
    s.filter = lod <= 0.0f ? tx.sm.magFilter : tx.sm.minFilter;
    if (tx.sm.mipFilter == BRMI_FILTER_POINT) {
        s.l0 = (uint32_t)floor_to_int(lod + 0.5f);
        if (s.l0 > tx.mipCount - 1u) s.l0 = tx.mipCount - 1u;
        s.l1 = s.l0; s.frac = 0.0f;
    } else {
        s.l0 = (uint32_t)floor_to_int(lod); s.frac = lod - floorf(lod);
        s.l1 = s.l0 + 1u > tx.mipCount - 1u ? tx.mipCount - 1u : s.l0 + 1u;
    }
    s.off0 = as_global(tx.mipOffset)[s.l0]; s.off1 = as_global(tx.mipOffset)[s.l1];
    return s;
}
// ---- the common case without branches (round 4) ---------------------------------------------------------------------------------------
// A wave-uniform binding (the G-buffer kernel's waterfall path: one material per step) of a texture with power-of-two sides, filtered
// linearly.  fetch_footprint decides per coordinate -- address mode, power of two or modulo, saturation of x0 + 1 -- with branches the
// compiler turns into exec-mask sections: ~600 VALU instructions and ~60 branches per trilinear sample, measured.  Here the address modes
// become three per-axis constants of level 0, shifted down per level (the level is per lane), and a coordinate is
//     c = clamp(i & mask, 0, hi);  index = min(c, fold - c)
//   wrap:   mask = n - 1,  hi = n - 1,  fold = INT_MAX  (no fold)            i & (n - 1)
//   clamp:  mask = ~0,     hi = n - 1,  fold = INT_MAX                       clamp(i, 0, n - 1)
//   mirror: mask = 2n - 1, hi = 2n - 1, fold = 2n - 1                        t = i & (2n - 1); t < n ? t : 2n - 1 - t
// -- the values address_texel returns, for every 32-bit i (two's complement `&` is the mathematical modulo of a power of two); shifting
// the level-0 constants right by the level (arithmetically: ~0 stays ~0) gives the level's, also where a side has shrunk to one texel.
// Texel indices fit 32 bits (sides <= 16384: the chain has < 2^29 texels), so addresses are a scalar base and a 32-bit byte offset.
struct Pow2Axis { int mask0, hi0, fold0; };
BRMI_DEV Pow2Axis pow2_axis(uint32_t n, uint32_t mode) {
    const int nm1 = (int)n - 1, m2 = (int)(2u * n) - 1;
    if (mode == BRMI_ADDRESS_MIRROR) return {m2, m2, m2};
    if (mode == BRMI_ADDRESS_CLAMP) return {-1, nm1, 0x7FFFFFFF};
    return {nm1, nm1, 0x7FFFFFFF};
}
BRMI_DEV bool pow2_fast_path(const TexBinding& tx) {
    return tx.waveUniform && (tx.width & (tx.width - 1u)) == 0u && (tx.height & (tx.height - 1u)) == 0u && tx.width <= 16384u && tx.height <= 16384u && tx.width != 0u && tx.height != 0u &&
           tx.sm.minFilter == BRMI_FILTER_LINEAR && tx.sm.magFilter == BRMI_FILTER_LINEAR;
}
BRMI_DEV int address_pow2(int i, int mask, int hi, int fold) { int c = i & mask; c = c < 0 ? 0 : c; c = c > hi ? hi : c; const int m = fold - c; return c < m ? c : m; }
BRMI_DEV int inc_sat(int x) { return x == 0x7FFFFFFF ? x : x + 1; }
BRMI_DEV Footprint fetch_footprint_pow2(const TexBinding& tx, const Pow2Axis& ax, const Pow2Axis& ay, uint32_t levelOffset, uint32_t level, f2 uv) {
    const uint32_t ws = tx.width >> level, hs = tx.height >> level;
    const uint32_t w = ws ? ws : 1u, h = hs ? hs : 1u;
    const float fx = uv.x * (float)(int)w - 0.5f, fy = uv.y * (float)(int)h - 0.5f;
    Footprint f;
    f.tx = fx - floorf(fx); f.ty = fy - floorf(fy);
    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
    const int mx = ax.mask0 >> level, hx = ax.hi0 >> level, ox = ax.fold0 >> level, my = ay.mask0 >> level, hy = ay.hi0 >> level, oy = ay.fold0 >> level;
    const uint32_t xa = (uint32_t)address_pow2(x0, mx, hx, ox), xb = (uint32_t)address_pow2(inc_sat(x0), mx, hx, ox);
    const uint32_t ya = (uint32_t)address_pow2(y0, my, hy, oy), yb = (uint32_t)address_pow2(inc_sat(y0), my, hy, oy);
    // rows and columns are < 2^14: 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction)
    const uint32_t rowA = __umul24(ya, w) + levelOffset, rowB = __umul24(yb, w) + levelOffset;
    const __attribute__((address_space(1))) char* base = (const __attribute__((address_space(1))) char*)tx.texels;
    auto texel = [&](uint32_t index) { return *(GlobalTexels)(base + (uint32_t)(index << 2)); };      // scalar base + 32-bit offset
    f.c00 = texel(rowA + xa); f.c10 = texel(rowA + xb); f.c01 = texel(rowB + xa); f.c11 = texel(rowB + xb);
    return f;
}
BRMI_DEV f4 sample_prepared_pow2(const TexelTables& tb, const TexBinding& tx, const LevelSetup& s, f2 uv) {
    const Pow2Axis ax = pow2_axis(tx.width, tx.sm.addressU), ay = pow2_axis(tx.height, tx.sm.addressV);      // scalar: a handful of SALU instructions per sample
    const Footprint f0 = fetch_footprint_pow2(tx, ax, ay, s.off0, s.l0, uv);
    if (!__any(s.frac != 0.0f)) return filter_footprint(tb, f0, tx.srgb);
    const Footprint f1 = fetch_footprint_pow2(tx, ax, ay, s.off1, s.l1, uv);
    const f4 a = filter_footprint(tb, f0, tx.srgb);
    return s.frac == 0.0f ? a : lerp4(a, filter_footprint(tb, f1, tx.srgb), s.frac);
}
BRMI_DEV f4 sample_prepared(const TexelTables& tb, const TexBinding& tx, const LevelSetup& s, f2 uv) {
    if (pow2_fast_path(tx)) return sample_prepared_pow2(tb, tx, s, uv);      // (a scalar branch; not compiled at all for per-lane bindings)
    const Footprint f0 = fetch_footprint(tx, s.off0, s.l0, uv, s.filter);
    if (!__any(s.frac != 0.0f)) return filter_footprint(tb, f0, tx.srgb);       // a + 0 * (b - a) for every lane
    const Footprint f1 = fetch_footprint(tx, s.off1, s.l1, uv, s.filter);
    const f4 a = filter_footprint(tb, f0, tx.srgb);
    return s.frac == 0.0f ? a : lerp4(a, filter_footprint(tb, f1, tx.srgb), s.frac);
}
BRMI_DEV f4 sample_level(const TexelTables& tb, const TexBinding& tx, f2 uv, float lodIn) {
    if (!tx.bound) return {1.0f, 1.0f, 1.0f, 1.0f};
    return sample_prepared(tb, tx, prepare_level(tx, lodIn), uv);
}

BRMI_DEV float log2_poly(float x) {
    const uint32_t b = as_u32(x);
    const int e = (int)((b >> 23) & 0xFFu) - 127;
    const float t = as_f32((b & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
    const float p = t * (1.442609190940857f + t * (-0.7168022990226746f + t * (0.44070422649383545f + t * (-0.2247820496559143f + t * 0.05827096104621887f))));
    return (float)e + p;
}
// Texture2D::SampleGrad: the level of detail of a gradient pair, then SampleLevel
BRMI_DEV float grad_lod(const TexBinding& tx, f2 dUVdx, f2 dUVdy) {
    const float W = (float)tx.width, H = (float)tx.height;
    const float dxx = dUVdx.x * W, dxy = dUVdx.y * H, dyx = dUVdy.x * W, dyy = dUVdy.y * H;
    const float rho2 = max2(dxx * dxx + dxy * dxy, dyx * dyx + dyy * dyy);
    if (!(rho2 >= 1.17549435e-38f)) return -127.0f;
    if (rho2 > 3.0e38f) return 128.0f;
    return 0.5f * log2_poly(rho2);
}
BRMI_DEV f4 sample_grad(const TexelTables& tb, const TexBinding& tx, f2 uv, f2 dUVdx, f2 dUVdy) {
    if (!tx.bound) return {1.0f, 1.0f, 1.0f, 1.0f};
    return sample_level(tb, tx, uv, grad_lod(tx, dUVdx, dUVdy));
}

// SWAlphaTestFailed (CLOD_SW_RASTER_DYNAMIC_ALPHA_TEST).  Everything that depends on the material alone -- the two texture
// bindings, SampleLevel(.., 0)'s level choice -- is resolved once per cluster / record, not per pixel.
// Packed (round 4): the record is per-lane state of the rasteriser's pixel loops -- 28 registers as {pointer, w, h} x 2 levels x 2 textures made
// k_raster_bins<true> spill; 16 now.  A level is its first texel and (w - 1) | (h - 1) << 16; the second level (read only when the sampler blends
// mips at level-of-detail 0, i.e. frac != 0) is an offset from the first, in texels; filter and address modes of both textures share one word.
struct AlphaTex { const uint32_t* base; uint32_t wh0; int32_t off1; uint32_t wh1; float frac; };      // 24 B
struct AlphaMaterial { AlphaTex baseColor, opacity; float alphaFactor, cutoff; uint32_t flags, pad; };    // 64 B
constexpr uint32_t ALPHA_FLAG_USED = 1u, ALPHA_FLAG_POINT = 2u, ALPHA_FLAG_ADDR_U_SHIFT = 2u, ALPHA_FLAG_ADDR_V_SHIFT = 4u, ALPHA_FLAG_POW2 = 64u, ALPHA_FLAG_OPACITY_SHIFT = 8u;      // per texture: used | point filter | addressU (2 bits) | addressV (2 bits) | sides are powers of two <= 16384
static_assert(sizeof(AlphaMaterial) == 64, "one cache line; the per-material table reserves 128 B per entry");
static_assert(BRMI_ADDRESS_WRAP < 4u && BRMI_ADDRESS_MIRROR < 4u && BRMI_ADDRESS_CLAMP < 4u, "two bits per address mode");
BRMI_DEV AlphaTex alpha_tex_of(const TexBinding& tx, bool enabled, uint32_t& flags) {
    AlphaTex t{};
    flags = 0u;
    if (!(enabled && tx.bound)) return t;
    float lod = min2(max2(0.0f + tx.sm.mipLodBias, tx.sm.minLod), tx.sm.maxLod);
    lod = min2(max2(lod, 0.0f), (float)(tx.mipCount - 1u));
    const uint32_t filter = lod <= 0.0f ? tx.sm.magFilter : tx.sm.minFilter;
    auto mode = [](uint32_t m) { return (m == BRMI_ADDRESS_CLAMP || m == BRMI_ADDRESS_MIRROR) ? m : BRMI_ADDRESS_WRAP; };      // address_texel: anything else wraps
    const bool pow2 = (tx.width & (tx.width - 1u)) == 0u && (tx.height & (tx.height - 1u)) == 0u && tx.width <= 16384u && tx.height <= 16384u && tx.width != 0u && tx.height != 0u;
    flags = ALPHA_FLAG_USED | (pow2 ? ALPHA_FLAG_POW2 : 0u) | (filter == BRMI_FILTER_POINT ? ALPHA_FLAG_POINT : 0u) | (mode(tx.sm.addressU) << ALPHA_FLAG_ADDR_U_SHIFT) | (mode(tx.sm.addressV) << ALPHA_FLAG_ADDR_V_SHIFT);
    uint32_t a, b;
    if (tx.sm.mipFilter == BRMI_FILTER_POINT) { a = (uint32_t)floor_to_int(lod + 0.5f); if (a > tx.mipCount - 1u) a = tx.mipCount - 1u; b = a; t.frac = 0.0f; }
    else { a = (uint32_t)floor_to_int(lod); t.frac = lod - floorf(lod); b = a + 1u > tx.mipCount - 1u ? tx.mipCount - 1u : a + 1u; }
    auto dims = [&](uint32_t l) { const uint32_t w = tx.width >> l ? tx.width >> l : 1u, h = tx.height >> l ? tx.height >> l : 1u; return ((w - 1u) & 0xFFFFu) | ((h - 1u) << 16); };      // sides up to 65536
    const uint32_t oa = as_global(tx.mipOffset)[a], ob = as_global(tx.mipOffset)[b];
    t.base = tx.texels + oa; t.wh0 = dims(a); t.off1 = (int32_t)(ob - oa); t.wh1 = dims(b);
    return t;
}
template <typename MatPtr, typename TexPtr, typename SampPtr>
BRMI_DEV AlphaMaterial load_alpha_material(MatPtr m, TexPtr textures, uint32_t textureCount, SampPtr samplers, uint32_t samplerCount) {
    AlphaMaterial r{};
    const uint32_t flags = m->materialFlags;
    r.alphaFactor = m->baseColorFactor[3]; r.cutoff = m->alphaCutoff;
    uint32_t fb = 0u, fo = 0u;
    r.baseColor = alpha_tex_of(bind_texture(textures, textureCount, samplers, samplerCount, m->baseColorTextureIndex, m->baseColorSamplerIndex), (flags & BRMI_MATERIAL_BASE_COLOR_TEXTURE) != 0u, fb);
    r.opacity = alpha_tex_of(bind_texture(textures, textureCount, samplers, samplerCount, m->opacityTextureIndex, m->opacitySamplerIndex), (flags & BRMI_MATERIAL_OPACITY_TEXTURE) != 0u, fo);
    r.flags = fb | (fo << ALPHA_FLAG_OPACITY_SHIFT);
    // a slot whose flag is set but whose descriptor is out of range reads as opaque white: alpha *= 1
    return r;
}
// one level of one texture: `flags` are the texture's six bits
// (measured at compile time, round 4: with this path beside the general one the rasteriser's alpha kernels need ~40 more registers and spill; off)
#ifndef BRMI_ALPHA_POW2_PATH
#define BRMI_ALPHA_POW2_PATH 0
#endif
// the bilinear footprint of a power-of-two level without a branch (the addressing of sample_prepared_pow2, its per-axis constants derived per lane:
// the lanes of the rasteriser's pixel loops belong to different materials)
BRMI_DEV float alpha_level_pow2(const float* unorm, const uint32_t* base, uint32_t wh, uint32_t flags, f2 uv) {
    const int wm1 = (int)(wh & 0xFFFFu), hm1 = (int)(wh >> 16);
    const uint32_t w = (uint32_t)wm1 + 1u;
    const float fx = uv.x * (float)(wm1 + 1) - 0.5f, fy = uv.y * (float)(hm1 + 1) - 0.5f;
    const float tx_ = fx - floorf(fx), ty_ = fy - floorf(fy);
    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
    const uint32_t modeU = (flags >> ALPHA_FLAG_ADDR_U_SHIFT) & 3u, modeV = (flags >> ALPHA_FLAG_ADDR_V_SHIFT) & 3u;
    const int m2x = 2 * wm1 + 1, m2y = 2 * hm1 + 1;
    const int hx = modeU == BRMI_ADDRESS_MIRROR ? m2x : wm1, ox = modeU == BRMI_ADDRESS_MIRROR ? m2x : 0x7FFFFFFF, mx = modeU == BRMI_ADDRESS_CLAMP ? -1 : hx;
    const int hy = modeV == BRMI_ADDRESS_MIRROR ? m2y : hm1, oy = modeV == BRMI_ADDRESS_MIRROR ? m2y : 0x7FFFFFFF, my = modeV == BRMI_ADDRESS_CLAMP ? -1 : hy;
    const uint32_t xa = (uint32_t)address_pow2(x0, mx, hx, ox), xb = (uint32_t)address_pow2(inc_sat(x0), mx, hx, ox);
    const uint32_t ya = (uint32_t)address_pow2(y0, my, hy, oy), yb = (uint32_t)address_pow2(inc_sat(y0), my, hy, oy);
    const uint32_t rowA = __umul24(ya, w), rowB = __umul24(yb, w);
    const __attribute__((address_space(1))) char* g = (const __attribute__((address_space(1))) char*)base;
    auto texel = [&](uint32_t index) { return *(GlobalTexels)(g + (uint32_t)(index << 2)); };
    const uint32_t c00 = texel(rowA + xa), c10 = texel(rowA + xb), c01 = texel(rowB + xa), c11 = texel(rowB + xb);
    const float a00 = unorm[c00 >> 24], a10 = unorm[c10 >> 24], a01 = unorm[c01 >> 24], a11 = unorm[c11 >> 24];
    const float top = a00 + tx_ * (a10 - a00), bot = a01 + tx_ * (a11 - a01);
    return top + ty_ * (bot - top);
}
BRMI_DEV float alpha_level(const float* unorm, const uint32_t* base, uint32_t wh, uint32_t flags, f2 uv) {
#if BRMI_ALPHA_POW2_PATH
    if ((flags & (ALPHA_FLAG_POW2 | ALPHA_FLAG_POINT)) == ALPHA_FLAG_POW2) return alpha_level_pow2(unorm, base, wh, flags, uv);      // (per lane; the general path below is skipped when no lane needs it)
#endif
    const int w = (int)(wh & 0xFFFFu) + 1, h = (int)(wh >> 16) + 1;
    const uint32_t addressU = (flags >> ALPHA_FLAG_ADDR_U_SHIFT) & 3u, addressV = (flags >> ALPHA_FLAG_ADDR_V_SHIFT) & 3u;
    GlobalTexels g = as_global(base);
    if (flags & ALPHA_FLAG_POINT)
        return unorm[g[(size_t)address_texel(floor_to_int(uv.y * (float)h), h, addressV) * (size_t)w + (size_t)address_texel(floor_to_int(uv.x * (float)w), w, addressU)] >> 24];
    const float fx = uv.x * (float)w - 0.5f, fy = uv.y * (float)h - 0.5f;
    const float tx_ = fx - floorf(fx), ty_ = fy - floorf(fy);
    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
    const int xa = address_texel(x0, w, addressU), xb = address_texel(x0 == 0x7FFFFFFF ? x0 : x0 + 1, w, addressU);
    const int ya = address_texel(y0, h, addressV), yb = address_texel(y0 == 0x7FFFFFFF ? y0 : y0 + 1, h, addressV);
    const uint32_t c00 = g[(size_t)ya * (size_t)w + (size_t)xa], c10 = g[(size_t)ya * (size_t)w + (size_t)xb], c01 = g[(size_t)yb * (size_t)w + (size_t)xa], c11 = g[(size_t)yb * (size_t)w + (size_t)xb];
    const float a00 = unorm[c00 >> 24], a10 = unorm[c10 >> 24], a01 = unorm[c01 >> 24], a11 = unorm[c11 >> 24];
    const float top = a00 + tx_ * (a10 - a00), bot = a01 + tx_ * (a11 - a01);
    return top + ty_ * (bot - top);
}
BRMI_DEV float alpha_sample(const float* unorm, const AlphaTex& t, uint32_t flags, f2 uv) {
    const float a = alpha_level(unorm, t.base, t.wh0, flags, uv);
    if (t.frac == 0.0f) return a;
    return a + t.frac * (alpha_level(unorm, t.base + t.off1, t.wh1, flags, uv) - a);
}
BRMI_DEV bool alpha_test_failed(const float* unorm, const AlphaMaterial& m, f2 uv) {
    float alpha = m.alphaFactor;
    if (m.flags & ALPHA_FLAG_USED) alpha *= alpha_sample(unorm, m.baseColor, m.flags, uv);
    if ((m.flags >> ALPHA_FLAG_OPACITY_SHIFT) & ALPHA_FLAG_USED) alpha *= alpha_sample(unorm, m.opacity, m.flags >> ALPHA_FLAG_OPACITY_SHIFT, uv);
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
