// orc_common.h -- shared helpers of the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  The oracle is a scalar CPU restatement of the reference HLSL for the
// visibility-buffer path; it exists to check libbrmi.so and to time a CPU baseline.  Nothing under
// basicrenderer_amd/ may include, link or call it.  PARITY UNPINNED: the reference has no tests,
// golden images or known-answer vectors for this path, and its HLSL and
// DX12 host code cannot be built here, so this restatement is anchored only on the cited source
// lines.
//
// Arithmetic contract (shared by specification, not by code, with the HIP kernels):
//   * IEEE-754 binary32, round-to-nearest-even, no FMA contraction, denormals kept;
//   * HLSL `mul(v, M)` (row vector x row-major matrix) accumulates k = 0..3 left to right;
//   * dot(a,b) = ((a.x*b.x + a.y*b.y) + a.z*b.z) (+ a.w*b.w);
//   * rcp(x) = 1/x, rsqrt(x) = 1/sqrt(x), normalize(v) = v * rsqrt(dot(v,v)), length = sqrt(dot);
//   * saturate/min/max/clamp as in HLSL (NaN-free inputs assumed);
//   * float -> unorm8: (uint)(saturate(x)*255 + 0.5); float -> half: round-to-nearest-even.
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <cmath>
#include <cstdint>
#include <cstring>

#include "brmi.h"

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct mat4 { float m[4][4]; };

inline uint32_t asuint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float asfloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

inline float3 make3(float x, float y, float z) { return {x, y, z}; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }

inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float rcp(float x) { return 1.0f / x; }
inline float rsqrt(float x) { return 1.0f / std::sqrt(x); }
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(float3 a) { return a * rsqrt(dot(a, a)); }
inline float fmin2(float a, float b) { return a < b ? a : b; }   // HLSL min
inline float fmax2(float a, float b) { return a > b ? a : b; }   // HLSL max
inline float saturate(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
inline float3 saturate(float3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
inline float clampf(float x, float a, float b) { return fmin2(fmax2(x, a), b); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float3 lerp(float3 a, float3 b, float t) { return {lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)}; }
inline float3 fmin3v(float3 a, float3 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)}; }
inline float3 fmax3v(float3 a, float3 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)}; }

inline const mat4& M(const float (&a)[4][4]) { return *reinterpret_cast<const mat4*>(&a[0][0]); }

// mul(float4 v, row_major M)
inline float4 mul(float4 v, const mat4& a) {
    float4 r;
    r.x = ((v.x * a.m[0][0] + v.y * a.m[1][0]) + v.z * a.m[2][0]) + v.w * a.m[3][0];
    r.y = ((v.x * a.m[0][1] + v.y * a.m[1][1]) + v.z * a.m[2][1]) + v.w * a.m[3][1];
    r.z = ((v.x * a.m[0][2] + v.y * a.m[1][2]) + v.z * a.m[2][2]) + v.w * a.m[3][2];
    r.w = ((v.x * a.m[0][3] + v.y * a.m[1][3]) + v.z * a.m[2][3]) + v.w * a.m[3][3];
    return r;
}
inline float4 mulPoint(float3 p, const mat4& a) { return mul(float4{p.x, p.y, p.z, 1.0f}, a); }
// mul(float3 v, (float3x3)M)
inline float3 mul3(float3 v, const mat4& a) {
    return {(v.x * a.m[0][0] + v.y * a.m[1][0]) + v.z * a.m[2][0],
            (v.x * a.m[0][1] + v.y * a.m[1][1]) + v.z * a.m[2][1],
            (v.x * a.m[0][2] + v.y * a.m[1][2]) + v.z * a.m[2][2]};
}
// mul(row_major A, row_major B)
inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
// mul(row_major A, float4 column v)
inline float4 mulCol(const mat4& a, float4 v) {
    return {((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z) + a.m[0][3] * v.w,
            ((a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z) + a.m[1][3] * v.w,
            ((a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z) + a.m[2][3] * v.w,
            ((a.m[3][0] * v.x + a.m[3][1] * v.y) + a.m[3][2] * v.z) + a.m[3][3] * v.w};
}
inline float3 xyz(float4 v) { return {v.x, v.y, v.z}; }

// float -> half, round to nearest even (the RGBA16F store)
inline uint16_t f32_to_f16(float f) {
    uint32_t x = asuint(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007FFFFFu;
    int32_t  exp = (int32_t)((x >> 23) & 0xFF);
    if (exp == 255) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u | (mant >> 13) : 0u));
    int32_t e = exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)(sign | half);
}
inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu;
    if (exp == 0) {
        if (mant == 0) return asfloat(sign);
        int e = -1; do { e++; mant <<= 1; } while ((mant & 0x400u) == 0);
        return asfloat(sign | (uint32_t)((127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13));
    }
    if (exp == 31) return asfloat(sign | 0x7F800000u | (mant << 13));
    return asfloat(sign | ((exp + 112u) << 23) | (mant << 13));
}
inline uint64_t pack_half4(float a, float b, float c, float d) {
    return (uint64_t)f32_to_f16(a) | ((uint64_t)f32_to_f16(b) << 16) | ((uint64_t)f32_to_f16(c) << 32) | ((uint64_t)f32_to_f16(d) << 48);
}
inline uint32_t unorm8(float x) { return (uint32_t)(saturate(x) * 255.0f + 0.5f); }
inline uint32_t pack_unorm4(float a, float b, float c, float d) { return unorm8(a) | (unorm8(b) << 8) | (unorm8(c) << 16) | (unorm8(d) << 24); }
inline float unorm8_to_float(uint32_t v) { return (float)(v & 0xFFu) / 255.0f; }

// ---- packed visible cluster (BR/shaders/Include/visibleClusterPacking.hlsli:83-122,220-235)
inline uint32_t vcViewID(const brmi_visible_cluster& c) { return c.x & 0xFFu; }
inline uint32_t vcInstanceID(const brmi_visible_cluster& c) { return (c.x >> 8) & 0xFFFFFFu; }
inline uint32_t vcLocalMeshlet(const brmi_visible_cluster& c) { return c.y & 0x3FFFu; }
inline uint32_t vcGroupID(const brmi_visible_cluster& c) { return ((c.y >> 14) & 0x3FFFFu) | ((c.z & 0x3u) << 18); }
inline uint32_t vcSlabDescriptor(const brmi_visible_cluster& c) { return (c.z >> 2) & 0xFFFFFu; }
inline uint32_t vcPageByteOffset(const brmi_visible_cluster& c) { return ((c.z >> 22) & 0x3FFu) << 18; }
inline brmi_visible_cluster packVisibleCluster(uint32_t view, uint32_t inst, uint32_t meshlet, uint32_t group, uint32_t slab, uint32_t pageByteOffset) {
    const uint32_t pageIndex = pageByteOffset >> 18;
    // CLodBuildVisibleClusterVsmPayloadFromClipmapIndex(INVALID) = 0x1F  (visibleClusterPacking.hlsli:153-160)
    return {(view & 0xFFu) | ((inst & 0xFFFFFFu) << 8), (meshlet & 0x3FFFu) | ((group & 0x3FFFFu) << 14),
            ((group >> 18) & 0x3u) | ((slab & 0xFFFFFu) << 2) | ((pageIndex & 0x3FFu) << 22), 0x1Fu};
}

// ---- slab access (BR/shaders/Include/clodPageAccess.hlsli:11-64)
inline uint32_t load32(const uint8_t* slab, uint32_t addr) { uint32_t v; std::memcpy(&v, slab + addr, 4); return v; }
inline const brmi_page_header* pageHeader(const uint8_t* slab, uint32_t pageOff) { return reinterpret_cast<const brmi_page_header*>(slab + pageOff); }
inline const brmi_meshlet_descriptor* meshletDesc(const uint8_t* slab, uint32_t pageOff, uint32_t descOff, uint32_t i) {
    return reinterpret_cast<const brmi_meshlet_descriptor*>(slab + pageOff + descOff + i * 64u);
}
inline uint32_t descVertexCount(const brmi_meshlet_descriptor& d) { return (d.bitsAndVertexCount >> 24) & 0xFFu; }
inline uint32_t descTriangleCount(const brmi_meshlet_descriptor& d) { return d.triangleCountAndRefinedGroup & 0xFFFFu; }
inline int32_t  descRefinedGroup(const brmi_meshlet_descriptor& d) { return (int32_t)(d.triangleCountAndRefinedGroup >> 16) - 1; }

// SWDecodeTriangle / DecodeTriangleCompact: three consecutive bytes of the triangle stream
inline void decodeTriangle(const uint8_t* slab, uint32_t triStreamBase, uint32_t triByteOffset, uint32_t t, uint32_t idx[3]) {
    const uint8_t* p = slab + triStreamBase + triByteOffset + t * 3u;
    idx[0] = p[0]; idx[1] = p[1]; idx[2] = p[2];
}
inline float3 loadPosition(const uint8_t* slab, uint32_t format, uint32_t streamBase, uint32_t byteOffset, uint32_t v) {
    if (format != BRMI_POSITION_FORMAT_FLOAT3) return {0, 0, 0};   // clodStructs.hlsli:115-129
    float3 p; std::memcpy(&p, slab + streamBase + byteOffset + v * 12u, 12); return p;
}

// ---- skinning (BR/shaders/Include/skinningCommon.hlsli:23-88)
mat4 buildSkinMatrix(const brmi_scene_buffers& sc, uint32_t slot, const uint32_t joints[8], const float weights[8]);

// MaxAxisScale_RowVector (workGraphCulling.hlsl:1397-1403)
inline float maxAxisScale(const mat4& m) {
    float3 ax{m.m[0][0], m.m[0][1], m.m[0][2]}, ay{m.m[1][0], m.m[1][1], m.m[1][2]}, az{m.m[2][0], m.m[2][1], m.m[2][2]};
    return fmax2(length(ax), fmax2(length(ay), length(az)));
}

}  // namespace orc
#endif
