// brmi_device.h -- device-side arithmetic and data-access helpers shared by the HIP kernels.
//
// Arithmetic contract (the same specification the CPU checker is written against, implemented
// here independently): IEEE-754 binary32, round-to-nearest-even, NO fused multiply-add (the
// library is compiled with -ffp-contract=off), correctly rounded division and square root
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved.  HLSL semantics restated:
//   mul(v, M)   row vector x row-major matrix, k accumulated 0..3 left to right
//   dot(a, b)   ((a.x*b.x + a.y*b.y) + a.z*b.z) (+ a.w*b.w)
//   rcp(x) = 1/x ; rsqrt(x) = 1/sqrt(x) ; normalize(v) = v * rsqrt(dot(v,v))
#ifndef BRMI_DEVICE_H
#define BRMI_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "brmi.h"

#define BRMI_DEV __device__ __forceinline__

namespace brmi {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct m4 { float m[4][4]; };

BRMI_DEV uint32_t as_u32(float f) { return __float_as_uint(f); }
BRMI_DEV float as_f32(uint32_t u) { return __uint_as_float(u); }

BRMI_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
BRMI_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
BRMI_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
BRMI_DEV f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
BRMI_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
BRMI_DEV f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
BRMI_DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
BRMI_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
BRMI_DEV f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
BRMI_DEV f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }

BRMI_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BRMI_DEV float dot4(f4 a, f4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
BRMI_DEV f3 cross3(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
BRMI_DEV float rcpf(float x) { return 1.0f / x; }
BRMI_DEV float rsqrtf_(float x) { return 1.0f / sqrtf(x); }
BRMI_DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
BRMI_DEV f3 normalize3(f3 a) { return a * rsqrtf_(dot3(a, a)); }
// Correctly rounded sqrt(x) and 1 / x for x well inside the normal range, 2^-63 <= x < 2^63.  The compiler's expansions of sqrtf and of
// the division carry an input-scaling prologue and a fix-up epilogue for subnormal / huge / special operands (7 of 16 and 4 of 11
// instructions); for an in-range operand those are the identity, and what is left -- restated here operation by operation -- is the
// hardware estimate plus the same FMA corrections, so the results are bit-identical to the IEEE ones.  The vectors of the shading pass
// (eye and light distances of a scene in metres) always take this path; anything else falls back to the general form.
BRMI_DEV bool in_range_pow63(float x) { return (as_u32(x) - 0x20000000u) < 0x3F000000u; }     // positive, finite, 2^-63 <= x < 2^63
BRMI_DEV float sqrt_rn_in_range(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float yDown = as_f32(as_u32(y) - 1u), yUp = as_f32(as_u32(y) + 1u);
    const float eDown = __builtin_fmaf(-yDown, y, x), eUp = __builtin_fmaf(-yUp, y, x);
    float r = eDown <= 0.0f ? yDown : y;
    r = eUp > 0.0f ? yUp : r;
    return r;
}
BRMI_DEV float rcp_rn_in_range(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    const float q = __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
    return __builtin_fmaf(__builtin_fmaf(-x, q, 1.0f), r1, q);
}
// length and normalize with the two helpers: `len` = sqrt(dot(a, a)), returns a * (1 / len), as normalize3 / length3 compute them
BRMI_DEV f3 normalize3_len(f3 a, float d2, float& len) {
    float inv;
    if (in_range_pow63(d2)) { len = sqrt_rn_in_range(d2); inv = rcp_rn_in_range(len); }
    else { len = sqrtf(d2); inv = 1.0f / len; }
    return a * inv;
}
BRMI_DEV f3 normalize3_q(f3 a) { float len; return normalize3_len(a, dot3(a, a), len); }
BRMI_DEV float min2(float a, float b) { return a < b ? a : b; }
BRMI_DEV float max2(float a, float b) { return a > b ? a : b; }
BRMI_DEV float sat(float x) { return min2(max2(x, 0.0f), 1.0f); }
BRMI_DEV f3 sat3(f3 v) { return {sat(v.x), sat(v.y), sat(v.z)}; }
BRMI_DEV float clampf(float x, float a, float b) { return min2(max2(x, a), b); }
BRMI_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
BRMI_DEV f3 lerp3(f3 a, f3 b, float t) { return {lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t)}; }
BRMI_DEV f3 min3v(f3 a, f3 b) { return {min2(a.x, b.x), min2(a.y, b.y), min2(a.z, b.z)}; }
BRMI_DEV f3 max3v(f3 a, f3 b) { return {max2(a.x, b.x), max2(a.y, b.y), max2(a.z, b.z)}; }
BRMI_DEV f3 xyz(f4 v) { return {v.x, v.y, v.z}; }

BRMI_DEV f4 mul_vm(f4 v, const m4& a) {
    f4 r;
    r.x = ((v.x * a.m[0][0] + v.y * a.m[1][0]) + v.z * a.m[2][0]) + v.w * a.m[3][0];
    r.y = ((v.x * a.m[0][1] + v.y * a.m[1][1]) + v.z * a.m[2][1]) + v.w * a.m[3][1];
    r.z = ((v.x * a.m[0][2] + v.y * a.m[1][2]) + v.z * a.m[2][2]) + v.w * a.m[3][2];
    r.w = ((v.x * a.m[0][3] + v.y * a.m[1][3]) + v.z * a.m[2][3]) + v.w * a.m[3][3];
    return r;
}
BRMI_DEV f4 mul_point(f3 p, const m4& a) { return mul_vm(f4{p.x, p.y, p.z, 1.0f}, a); }
BRMI_DEV f3 mul_v3m3(f3 v, const m4& a) {
    return {(v.x * a.m[0][0] + v.y * a.m[1][0]) + v.z * a.m[2][0],
            (v.x * a.m[0][1] + v.y * a.m[1][1]) + v.z * a.m[2][1],
            (v.x * a.m[0][2] + v.y * a.m[1][2]) + v.z * a.m[2][2]};
}
BRMI_DEV m4 mul_mm(const m4& a, const m4& b) {
    m4 r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
BRMI_DEV f4 mul_mcol(const m4& a, f4 v) {
    return {((a.m[0][0] * v.x + a.m[0][1] * v.y) + a.m[0][2] * v.z) + a.m[0][3] * v.w,
            ((a.m[1][0] * v.x + a.m[1][1] * v.y) + a.m[1][2] * v.z) + a.m[1][3] * v.w,
            ((a.m[2][0] * v.x + a.m[2][1] * v.y) + a.m[2][2] * v.z) + a.m[2][3] * v.w,
            ((a.m[3][0] * v.x + a.m[3][1] * v.y) + a.m[3][2] * v.z) + a.m[3][3] * v.w};
}
BRMI_DEV const m4& as_m4(const float (&a)[4][4]) { return *reinterpret_cast<const m4*>(&a[0][0]); }
BRMI_DEV m4 load_m4(const float* p) {   // 64 B, 16 B aligned
    m4 r;
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) { float4 v = q[i]; r.m[i][0] = v.x; r.m[i][1] = v.y; r.m[i][2] = v.z; r.m[i][3] = v.w; }
    return r;
}
BRMI_DEV float max_axis_scale(const m4& m) {   // MaxAxisScale_RowVector
    f3 ax{m.m[0][0], m.m[0][1], m.m[0][2]}, ay{m.m[1][0], m.m[1][1], m.m[1][2]}, az{m.m[2][0], m.m[2][1], m.m[2][2]};
    return max2(length3(ax), max2(length3(ay), length3(az)));
}

// float -> int with the saturating semantics of v_cvt_i32_f32, stated explicitly
// (round 4: the instruction itself.  Written as three compares and a cast the compiler kept the compares and wrapped the conversion in two
// exec-mask sections -- a dozen instructions and two branches for each of the rasteriser's six clip divisions per row and the sampler's four
// coordinates per footprint.  NaN -> 0, values beyond the range saturate, everything else truncates toward zero: the oracle's to_int_sat.)
BRMI_DEV int to_int_sat(float f) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// Issue priority of the geometry half's waves (s_setprio: which wave of a SIMD the arbiter picks first).  The geometry kernels are short chains of
// dependent loads; when they share a SIMD with another frame's shading waves (hundreds of VALU instructions between loads) every instruction of
// theirs that waits its turn lengthens the chain the NEXT frame waits for.  0 = off (the hardware default for every wave).
#ifndef BRMI_GEOM_PRIO
#define BRMI_GEOM_PRIO 0
#endif
#ifndef BRMI_PRIO_MASK
#define BRMI_PRIO_MASK 0
#endif
enum : int { PRIO_CULL = 1, PRIO_SCAN = 2, PRIO_RASTER = 4, PRIO_BINS = 8, PRIO_HZB = 16, PRIO_SETUP = 32, PRIO_GBUFFER = 64, PRIO_SHADE = 128 };
template <int WHICH> BRMI_DEV void wave_prio() {
#if BRMI_GEOM_PRIO
    if (BRMI_PRIO_MASK & WHICH) __builtin_amdgcn_s_setprio(BRMI_GEOM_PRIO);
#endif
}

// ---- packing ----------------------------------------------------------------------------------
BRMI_DEV uint32_t f32_to_f16_bits(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f); }   // v_cvt_f16_f32, RTNE
BRMI_DEV float f16_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
BRMI_DEV uint64_t pack_half4(float a, float b, float c, float d) {
    return (uint64_t)f32_to_f16_bits(a) | ((uint64_t)f32_to_f16_bits(b) << 16) | ((uint64_t)f32_to_f16_bits(c) << 32) | ((uint64_t)f32_to_f16_bits(d) << 48);
}
BRMI_DEV uint32_t unorm8(float x) { return (uint32_t)(sat(x) * 255.0f + 0.5f); }
BRMI_DEV uint32_t pack_unorm4(float a, float b, float c, float d) { return unorm8(a) | (unorm8(b) << 8) | (unorm8(c) << 16) | (unorm8(d) << 24); }
BRMI_DEV float unorm8_to_f32(uint32_t v) { return (float)(v & 0xFFu) / 255.0f; }

// visibility key: 31 bits depth | 26 bits cluster | 7 bits triangle
BRMI_DEV uint64_t pack_vis_key(float depth, uint32_t cluster, uint32_t tri) {
    const uint64_t depthBits = as_u32(depth) >> 1;
    return (depthBits << BRMI_VIS_META_BITS) | ((uint64_t)(cluster & 0x3FFFFFFu) << BRMI_VIS_TRI_BITS) | (uint64_t)(tri & 0x7Fu);
}

// packed visible cluster accessors
BRMI_DEV uint32_t vc_view(const uint4& c) { return c.x & 0xFFu; }
BRMI_DEV uint32_t vc_instance(const uint4& c) { return (c.x >> 8) & 0xFFFFFFu; }
BRMI_DEV uint32_t vc_meshlet(const uint4& c) { return c.y & 0x3FFFu; }
BRMI_DEV uint32_t vc_group(const uint4& c) { return ((c.y >> 14) & 0x3FFFFu) | ((c.z & 0x3u) << 18); }
BRMI_DEV uint32_t vc_slab(const uint4& c) { return (c.z >> 2) & 0xFFFFFu; }
BRMI_DEV uint32_t vc_page_offset(const uint4& c) { return ((c.z >> 22) & 0x3FFu) << 18; }
BRMI_DEV uint4 pack_visible_cluster(uint32_t view, uint32_t inst, uint32_t meshlet, uint32_t group, uint32_t slab, uint32_t pageByteOffset) {
    const uint32_t page = pageByteOffset >> 18;
    return make_uint4((view & 0xFFu) | ((inst & 0xFFFFFFu) << 8), (meshlet & 0x3FFFu) | ((group & 0x3FFFFu) << 14),
                      ((group >> 18) & 0x3u) | ((slab & 0xFFFFFu) << 2) | ((page & 0x3FFu) << 22), 0x1Fu);
}

// tiled 8x8 surface addressing, column-major inside the tile: element index of pixel (x, y)
BRMI_DEV uint32_t tiled_index(uint32_t x, uint32_t y, uint32_t tilesX) { return (((y >> 3) * tilesX + (x >> 3)) << 6) | ((x & 7u) << 3) | (y & 7u); }

// ---- interleaved screen partition (brmi_config::stripe*): rows of the frame <-> rows of this GPU's compact surfaces -----------------------
// The frame is cut into chunks of `rows` rows; `count` consecutive chunks form a group, every GPU owns one chunk of every group: chunk
// `index` in even groups, chunk count - 1 - index in odd ones (the order runs back and forth, so a vertical gradient of the frame's cost --
// sky at the top, the horizon's detail in the middle -- does not favour one GPU in every group: measured max / min over 8 ranks 1.16 -> see
// DESIGN.md section 6).
struct StripeMap { uint32_t rows, count, index, fullHeight; };      // count <= 1: the identity (fullHeight = the surface height)
BRMI_DEV bool stripe_on(const StripeMap& m) { return m.count > 1u; }
BRMI_DEV uint32_t stripe_slot(const StripeMap& m, uint32_t group) { return (group & 1u) ? m.count - 1u - m.index : m.index; }      // this GPU's chunk inside a group
BRMI_DEV bool stripe_owns(const StripeMap& m, uint32_t py) {
    if (m.count <= 1u) return true;
    const uint32_t c = py / m.rows, g = c / m.count;
    return c - g * m.count == stripe_slot(m, g);
}
BRMI_DEV uint32_t stripe_vrow(const StripeMap& m, uint32_t py) {      // surface row of an owned frame row
    if (m.count <= 1u) return py;
    const uint32_t c = py / m.rows;
    return (c / m.count) * m.rows + (py - c * m.rows);
}
BRMI_DEV uint32_t stripe_rrow(const StripeMap& m, uint32_t v) {       // frame row of a surface row
    if (m.count <= 1u) return v;
    const uint32_t g = v / m.rows;
    return (g * m.count + stripe_slot(m, g)) * m.rows + (v - g * m.rows);
}
// first owned frame row >= y / last owned frame row <= y (0xFFFFFFFF: none)
BRMI_DEV uint32_t stripe_first_owned(const StripeMap& m, uint32_t y) {
    if (m.count <= 1u) return y;
    const uint32_t c = y / m.rows, g = c / m.count, mine = g * m.count + stripe_slot(m, g);
    if (c == mine) return y;
    if (c < mine) return mine * m.rows;
    return ((g + 1u) * m.count + stripe_slot(m, g + 1u)) * m.rows;
}
BRMI_DEV uint32_t stripe_last_owned(const StripeMap& m, uint32_t y) {
    if (m.count <= 1u) return y;
    const uint32_t c = y / m.rows, g = c / m.count, mine = g * m.count + stripe_slot(m, g);
    if (c == mine) return y;
    if (c > mine) return mine * m.rows + m.rows - 1u;
    if (g == 0u) return 0xFFFFFFFFu;
    return ((g - 1u) * m.count + stripe_slot(m, g - 1u)) * m.rows + m.rows - 1u;
}

// wave64 helpers
// force a wave-uniform value into an SGPR (frame constants loaded through a pointer otherwise occupy VGPRs)
BRMI_DEV float uni(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
BRMI_DEV m4 uni_m4(const m4& a) { m4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = uni(a.m[i][j]); return r; }
BRMI_DEV uint32_t lane_id() { return __lane_id(); }
BRMI_DEV uint32_t lane_rank(uint64_t mask) { return __popcll(mask & ((1ull << lane_id()) - 1ull)); }
// LDS visibility inside ONE wave (single-wave workgroups, or waves of one workgroup that run different amounts of work: no workgroup barrier between their
// steps).  __syncthreads() would also wait for every global store and atomic the wave has in flight: a memory round trip per hand-off.
BRMI_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ---- skinning (BR/shaders/Include/skinningCommon.hlsli:23-88) -------------------------------------------------------------
// `skinningMatrices` holds bone * inverseBind per (slot, joint), 64 joints per slot; LoadBoneSkinMatrix is its transpose.
BRMI_DEV m4 load_bone_skin_matrix(const float* skinningMatrices, uint32_t slot, uint32_t joint) {
    const m4 prod = load_m4(skinningMatrices + ((size_t)slot * 64u + joint) * 16u);
    m4 r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r.m[i][j] = prod.m[j][i];
    return r;
}
// BuildSkinMatrix: w0*M0 + w1*M1 + ... + w7*M7, left to right
BRMI_DEV m4 build_skin_matrix(const float* skinningMatrices, uint32_t slot, const uint32_t* joints, const float* weights) {
    m4 r;
    if (slot == 0xFFFFFFFFu || skinningMatrices == nullptr) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return r;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const m4 b = load_bone_skin_matrix(skinningMatrices, slot, joints[k]);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { const float t = weights[k] * b.m[i][j]; r.m[i][j] = (k == 0) ? t : r.m[i][j] + t; }
    }
    return r;
}
// joints / weights of one vertex (8 x u32, 8 x f32); missing arrays read as zero
BRMI_DEV void load_skin_influences(const uint8_t* jointPtr, const uint8_t* weightPtr, uint32_t joints[8], float weights[8]) {
#pragma unroll
    for (int k = 0; k < 8; k++) { joints[k] = jointPtr ? reinterpret_cast<const uint32_t*>(jointPtr)[k] : 0u; weights[k] = weightPtr ? reinterpret_cast<const float*>(weightPtr)[k] : 0.0f; }
}

BRMI_DEV float ior_to_f0(float ior) { const float s = max2(ior, 1.0f); const float f = (s - 1.0f) / (s + 1.0f); return f * f; }
// Read-only data produced by an earlier kernel, viewed through the constant address space: with a wave-uniform address
// the compiler then selects scalar (s_load) instead of vector loads.
template <typename T> BRMI_DEV const __attribute__((address_space(4))) T* kconst(const T* p) { return (const __attribute__((address_space(4))) T*)p; }
// A whole record through the constant address space (word by word: the compiler merges the words into s_load_dwordxN when the address is
// wave-uniform).  For records an EARLIER kernel wrote or the host uploaded -- the scalar cache is not coherent with this launch's own stores.
template <typename T> BRMI_DEV T load_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    T r;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
    const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)p;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; i++) dst[i] = src[i];
    return r;
}
// HasOpenPBRTexture (utilities.hlsli:643-646) for any of the six coat / fuzz slots of an OpenPBR record
template <typename Op> BRMI_DEV bool openpbr_has_textures(Op op) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 6; k++) any = any || (op->textureBindings[2 * k] != 0xFFFFFFFFu && op->textureBindings[2 * k + 1] != 0xFFFFFFFFu);
    return any;
}
// `p` in the address space of `like`: tables indexed by fields of a record take the scalar path when the record does
template <typename L, typename T> BRMI_DEV const T* as_space_of(const L*, const T* p) { return p; }
template <typename L, typename T> BRMI_DEV const __attribute__((address_space(4))) T* as_space_of(const __attribute__((address_space(4))) L*, const T* p) { return kconst(p); }
// wave-aggregated append: one atomic per wave; returns the slot of this lane (valid when pred)
BRMI_DEV uint32_t wave_append(uint32_t* counter, bool pred) {
    const uint64_t mask = __ballot(pred);
    if (mask == 0) return 0;
    const uint32_t leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + lane_rank(mask);
}

}  // namespace brmi
#endif
