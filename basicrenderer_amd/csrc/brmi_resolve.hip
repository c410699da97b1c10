// brmi_resolve.hip -- G-buffer reconstruction from the visibility buffer (K7 + K8) for gfx950.
//
// Computes what EvaluateGBufferOptimized -> ResolveClodCommonSampleFromVisKeyWithFace does
// (BR/shaders/gbuffer.hlsl:4-35, BR/shaders/Include/clodResolveCommon.hlsli:1414-1721): unpack the
// key, re-fetch the triangle's three vertices, re-project, analytic perspective-correct
// barycentrics at the pixel centre (CalcFullBary, clodResolveCommon.hlsli:104-143), interpolate
// position and normal, evaluate the constant-factor material, write the seven G-buffer surfaces.
// MI355X-first differences from the reference's schedule:
//   * The reference bins pixels per material permutation (histogram -> scan -> pixel list ->
//     per-material ExecuteIndirect, VisUtil.hlsl:37-243) only to run one specialised PSO per bin.
//     One kernel handles every constant-factor material, so the four extra full-screen passes over
//     the 8 B/px surface disappear; the linear-depth write of K6 is fused in too.
//   * one lane per pixel in tile order: a wave64 is exactly one 8x8 tile, so every surface read
//     or written by the wave is one contiguous 256 B..1 KB segment.
#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_texture.h"

namespace brmi {

struct GBufferArgs {
    brmi_scene_buffers sc;
    const uint4* clusters;
    const uint32_t* counters;
    const unsigned long long* vis;
    float4* normals; uint32_t* albedo; unsigned long long* coat; unsigned long long* emissive; unsigned long long* fuzz;
    uint32_t* metallicRoughness; uint32_t* motion; float* depth;
    uint32_t W, H, tilesX, bandY0, bandY1; uint64_t firstPixel, pixelCount;
    StripeMap stripes;      // interleaved partition: surface row -> frame row
    uint32_t clusterCapacity;
    const m4* frameConst; const float* objConst;
    const ClusterSetup* setup; ResolveVertex* verts; ResolveTriangle* tris; uint32_t vertCapacity, triCapacity;
    MaterialWords* matWords;
    uint32_t* hostFeedback;        // host-mapped words (brmi_pass::ensureFeedback) or null
    const ClusterUv* clusterUv; float2* uvs;      // textured scenes: where the UV sets of every visible cluster live, decoded texcoords of the arena's vertices
    uint32_t uvSets;                              // sets the materials of the scene address (1 unless one names a set > 0): uvs holds [set][vertCapacity]
    uint32_t* colors;                             // scenes with vertex colours: the RGBA8 colour of the arena's vertices
    uint32_t setupPart;            // k_resolve_setup: 0 = every visible cluster, 1 = the phase-1 clusters only (launched beside the rasteriser), 2 = the phase-2 clusters only
    uint32_t inlineRatio;          // a frame with more than 1 / inlineRatio cluster triangles per pixel resolves without the tables (resolve_inline_frame): what the kernels tell the host
    uint32_t variantSelect;        // 0: run; 1: run only when no cluster spilled out of the arena; 2: only when one did (counters[CNT_RESOLVE_SPILL])
};

BRMI_DEV f3 oct_decode_normal(uint32_t packed) {
    const int sp = (int)packed;
    const int x = (int)((uint32_t)sp << 16) >> 16, y = sp >> 16;
    const float ex = max2(-1.0f, (float)x / 32767.0f), ey = max2(-1.0f, (float)y / 32767.0f);
    f3 v{ex, ey, 1.0f - fabsf(ex) - fabsf(ey)};
    if (v.z < 0.0f) {
        const float fx = (1.0f - fabsf(v.y)) * (v.x >= 0.0f ? 1.0f : -1.0f);
        const float fy = (1.0f - fabsf(v.x)) * (v.y >= 0.0f ? 1.0f : -1.0f);
        v.x = fx; v.y = fy;
    }
    return normalize3(v);
}

// Per-cluster resolve tables, one wave64 per visible cluster.  Everything CalcFullBary (clodResolveCommon.hlsli:104-143)
// derives from the triangle alone -- the three projected vertices, 1/w, the screen-space derivatives of the barycentrics --
// and the decoded vertex normals are evaluated once per triangle / vertex here with the shader's operation order; the pixel
// pass then only evaluates the part that depends on the pixel.  (The reference shader recomputes all of it per pixel.)
__global__ void __launch_bounds__(64) k_resolve_setup(GBufferArgs a) {
    wave_prio<PRIO_SETUP>();
    __shared__ float cx[BRMI_MESHLET_MAX_VERTS], cy[BRMI_MESHLET_MAX_VERTS], cw[BRMI_MESHLET_MAX_VERTS];
    const uint32_t lane = threadIdx.x;
    // part 1 runs while the rasteriser and the phase-2 culling are still at work: it reads the phase-1 count only (final since the compaction)
    const uint32_t firstCluster = a.setupPart == 2u ? min(a.counters[CNT_VISIBLE], a.clusterCapacity) : 0u;
    const uint32_t clusterCount = a.setupPart == 1u ? min(a.counters[CNT_VISIBLE], a.clusterCapacity) : min(a.counters[CNT_VISIBLE] + a.counters[CNT_VISIBLE2], a.clusterCapacity);
    if (blockIdx.x == 0u && lane == 0u && a.hostFeedback && a.setupPart != 1u) {      // tell the host whether frames like this one should resolve without the tables (its hint for the next frames)
        const uint64_t tris = (uint64_t)a.counters[CNT_SUM_VERTS_HI];
        __hip_atomic_store(a.hostFeedback + 1, tris * a.inlineRatio > a.pixelCount ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (uint32_t c = firstCluster + blockIdx.x; c < clusterCount; c += gridDim.x) {
        const ClusterSetup cs = a.setup[c];
        if (cs.vertBase == BRMI_ARENA_NONE) continue;          // arena full: the pixel pass walks this cluster's data itself
        const uint32_t vertCount = cs.counts & 0xFFu, triCount = (cs.counts >> 8) & 0xFFu, posFormat = (cs.counts >> 16) & 0xFFu;
        const m4 objectToClip = load_m4(a.objConst + (size_t)cs.perObjectIndex * OBJ_CONST_FLOATS + 16u);
        const bool skinned = (cs.counts & BRMI_CS_SKINNED) != 0u;
        const uint32_t skinSlot = skinned ? a.sc.perMeshInstance[cs.instanceIndex].skinningInstanceSlot : 0xFFFFFFFFu;
        for (uint32_t v = lane; v < vertCount; v += 64) {
            f3 p{0.0f, 0.0f, 0.0f};
            if (posFormat == BRMI_POSITION_FORMAT_FLOAT3) { const float* pp = reinterpret_cast<const float*>(cs.posBase + v * 12u); p = f3{pp[0], pp[1], pp[2]}; }
            f3 n = oct_decode_normal(*reinterpret_cast<const uint32_t*>(cs.nrmBase + v * 4u));
            if (skinned) {      // ApplyClodSkinning (clodResolveCommon.hlsli:702-714): once per vertex here, not once per pixel and corner
                uint32_t joints[8]; float weights[8];
                load_skin_influences((cs.counts & BRMI_CS_JOINTS) ? cs.nrmBase + cs.jointDelta + v * 32u : nullptr, (cs.counts & BRMI_CS_WEIGHTS) ? cs.nrmBase + cs.weightDelta + v * 32u : nullptr, joints, weights);
                const m4 skin = build_skin_matrix(a.sc.skinningMatrices, skinSlot, joints, weights);
                p = xyz(mul_point(p, skin)); n = mul_v3m3(n, skin);
            }
            const f4 clip = mul_point(p, objectToClip);
            cx[v] = clip.x; cy[v] = clip.y; cw[v] = clip.w;
            a.verts[cs.vertBase + v] = ResolveVertex{p.x, p.y, p.z, n.x, n.y, n.z};
            if (a.uvs && (cs.counts & BRMI_CS_TEXTURED))
                for (uint32_t set = 0; set < a.uvSets; set++) { const f2 uv = decode_uv_set(a.clusterUv[c], set, v); a.uvs[(size_t)set * a.vertCapacity + cs.vertBase + v] = make_float2(uv.x, uv.y); }
            if (a.colors && (cs.counts & BRMI_CS_COLOR)) a.colors[cs.vertBase + v] = reinterpret_cast<const uint32_t*>(a.clusterUv[c].color)[v];
        }
        wave_lds_sync();      // (one wave per workgroup: the LDS hand-off must not wait for the arena stores in flight, round 5)
        for (uint32_t t = lane; t < triCount; t += 64) {
            const uint8_t* tb = cs.triBase + t * 3u;
            const uint32_t i0 = tb[0], i1 = tb[1], i2 = tb[2];
            // CalcFullBary, triangle part
            const f3 invW{rcpf(cw[i0]), rcpf(cw[i1]), rcpf(cw[i2])};
            const float n0x = cx[i0] * invW.x, n0y = cy[i0] * invW.x, n1x = cx[i1] * invW.y, n1y = cy[i1] * invW.y, n2x = cx[i2] * invW.z, n2y = cy[i2] * invW.z;
            const float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
            const float invDet = rcpf(ax * by - ay * bx);
            const f3 ddx = f3{n1y - n2y, n2y - n0y, n0y - n1y} * invDet * invW;
            const f3 ddy = f3{n2x - n1x, n0x - n2x, n1x - n0x} * invDet * invW;
            ResolveTriangle r;
            r.n0x = n0x; r.n0y = n0y; r.invW0 = invW.x;
            r.ddx[0] = ddx.x; r.ddx[1] = ddx.y; r.ddx[2] = ddx.z; r.ddy[0] = ddy.x; r.ddy[1] = ddy.y; r.ddy[2] = ddy.z;
            r.ddxSum = dot3(ddx, f3{1.0f, 1.0f, 1.0f}); r.ddySum = dot3(ddy, f3{1.0f, 1.0f, 1.0f});
            r.indices = i0 | (i1 << 8) | (i2 << 16);
            a.tris[cs.triBase32 + t] = r;
        }
        wave_lds_sync();
    }
}

// CalcFullBary, pixel part: only lambda is consumed when no texture / normal map is bound
BRMI_DEV f3 bary_lambda(const ResolveTriangle& r, float ndcX, float ndcY) {
    const float dx = ndcX - r.n0x, dy = ndcY - r.n0y;
    const float interpInvW = r.invW0 + dx * r.ddxSum + dy * r.ddySum;
    const float interpW = rcpf(interpInvW);
    f3 l;
    l.x = interpW * (r.invW0 + dx * r.ddx[0] + dy * r.ddy[0]);
    l.y = interpW * (0.0f + dx * r.ddx[1] + dy * r.ddy[1]);
    l.z = interpW * (0.0f + dx * r.ddx[2] + dy * r.ddy[2]);
    return l;
}

// CalcFullBary, the rest of the pixel part (clodResolveCommon.hlsli:128-141): screen-space derivatives of the barycentrics
struct BaryDeriv { f3 ddx, ddy; };
BRMI_DEV BaryDeriv bary_derivatives(const ResolveTriangle& r, f3 lambda, float ndcX, float ndcY, float winX, float winY) {
    const float dx = ndcX - r.n0x, dy = ndcY - r.n0y;
    const float interpInvW = r.invW0 + dx * r.ddxSum + dy * r.ddySum;
    const float sx = 2.0f / winX, sy = 2.0f / winY;
    const f3 ddx = f3{r.ddx[0], r.ddx[1], r.ddx[2]} * sx;
    const f3 ddy = (f3{r.ddy[0], r.ddy[1], r.ddy[2]} * sy) * -1.0f;
    const float ddxSum = r.ddxSum * sx, ddySum = (r.ddySum * sy) * -1.0f;
    const float interpW_ddx = 1.0f / (interpInvW + ddxSum), interpW_ddy = 1.0f / (interpInvW + ddySum);
    BaryDeriv d;
    d.ddx = interpW_ddx * (lambda * interpInvW + ddx) - lambda;
    d.ddy = interpW_ddy * (lambda * interpInvW + ddy) - lambda;
    return d;
}
// getContactRefinementParallaxCoordsAndHeight (parallax.hlsli:46-120; oracle: parallaxCoords in orc_resolve.cpp).  16 coarse steps
// along the tangent-space view ray, one refinement pass after the first hit, a secant between the last two points; p1 / p2 /
// parallaxAmount start as zero where the HLSL leaves them uninitialised.
BRMI_DEV float wrap1(float x) { const float y = x + 1.0f; return y - floorf(y); }
// Steps of the march whose fetches are issued together.  1 = the HLSL's loop as it is.  Larger batches were tried to overlap the fetches'
// latency (Sponza 4K, parallax on half of the textured materials: G-buffer 1.31 ms at 1, 1.42 at 2, 1.53 at 4, 1.90 at 8): the steps behind
// the first hit of a batch are wasted work and the pass is bound by its instruction count (texel addressing, decode, filter), not by latency.
#ifndef PARALLAX_BATCH
#define PARALLAX_BATCH 1
#endif
BRMI_DEV f2 parallax_coords(const TexelTables& tb, const TexBinding& height, f3 T, f3 B, f3 N, f2 uv, f3 viewDirWS, float heightmapScale, f2 dUVdx, f2 dUVdy) {
    uv.y = 1.0f - uv.y;
    const f3 viewDir = normalize3(f3{dot3(T, viewDirWS), dot3(B, viewDirWS), dot3(N, viewDirWS)});
    const float maxHeight = heightmapScale, minHeight = maxHeight * 0.5f;
    int numSteps = 16;
    const float viewCorrection = (-viewDir.z) + 2.0f;
    float stepSize = 1.0f / 17.0f;
    f2 stepOffset{viewDir.x * maxHeight * stepSize, viewDir.y * maxHeight * stepSize};
    f2 lastOffset{wrap1(viewDir.x * minHeight + uv.x), wrap1(viewDir.y * minHeight + uv.y)};
    float lastRayDepth = 1.0f, lastHeight = 1.0f;
    f2 p1{0.0f, 0.0f}, p2{0.0f, 0.0f};
    bool refine = false;
    // Every fetch of the march uses the pixel's gradients: the level of detail, the filter and the two mip levels are settled once.
    // The march itself is the HLSL's loop evaluated PARALLAX_BATCH steps at a time: the ray offsets of the next steps do not depend on the
    // heights (only the decision where to stop does), so their fetches are issued together -- one memory round trip per batch instead of
    // one per step -- and then examined in order; a step behind the first hit of its batch was fetched for nothing and changes nothing.
    const bool bound = height.bound;
    const LevelSetup ls = bound ? prepare_level(height, grad_lod(height, dUVdx, dUVdy)) : LevelSetup{};
    bool done = false;
#pragma nounroll
    while (numSteps > 0 && !done) {
        const int n = numSteps < PARALLAX_BATCH ? numSteps : PARALLAX_BATCH;
        f2 cand[PARALLAX_BATCH]; float depthAt[PARALLAX_BATCH], heightAt[PARALLAX_BATCH];
        f2 o = lastOffset; float dpt = lastRayDepth;
#pragma unroll
        for (int i = 0; i < PARALLAX_BATCH; i++) {
            o = f2{wrap1(o.x - stepOffset.x), wrap1(o.y - stepOffset.y)}; dpt = dpt - stepSize;
            cand[i] = o; depthAt[i] = dpt;
        }
#pragma unroll
        for (int i = 0; i < PARALLAX_BATCH; i++) heightAt[i] = (i < n) ? viewCorrection * (bound ? sample_prepared(tb, height, ls, cand[i]).x : 1.0f) : 0.0f;
#pragma unroll
        for (int i = 0; i < PARALLAX_BATCH; i++) {
            if (i >= n || done) continue;
            const float currentRayDepth = depthAt[i], currentHeight = heightAt[i];
            if (currentHeight > currentRayDepth) {
                p1 = f2{currentRayDepth, currentHeight};
                p2 = f2{lastRayDepth, lastHeight};
                if (refine) { done = true; continue; }
                refine = true;
                lastRayDepth = p2.x;
                stepSize = stepSize / (float)numSteps;
                stepOffset = f2{stepOffset.x / (float)numSteps, stepOffset.y / (float)numSteps};
                break;                                   // the rest of the batch was laid out with the coarse step: start over from here
            }
            lastOffset = cand[i];
            lastRayDepth = currentRayDepth;
            lastHeight = currentHeight;
            numSteps -= 1;
        }
    }
    const float diff1 = p1.x - p1.y, diff2 = p2.x - p2.y;
    const float denominator = diff2 - diff1;
    float parallaxAmount = 0.0f;
    if (denominator != 0.0f) parallaxAmount = (p1.x * diff2 - p2.x * diff1) / denominator;
    const float offset = ((1.0f - parallaxAmount) * -maxHeight) + minHeight;
    return f2{viewDir.x * offset + uv.x, viewDir.y * offset + uv.y};
}

BRMI_DEV float swizzle4(f4 v, uint32_t idx) { return idx == 0u ? v.x : idx == 1u ? v.y : idx == 2u ? v.z : v.w; }

// triangle + vertex tables of one pixel when its cluster has no arena space: the chain the setup kernel walks, per pixel
BRMI_DEV void resolve_tables_inline(const GBufferArgs& a, const ClusterSetup& cs, uint32_t triId, ResolveTriangle& r, f3 p[3], f3 n[3]) {
    const uint8_t* tb = cs.triBase + triId * 3u;
    const uint32_t ti[3] = {tb[0], tb[1], tb[2]};
    const m4 objectToClip = load_m4(a.objConst + (size_t)cs.perObjectIndex * OBJ_CONST_FLOATS + 16u);
    const bool skinned = (cs.counts & BRMI_CS_SKINNED) != 0u;
    const uint32_t skinSlot = skinned ? a.sc.perMeshInstance[cs.instanceIndex].skinningInstanceSlot : 0xFFFFFFFFu;
    f4 clip[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (((cs.counts >> 16) & 0xFFu) == BRMI_POSITION_FORMAT_FLOAT3) { const float* pp = reinterpret_cast<const float*>(cs.posBase + ti[k] * 12u); p[k] = f3{pp[0], pp[1], pp[2]}; }
        else p[k] = f3{0.0f, 0.0f, 0.0f};
        n[k] = oct_decode_normal(*reinterpret_cast<const uint32_t*>(cs.nrmBase + ti[k] * 4u));
        if (skinned) {
            uint32_t joints[8]; float weights[8];
            load_skin_influences((cs.counts & BRMI_CS_JOINTS) ? cs.nrmBase + cs.jointDelta + ti[k] * 32u : nullptr, (cs.counts & BRMI_CS_WEIGHTS) ? cs.nrmBase + cs.weightDelta + ti[k] * 32u : nullptr, joints, weights);
            const m4 skin = build_skin_matrix(a.sc.skinningMatrices, skinSlot, joints, weights);
            p[k] = xyz(mul_point(p[k], skin)); n[k] = mul_v3m3(n[k], skin);
        }
        clip[k] = mul_point(p[k], objectToClip);
    }
    const f3 invW{rcpf(clip[0].w), rcpf(clip[1].w), rcpf(clip[2].w)};
    const float n0x = clip[0].x * invW.x, n0y = clip[0].y * invW.x, n1x = clip[1].x * invW.y, n1y = clip[1].y * invW.y, n2x = clip[2].x * invW.z, n2y = clip[2].y * invW.z;
    const float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
    const float invDet = rcpf(ax * by - ay * bx);
    const f3 ddx = f3{n1y - n2y, n2y - n0y, n0y - n1y} * invDet * invW;
    const f3 ddy = f3{n2x - n1x, n0x - n2x, n1x - n0x} * invDet * invW;
    r.n0x = n0x; r.n0y = n0y; r.invW0 = invW.x;
    r.ddx[0] = ddx.x; r.ddx[1] = ddx.y; r.ddx[2] = ddx.z; r.ddy[0] = ddy.x; r.ddy[1] = ddy.y; r.ddy[2] = ddy.z;
    r.ddxSum = dot3(ddx, f3{1.0f, 1.0f, 1.0f}); r.ddySum = dot3(ddy, f3{1.0f, 1.0f, 1.0f});
    r.indices = ti[0] | (ti[1] << 8) | (ti[2] << 16);
}

template <typename M> BRMI_DEV m4 load_m4_any(const M* p) {     // p: float in the global or the constant address space
    m4 r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r.m[i][j] = p[i * 4 + j];
    return r;
}

#ifndef BRMI_RESOLVE_WATERFALL
#define BRMI_RESOLVE_WATERFALL 4
#endif
constexpr int RESOLVE_WATERFALL = BRMI_RESOLVE_WATERFALL;     // distinct mesh instances per 8x8 tile handled with scalar loads before falling back

// INLINE_TABLES: the arena may be too small for the frame, keep the per-pixel table path (it doubles the register count, so
// the host only picks this variant when the arena cannot hold every cluster the configuration allows).
// TEXTURED: some material of the scene samples textures (SampleMaterialEvalFromUvCache with its PSO_*_TEXTURE branches, evaluated
// per pixel from the material's flags); scenes of constant-factor materials run the lean instantiation.
#ifndef BRMI_GB_WAVES
#define BRMI_GB_WAVES 6
#endif
// MULTI_UV: some texture slot of the scene names a UV set > 0 (brmi_set_scene); the common single-set scenes run the variant without the set switching
// Waves per SIMD of the textured variants, re-measured on the build without packed pairs (Sponza 4K, G-buffer stage): single-set textured
// 3 / 4 waves 572 / 555 us; parallax 2 / 3 / 4 waves 1009 / 1055 / 1047 us; the MULTI_UV variants need their registers (three sets: 720 us at
// 3 waves against 752 at 4; with parallax 1387 at 3 against 1639 at 2)
#ifndef BRMI_GBT_WAVES
#define BRMI_GBT_WAVES 4
#endif
#ifndef BRMI_GBP_WAVES
#define BRMI_GBP_WAVES 2
#endif
#ifndef FRAME_EARLY
#define FRAME_EARLY 1
#endif
#ifndef BRMI_GBPM_WAVES
#define BRMI_GBPM_WAVES 3
#endif
#ifndef BRMI_GBM_WAVES
#define BRMI_GBM_WAVES 3
#endif
// Epi: what happens to a pixel's G-buffer words besides being stored.  `pixel()` is called once per lane and tile, in converged control flow,
// after the tile's waterfall: nothing for the plain kernels (round 5 removed the fused G-buffer + shading kernel, which lost on measurement twice).
struct NoEpilogue {
    static constexpr bool kWanted = false;
    BRMI_DEV void pixel(bool, unsigned long long, bool, uint32_t, uint32_t, uint64_t, const float4&, uint32_t, uint32_t, unsigned long long, unsigned long long) const {}
};
template <bool INLINE_TABLES, bool TEXTURED, bool PARALLAX, bool MULTI_UV, int SLIM, class Epi>
BRMI_DEV void gbuffer_body(const GBufferArgs& a, const Epi& epi) {
    const brmi_scene_buffers& sc = a.sc;
    if (a.variantSelect != 0u && (a.counters[CNT_RESOLVE_SPILL] != 0u) != (a.variantSelect == 2u)) return;     // the other variant's frame
    if (INLINE_TABLES && a.hostFeedback && blockIdx.x == 0u && threadIdx.x == 0u)      // a frame without the setup launch: this kernel tells the host whether frames like this one are still of that kind
        __hip_atomic_store(a.hostFeedback + 1, (uint64_t)a.counters[CNT_SUM_VERTS_HI] * a.inlineRatio > a.pixelCount ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ float texelTables[TEXTURED ? 512 : 1];          // code -> float: unorm, sRGB decode
    if (TEXTURED) { stage_texel_tables(texelTables, sc.srgbToLinear, threadIdx.x, 256u); __syncthreads(); }
    TexelTables tb; tb.t = texelTables;
    const brmi_per_frame* pf = sc.perFrame;
    const uint32_t clusterCount = min(a.counters[CNT_VISIBLE] + a.counters[CNT_VISIBLE2], a.clusterCapacity);
    // SLIM (chosen by the host, brmi_execute only): 1 = the depth map is final already (the chain build wrote it from the keys); 2 = also the
    // coat and fuzz planes hold the one word every material of the scene stores (brmi_config::keepUniformLayerPlanes: filled once after
    // brmi_setup) -- 20 of 56 B per pixel are not stored.  Compile-time: the same skips as run-time branches made the kernel slower than
    // storing everything (107 against 103 us; without the stores: 93)
    constexpr bool skipCoat = SLIM == 2, skipFuzz = SLIM == 2;
    // view-projection products are frame constants; every lane derives them the way the shader does
    // (the texture-sampling variants re-load the two products per tile through the scalar cache instead of keeping 32 SGPRs over the fetch loops:
    // their scalar registers spill into vector registers, and those are what the variant is short of)
    m4 unjVP_{}, prevVP_{};
    if (!TEXTURED) { unjVP_ = uni_m4(a.frameConst[1]); prevVP_ = uni_m4(a.frameConst[2]); }
    const float winX = uni((float)pf->screenResX), winY = uni((float)pf->screenResY);
    const uint64_t end = (a.pixelCount + 63ull) & ~63ull, stride = (uint64_t)gridDim.x * blockDim.x;
    // A wave is one 8x8 tile (firstPixel, the block size and the stride are multiples of 64).  The tile index, its row and column and every
    // plane's address of the tile are wave-uniform: kept in scalar registers (round 4; as per-lane 64-bit indices they were a dozen vector
    // registers of loop state, which the texture-sampling variants do not have), the lane contributes its constant offset inside the tile.
    const uint32_t lane0 = threadIdx.x & 63u;
    const uint64_t waveFirst = (uint64_t)blockIdx.x * blockDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    auto in_band = [&](uint64_t jb, uint32_t lane, uint32_t& px, uint32_t& py) {       // jb: first pixel of the wave's tile, relative to firstPixel (scalar)
        const uint32_t tile = (uint32_t)((a.firstPixel + jb) >> 6);
        px = (tile % a.tilesX) * 8u + (lane >> 3); py = (tile / a.tilesX) * 8u + (lane & 7u);
        return jb + lane < a.pixelCount && px < a.W && py < a.H && py >= a.bandY0 && py < a.bandY1;
    };
    // software pipeline: the key of the next tile is requested while this one is resolved (the chain key -> cluster -> triangle -> vertices is
    // four dependent loads)
    uint32_t px = 0, py = 0;
    bool nvalid = waveFirst < end && in_band(waveFirst, lane0, px, py);
    unsigned long long nkey = nvalid ? __builtin_nontemporal_load(a.vis + a.firstPixel + waveFirst + lane0) : BRMI_VIS_EMPTY;
    for (uint64_t jb = waveFirst; jb < end; jb += stride) {
        const uint64_t ib = a.firstPixel + jb;             // scalar; the lane's pixel is ib + lane
        // (the lane index behind an opaque copy: what derives from it -- row and column inside the tile, the byte offsets of the plane stores -- is
        // then computed where it is used, a VALU instruction each, instead of being hoisted out of the tile loop into registers of their own)
        uint32_t lane = lane0;
        asm volatile("" : "+v"(lane));
        const uint64_t i = ib + lane;
        bool valid = in_band(jb, lane, px, py);
        const bool inBand = valid;
        const unsigned long long key = nkey;
        float4 outN = make_float4(0.0f, 0.0f, 0.0f, 0.0f); uint32_t outAl = 0u, outMr = 0u; unsigned long long outCoat = 0ull, outEmis = 0ull;      // the pixel's words, for the epilogue
        if (jb + stride < end) { uint32_t qx, qy; nvalid = in_band(jb + stride, lane, qx, qy); nkey = nvalid ? __builtin_nontemporal_load(a.vis + a.firstPixel + jb + stride + lane) : BRMI_VIS_EMPTY; }
        if (SLIM == 0 && valid && a.depth) __builtin_nontemporal_store((key == BRMI_VIS_EMPTY) ? as_f32(BRMI_DEPTH_EMPTY_BITS) : as_f32(((uint32_t)(key >> BRMI_VIS_META_BITS)) << 1), a.depth + ib + lane);
        const uint32_t triId = (uint32_t)(key & 0x7Full);
        const uint32_t clusterIndex = (uint32_t)((key >> BRMI_VIS_TRI_BITS) & 0x3FFFFFFull);
        valid = valid && key != BRMI_VIS_EMPTY && clusterIndex < clusterCount;
        ClusterSetup cs{};
        if (valid) { cs = a.setup[clusterIndex]; valid = triId < ((cs.counts >> 8) & 0xFFu); }
        // per-pixel part of the tables
        ResolveTriangle r{}; f3 p[3] = {}, n[3] = {}; f2 tc[3] = {}; uint32_t vc[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (valid) {
            const bool wantUv = TEXTURED && (cs.counts & BRMI_CS_TEXTURED) != 0u, wantColor = TEXTURED && (cs.counts & BRMI_CS_COLOR) != 0u;
            if (cs.vertBase != BRMI_ARENA_NONE) {
                r = a.tris[cs.triBase32 + triId];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t vi = cs.vertBase + ((r.indices >> (8 * k)) & 0xFFu);
                    const ResolveVertex v = a.verts[vi];
                    p[k] = f3{v.px, v.py, v.pz}; n[k] = f3{v.nx, v.ny, v.nz};
                    if (wantUv) { const float2 u = a.uvs[vi]; tc[k] = f2{u.x, u.y}; }
                    if (wantColor) vc[k] = a.colors[vi];
                }
            } else if (INLINE_TABLES) {
                resolve_tables_inline(a, cs, triId, r, p, n);
                if (wantUv) {
                    const ClusterUv cu = a.clusterUv[clusterIndex];
#pragma unroll
                    for (int k = 0; k < 3; k++) tc[k] = decode_uv(cu, (r.indices >> (8 * k)) & 0xFFu);
                }
                if (wantColor) {
                    const uint32_t* col = reinterpret_cast<const uint32_t*>(a.clusterUv[clusterIndex].color);
#pragma unroll
                    for (int k = 0; k < 3; k++) vc[k] = col[(r.indices >> (8 * k)) & 0xFFu];
                }
            } else valid = false;      // cannot happen: the arena holds every cluster of this configuration
        }
        // the corners' texcoords of a set > 0 (rare: fetched only by the slots that name one)
        auto corner_uvs = [&](uint32_t set, f2 (&o)[3]) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t local = (r.indices >> (8 * k)) & 0xFFu;
                if (cs.vertBase != BRMI_ARENA_NONE) { const float2 u = set < a.uvSets ? a.uvs[(size_t)set * a.vertCapacity + cs.vertBase + local] : make_float2(0.0f, 0.0f); o[k] = f2{u.x, u.y}; }
                else o[k] = decode_uv_set(a.clusterUv[clusterIndex], set, local);
            }
        };
        const float uvx = ((float)px + 0.5f) / winX, uvy = ((float)(stripe_rrow(a.stripes, (uint32_t)__builtin_amdgcn_readfirstlane((int)(py & ~7u))) + (py & 7u)) + 0.5f) / winY;      // the pixel's row in the FRAME (chunks are multiples of 16 rows: the tile's eight rows lie in one, so its first row is mapped in scalar arithmetic)
        const float ndcX = uvx * 2.0f - 1.0f, ndcY = (1.0f - uvy) * 2.0f - 1.0f;
        const f3 l = bary_lambda(r, ndcX, ndcY);
        const f3 posOS{dot3(f3{p[0].x, p[1].x, p[2].x}, l), dot3(f3{p[0].y, p[1].y, p[2].y}, l), dot3(f3{p[0].z, p[1].z, p[2].z}, l)};
        const f3 normalOS = normalize3(f3{dot3(f3{n[0].x, n[1].x, n[2].x}, l), dot3(f3{n[0].y, n[1].y, n[2].y}, l), dot3(f3{n[0].z, n[1].z, n[2].z}, l)});

        // object / material part: model, previous model, normal matrix and the material's packed words.  `obj`, `nm`, `mw`
        // are either scalar (constant address space, wave-uniform index) or per-lane pointers.
        auto finish = [&](auto obj, auto nm, auto mw, auto mat) {
            const m4 model = load_m4_any(&obj->model[0][0]);
            const f3 worldPosition = xyz(mul_point(posOS, model));
            const m4 normalMatrix = load_m4_any(nm);
            const f3 worldNormal = normalize3(mul_v3m3(normalOS, normalMatrix));
            // ComputeClodMotionVector
            const m4 unjVP = TEXTURED ? load_m4_any(&kconst(a.frameConst)[1].m[0][0]) : unjVP_, prevVP = TEXTURED ? load_m4_any(&kconst(a.frameConst)[2].m[0][0]) : prevVP_;
            const f4 clipCur = mul_point(worldPosition, unjVP);
            const f3 prevWorld = xyz(mul_point(posOS, load_m4_any(&obj->prevModel[0][0])));
            const f4 clipPrev = mul_point(prevWorld, prevVP);
            const float mvx = clipCur.x / clipCur.w - clipPrev.x / clipPrev.w, mvy = clipCur.y / clipCur.w - clipPrev.y / clipPrev.w;
            // (stored here, not with the other planes: the word does not depend on the material, and its two inputs would stay live over the texture fetches)
            __builtin_nontemporal_store((uint32_t)(f32_to_f16_bits(mvx) | (f32_to_f16_bits(mvy) << 16)), a.motion + ib + lane);
            uint32_t albedoW = mw->albedo, mrW = mw->metallicRoughness;
            unsigned long long emissiveW = mw->emissive, coatW = mw->coat, fuzzW = mw->fuzz;
            f3 normalWS = worldNormal;
            if (TEXTURED) {
                const uint32_t flags = mat->materialFlags;
                // DecodeCompressedColor + interpolation (clodResolveCommon.hlsli:657-667,1641-1648); white where the page has no colours
                const bool colored = (cs.counts & BRMI_CS_COLOR) != 0u;
                f3 vertexColor{1.0f, 1.0f, 1.0f};
                if (colored) vertexColor = f3{dot3(f3{tb.t[vc[0] & 0xFFu], tb.t[vc[1] & 0xFFu], tb.t[vc[2] & 0xFFu]}, l), dot3(f3{tb.t[(vc[0] >> 8) & 0xFFu], tb.t[(vc[1] >> 8) & 0xFFu], tb.t[(vc[2] >> 8) & 0xFFu]}, l),
                                              dot3(f3{tb.t[(vc[0] >> 16) & 0xFFu], tb.t[(vc[1] >> 16) & 0xFFu], tb.t[(vc[2] >> 16) & 0xFFu]}, l)};
                if (!(flags & BRMI_MATERIAL_ANY_TEXTURE) && colored)
                    albedoW = pack_unorm4(mat->baseColorFactor[0] * vertexColor.x, mat->baseColorFactor[1] * vertexColor.y, mat->baseColorFactor[2] * vertexColor.z, 1.0f);
                const bool layerTex = (mw->pad & 1u) != 0u;
                if ((flags & BRMI_MATERIAL_ANY_TEXTURE) || layerTex) {
                    // BuildClodMaterialUvData + SampleMaterialEvalFromUvCache (utilities.hlsli:1850-2075).  uv / dUVdx / dUVdy are the texcoord and
                    // gradients of `curSet`; use_set() switches them to the set a slot names (AppendClodMaterialUvSample: indices >= 8 mean set 0).
                    const BaryDeriv bd = bary_derivatives(r, l, ndcX, ndcY, winX, winY);
                    f2 uv, dUVdx, dUVdy;
                    uint32_t curSet = 0u, parallaxSet = 0xFFFFFFFFu; f2 parallaxUv{};
                    auto interpolate = [&](const f2 (&c)[3]) {
                        const f3 us{c[0].x, c[1].x, c[2].x}, vs{c[0].y, c[1].y, c[2].y};
                        uv = f2{dot3(us, l), dot3(vs, l)};
                        dUVdx = f2{dot3(us, bd.ddx), dot3(vs, bd.ddx)}; dUVdy = f2{dot3(us, bd.ddy), dot3(vs, bd.ddy)};
                    };
                    auto use_set = [&](uint32_t setIndex) {
                        if (!MULTI_UV) return;
                        const uint32_t set = setIndex < 8u ? setIndex : 0u;
                        if (set == curSet) return;
                        curSet = set;
                        if (set == 0u) interpolate(tc); else { f2 c[3]; corner_uvs(set, c); interpolate(c); }
                        if (PARALLAX && set == parallaxSet) uv = parallaxUv;      // ResolveMaterialUvSample: the displaced texcoord, the set's own gradients
                    };
                    interpolate(tc);
                    // texture / sampler tables in the address space of the material pointer: scalar loads on the waterfall path
                    auto texturesP = as_space_of(mat, sc.textures); auto samplersP = as_space_of(mat, sc.samplers);
                    auto bind = [&](uint32_t ti, uint32_t si) { return bind_texture(texturesP, sc.textureCount, samplersP, sc.samplerCount, ti, si); };
                    f3 Tn{}, Bn{};
                    auto cotangent_frame = [&]() {
                        // dpdx / dpdy through the model's 3x3 (clodResolveCommon.hlsli:1607-1624), cotangent_frame_from_derivs (utilities.hlsli:323-336)
                        const f3 pxs{p[0].x, p[1].x, p[2].x}, pys{p[0].y, p[1].y, p[2].y}, pzs{p[0].z, p[1].z, p[2].z};
                        const f3 dpdx = mul_v3m3(f3{dot3(pxs, bd.ddx), dot3(pys, bd.ddx), dot3(pzs, bd.ddx)}, model);
                        const f3 dpdy = mul_v3m3(f3{dot3(pxs, bd.ddy), dot3(pys, bd.ddy), dot3(pzs, bd.ddy)}, model);
                        const f3 dp2perp = cross3(dpdy, worldNormal), dp1perp = cross3(worldNormal, dpdx);
                        const f3 T = dp2perp * dUVdx.x + dp1perp * dUVdy.x, B = dp2perp * dUVdx.y + dp1perp * dUVdy.y;
                        const float invmax = rsqrtf_(max2(dot3(T, T), dot3(B, B)));
                        Tn = T * invmax; Bn = B * invmax;
                    };
                    // with parallax the frame is needed before the fetches; without, building it next to its only use keeps six registers free over them.
                    // BuildMaterialUvBindings: the frame follows the normal slot's UV set, else the height slot's
                    if (PARALLAX && (flags & (BRMI_MATERIAL_NORMAL_MAP | BRMI_MATERIAL_PARALLAX))) { use_set((flags & BRMI_MATERIAL_NORMAL_MAP) ? mat->normalUvSetIndex : mat->heightUvSetIndex); cotangent_frame(); }
                    // single-set scenes: the frame before the fetches too -- six values live over them instead of the corner positions and the barycentric derivatives (15)
                    if (!PARALLAX && FRAME_EARLY && (flags & BRMI_MATERIAL_NORMAL_MAP)) { use_set(mat->normalUvSetIndex); cotangent_frame(); }
                    if (PARALLAX && (flags & BRMI_MATERIAL_PARALLAX)) {       // PSO_PARALLAX (utilities.hlsli:1869-1897): every slot on the height map's UV set moves with it
                        const brmi_camera* cam = sc.cameras + sc.perFrame->mainCameraIndex;
                        const f3 camPos{cam->positionWorldSpace[0], cam->positionWorldSpace[1], cam->positionWorldSpace[2]};
                        use_set(mat->heightUvSetIndex);
                        parallaxUv = parallax_coords(tb, bind(mat->heightMapIndex, mat->heightSamplerIndex), Tn, Bn, worldNormal, uv, normalize3(camPos - worldPosition), mat->heightMapScale, dUVdx, dUVdy);
                        parallaxSet = curSet; uv = parallaxUv;
                    }
                    // One loop over the six material slots (one copy of the sampler code).  Metallic, roughness and occlusion usually are
                    // channels of ONE texture (glTF packing): a slot bound like the previous one reuses its fetch.  A slot's sample is consumed
                    // where it arrives -- a channel, a product with the factor -- so that 13 values instead of six float4 stay live over the
                    // fetches (round 4: the kernel spilled 67 registers at four waves per SIMD and half of its HBM writes were scratch).
                    f4 baseColor{mat->baseColorFactor[0], mat->baseColorFactor[1], mat->baseColorFactor[2], mat->baseColorFactor[3]};
                    float metallic = mat->metallicFactor, roughness = mat->roughnessFactor, ao = 1.0f;
                    f3 sNormal{}, sEmis{};
                    const auto* opRec = as_space_of(mat, sc.openpbrMaterials) + mat->openPBRMaterialDataIndex;
                    {
                        uint32_t prevTi = 0xFFFFFFFFu, prevSi = 0xFFFFFFFFu, prevSet = 0xFFFFFFFFu; f4 prevSample{};      // same (texture, sampler, UV set) as the previous slot: same fetch
#pragma nounroll
                        for (uint32_t slot = 0; slot < 6u; slot++) {
                            uint32_t bit, ti, si, set;
                            switch (slot) {
                                case 0: bit = BRMI_MATERIAL_BASE_COLOR_TEXTURE; ti = mat->baseColorTextureIndex; si = mat->baseColorSamplerIndex; set = mat->baseColorUvSetIndex; break;
                                case 1: bit = BRMI_MATERIAL_METALLIC_TEXTURE; ti = mat->metallicTextureIndex; si = mat->metallicSamplerIndex; set = mat->metallicUvSetIndex; break;
                                case 2: bit = BRMI_MATERIAL_ROUGHNESS_TEXTURE; ti = mat->roughnessTextureIndex; si = mat->roughnessSamplerIndex; set = mat->roughnessUvSetIndex; break;
                                case 3: bit = BRMI_MATERIAL_AO_TEXTURE; ti = mat->aoMapIndex; si = mat->aoSamplerIndex; set = mat->aoUvSetIndex; break;
                                case 4: bit = BRMI_MATERIAL_NORMAL_MAP; ti = mat->normalTextureIndex; si = mat->normalSamplerIndex; set = mat->normalUvSetIndex; break;
                                default: bit = BRMI_MATERIAL_EMISSIVE_TEXTURE; ti = mat->emissiveTextureIndex; si = mat->emissiveSamplerIndex; set = mat->emissiveUvSetIndex; break;
                            }
                            if (!(flags & bit)) continue;
                            use_set(set);
                            f4 t = prevSample;
                            if (ti != prevTi || si != prevSi || (MULTI_UV && curSet != prevSet) || ti >= sc.textureCount || si >= sc.samplerCount) t = sample_grad(tb, bind(ti, si), uv, dUVdx, dUVdy);
                            prevTi = ti; prevSi = si; prevSet = curSet; prevSample = t;
                            if (slot == 0u) baseColor = f4{baseColor.x * t.x, baseColor.y * t.y, baseColor.z * t.z, baseColor.w * t.w};
                            else if (slot == 1u) metallic = swizzle4(t, mat->metallicChannel) * mat->metallicFactor;
                            else if (slot == 2u) roughness = swizzle4(t, mat->roughnessChannel) * mat->roughnessFactor;
                            else if (slot == 3u) ao = swizzle4(t, mat->aoChannel);
                            else if (slot == 4u) sNormal = f3{t.x, t.y, t.z};
                            else sEmis = f3{swizzle4(t, mat->emissiveChannels[0]), swizzle4(t, mat->emissiveChannels[1]), swizzle4(t, mat->emissiveChannels[2])};
                        }
                    }
                    if (flags & BRMI_MATERIAL_NORMAL_MAP) {
                        if (!PARALLAX && !FRAME_EARLY) { use_set(mat->normalUvSetIndex); cotangent_frame(); }
                        f3 tn = normalize3(sNormal * 2.0f - f3{1.0f, 1.0f, 1.0f});
                        if (flags & BRMI_MATERIAL_NEGATE_NORMALS) tn = -tn;
                        if (flags & BRMI_MATERIAL_INVERT_NORMAL_GREEN) tn.y = -tn.y;
                        normalWS = normalize3(f3{(tn.x * Tn.x + tn.y * Bn.x) + tn.z * worldNormal.x, (tn.x * Tn.y + tn.y * Bn.y) + tn.z * worldNormal.y, (tn.x * Tn.z + tn.y * Bn.z) + tn.z * worldNormal.z});
                    }
                    if (flags & BRMI_MATERIAL_EMISSIVE_TEXTURE) {
                        const f3 emissiveIn = sEmis * f3{mat->emissiveFactor[0], mat->emissiveFactor[1], mat->emissiveFactor[2]};
                        // ResolveCanonicalOpenPBRSurface: an all-zero sampled emissive falls back to the OpenPBR record's
                        const f3 canonical = f3{opRec->emissionColor[0], opRec->emissionColor[1], opRec->emissionColor[2]} * opRec->emissionLuminance;
                        const f3 e = dot3(emissiveIn, emissiveIn) > 0.0f ? emissiveIn : canonical;
                        emissiveW = pack_half4(e.x, e.y, e.z, 0.0f);
                    }
                    albedoW = pack_unorm4(baseColor.x * vertexColor.x, baseColor.y * vertexColor.y, baseColor.z * vertexColor.z, ao);
                    mrW = (pack_unorm4(metallic, roughness, 0.0f, 0.0f) & 0xFFFFu) | (mrW & 0xFFFF0000u);       // coat roughness / fuzz weight stay the material's
                    if (layerTex) {
                        // ApplyOpenPBRTextureSampling (utilities.hlsli:720-846): the six coat / fuzz slots, bound when both indices are valid.  A loop of
                        // its own behind the packing of the base words: its six samples are not live over the material slots' fetches.
                        // surface.x = saturate(record.x) [ResolveCanonicalOpenPBRSurface]; x *= sample (1 when the slot is unbound); x = saturate(x) -- applied
                        // where the sample arrives: ten values live over the fetches instead of six float4
                        auto tbw = [&](uint32_t w) { return opRec->textureBindings[w]; };
                        const f3 coatColor = sat3(f3{opRec->coatColor[0], opRec->coatColor[1], opRec->coatColor[2]}), fuzzColor = sat3(f3{opRec->fuzzColor[0], opRec->fuzzColor[1], opRec->fuzzColor[2]});
                        f3 cc = sat3(coatColor * f3{1.0f, 1.0f, 1.0f}), fc = sat3(fuzzColor * f3{1.0f, 1.0f, 1.0f});
                        float cw = sat(sat(opRec->coatWeight) * 1.0f), cr = sat(sat(opRec->coatRoughness) * 1.0f), fw = sat(sat(opRec->fuzzWeight) * 1.0f), fr = sat(sat(opRec->fuzzRoughness) * 1.0f);
                        {
                            uint32_t prevTi = 0xFFFFFFFFu, prevSi = 0xFFFFFFFFu, prevSet = 0xFFFFFFFFu; f4 prevSample{};
#pragma nounroll
                            for (uint32_t slot = 0; slot < 6u; slot++) {
                                const uint32_t ti = opRec->textureBindings[2u * slot], si = opRec->textureBindings[2u * slot + 1u], set = opRec->textureBindings[26u + slot];
                                if (ti == 0xFFFFFFFFu || si == 0xFFFFFFFFu) continue;
                                use_set(set);
                                f4 t = prevSample;
                                if (ti != prevTi || si != prevSi || (MULTI_UV && curSet != prevSet) || ti >= sc.textureCount || si >= sc.samplerCount) t = sample_grad(tb, bind(ti, si), uv, dUVdx, dUVdy);
                                prevTi = ti; prevSi = si; prevSet = curSet; prevSample = t;
                                if (slot == 0u) cc = sat3(coatColor * f3{swizzle4(t, tbw(12)), swizzle4(t, tbw(13)), swizzle4(t, tbw(14))});
                                else if (slot == 1u) cw = sat(sat(opRec->coatWeight) * swizzle4(t, tbw(16)));
                                else if (slot == 2u) cr = sat(sat(opRec->coatRoughness) * swizzle4(t, tbw(17)));
                                else if (slot == 3u) fc = sat3(fuzzColor * f3{swizzle4(t, tbw(19)), swizzle4(t, tbw(20)), swizzle4(t, tbw(21))});
                                else if (slot == 4u) fw = sat(sat(opRec->fuzzWeight) * swizzle4(t, tbw(23)));
                                else fr = sat(sat(opRec->fuzzRoughness) * swizzle4(t, tbw(24)));
                            }
                        }
                        coatW = pack_half4(cc.x, cc.y, cc.z, cw);
                        fuzzW = pack_half4(fc.x, fc.y, fc.z, fr);
                        mrW = (mrW & 0xFFFFu) | (pack_unorm4(0.0f, 0.0f, cr, fw) & 0xFFFF0000u);
                    }
                }
            }
            // streaming stores: 52 B per pixel written once and read once by the shading pass; keeping them out of the caches
            // leaves L2 / the memory-side cache to the vertex tables, the keys and the shading pass's own reads (0.149 -> 0.123 ms,
            // and 0.339 -> 0.324 ms for k_shade)
            uint32_t laneS = lane0;
            asm volatile("" : "+v"(laneS));
            { float4 nv = make_float4(normalWS.x, normalWS.y, normalWS.z, mw->opIndexF); __builtin_nontemporal_store(nv.x, &(a.normals + ib)[laneS].x); __builtin_nontemporal_store(nv.y, &(a.normals + ib)[laneS].y); __builtin_nontemporal_store(nv.z, &(a.normals + ib)[laneS].z); __builtin_nontemporal_store(nv.w, &(a.normals + ib)[laneS].w); }
            __builtin_nontemporal_store(albedoW, a.albedo + ib + laneS);
            if (!skipCoat) __builtin_nontemporal_store(coatW, a.coat + ib + laneS);
            __builtin_nontemporal_store(emissiveW, a.emissive + ib + laneS);
            if (!skipFuzz) __builtin_nontemporal_store(fuzzW, a.fuzz + ib + laneS);
            __builtin_nontemporal_store(mrW, a.metallicRoughness + ib + laneS);
            if (Epi::kWanted) { outN = make_float4(normalWS.x, normalWS.y, normalWS.z, mw->opIndexF); outAl = albedoW; outMr = mrW; outCoat = coatW; outEmis = emissiveW; }
        };
        // waterfall over the distinct mesh instances of the tile (usually one or two)
        uint64_t pending = __ballot(valid);
        for (int it = 0; pending != 0ull; it++) {
            if (RESOLVE_WATERFALL > 0 && it == RESOLVE_WATERFALL) {      // a tile of many small instances: per-lane loads for the rest
                if ((pending >> lane_id()) & 1ull)
                    finish(sc.perObject + cs.perObjectIndex, sc.normalMatrices + (size_t)cs.normalMatrixIndex * 16u, a.matWords + cs.materialDataIndex, sc.materials + cs.materialDataIndex);
                break;
            }
            const int lead = __ffsll((unsigned long long)pending) - 1;
            const uint32_t uInst = (uint32_t)__builtin_amdgcn_readlane((int)cs.instanceIndex, lead);
            const uint32_t uObj = (uint32_t)__builtin_amdgcn_readlane((int)cs.perObjectIndex, lead);
            const uint32_t uNm = (uint32_t)__builtin_amdgcn_readlane((int)cs.normalMatrixIndex, lead);
            const uint32_t uMat = (uint32_t)__builtin_amdgcn_readlane((int)cs.materialDataIndex, lead);
            const uint64_t same = __ballot(valid && cs.instanceIndex == uInst);
            pending &= ~same;
            if ((same >> lane_id()) & 1ull) finish(kconst(sc.perObject) + uObj, kconst(sc.normalMatrices) + (size_t)uNm * 16u, kconst(a.matWords) + uMat, kconst(sc.materials) + uMat);
        }
        if (Epi::kWanted) epi.pixel(inBand, key, valid, px, py, i, outN, outAl, outMr, outCoat, outEmis);
    }
}
template <bool INLINE_TABLES, bool TEXTURED, bool PARALLAX = false, bool MULTI_UV = false, int SLIM = 0>
__global__ void __launch_bounds__(256, INLINE_TABLES ? 1 : (MULTI_UV ? (PARALLAX ? BRMI_GBPM_WAVES : BRMI_GBM_WAVES) : (PARALLAX ? BRMI_GBP_WAVES : (TEXTURED ? BRMI_GBT_WAVES : BRMI_GB_WAVES)))) k_gbuffer(GBufferArgs a) {
    wave_prio<PRIO_GBUFFER>();
    gbuffer_body<INLINE_TABLES, TEXTURED, PARALLAX, MULTI_UV, SLIM>(a, NoEpilogue{});
}

static GBufferArgs gbuffer_args_of(brmi_pass* p) {
    GBufferArgs a;
    a.hostFeedback = nullptr; a.setupPart = 0u; a.inlineRatio = resolve_inline_ratio(p);
    a.sc = shading_scene_of(p);      // the frame's camera / per-frame record as the constants kernel saw them (FrameSnapshot)
    a.clusters = static_cast<const uint4*>(p->res[BRMI_RES_VISIBLE_CLUSTERS]); a.counters = p->counters();
    a.vis = static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]);
    a.normals = static_cast<float4*>(p->res[BRMI_RES_GBUF_NORMALS]); a.albedo = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_ALBEDO]);
    a.coat = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_COAT]); a.emissive = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_EMISSIVE]);
    a.fuzz = static_cast<unsigned long long*>(p->res[BRMI_RES_GBUF_FUZZ]); a.metallicRoughness = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_METALLIC_ROUGHNESS]);
    a.motion = static_cast<uint32_t*>(p->res[BRMI_RES_GBUF_MOTION_VECTORS]); a.depth = p->depthFinal ? nullptr : static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]);
    a.W = p->cfg.width; a.H = p->cfg.height; a.tilesX = p->tilesX; a.bandY0 = p->bandY0; a.bandY1 = p->bandY1; a.firstPixel = p->bandFirstPixel; a.pixelCount = p->bandPixelCount;
    a.stripes = p->stripes;
    a.clusterCapacity = p->cfg.maxVisibleClusters;
    a.frameConst = p->wsPtr<m4>(p->ws.frameConst); a.objConst = p->wsPtr<float>(p->ws.objConst);
    a.setup = p->wsPtr<ClusterSetup>(p->ws.clusterSetup); a.verts = p->wsPtr<ResolveVertex>(p->ws.resolveVerts); a.tris = p->wsPtr<ResolveTriangle>(p->ws.resolveTris);
    a.vertCapacity = p->resolveCapacity; a.triCapacity = p->resolveCapacity;
    a.matWords = p->wsPtr<MaterialWords>(p->ws.matWords);
    a.clusterUv = p->sceneHasTextures ? p->wsPtr<ClusterUv>(p->ws.clusterUv) : nullptr;
    a.uvs = p->sceneHasTextures ? p->wsPtr<float2>(p->ws.resolveUVs) : nullptr; a.uvSets = p->sceneUvSets;
    a.colors = p->sceneHasVertexColors ? p->wsPtr<uint32_t>(p->ws.resolveColors) : nullptr;
    if (p->sceneHasVertexColors) a.clusterUv = p->wsPtr<ClusterUv>(p->ws.clusterUv);
    return a;
}

// The per-cluster tables of the pixel pass.  Part of brmi_gbuffer; brmi_execute_split runs it at the end of the geometry half instead (it
// needs the final cluster list and keys only), so that the shading half starts with the pixel pass.
// whether frames like the recent ones want the marking pass (which needs the final keys: the setup then cannot start before the rasteriser is done)
// Frames of many triangles per pixel (Zorah-class: 62 M cluster triangles for 33 M pixels, 2.3 M of them owning a pixel) resolve WITHOUT the per-cluster tables (round 5): a triangle's table entry is
// read by a few pixels or, mostly, by none, so making it in a pass of its own -- 0.29 ms at the end of the 8K frame's geometry half, 830 MB written and read back -- costs more than
// deriving it where the pixel needs it (the G-buffer kernel's INLINE_TABLES form, until then the fallback for a full arena): 8K frame in flight 2.96 -> 2.66 ms, the
// dense 4K frame (0.34 triangles per pixel) 0.724 -> 0.684; Bistro- / Sponza- / San-Miguel-class frames (0.12 - 0.2) keep the tables (0.50 -> 0.55, 0.385 -> 0.464, 0.82 -> 1.00 without).
// The ratio: a quarter of a triangle per pixel; half for scenes whose pixels fetch texcoords or vertex colours too (the in-place form decodes them per pixel).  No cut through the
// scene's DAGs below the ratio can be such a frame, and whether recent frames were is a word the device stores for the host (no wait; a stale answer picks the other, equally
// exact, form).  BRMI_TUNING=resolve_inline=0 / 1: never / always (tests run scenes both ways).  (Replaces round 4's marking pass, which only skipped whole clusters.)
uint32_t resolve_inline_ratio(const brmi_pass* p) { return (p->sceneHasTextures || p->sceneHasVertexColors) ? 2u : 4u; }
bool resolve_inline_frame(brmi_pass* p) {
    if (p->resolveInlineMode >= 0) return p->resolveInlineMode != 0;
    return (uint64_t)p->totalBits * BRMI_MESHLET_MAX_TRIS * resolve_inline_ratio(p) > p->bandPixelCount && p->ensureFeedback() && reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[1] != 0u;
}
int launch_resolve_setup(brmi_pass* p, hipStream_t s, uint32_t part) {
    if (int rc = ensure_frame_constants(p, s)) return rc;
    if (p->inlineResolve) return BRMI_OK;      // (the compaction marked every cluster "no tables")
    GBufferArgs a = gbuffer_args_of(p);
    a.hostFeedback = p->ensureFeedback() ? p->phase2FeedbackDev : nullptr;
    a.setupPart = part;
    if (part != 0u) {      // (brmi_execute_split: phase-1 clusters beside the rasteriser on the shading stream, phase-2 clusters -- usually a handful -- at the end of the geometry half)
        hipLaunchKernelGGL(k_resolve_setup, dim3(part == 1u ? 8192 : 512), dim3(64), 0, s, a);
        BRMI_LAUNCH_CHECK(p, "k_resolve_setup");
        return BRMI_OK;
    }
    hipLaunchKernelGGL(k_resolve_setup, dim3(8192), dim3(64), 0, s, a);
    BRMI_LAUNCH_CHECK(p, "k_resolve_setup");
    return BRMI_OK;
}

int launch_gbuffer(brmi_pass* p, hipStream_t s) {
    if (int rc = ensure_frame_constants(p, s)) return rc;
    if (p->resolveSetupDone) p->resolveSetupDone = false;
    else if (int rc = launch_resolve_setup(p, s, 0u)) return rc;
    GBufferArgs a = gbuffer_args_of(p);
    // inside brmi_execute with occlusion culling the depth map is final; with the layer planes holding the scene's one coat / fuzz word the
    // slim instantiations leave 20 B per pixel unwritten.  The fallback variant (arena overflow) writes everything: same values.
    const int slim = p->depthFinal ? (p->layerPlanesUniform ? 2 : 1) : 0;
    const bool lean = (uint64_t)p->resolveCapacity >= (uint64_t)p->cfg.maxVisibleClusters * BRMI_MESHLET_MAX_TRIS;
    // `lean`: the arena holds every visible cluster even at 128 vertices / triangles each.  Otherwise whether one spilled is only
    // known on the device: both variants are launched and each leaves at once when the frame is the other one's (a ~5 us empty
    // launch against the two-waves-per-SIMD fallback variant on frames that do not need it: San-Miguel-class 4K 1.04 -> 0.98 ms).
    auto launch = [&](auto leanKernel, auto fallbackKernel) {
        if (p->inlineResolve) {      // every cluster without tables: the in-place form alone
            a.variantSelect = 0u; a.hostFeedback = p->ensureFeedback() ? p->phase2FeedbackDev : nullptr;
            hipLaunchKernelGGL(fallbackKernel, dim3(8192), dim3(256), 0, s, a);
            return;
        }
        a.variantSelect = lean ? 0u : 1u;
        // the texture-sampling variants' waves live long: twice the workgroups shorten the tail (Sponza 4K textured 486 -> 472 us, parallax 1001 -> 963);
        // the constant-factor variant is best at 4096
        const uint32_t grid = (p->sceneHasTextures || p->sceneHasVertexColors) ? 8192u : (p->shadeSharesChip ? p->gbufferGridShared : 4096u);
        hipLaunchKernelGGL(leanKernel, dim3(grid), dim3(256), 0, s, a);
        if (!lean) { a.variantSelect = 2u; hipLaunchKernelGGL(fallbackKernel, dim3(4096), dim3(256), 0, s, a); }
    };
    if (p->sceneHasTextures || p->sceneHasVertexColors) {
        const bool multiUv = p->sceneUvSets > 1;
        // (SLIM: 0 / 1 / 2 as decided above)
#define BRMI_GB_LAUNCH(T, P, M) do { if (slim == 2) launch(k_gbuffer<false, T, P, M, 2>, k_gbuffer<true, T, P, M>); else if (slim == 1) launch(k_gbuffer<false, T, P, M, 1>, k_gbuffer<true, T, P, M>); \
                                     else launch(k_gbuffer<false, T, P, M>, k_gbuffer<true, T, P, M>); } while (0)
        if (p->sceneHasParallax) { if (multiUv) BRMI_GB_LAUNCH(true, true, true); else BRMI_GB_LAUNCH(true, true, false); }      // its own variants: the ray march costs the others registers they would spill
        else if (multiUv) BRMI_GB_LAUNCH(true, false, true);
        else BRMI_GB_LAUNCH(true, false, false);
    } else if (p->inlineResolve && slim == 2) launch(k_gbuffer<false, false, false, false, 2>, k_gbuffer<true, false, false, false, 2>);      // (the in-place form of scenes without textures has the slim instantiations too)
    else if (p->inlineResolve && slim == 1) launch(k_gbuffer<false, false, false, false, 1>, k_gbuffer<true, false, false, false, 1>);
    else BRMI_GB_LAUNCH(false, false, false);
#undef BRMI_GB_LAUNCH
    BRMI_LAUNCH_CHECK(p, "k_gbuffer");
    return BRMI_OK;
}

}  // namespace brmi
