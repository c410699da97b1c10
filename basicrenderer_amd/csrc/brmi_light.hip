// brmi_light.hip -- light clustering (K9 + K10) and clustered OpenPBR deferred shading (K11) for gfx950.
//
// Reference: BR/shaders/clustering.hlsl:31-107, BR/shaders/lightCulling.hlsl:40-126,
// BR/shaders/deferred.hlsl:11-106, BR/shaders/Include/lighting.hlsli:81-196,391-661,
// BR/shaders/Include/IBL.hlsli:94-672, BR/shaders/Include/PBR.hlsli:8-190,
// BR/shaders/Include/utilities.hlsli:2590-2709.
// MI355X-first differences:
//   * K9 (1 thread per group, 3456 groups) and K10 (global atomic page allocator) become: light
//     spheres to view space once, one lane per cluster for AABB + page demand, a prefix scan that
//     hands every cluster a contiguous, deterministic page range (= the allocation order of a
//     serial run of the reference), and the fill pass.  Page contents and list order are exactly the reference's; page numbers no longer
//     depend on atomic ordering.
//   * slice plane depths come from the host (see brmi_update): log()/exp() results differ in the
//     last bit between math libraries and would make light lists irreproducible.
//   * K11 runs one lane per pixel in tile order (a wave = one 8x8 tile): all G-buffer reads and the
//     HDR write are contiguous per wave.
//   * OpenPBR lookup tables are injected R16_UNORM / float tables, bilinearly filtered in fp32.
#include <algorithm>
#include <cstdlib>

#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_shade_math.h"
#include "brmi_lightgrid.h"
#include "brmi_shade.h"

namespace brmi {


// =================================== K9 + K10 ==================================================
// (bodies: brmi_lightgrid.h)
__global__ void __launch_bounds__(256) k_lc_count(ClusterArgs a) { lc_count_wave(a, blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63u); }
__global__ void __launch_bounds__(256) k_lc_fill(ClusterArgs a) { lc_fill_block(a, blockIdx.x, threadIdx.x); }

// =================================== K11 (device code: brmi_shade.h) ===========================
// Waves per SIMD the plain variant is compiled for.  Alone on the chip the kernel wants four (128 VGPRs, 3 of them spilled: 218 -> 203 us); beside
// another frame's geometry half it wants three (132 VGPRs): with 4 x 128 registers taken a retiring workgroup frees 128 per SIMD, k_raster's
// 140-register waves find no room, and the frame in flight gets 10 % slower (profiles/r03_experiments.md).  brmi_execute_split picks.
#ifndef BRMI_SHADE_WAVES
#define BRMI_SHADE_WAVES 3
#endif
#ifndef BRMI_SHADE_STASH_SHARED
#define BRMI_SHADE_STASH_SHARED 0      // floats the in-flight variant parks (experiments: 6)
#endif
// Round 4: the stand-alone variant (brmi_execute, the stage entry point) runs FIVE waves per SIMD: the lane index goes through an opaque copy per tile
// (what derives from it is then recomputed where it is used instead of living in hoisted registers: 125 -> 110 VGPRs), and the next tile's G-buffer words
// are requested behind this tile's shading instead of in front of it (-14 registers: 95 VGPRs, no scratch) -- the fifth wave hides more latency than
// the prefetch did (serial shading 0.191 -> 0.174 ms Bistro-class, 0.182 -> 0.164 Sponza-class).  The variant that shares the chip with another frame's
// geometry half keeps the round-3 form on purpose: at 117 VGPRs it fits four waves per SIMD, runs faster itself (0.41 -> 0.33 ms in flight) and
// starves the geometry stream -- period 0.522 -> 0.585 ms Bistro-class (profiles/r04_experiments.md).
#ifndef BRMI_SHADE_OPAQUE_LANE_SHARED
#define BRMI_SHADE_OPAQUE_LANE_SHARED 0
#endif
#ifndef BRMI_SHADE_PREFETCH_ALONE
#define BRMI_SHADE_PREFETCH_ALONE 0
#endif
#ifndef BRMI_SHADE_STASH_ALONE
#define BRMI_SHADE_STASH_ALONE 9       // floats the stand-alone variant parks in LDS per pixel: 9 = the metal lobe's inputs, 17 = + emissive + the diffuse fit's coefficients
#endif
#ifndef BRMI_SHADE_WAVES_ALONE
#define BRMI_SHADE_WAVES_ALONE 5
#endif
// BRMI_SHADE_SHARED_MAXWAVES (experiments): the variant that shares the chip is capped at this many waves per SIMD -- amdgpu_waves_per_eu's upper bound makes
// the kernel descriptor claim enough registers that no more waves fit -- 0 = no cap; BRMI_SHADE_SHARED_LEAN: that variant in the stand-alone one's form
// (opaque lane index, no prefetch: 95 VGPRs)
#ifndef BRMI_SHADE_SHARED_MAXWAVES
#define BRMI_SHADE_SHARED_MAXWAVES 0
#endif
#ifndef BRMI_SHADE_SHARED_LEAN
#define BRMI_SHADE_SHARED_LEAN 0
#endif
template <int MODE, int WAVES = BRMI_SHADE_WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE != 0 ? 1 : WAVES, (MODE == 0 && WAVES == BRMI_SHADE_WAVES && BRMI_SHADE_SHARED_MAXWAVES != 0) ? BRMI_SHADE_SHARED_MAXWAVES : 8)))
k_shade(ShadeArgs a) {
    wave_prio<PRIO_SHADE>();
#ifndef BRMI_SHADE_SHARED_RESERVE
#define BRMI_SHADE_SHARED_RESERVE 1
#endif
    // The variant that shares the chip holds 136 registers per wave ON PURPOSE (round 5: the arithmetic needs 123): three of its waves then leave a SIMD
    // 104 registers for the other frame's geometry waves, four waves of 128 leave none -- the shading half gets faster and the frame slower (DESIGN.md 4.6:
    // period 0.522 -> 0.585 ms when tried with a leaner kernel in round 4; 0.510 -> 0.519 in round 5).
    if (MODE == 0 && WAVES == BRMI_SHADE_WAVES && BRMI_SHADE_WAVES_ALONE != BRMI_SHADE_WAVES && BRMI_SHADE_SHARED_RESERVE) asm volatile("" ::: "v131");
    const ShadeFrame k = make_shade_frame(a);
    __shared__ float sliceStart[64];
    __shared__ float unormT[256];
    __shared__ float4 camK[9];                            // rows of projectionInverse, rows of viewInverse, camera position
    shade_stage_lds(a, k, sliceStart, unormT, camK);
    constexpr bool ALONE = WAVES == BRMI_SHADE_WAVES_ALONE && BRMI_SHADE_WAVES_ALONE != BRMI_SHADE_WAVES;
    constexpr bool OPAQUE_LANE = ALONE || BRMI_SHADE_OPAQUE_LANE_SHARED || BRMI_SHADE_SHARED_LEAN, PREFETCH = (!ALONE && !BRMI_SHADE_SHARED_LEAN) || (ALONE && BRMI_SHADE_PREFETCH_ALONE);
    if (MODE == 0) {
        // One 8x8 tile per wave and iteration: the tile index is wave-uniform, so the base address of every plane is scalar arithmetic and a
        // lane only adds its own constant offset (no per-lane 64-bit address math, no integer division per pixel).  Software pipeline: the
        // G-buffer words of the next tile are requested before the current one is shaded.
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wavesInGrid = gridDim.x * (blockDim.x >> 6);
        const uint32_t tileCount = (uint32_t)((a.pixelCount + 63ull) >> 6), firstTile = (uint32_t)(a.firstPixel >> 6);
#ifndef BRMI_SHADE_TILE_RUNS
#define BRMI_SHADE_TILE_RUNS 1
#endif
        // Round 5: a wave takes a RUN of neighbouring tiles (tiles per wave = ceil(tiles / waves) of them, left to right) instead of every
        // `wavesInGrid`-th tile of the band.  A light cluster of the 12 x 12 grid is 40 tiles wide at 4K, so the tiles of a run sit in one cluster nearly
        // always: the list head and the 64 B light records the first tile fetched through the scalar cache (16 KB) are still there for the others.  Strided,
        // every CU met every cluster of the frame -- 0.9 MB of records in list order -- and each light evaluation began with a scalar load that went to L2
        // (an ablation that let every tile read the head of one table: k_shade 0.170 -> 0.138 ms).
        const uint32_t waveInGrid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
        // (the stand-alone variant only: 0.1665 -> 0.1635 ms; beside another frame's geometry half the strided order is 2 % better -- tiles of one
        // workgroup then end at different times and slots come free more evenly: Bistro-class 0.504 against 0.514 ms per frame, Sponza-class 0.388 / 0.397)
        constexpr bool RUNS = BRMI_SHADE_TILE_RUNS && ALONE;
        const uint32_t run = RUNS ? (tileCount + wavesInGrid - 1u) / wavesInGrid : 1u;
        uint32_t t = RUNS ? waveInGrid * run : waveInGrid;
        const uint32_t tEnd = RUNS ? min(t + run, tileCount) : tileCount;
        uint32_t tx = (firstTile + t) % a.tilesX, ty = (firstTile + t) / a.tilesX;
        const uint32_t stepX = RUNS ? 1u : wavesInGrid % a.tilesX, stepY = RUNS ? 0u : wavesInGrid / a.tilesX;
        const uint32_t stepT = RUNS ? 1u : wavesInGrid;
        // (the lane index behind an opaque copy per tile: its row / column inside the tile and the byte offsets of the plane loads are then a VALU
        // instruction each where they are used, not loop invariants in registers of their own -- the trick that took the G-buffer kernel from 73 to 59 VGPRs)
        auto fetch = [&](uint32_t tt, uint32_t ttx, uint32_t tty, bool& ok) {
            uint32_t ln = lane;
            if (OPAQUE_LANE) asm volatile("" : "+v"(ln));
            const uint32_t px = ttx * 8u + (ln >> 3), py = tty * 8u + (ln & 7u);
            ok = tt < tEnd && px < a.W && py < a.H && py >= a.bandY0 && py < a.bandY1;
            return ok ? load_raw_pixel_plain(a, ((uint64_t)(firstTile + tt) << 6), ln, px, py) : empty_raw_pixel();
        };
        bool ok = false;
        RawPixel cur = fetch(t, tx, ty, ok);
        while (t < tEnd) {
            uint32_t nt = t + stepT, ntx = tx + stepX, nty = ty + stepY;
            if (ntx >= a.tilesX) { ntx -= a.tilesX; nty++; }
            bool nok = false;
            RawPixel nxt = empty_raw_pixel();
            if (PREFETCH) nxt = fetch(nt, ntx, nty, nok);
            const uint64_t tileBase = (uint64_t)(firstTile + t) << 6;
            uint32_t ls = lane;
            if (OPAQUE_LANE) asm volatile("" : "+v"(ls));
            const uint32_t cls = shade_pixel<0, (BRMI_SHADE_METAL_STASH && WAVES == BRMI_SHADE_WAVES_ALONE && BRMI_SHADE_WAVES_ALONE != BRMI_SHADE_WAVES) ? BRMI_SHADE_STASH_ALONE : BRMI_SHADE_STASH_SHARED>(a, k, sliceStart, unormT, camK, cur, ok, tileBase, ls);
            shade_defer(a, t, cls, ls);
            if (!PREFETCH) nxt = fetch(nt, ntx, nty, nok);      // (experiments: the next tile's words requested behind this tile's shading, ~14 registers less in it)
            cur = nxt; ok = nok; t = nt; tx = ntx; ty = nty;
        }
    } else {
        if (blockIdx.x == 0 && threadIdx.x < CNT_STRIPE_COUNT) a.counters[CNT_STRIPES + threadIdx.x * CNT_STRIPE_WORDS + a.nextDeferredWord + (uint32_t)(MODE - 1)] = 0u;   // the next shading call starts with empty lists
        // the 64 lists as one index space: their lengths are read side by side and scanned, a work item finds its stripe by search
        __shared__ uint32_t stripeStart[CNT_STRIPE_COUNT + 1];
        if (threadIdx.x < 64u) {
            const uint32_t n = min(a.counters[CNT_STRIPES + threadIdx.x * CNT_STRIPE_WORDS + a.deferredWord + (uint32_t)(MODE - 1)], a.stripeCapacity);
            uint32_t incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (threadIdx.x >= (uint32_t)o) incl += v; }
            stripeStart[threadIdx.x] = incl - n;
            if (threadIdx.x == 63u) stripeStart[64] = incl;
        }
        __syncthreads();
        const uint32_t total = stripeStart[64];
        // wave-uniform loop (q0 is the same for the 64 lanes of a wave): the lanes past the end of the list stay in step with the others
        for (uint32_t q0 = blockIdx.x * blockDim.x; q0 < total; q0 += gridDim.x * blockDim.x) {
            const uint32_t q = q0 + threadIdx.x;
            const bool ok = q < total;
            uint64_t i = a.firstPixel; uint32_t px = 0, py = 0;
            RawPixel raw = empty_raw_pixel();
            if (ok) {
                uint32_t stripe = 0;
#pragma unroll
                for (uint32_t step = 32; step > 0; step >>= 1) if (stripeStart[stripe + step] <= q) stripe += step;
                i = a.firstPixel + a.deferred[((size_t)(MODE - 1) * CNT_STRIPE_COUNT + stripe) * a.stripeCapacity + (q - stripeStart[stripe])];
                const uint32_t tile = (uint32_t)(i >> 6), within = (uint32_t)(i & 63u);
                px = (tile % a.tilesX) * 8u + (within >> 3); py = (tile / a.tilesX) * 8u + (within & 7u);
                raw = load_raw_pixel(a, i, px, py);
            }
            shade_pixel<MODE>(a, k, sliceStart, unormT, camK, raw, ok, i & ~63ull, (uint32_t)(i & 63ull));
        }
    }
}

__global__ void __launch_bounds__(256) k_expand_luts(const uint16_t* odE, const uint16_t* odAvg, const uint16_t* imE, const uint16_t* imAvg, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 32768u) out[i] = (float)odE[i] / 65535.0f;
    else if (i < 32768u + 1024u) out[i] = (float)odAvg[i - 32768u] / 65535.0f;
    else if (i < 32768u + 2048u) out[i] = (float)imE[i - 32768u - 1024u] / 65535.0f;
    else if (i < 32768u + 2048u + 32u) out[i] = (float)imAvg[i - 32768u - 2048u] / 65535.0f;
    else if (i < 32768u + 2048u + 32u + 256u) out[i] = (float)(i - (32768u + 2048u + 32u)) / 255.0f;
}

int launch_expand_luts(brmi_pass* p, hipStream_t s) {
    const brmi_scene_buffers& sc = p->scene;
    hipLaunchKernelGGL(k_expand_luts, dim3((35104 + 255) / 256), dim3(256), 0, s, sc.lutOpaqueDielectricEnergyComplement, sc.lutOpaqueDielectricAvgEnergyComplement,
                       sc.lutIdealMetalEnergyComplement, sc.lutIdealMetalAvgEnergyComplement, p->wsPtr<float>(p->ws.lutF));
    BRMI_LAUNCH_CHECK(p, "k_expand_luts");
    return BRMI_OK;
}

ClusterArgs cluster_args_of(brmi_pass* p) {
    ClusterArgs a;
    a.sc = p->scene;
    for (size_t k = 0; k < p->planesHost.size() && k < 2 * 62; k++) a.planes[k] = p->planesHost[k];
    a.clusters = static_cast<brmi_light_cluster*>(p->res[BRMI_RES_LIGHT_CLUSTERS]); a.pages = static_cast<brmi_light_page*>(p->res[BRMI_RES_LIGHT_PAGES]);
    a.poolSize = p->lightPagePool; a.counters = p->counters();
    a.lightVS = p->wsPtr<float4>(p->ws.lightVS); a.lightMeta = p->wsPtr<uint32_t>(p->ws.lightMeta); a.clusterPages = p->wsPtr<uint32_t>(p->ws.clusterPages);
    a.clusterHits = p->wsPtr<uint32_t>(p->ws.clusterHits); a.pageTotal = p->wsPtr<uint32_t>(p->ws.pageTotal);
    a.hitMasks = p->wsPtr<uint64_t>(p->ws.lightHitMasks); a.maskWords = (std::max(1u, p->scene.lightCount) + 63u) / 64u;
    a.clusterList = p->wsPtr<uint2>(p->ws.clusterList); a.listEntries = p->wsPtr<uint32_t>(p->ws.listEntries);
    a.shadeLights = p->wsPtr<float4>(p->ws.shadeLights); a.listRecords = p->wsPtr<float4>(p->ws.listRecords);
    return a;
}

int launch_light_clustering(brmi_pass* p, hipStream_t s) {
    if (int rc = ensure_frame_constants(p, s)) return rc;
    ClusterArgs a = cluster_args_of(p);
    a.sc = shading_scene_of(p);      // may run on the shading stream of a split frame: the frame's own camera (FrameSnapshot), not the caller's buffer
    const uint32_t nc = p->numLightClusters;
    hipLaunchKernelGGL(k_lc_count, dim3((nc + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_lc_fill, dim3((nc + 3) / 4), dim3(256), 0, s, a);
    BRMI_LAUNCH_CHECK(p, "light clustering");
    return BRMI_OK;
}

ShadeArgs shade_args_of(brmi_pass* p) {
    ShadeArgs a;
    { const brmi_scene_buffers ssc = shading_scene_of(p); a.perFrame = ssc.perFrame; a.cameras = ssc.cameras; }
    a.openpbrMaterialCount = p->scene.openpbrMaterialCount; a.lutFuzzLTC = p->scene.lutFuzzLTC;
    a.shadeLights = p->wsPtr<float4>(p->ws.shadeLights); a.clusterList = p->wsPtr<uint2>(p->ws.clusterList); a.listEntries = p->wsPtr<uint32_t>(p->ws.listEntries);
    a.listRecords = p->wsPtr<float4>(p->ws.listRecords);
    a.depth = static_cast<const float*>(p->res[BRMI_RES_LINEAR_DEPTH]); a.normals = static_cast<const float4*>(p->res[BRMI_RES_GBUF_NORMALS]);
    a.albedo = static_cast<const uint32_t*>(p->res[BRMI_RES_GBUF_ALBEDO]); a.coat = static_cast<const unsigned long long*>(p->res[BRMI_RES_GBUF_COAT]);
    a.emissive = static_cast<const unsigned long long*>(p->res[BRMI_RES_GBUF_EMISSIVE]); a.fuzz = static_cast<const unsigned long long*>(p->res[BRMI_RES_GBUF_FUZZ]);
    a.metallicRoughness = static_cast<const uint32_t*>(p->res[BRMI_RES_GBUF_METALLIC_ROUGHNESS]);
    a.hdr = static_cast<unsigned long long*>(p->res[BRMI_RES_HDR_COLOR]);
    a.W = p->cfg.width; a.H = p->cfg.height; a.tilesX = p->tilesX; a.bandY0 = p->bandY0; a.bandY1 = p->bandY1; a.firstPixel = p->bandFirstPixel; a.pixelCount = p->bandPixelCount;
    a.enablePunctual = p->cfg.enablePunctualLights; a.clustered = p->cfg.enableClusteredLighting;
    a.lutF = p->wsPtr<float>(p->ws.lutF);
    a.matConst = p->wsPtr<MatConst>(p->ws.matConst);
    a.shadeRows = p->wsPtr<ShadeRows>(p->ws.shadeRows); a.shadeAvgs = p->wsPtr<ShadeAverages>(p->ws.shadeAvgs); a.ggxQuads = p->wsPtr<GgxQuad>(p->ws.ggxQuads);
    a.sceneHasCoat = p->sceneHasCoat ? 1u : 0u;
    a.tables = shade_tables_of(p);
    a.counters = p->counters(); a.deferred = p->wsPtr<uint32_t>(p->ws.deferredPixels);
    a.deferredWord = (p->shadeSerial & 1u) ? STRIPE_DEFERRED_B : STRIPE_DEFERRED_A;
    a.nextDeferredWord = (p->shadeSerial & 1u) ? STRIPE_DEFERRED_A : STRIPE_DEFERRED_B;
    a.stripeCapacity = p->deferredStripeCapacity;
    return a;
}

static int launch_shade_range(brmi_pass* p, hipStream_t s, uint32_t row0, uint32_t row1, uint32_t share);
int launch_shade(brmi_pass* p, hipStream_t s) {
    if (int rc = ensure_frame_constants(p, s)) return rc;
    if (p->shadeSlabs > 1u && p->shadeSlabFn) {
        // brmi_set_shade_slabs: the band in slabs of whole 8-row tile rows, top to bottom; after each slab's launches the host hook (the composition of those rows)
        const uint32_t y0 = p->bandY0, y1 = p->bandY1, tileRows = (y1 - y0 + 7u) / 8u, n = std::min(p->shadeSlabs, std::max(1u, tileRows));
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t r0 = y0 + (uint32_t)((uint64_t)tileRows * k / n) * 8u, r1 = std::min(y1, y0 + (uint32_t)((uint64_t)tileRows * (k + 1u) / n) * 8u);
            if (r1 <= r0) continue;
            if (int rc = launch_shade_range(p, s, r0, r1, n)) return rc;
            p->shadeSlabFn(p->shadeSlabUser, r0, r1, static_cast<brmi_stream>(s));
        }
        return BRMI_OK;
    }
    return launch_shade_range(p, s, p->bandY0, p->bandY1, 1u);
}
// the deferred shading of the band's rows [row0, row1) (row0 a multiple of 8 above the band's first row); share: the launch is one of that many
static int launch_shade_range(brmi_pass* p, hipStream_t s, uint32_t row0, uint32_t row1, uint32_t share) {
    ShadeArgs a = shade_args_of(p);
    if (share > 1u) {
        const uint64_t first = a.firstPixel + (uint64_t)((row0 - p->bandY0) / 8u) * p->tilesX * 64ull;
        a.pixelCount = (uint64_t)((row1 - row0 + 7u) / 8u) * p->tilesX * 64ull; a.firstPixel = first;
        a.bandY0 = row0; a.bandY1 = row1;
    }
    p->shadeSerial++;
    // 8192 workgroups of four waves, four tiles per wave at 4K: against 4096 (eight tiles per wave) the kernel's tail is shorter (233 -> 226 us)
    // and, with another frame's geometry half in flight beside it, slots come free twice as often for that half's high-priority launches
    // (Bistro 4K, two frames in flight: 0.436 -> 0.413 ms per frame; 16384: 0.425, 2048: 0.49)
    {
        static const uint32_t pad = (uint32_t)experiment("shade_lds_pad", 0);   // (builds with -DBRMI_EXPERIMENTS: unused dynamic LDS caps the kernel's occupancy)
        if (p->shadeSharesChip || BRMI_SHADE_WAVES_ALONE == BRMI_SHADE_WAVES) hipLaunchKernelGGL((k_shade<0, BRMI_SHADE_WAVES>), dim3(std::max(256u, p->shadeGridShared / share)), dim3(256), pad, s, a);
        else hipLaunchKernelGGL((k_shade<0, BRMI_SHADE_WAVES_ALONE>), dim3(std::max(256u, 8192u / share)), dim3(256), pad, s, a);
    }
    // deferred pixels by class: coat, fuzz, both -- only the variants some material of the scene can need
    if (p->sceneHasCoat) hipLaunchKernelGGL(k_shade<1>, dim3(512), dim3(256), 0, s, a);
    if (p->sceneHasFuzz) hipLaunchKernelGGL(k_shade<2>, dim3(512), dim3(256), 0, s, a);
    if (p->sceneHasCoat && p->sceneHasFuzz) hipLaunchKernelGGL(k_shade<3>, dim3(512), dim3(256), 0, s, a);
    BRMI_LAUNCH_CHECK(p, "k_shade");
    return BRMI_OK;
}

}  // namespace brmi
