// brmi_lightgrid.h -- light clustering (K9 + K10) as device functions: the kernels of brmi_light.hip call them, and inside brmi_execute they
// ride on launches of the culling pass (brmi_cull.hip) instead of being launched on their own.
#ifndef BRMI_LIGHTGRID_H
#define BRMI_LIGHTGRID_H
#include "brmi_device.h"
#include "brmi_internal.h"

namespace brmi {

// =================================== K9 + K10 ==================================================
struct ClusterArgs {
    brmi_scene_buffers sc;
    float planes[2 * 62];         // near/far per slice, 2 * gridZ (brmi_update evaluates them on the host; they travel as kernel arguments, not through a copy)
    brmi_light_cluster* clusters;
    brmi_light_page* pages;
    uint32_t poolSize;
    uint32_t* counters;
    float4* lightVS;              // per active light: view-space bounding sphere (xyz, r)
    uint32_t* lightMeta;          // per active light: type | lightIndex << 2
    uint32_t* clusterPages;       // per cluster: page demand
    uint32_t* clusterHits;        // per cluster: lights that touch it
    uint32_t* pageTotal;          // [0]: pages demanded by all clusters (unclamped)
    uint64_t* hitMasks; uint32_t maskWords;   // per cluster: one bit per light of the list
    // the same lists once more for the shading pass, flat: clusterList[c] = {first entry, length}, entries = positions in the active-light
    // list in the order the page walk of the reference visits them (newest page first)
    uint2* clusterList; uint32_t* listEntries;
    const float4* shadeLights; float4* listRecords;      // the shading pass's 64 B light records, and a copy of them in list order (one hop less per tile)
};

BRMI_DEV bool light_hits_cluster(float4 sphere, uint32_t type, f3 mn, f3 mx) {
    if (type == BRMI_LIGHT_DIRECTIONAL) return true;
    if (type != BRMI_LIGHT_POINT && type != BRMI_LIGHT_SPOT) return false;
    const f3 center{sphere.x, sphere.y, sphere.z};
    const f3 closest = max3v(mn, min3v(center, mx));
    const f3 d = closest - center;
    return dot3(d, d) <= sphere.w * sphere.w;
}

// One wave64 per cluster, one lane per light: AABB (clustering.hlsl:31-107), the lights that touch it as bit masks, and the
// page demand of the reference's serial allocator (lightCulling.hlsl:70-118) in closed form.  The serial loop opens a new page
// whenever it reaches a light (hit or not) with 12 entries in the current page, so with T hits in total it allocates
// 1 + T / 12 pages, minus one when the page filled exactly at the last light of the list.
BRMI_DEV void lc_count_wave(const ClusterArgs& a, uint32_t idx, uint32_t lane) {
    const brmi_scene_buffers& sc = a.sc;
    const brmi_per_frame* pf = sc.perFrame;
    const brmi_camera* cam = sc.cameras + pf->mainCameraIndex;
    const uint32_t gx = pf->lightClusterGridSizeX, gy = pf->lightClusterGridSizeY, gz = pf->lightClusterGridSizeZ;
    const uint32_t total = gx * gy * gz, lightCount = pf->numLights;
    if (idx >= total) return;
    const float W = (float)pf->screenResX, H = (float)pf->screenResY;
    const m4 invProj = load_m4(&cam->projectionInverse[0][0]);
    const float tsx = W / (float)gx, tsy = H / (float)gy;
    const uint32_t x = idx % gx, y = (idx / gx) % gy, z = idx / (gx * gy);
    f3 tileV[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float sxp = ((float)x + (k ? 1.0f : 0.0f)) * tsx, syp = ((float)y + (k ? 1.0f : 0.0f)) * tsy;
        const f4 ndc{2.0f * sxp / W - 1.0f, 2.0f * (H - syp - 1.0f) / H - 1.0f, 1.0f, 1.0f};
        const f4 v = mul_vm(ndc, invProj);
        tileV[k] = f3{v.x / v.w, v.y / v.w, v.z / v.w};
    }
    const float pn = a.planes[2 * z], pfar = a.planes[2 * z + 1];
    f3 pts[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const f3 e = tileV[k & 1];
        const float t = ((k & 2) ? pfar : pn) / e.z;
        pts[k] = f3{t * e.x, t * e.y, t * e.z};
    }
    const f3 mn = min3v(min3v(pts[0], pts[1]), min3v(pts[2], pts[3])), mx = max3v(max3v(pts[0], pts[1]), max3v(pts[2], pts[3]));
    brmi_light_cluster* c = a.clusters + idx;
    if (lane == 0) {
        *reinterpret_cast<float4*>(c->minPoint) = make_float4(mn.x, mn.y, mn.z, 0.0f);
        *reinterpret_cast<float4*>(c->maxPoint) = make_float4(mx.x, mx.y, mx.z, 0.0f);
    }
    uint32_t hits = 0;                  // wave-uniform
    int lastHit = -1;                   // position in the light list of the last hit
    uint64_t* masks = a.hitMasks + (size_t)idx * a.maskWords;
    for (uint32_t base = 0, w = 0; base < lightCount; base += 64, w++) {
        const uint32_t li = base + lane;
        bool hit = false;
        if (li < lightCount) hit = light_hits_cluster(a.lightVS[li], a.lightMeta[li] & 3u, mn, mx);
        const uint64_t m = __ballot(hit);
        if (lane == 0) masks[w] = m;
        if (m != 0ull) { hits += (uint32_t)__popcll(m); lastHit = (int)base + 63 - __clzll((long long)m); }
    }
    if (lane == 0) {
        uint32_t pagesNeeded = 1u + hits / BRMI_LIGHTS_PER_PAGE;
        if (hits != 0u && hits % BRMI_LIGHTS_PER_PAGE == 0u && lastHit == (int)lightCount - 1) pagesNeeded--;
        a.clusterPages[idx] = pagesNeeded;
        a.clusterHits[idx] = hits;
    }
}

// fill, one wave64 per cluster: hit j of the cluster (in light-list order, from the bit masks) goes to entry j % 12 of the
// cluster's page j / 12; pages come from the scan, are chained newest -> oldest like the serial allocator chains them, and
// stop where the pool ends (the reference's `break`: the cluster then keeps its full pages only).
// The first page of a cluster is the exclusive prefix of the page demand in cluster order = the serial allocation order.  Every workgroup
// sums the demand of the clusters before its own four (a few thousand L2-resident words over 256 threads) instead of a scan kernel of
// its own between count and fill: one ~6 us launch less per frame.
// one workgroup of 256 threads = four clusters (vblock = its index among the fill's workgroups)
BRMI_DEV void lc_fill_block(const ClusterArgs& a, uint32_t vblock, uint32_t tid) {
    __shared__ uint32_t waveSum[4], waveDemand[4];
    const brmi_per_frame* pf = a.sc.perFrame;
    const uint32_t total = pf->lightClusterGridSizeX * pf->lightClusterGridSizeY * pf->lightClusterGridSizeZ, lightCount = pf->numLights;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t idx0 = vblock * 4u, idx = idx0 + wave;
    uint32_t part = 0;
    for (uint32_t i = tid; i < min(idx0, total); i += 256u) part += a.clusterPages[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o);
    const uint32_t demand = idx < total ? a.clusterPages[idx] : 0u;
    if (lane == 0) { waveSum[wave] = part; waveDemand[wave] = demand; }
    __syncthreads();
    uint32_t base = waveSum[0] + waveSum[1] + waveSum[2] + waveSum[3];         // first page of this wave's cluster
    for (uint32_t w = 0; w < wave; w++) base += waveDemand[w];
    if (idx + 1u == total && lane == 0) { a.counters[CNT_LIGHT_PAGES] = min(base + demand, a.poolSize); a.pageTotal[0] = base + demand; }
    if (idx >= total) return;
    const uint32_t hits = a.clusterHits[idx];
    const uint32_t valid = base >= a.poolSize ? 0u : min(demand, a.poolSize - base);   // pages that exist
    const uint64_t* masks = a.hitMasks + (size_t)idx * a.maskWords;
    // the walk starts at the newest page (base + valid - 1), the only one that may be partly filled
    const uint32_t newestCount = (valid < demand || valid == 0u) ? BRMI_LIGHTS_PER_PAGE : hits - BRMI_LIGHTS_PER_PAGE * (demand - 1u);
    const uint32_t listBase = base * BRMI_LIGHTS_PER_PAGE;
    uint32_t before = 0;
    for (uint32_t lb = 0, w = 0; lb < lightCount; lb += 64, w++) {
        const uint64_t m = masks[w];
        if ((m >> lane) & 1ull) {
            const uint32_t j = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const uint32_t pg = j / BRMI_LIGHTS_PER_PAGE, e = j % BRMI_LIGHTS_PER_PAGE;
            if (pg < valid) {
                a.pages[base + pg].lightIndices[e] = a.lightMeta[lb + lane] >> 2;
                const uint32_t pos = listBase + (pg + 1u == valid ? e : newestCount + BRMI_LIGHTS_PER_PAGE * (valid - 2u - pg) + e);
                a.listEntries[pos] = lb + lane;
                // the light's shading record beside its list entry: k_shade stages a cluster's lights with ONE dependent load (list position
                // -> record) instead of two (-> light index -> record); the chain is on every tile's critical path
                const float4* src = a.shadeLights + (size_t)(lb + lane) * 4u; float4* dst = a.listRecords + (size_t)pos * 4u;
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
            }
        }
        before += (uint32_t)__popcll(m);
    }
    for (uint32_t pg = lane; pg < valid; pg += 64) {
        brmi_light_page* p = a.pages + base + pg;
        p->ptrNextPage = pg == 0u ? BRMI_LIGHT_PAGE_NULL : base + pg - 1u;
        // every page but the newest is full; with the pool exhausted all surviving pages are full
        p->numLightsInPage = (valid < demand || pg + 1u < demand) ? BRMI_LIGHTS_PER_PAGE : hits - BRMI_LIGHTS_PER_PAGE * (demand - 1u);
    }
    if (lane == 0) {
        brmi_light_cluster* c = a.clusters + idx;
        c->numLights = valid < demand ? BRMI_LIGHTS_PER_PAGE * valid : hits;
        c->ptrFirstPage = valid == 0u ? BRMI_LIGHT_PAGE_NULL : base + valid - 1u;
        c->pad[0] = 0; c->pad[1] = 0;
        // The reference's walk (lighting.hlsli:625-655) stops at a page with no lights in it: a cluster whose newest page was opened by a
        // light that then missed (12, 24, ... hits and more lights behind the last one) shades NO light at all.  Reproduced as it is.
        a.clusterList[idx] = make_uint2(listBase, (valid == 0u || newestCount == 0u) ? 0u : c->numLights);
    }
}


ClusterArgs cluster_args_of(brmi_pass* p);      // host (brmi_light.hip)

}  // namespace brmi
#endif
