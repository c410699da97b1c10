// brmi_hzb.hip -- linear-depth mip chain for occlusion culling (LinearDepthDownsamplePass).
//
// The reference builds the chain with FidelityFX SPD (BR/shaders/downsample.hlsl:108-112: SpdReduce4 = max of
// four texels) over the LinearDepthMap padded to the next power of two (SceneRenderBridge.cpp:223-236).  A max
// pyramid does not depend on the order of reduction, so only the result is mirrored:
//   mip m texel (x, y) = max over the 2^m x 2^m block of the padded depth map; padding reads as "empty" (0x7F7FFFFF).
// MI355X: pure streaming work.  One workgroup reduces a 32 x 32 block of the depth map (whole 8x8 tiles, 8-byte
// contiguous loads) through LDS into mips 1..5 in one launch; a single workgroup finishes the small tail.
#include "brmi_device.h"
#include "brmi_internal.h"

namespace brmi {

BRMI_DEV float hzb_depth_texel(const HzbDesc& h, uint32_t x, uint32_t y) {
    return (x < h.width && y >= h.rowLo && y < h.rowHi) ? h.depth[tiled_index(x, y, h.tilesX)] : __uint_as_float(BRMI_DEPTH_EMPTY_BITS);
}
BRMI_DEV float key_depth(unsigned long long k) { return (k == BRMI_VIS_EMPTY) ? as_f32(BRMI_DEPTH_EMPTY_BITS) : as_f32(((uint32_t)(k >> BRMI_VIS_META_BITS)) << 1); }

// mips firstMip..last, one workgroup, level by level (source texels clamped to the source extent: a dimension that
// reached 1 stays 1)
BRMI_DEV void hzb_tail_levels(const HzbDesc& h, uint32_t firstMip, uint32_t threads) {
    for (uint32_t mip = firstMip; mip < h.mipCount; mip++) {
        const uint32_t sw = max(1u, h.paddedW >> (mip - 1u)), sh = max(1u, h.paddedH >> (mip - 1u));
        const uint32_t w = max(1u, h.paddedW >> mip), hh = max(1u, h.paddedH >> mip);
        const float* src = h.mips + h.mipOffset[mip - 1u];
        float* dst = h.mips + h.mipOffset[mip];
        for (uint32_t i = threadIdx.x; i < w * hh; i += threads) {
            const uint32_t x = i % w, y = i / w;
            const uint32_t x0 = min(2u * x, sw - 1u), x1 = min(2u * x + 1u, sw - 1u), y0 = min(2u * y, sh - 1u), y1 = min(2u * y + 1u, sh - 1u);
            float a, b, c, d;
            if (mip == 1u) { a = hzb_depth_texel(h, x0, y0); b = hzb_depth_texel(h, x1, y0); c = hzb_depth_texel(h, x0, y1); d = hzb_depth_texel(h, x1, y1); }
            else { a = src[(size_t)y0 * sw + x0]; b = src[(size_t)y0 * sw + x1]; c = src[(size_t)y1 * sw + x0]; d = src[(size_t)y1 * sw + x1]; }
            dst[i] = max2(max2(a, b), max2(c, d));
        }
        __syncthreads();   // also orders this block's global writes before the next level's reads
    }
}

// mips 1..5 from the depth map: block = 16 x 16 texels of mip 1 (requires paddedW, paddedH >= 32)
// FROM_VIS: the source is the visibility buffer; the linear depth of the four texels (K6, gbuffer.hlsl:114-161) is written to
// the depth map on the way.  `skipUnless` (may be null): the launch does nothing when that counter is zero.
// (FidelityFX SPD's single-pass scheme -- the last workgroup to finish builds the tail -- was tried: every workgroup needs a device-scope fence
// before it takes its ticket, which on this part writes back the XCD's L2; 4,352 of them turned a 20 us kernel into 1.1 ms.  The tail stays
// a launch of its own.)
// DirtyBlocks: the second build of a frame only redoes the blocks phase 2's triangles may have touched (brmi_raster.hip); null = all
struct DirtyBlocks { const uint8_t* chainDirty; uint32_t chainBlocksX; };
template <bool FROM_VIS>
__global__ void __launch_bounds__(256) k_hzb_head(HzbDesc h, const unsigned long long* vis, float* depthOut, const uint32_t* skipUnless, uint32_t blockRow0, DirtyBlocks mk) {
    wave_prio<PRIO_HZB>();
    if (skipUnless && *skipUnless == 0u) return;
    if (mk.chainDirty && mk.chainDirty[0] == 0u) {      // (workgroup-uniform) phase 1's values of this block's texels of the depth map and of mips 1 - 5 still stand
        if (blockIdx.x >= mk.chainBlocksX || mk.chainDirty[4u + (blockIdx.y + blockRow0) * mk.chainBlocksX + blockIdx.x] == 0u) return;
    }
    __shared__ float lvl[16 * 16];
    const uint32_t tx = threadIdx.x >> 4, ty = threadIdx.x & 15u;             // ty fastest: follows the column-major tile layout
    const uint32_t bx = blockIdx.x, by = blockIdx.y + blockRow0;      // only the 32-row strips that touch this GPU's band are launched
    const uint32_t x1 = bx * 16u + tx, y1 = by * 16u + ty;                    // mip-1 texel
    float v;
    {
        const uint32_t x0 = x1 * 2u, y0 = y1 * 2u;
        if (FROM_VIS) {
            // rows y0, y0 + 1 of a column are adjacent in the tile: one 16 B key load and one 8 B depth store per column
            float d[2][2];
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                const uint32_t x = x0 + c;
                if (x < h.width && y0 >= h.rowLo && y0 + 1u < h.rowHi) {       // (bands are multiples of 8 rows: both rows inside or both outside)
                    const uint32_t ti = tiled_index(x, y0, h.tilesX);
                    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(vis + ti);
                    d[c][0] = key_depth(k2.x); d[c][1] = key_depth(k2.y);
                    *reinterpret_cast<float2*>(depthOut + ti) = make_float2(d[c][0], d[c][1]);
                } else {
                    for (uint32_t r = 0; r < 2; r++) {
                        const uint32_t y = y0 + r;
                        const bool in = x < h.width && y >= h.rowLo && y < h.rowHi;
                        const unsigned long long k1 = in ? vis[tiled_index(x, y, h.tilesX)] : BRMI_VIS_EMPTY;
                        d[c][r] = key_depth(k1);
                        if (in) depthOut[tiled_index(x, y, h.tilesX)] = d[c][r];
                    }
                }
            }
            v = max2(max2(d[0][0], d[1][0]), max2(d[0][1], d[1][1]));
        } else if (x0 + 1u < h.width && y0 >= h.rowLo && y0 + 1u < h.rowHi) {
            // both rows of a column are adjacent in the tile: one 8-byte load per column
            const float2 c0 = *reinterpret_cast<const float2*>(h.depth + tiled_index(x0, y0, h.tilesX));
            const float2 c1 = *reinterpret_cast<const float2*>(h.depth + tiled_index(x0 + 1u, y0, h.tilesX));
            v = max2(max2(c0.x, c1.x), max2(c0.y, c1.y));
        } else {
            v = max2(max2(hzb_depth_texel(h, x0, y0), hzb_depth_texel(h, x0 + 1u, y0)), max2(hzb_depth_texel(h, x0, y0 + 1u), hzb_depth_texel(h, x0 + 1u, y0 + 1u)));
        }
    }
    uint32_t w = h.paddedW >> 1;
    if (h.mipCount > 1) h.mips[h.mipOffset[1] + (size_t)y1 * w + x1] = v;
    lvl[tx * 16u + ty] = v;
    // mips 2..5 inside the block: side 8, 4, 2, 1
    uint32_t side = 16;
#pragma unroll
    for (uint32_t mip = 2; mip <= 5; mip++) {
        __syncthreads();
        side >>= 1;
        float r = 0.0f;
        const uint32_t cx = threadIdx.x / side, cy = threadIdx.x % side;
        const bool active = threadIdx.x < side * side;
        if (active) {
            const uint32_t s2 = side * 2u;   // row stride of the previous level inside lvl (stored [x][y])
            r = max2(max2(lvl[(2u * cx) * s2 + 2u * cy], lvl[(2u * cx + 1u) * s2 + 2u * cy]), max2(lvl[(2u * cx) * s2 + 2u * cy + 1u], lvl[(2u * cx + 1u) * s2 + 2u * cy + 1u]));
        }
        __syncthreads();
        if (active) {
            lvl[cx * side + cy] = r;
            if (mip < h.mipCount) h.mips[h.mipOffset[mip] + (size_t)(by * side + cy) * (h.paddedW >> mip) + (bx * side + cx)] = r;
        }
    }
}

// `seedCounters` (brmi_execute's phase-1 build only): the block also does k_seed_phase2's work for the culling pass that follows.
__global__ void __launch_bounds__(1024) k_hzb_tail(HzbDesc h, uint32_t firstMip, const uint32_t* skipUnless, uint32_t* seedCounters, uint32_t seedCapacity) {
    wave_prio<PRIO_HZB>();
    if (seedCounters) seed_phase2(seedCounters, seedCapacity, threadIdx.x);      // (whatever the build does: the culling pass that follows starts from these)
    if (skipUnless && *skipUnless == 0u) return;
    hzb_tail_levels(h, firstMip, 1024u);
}

int launch_hzb(brmi_pass* p, hipStream_t s, bool fromVisibility, bool onlyIfPhase2Drew) {
    const HzbDesc h = p->hzbDesc();
    const bool head = h.paddedW >= 32 && h.paddedH >= 32;
    if (fromVisibility && !head) { int rc = launch_depth_copy(p, s); if (rc) return rc; fromVisibility = false; }   // tiny targets: unfused
    if (p->hzbMipCount < 2) return BRMI_OK;     // 1 x 1 target: mip 0 is all there is
    // Round 6: a frame whose phase-1 rasteriser stage built the chain itself (brmi_raster.hip: the re-test reads it) has the chain of everything but the late pass; the
    // build from the visibility keys that follows that stage redoes the late pass's blocks only -- nothing at all when the late list stayed empty.  (A build from the
    // depth map, brmi_build_hzb after brmi_depth_copy, redoes everything as always.)
    const bool lateOnly = p->chainBuiltInRaster && fromVisibility && !onlyIfPhase2Drew;
    p->chainBuiltInRaster = false;
    const uint32_t* skip = onlyIfPhase2Drew ? p->counters() + CNT_VISIBLE2 : (lateOnly ? p->counters() + CNT_LATE1 : nullptr);
    uint32_t first = 1;
    if (head) {
        // texels of mips 1-5 outside the band's 32-row strips never change: brmi_setup filled the chain with "empty"
        uint32_t row0 = h.rowLo / 32u, row1 = std::min((h.rowHi + 31u) / 32u, h.paddedH / 32u);
        if (!onlyIfPhase2Drew && !lateOnly) {
            // a full build; a band that moved since the last one (brmi_set_band): the strips it left are rebuilt too -- rows outside the band read "empty", so their texels
            // stop occluding -- and from here on the chain holds this band's strips
            const uint32_t b0 = row0, b1 = row1;
            if (p->chainStripLo < p->chainStripHi) { row0 = std::min(row0, p->chainStripLo); row1 = std::max(row1, std::min(p->chainStripHi, h.paddedH / 32u)); }
            p->chainStripLo = b0; p->chainStripHi = b1;
        }
        const dim3 grid(h.paddedW / 32, std::max(1u, row1 - row0));
        DirtyBlocks mk{nullptr, (p->cfg.width + 31u) / 32u};
        if (fromVisibility && (onlyIfPhase2Drew || lateOnly) && p->chainDirtyTracked) mk.chainDirty = p->wsPtr<uint8_t>(p->ws.chainDirty);
        if (fromVisibility) hipLaunchKernelGGL(k_hzb_head<true>, grid, dim3(256), 0, s, h, static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]), static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]), skip, row0, mk);
        else hipLaunchKernelGGL(k_hzb_head<false>, grid, dim3(256), 0, s, h, (const unsigned long long*)nullptr, (float*)nullptr, skip, row0, mk);
        first = 6;
    }
    if (first < h.mipCount) {
        const bool seed = p->seedInHzbTail && !onlyIfPhase2Drew;      // (seeding happens whether or not the levels are rebuilt)
        hipLaunchKernelGGL(k_hzb_tail, dim3(1), dim3(1024), 0, s, h, first, skip, seed ? p->counters() : nullptr, p->cfg.maxTraversalRecords);
        if (seed) p->phase2Seeded = true;
    }
    BRMI_LAUNCH_CHECK(p, "k_hzb");
    return BRMI_OK;
}

}  // namespace brmi
