// brmi_hzb.h -- the depth chain's block functions: the kernels of brmi_hzb.hip call them, and inside brmi_execute the rebuild after phase 2
// rides on the G-buffer and shading launches (brmi_resolve.hip, brmi_light.hip).
#ifndef BRMI_HZB_H
#define BRMI_HZB_H
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
// WRITE_DEPTH (with FROM_VIS): the linear depth of the texels is written to the depth map on the way.  (bx, by): the 32 x 32 px block; `tid` of 256.
template <bool FROM_VIS, bool WRITE_DEPTH>
BRMI_DEV void hzb_head_block(const HzbDesc& h, const unsigned long long* vis, float* depthOut, uint32_t bx, uint32_t by, uint32_t tid) {
    __shared__ float lvl[16 * 16];
    const uint32_t tx = tid >> 4, ty = tid & 15u;             // ty fastest: follows the column-major tile layout
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
                    if (WRITE_DEPTH) *reinterpret_cast<float2*>(depthOut + ti) = make_float2(d[c][0], d[c][1]);
                } else {
                    for (uint32_t r = 0; r < 2; r++) {
                        const uint32_t y = y0 + r;
                        const bool in = x < h.width && y >= h.rowLo && y < h.rowHi;
                        d[c][r] = in ? key_depth(vis[tiled_index(x, y, h.tilesX)]) : __uint_as_float(BRMI_DEPTH_EMPTY_BITS);
                        if (WRITE_DEPTH && in) depthOut[tiled_index(x, y, h.tilesX)] = d[c][r];
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
        const uint32_t cx = tid / side, cy = tid % side;
        const bool active = tid < side * side;
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


// The rebuild of the chain after phase 2 inside brmi_execute (it only acts when phase 2 drew something): the head's workgroups ride behind the
// G-buffer kernel's (reading the final keys), the tail's one workgroup behind the shading kernel's -- two launches that usually find nothing to do
// become none.  `on` = 0: no ride.
struct HzbRide { HzbDesc h; const uint32_t* skipUnless; uint32_t mainBlocks, gridX, row0, firstTailMip, on; };

}  // namespace brmi
#endif
