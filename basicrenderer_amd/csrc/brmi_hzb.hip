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
    return (x < h.width && y < h.height) ? h.depth[tiled_index(x, y, h.tilesX)] : __uint_as_float(BRMI_DEPTH_EMPTY_BITS);
}

// mips 1..5 from the depth map: block = 16 x 16 texels of mip 1 (requires paddedW, paddedH >= 32)
__global__ void __launch_bounds__(256) k_hzb_head(HzbDesc h) {
    __shared__ float lvl[16 * 16];
    const uint32_t tx = threadIdx.x >> 4, ty = threadIdx.x & 15u;             // ty fastest: follows the column-major tile layout
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
    const uint32_t x1 = bx * 16u + tx, y1 = by * 16u + ty;                    // mip-1 texel
    float v;
    {
        const uint32_t x0 = x1 * 2u, y0 = y1 * 2u;
        if (x0 + 1u < h.width && y0 + 1u < h.height) {
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

// mips firstMip..last, one workgroup, level by level (source texels clamped to the source extent: a dimension that
// reached 1 stays 1)
__global__ void __launch_bounds__(1024) k_hzb_tail(HzbDesc h, uint32_t firstMip) {
    for (uint32_t mip = firstMip; mip < h.mipCount; mip++) {
        const uint32_t sw = max(1u, h.paddedW >> (mip - 1u)), sh = max(1u, h.paddedH >> (mip - 1u));
        const uint32_t w = max(1u, h.paddedW >> mip), hh = max(1u, h.paddedH >> mip);
        const float* src = h.mips + h.mipOffset[mip - 1u];
        float* dst = h.mips + h.mipOffset[mip];
        for (uint32_t i = threadIdx.x; i < w * hh; i += 1024u) {
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

int launch_hzb(brmi_pass* p, hipStream_t s) {
    if (p->hzbMipCount < 2) return BRMI_OK;     // 1 x 1 target: mip 0 is all there is
    const HzbDesc h = p->hzbDesc();
    uint32_t first = 1;
    if (h.paddedW >= 32 && h.paddedH >= 32) {
        hipLaunchKernelGGL(k_hzb_head, dim3(h.paddedW / 32, h.paddedH / 32), dim3(256), 0, s, h);
        first = 6;
    }
    if (first < h.mipCount) hipLaunchKernelGGL(k_hzb_tail, dim3(1), dim3(1024), 0, s, h, first);
    BRMI_LAUNCH_CHECK(p, "k_hzb");
    return BRMI_OK;
}

}  // namespace brmi
