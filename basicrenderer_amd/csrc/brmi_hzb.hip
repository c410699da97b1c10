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
#include "brmi_hzb.h"

namespace brmi {

template <bool FROM_VIS>
__global__ void __launch_bounds__(256) k_hzb_head(HzbDesc h, const unsigned long long* vis, float* depthOut, const uint32_t* skipUnless, uint32_t blockRow0) {
    if (skipUnless && *skipUnless == 0u) return;
    hzb_head_block<FROM_VIS, FROM_VIS>(h, vis, depthOut, blockIdx.x, blockIdx.y + blockRow0, threadIdx.x);      // only the 32-row strips that touch this GPU's band are launched
}

// `seedCounters` (brmi_execute's phase-1 build only): the block also does k_seed_phase2's work for the culling pass that follows.
__global__ void __launch_bounds__(1024) k_hzb_tail(HzbDesc h, uint32_t firstMip, const uint32_t* skipUnless, uint32_t* seedCounters, uint32_t seedCapacity) {
    if (skipUnless && *skipUnless == 0u) return;
    if (seedCounters) seed_phase2(seedCounters, seedCapacity, threadIdx.x);
    hzb_tail_levels(h, firstMip, 1024u);
}

int launch_hzb(brmi_pass* p, hipStream_t s, bool fromVisibility, bool onlyIfPhase2Drew) {
    const HzbDesc h = p->hzbDesc();
    const bool head = h.paddedW >= 32 && h.paddedH >= 32;
    if (fromVisibility && !head) { int rc = launch_depth_copy(p, s); if (rc) return rc; fromVisibility = false; }   // tiny targets: unfused
    if (p->hzbMipCount < 2) return BRMI_OK;     // 1 x 1 target: mip 0 is all there is
    const uint32_t* skip = onlyIfPhase2Drew ? p->counters() + CNT_VISIBLE2 : nullptr;
    uint32_t first = 1;
    if (head) {
        // texels of mips 1-5 outside the band's 32-row strips never change: brmi_setup filled the chain with "empty"
        const uint32_t row0 = h.rowLo / 32u, row1 = std::min((h.rowHi + 31u) / 32u, h.paddedH / 32u);
        const dim3 grid(h.paddedW / 32, std::max(1u, row1 - row0));
        if (fromVisibility) hipLaunchKernelGGL(k_hzb_head<true>, grid, dim3(256), 0, s, h, static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]), static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]), skip, row0);
        else hipLaunchKernelGGL(k_hzb_head<false>, grid, dim3(256), 0, s, h, (const unsigned long long*)nullptr, (float*)nullptr, skip, row0);
        first = 6;
    }
    if (first < h.mipCount) {
        const bool seed = p->seedInHzbTail && !onlyIfPhase2Drew;
        hipLaunchKernelGGL(k_hzb_tail, dim3(1), dim3(1024), 0, s, h, first, skip, seed ? p->counters() : nullptr, p->cfg.maxTraversalRecords);
        if (seed) p->phase2Seeded = true;
    }
    BRMI_LAUNCH_CHECK(p, "k_hzb");
    return BRMI_OK;
}

}  // namespace brmi
