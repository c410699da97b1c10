// orc_api.cpp -- small exported helpers so tests can check the oracle's own primitives.
// TEST INFRASTRUCTURE ONLY (see orc_common.h).
#include "orc_common.h"

using namespace orc;

extern "C" {

int orc_f32_to_f16(const float* in, uint16_t* out, uint64_t n) { for (uint64_t i = 0; i < n; i++) out[i] = f32_to_f16(in[i]); return 0; }
int orc_f16_to_f32(const uint16_t* in, float* out, uint64_t n) { for (uint64_t i = 0; i < n; i++) out[i] = f16_to_f32(in[i]); return 0; }
int orc_unorm8(const float* in, uint32_t* out, uint64_t n) { for (uint64_t i = 0; i < n; i++) out[i] = unorm8(in[i]); return 0; }

// PackVisKey / UnpackVisKey round trip (visibilityPacking.hlsli:11-37)
uint64_t orc_pack_vis_key(float depth, uint32_t cluster, uint32_t tri) {
    uint64_t depthBits = asuint(depth) >> 1;
    return (depthBits << BRMI_VIS_META_BITS) | ((uint64_t)(cluster & 0x3FFFFFFu) << BRMI_VIS_TRI_BITS) | (uint64_t)(tri & 0x7Fu);
}

}  // extern "C"
