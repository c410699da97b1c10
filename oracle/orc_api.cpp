// orc_api.cpp -- small exported helpers so tests can check the oracle's own primitives.
// TEST INFRASTRUCTURE ONLY (see orc_common.h).
#include "orc_common.h"
#include "orc_texture.h"

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

// the software sampler, one call per sample (orc_texture.h); out = float4 per sample
int orc_sample_level(const brmi_scene_buffers* sc, uint32_t textureIndex, uint32_t samplerIndex, const float* uv, const float* lod, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) { const float4 r = sampleLevel(*sc, textureIndex, samplerIndex, float2{uv[i * 2], uv[i * 2 + 1]}, lod[i]); std::memcpy(out + i * 4, &r, 16); }
    return 0;
}
int orc_sample_grad(const brmi_scene_buffers* sc, uint32_t textureIndex, uint32_t samplerIndex, const float* uv, const float* ddx, const float* ddy, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float4 r = sampleGrad(*sc, textureIndex, samplerIndex, float2{uv[i * 2], uv[i * 2 + 1]}, float2{ddx[i * 2], ddx[i * 2 + 1]}, float2{ddy[i * 2], ddy[i * 2 + 1]});
        std::memcpy(out + i * 4, &r, 16);
    }
    return 0;
}
uint32_t clusterSlice(float z, float zNear, float zFar, float zSplit, uint32_t nearSlices, uint32_t gz);      // orc_light.cpp
int orc_cluster_slice(const float* z, uint64_t n, float zNear, float zFar, float zSplit, uint32_t nearSlices, uint32_t gz, uint32_t* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = clusterSlice(z[i], zNear, zFar, zSplit, nearSlices, gz);
    return 0;
}
int orc_log2_poly(const float* in, float* out, uint64_t n) { for (uint64_t i = 0; i < n; i++) out[i] = log2Poly(in[i]); return 0; }
// decoded UV set `uvSet` of every vertex of a visible cluster's meshlet; returns the vertex count
int orc_cluster_uvs(const brmi_scene_buffers* sc, const brmi_visible_cluster* cluster, uint32_t uvSet, float* out) {
    const uint8_t* slab = sc->slabs[vcSlabDescriptor(*cluster)];
    const uint32_t pageOff = vcPageByteOffset(*cluster), meshlet = vcLocalMeshlet(*cluster);
    const brmi_page_header& hdr = *pageHeader(slab, pageOff);
    const uint32_t n = descVertexCount(*meshletDesc(slab, pageOff, hdr.descriptorOffset, meshlet));
    for (uint32_t v = 0; v < n; v++) { const float2 uv = decodeCompressedUV(slab, pageOff, hdr, meshlet, uvSet, v); out[v * 2] = uv.x; out[v * 2 + 1] = uv.y; }
    return (int)n;
}
int orc_alpha_test_failed(const brmi_scene_buffers* sc, uint32_t materialDataIndex, const float* uv, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = alphaTestFailed(*sc, float2{uv[i * 2], uv[i * 2 + 1]}, materialDataIndex) ? 1 : 0;
    return 0;
}

}  // extern "C"
