// orc_skin.cpp -- skin-matrix blend used by the raster, culling-bounds and resolve restatements.
// TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED.
//
// Follows BuildSkinMatrix / LoadBoneSkinMatrix (BR/shaders/Include/skinningCommon.hlsli:23-88):
//   M = sum_{i<8} w_i * transpose(bone[j_i] * invBind[j_i])   ;  p' = mul(float4(p,1), M)
// The frame is static, so `skinningMatrices` already holds the per-(slot, joint) product
// bone*invBind (64 joints per slot); the transpose and the 8-term blend happen here per vertex.
#include "orc_common.h"

namespace orc {

mat4 buildSkinMatrix(const brmi_scene_buffers& sc, uint32_t slot, const uint32_t joints[8], const float weights[8]) {
    mat4 r;
    if (slot == 0xFFFFFFFFu || sc.skinningMatrices == nullptr) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return r;
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = 0.0f;
    for (int k = 0; k < 8; k++) {
        const float* b = sc.skinningMatrices + ((size_t)slot * 64u + joints[k]) * 16u;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                const float t = weights[k] * b[j * 4 + i];   // transpose(product)
                r.m[i][j] = (k == 0) ? t : r.m[i][j] + t;
            }
    }
    return r;
}

}  // namespace orc
