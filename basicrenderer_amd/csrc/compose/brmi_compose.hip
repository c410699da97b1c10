// brmi_compose.hip -- libbrmi_compose.so: RCCL composition of the row-band partition (include/brmi_compose.h).
//
// One all-gather per frame on the composer's own stream, fed from a staging copy made on the render stream; events order the two
// streams, the host never waits.  xGMI is point to point (7 links per GPU): an all-gather of equal bands moves one band over every
// link in each direction, so the collective is link-bound by the band size, not by the rank count.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "brmi_compose.h"

static_assert(sizeof(ncclUniqueId) == BRMI_COMPOSE_ID_BYTES, "ncclUniqueId size");

struct brmi_composer {
    brmi_compose_config cfg{};
    ncclComm_t comm = nullptr;
    hipStream_t collStream = nullptr;
    std::vector<hipEvent_t> staged, done;       // per slot: staging copy finished (render stream) / collective finished (collective stream)
    std::vector<bool> inFlight;
    uint8_t* staging = nullptr; uint8_t* output = nullptr;
    uint64_t bandOffset = 0, bandBytes = 0, stagingBytes = 0, outputBytes = 0, pixels = 0;
    uint64_t frames = 0;
    std::string err;
};

namespace {

int fail(brmi_composer* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define CHECK_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), -2, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define CHECK_NCCL(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail((c), -5, "%s: %s", #call, ncclGetErrorString(r_)); } while (0)

// RGBA16F -> RGB16F, four pixels per thread: 32 B in, 24 B out
__global__ void __launch_bounds__(256) k_pack_rgb16f(const uint4* src, uint2* dst, uint64_t quads) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 a = src[2 * i], b = src[2 * i + 1];      // a = px0 (x, y), px1 (z, w); b = px2, px3; a pixel is {r | g << 16, b | alpha << 16}
        const uint32_t r0g0 = a.x, b0 = a.y & 0xFFFFu, r1g1 = a.z, b1 = a.w & 0xFFFFu, r2g2 = b.x, b2 = b.y & 0xFFFFu, r3g3 = b.z, b3 = b.w & 0xFFFFu;
        dst[3 * i]     = make_uint2(r0g0, b0 | (r1g1 << 16));
        dst[3 * i + 1] = make_uint2((r1g1 >> 16) | (b1 << 16), r2g2);
        dst[3 * i + 2] = make_uint2(b2 | (r3g3 << 16), (r3g3 >> 16) | (b3 << 16));
    }
}

}  // namespace

extern "C" {

int brmi_compose_unique_id(uint8_t id[BRMI_COMPOSE_ID_BYTES]) {
    if (!id) return -1;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return -5;
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int brmi_compose_create(const brmi_compose_config* cfg, const uint8_t id[BRMI_COMPOSE_ID_BYTES], brmi_composer** out) {
    if (!cfg || !id || !out || cfg->structSize != sizeof(brmi_compose_config)) return -1;
    if (cfg->nRanks == 0 || cfg->rank >= cfg->nRanks || cfg->depth == 0 || cfg->depth > 8 || cfg->width == 0 || cfg->bytesPerPixel == 0) return -1;
    if (cfg->bandY0 % 8u || cfg->bandY1 % 8u || cfg->bandY1 <= cfg->bandY0) return -1;
    if (cfg->transport > BRMI_TRANSPORT_RGB16F || (cfg->transport == BRMI_TRANSPORT_RGB16F && cfg->bytesPerPixel != 8)) return -1;
    brmi_composer* c = new brmi_composer();
    c->cfg = *cfg;
    const uint64_t tilesX = (cfg->width + 7u) / 8u, rowBytes = tilesX * 64u * cfg->bytesPerPixel;      // one 8-row tile row of the surface
    c->bandOffset = (uint64_t)(cfg->bandY0 / 8u) * rowBytes; c->bandBytes = (uint64_t)((cfg->bandY1 - cfg->bandY0) / 8u) * rowBytes;
    c->pixels = c->bandBytes / cfg->bytesPerPixel;
    c->stagingBytes = cfg->transport == BRMI_TRANSPORT_RGB16F ? c->pixels * 6u : c->bandBytes;
    c->outputBytes = c->stagingBytes * cfg->nRanks;
    *out = c;
    CHECK_HIP(c, hipSetDevice(cfg->device));
    ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
    CHECK_NCCL(c, ncclCommInitRank(&c->comm, (int)cfg->nRanks, u, (int)cfg->rank));
    // highest priority: from two GPUs on the gather bounds the frame rate, and with two frames in flight the render
    // streams keep every CU busy -- the collective's few workgroups must not queue behind them
    { int least = 0, greatest = 0; CHECK_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest)); CHECK_HIP(c, hipStreamCreateWithPriority(&c->collStream, hipStreamNonBlocking, greatest)); }
    c->staged.resize(cfg->depth); c->done.resize(cfg->depth); c->inFlight.assign(cfg->depth, false);
    for (uint32_t i = 0; i < cfg->depth; i++) { CHECK_HIP(c, hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming)); CHECK_HIP(c, hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming)); }
    return 0;
}

uint64_t brmi_compose_staging_bytes(const brmi_composer* c) { return c ? c->stagingBytes : 0; }
uint64_t brmi_compose_output_bytes(const brmi_composer* c) { return c ? c->outputBytes : 0; }

int brmi_compose_bind(brmi_composer* c, void* staging, uint64_t stagingBytes, void* output, uint64_t outputBytes) {
    if (!c || !staging || !output) return -1;
    if (stagingBytes < c->stagingBytes * c->cfg.depth || outputBytes < c->outputBytes * c->cfg.depth) return fail(c, -3, "brmi_compose_bind: %u buffers of %llu B (staging) and %llu B (output) are needed",
                                                                                                                      c->cfg.depth, (unsigned long long)c->stagingBytes, (unsigned long long)c->outputBytes);
    if ((reinterpret_cast<uintptr_t>(staging) | reinterpret_cast<uintptr_t>(output)) & 15u) return fail(c, -1, "brmi_compose_bind: buffers must be 16-byte aligned");
    c->staging = static_cast<uint8_t*>(staging); c->output = static_cast<uint8_t*>(output);
    return 0;
}

int brmi_compose_submit(brmi_composer* c, const void* surface, brmi_compose_stream renderStream) {
    if (!c || !surface) return -1;
    if (!c->staging) return fail(c, -4, "brmi_compose_submit: call brmi_compose_bind first");
    hipStream_t rs = static_cast<hipStream_t>(renderStream);
    const uint32_t slot = (uint32_t)(c->frames % c->cfg.depth);
    uint8_t* st = c->staging + (uint64_t)slot * c->stagingBytes; uint8_t* dst = c->output + (uint64_t)slot * c->outputBytes;
    const uint8_t* band = static_cast<const uint8_t*>(surface) + c->bandOffset;
    if (c->inFlight[slot]) CHECK_HIP(c, hipStreamWaitEvent(rs, c->done[slot], 0));       // the slot's previous collective has read the staging buffer
    if (c->cfg.transport == BRMI_TRANSPORT_RGB16F) {
        const uint64_t quads = c->pixels / 4u;                                            // a band is whole 8x8 tiles
        hipLaunchKernelGGL(k_pack_rgb16f, dim3((unsigned)std::min<uint64_t>(4096, (quads + 255) / 256)), dim3(256), 0, rs, reinterpret_cast<const uint4*>(band), reinterpret_cast<uint2*>(st), quads);
        CHECK_HIP(c, hipGetLastError());
    } else CHECK_HIP(c, hipMemcpyAsync(st, band, c->bandBytes, hipMemcpyDeviceToDevice, rs));
    CHECK_HIP(c, hipEventRecord(c->staged[slot], rs));
    CHECK_HIP(c, hipStreamWaitEvent(c->collStream, c->staged[slot], 0));
    CHECK_NCCL(c, ncclAllGather(st, dst, c->stagingBytes, ncclUint8, c->comm, c->collStream));
    CHECK_HIP(c, hipEventRecord(c->done[slot], c->collStream));
    c->inFlight[slot] = true;
    c->frames++;
    return (int)slot;
}

int brmi_compose_finish(brmi_composer* c, brmi_compose_stream stream, void** composed) {
    if (!c) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (uint32_t i = 0; i < c->cfg.depth; i++) if (c->inFlight[i]) { CHECK_HIP(c, hipStreamWaitEvent(s, c->done[i], 0)); c->inFlight[i] = false; }
    if (composed) *composed = c->frames ? c->output + (uint64_t)((c->frames - 1) % c->cfg.depth) * c->outputBytes : nullptr;
    return 0;
}

void brmi_compose_destroy(brmi_composer* c) {
    if (!c) return;
    if (c->collStream) (void)hipStreamSynchronize(c->collStream);
    for (hipEvent_t e : c->staged) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->collStream) (void)hipStreamDestroy(c->collStream);
    delete c;
}

const char* brmi_compose_last_error(const brmi_composer* c) { return c ? c->err.c_str() : "null composer"; }

}  // extern "C"
