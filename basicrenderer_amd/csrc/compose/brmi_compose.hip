// brmi_compose.hip -- libbrmi_compose.so: RCCL composition of the row-band partition (include/brmi_compose.h).
//
// One all-gather per frame on the composer's own stream, fed from a staging copy made on the render stream; events order the two
// streams, the host never waits.  xGMI is point to point (7 links per GPU): an all-gather of equal bands moves one band over every
// link in each direction, so the collective is link-bound by the band size, not by the rank count.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "brmi_compose.h"

static_assert(sizeof(ncclUniqueId) == BRMI_COMPOSE_ID_BYTES, "ncclUniqueId size");

// peer-write path: what a rank exports, and how a rank sees a peer
struct PeerHandle { hipIpcMemHandle_t output, flags; uint64_t outputBytes; uint32_t rank, pad; };
static_assert(sizeof(PeerHandle) <= BRMI_COMPOSE_HANDLE_BYTES, "BRMI_COMPOSE_HANDLE_BYTES");
constexpr uint32_t kMaxRanks = 16;
// flag words of one rank (in its own memory, written by its peers with system-scope stores): landed[slot][writer] = newest frame whose band
// the writer has stored into this rank's output[slot]; submitted[reader] = newest frame that reader has submitted itself (it no longer
// needs frame - depth); status = 0 or the frame a wait gave up on
struct FlagBlock { uint32_t landed[8][kMaxRanks]; uint32_t submitted[kMaxRanks]; uint32_t status; uint32_t pad[15]; };
struct PeerTable { uint8_t* output[kMaxRanks]; FlagBlock* flags[kMaxRanks]; };

struct brmi_composer {
    brmi_compose_config cfg{};
    bool peerWrite = false, ownsShared = false, imported = false;
    FlagBlock* flags = nullptr;                 // this rank's (device memory)
    PeerTable peers{};                          // [rank] -> mapped pointers ([own rank] = own buffers)
    std::vector<void*> opened;                  // hipIpcOpenMemHandle results, closed on destroy
    ncclComm_t comm = nullptr;
    hipStream_t collStream = nullptr;
    std::vector<hipEvent_t> staged, done;       // per slot: staging copy finished (render stream) / collective finished (collective stream)
    std::vector<bool> inFlight;
    uint8_t* staging = nullptr; uint8_t* output = nullptr;
    uint64_t bandOffset = 0, bandBytes = 0, stagingBytes = 0, outputBytes = 0, pixels = 0;
    bool dynamic = false;                       // cfg.frameHeight > 0: bands of any height at their own rows of the composed frame (brmi_compose_set_bounds)
    std::vector<uint32_t> bounds;               // [nRanks + 1] rows, this frame's partition
    uint64_t rowBytes = 0, rowBytesOut = 0;     // one 8-row tile row of the surface / of the composed image (transport form)
    uint64_t bandOut() const { return dynamic ? (uint64_t)(cfg.bandY0 / 8u) * rowBytesOut : (uint64_t)cfg.rank * stagingBytes; }      // where this rank's band starts in a composed image
    uint64_t frames = 0;
    uint32_t openRow = 0;                       // submit_rows: next row expected (0 = no frame open)
    hipEvent_t slabReady = nullptr;             // submit_rows: the slab's shading is enqueued (render stream -> composer stream)
    hipEvent_t ownStores = nullptr; bool ownStoresRecorded = false;      // submit_rows: the rank's own stores of the newest frame (composer stream -> finish)
    // submit_rows reads the caller's surface on the COMPOSER's stream: the newest of those reads per surface (a renderer with frames in flight hands over
    // the surfaces of several passes in turn), for brmi_compose_wait_source -- what the stream that next writes the surface has to wait for
    struct SourceRead { const void* surface; hipEvent_t read; };
    std::vector<SourceRead> sourceReads;
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

// ---- peer-write path -----------------------------------------------------------------------------------------------------------------
// waits (one wave, system-scope loads, s_sleep between polls) until every word of `words[0..n)` except [skip] is >= `want`; gives up after
// `timeoutTicks` of the 100 MHz clock and latches `want` into *status
__global__ void k_wait_flags(const uint32_t* words, uint32_t n, uint32_t skip, uint32_t want, uint32_t* status, unsigned long long timeoutTicks) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = lane >= n || lane == skip;
    while (!__all(ok)) {
        if (!ok) ok = (int32_t)(__hip_atomic_load(&words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeoutTicks) { if (lane == 0) __hip_atomic_store(status, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: the bands behind the flags
}
// A store that goes THROUGH this GPU's caches to the memory it addresses (system-scope relaxed atomic store: global_store ... sc0 sc1 on gfx950; 8 B at a time, the widest
// an atomic store comes in).  The images are ordinary coarse-grained device memory of OTHER GPUs mapped over hipIpc: a plain store may sit dirty in this GPU's L2 past the
// end of the kernel (HIP only promises agent scope between kernels of a stream), and the `landed` flag -- uncached memory, written by the NEXT kernel with a relaxed
// store -- could then reach the peer before the band it announces.  With write-through stores the kernel boundary (every store of the kernel has been acknowledged by
// the memory it went to before the next kernel starts) is all the release the flag needs; a system-scope release fence instead writes back every XCD's whole L2 --
// G-buffer planes and all -- and cost a frame 0.45 ms in round 5.
__device__ __forceinline__ void store_through(uint2* d, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(d), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the band into every rank's image (own copy included).  RGB = drop the alpha on the way: four pixels (32 B) in, 24 B out per thread and peer.
template <bool RGB>
__global__ void __launch_bounds__(256) k_peer_write(const uint4* src, PeerTable peers, uint32_t nRanks, uint64_t slotOffset, uint64_t bandOffsetOut, uint64_t units) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (uint64_t)gridDim.x * blockDim.x) {
        if (RGB) {
            const uint4 a = src[2 * i], b = src[2 * i + 1];
            const uint32_t r0g0 = a.x, b0 = a.y & 0xFFFFu, r1g1 = a.z, b1 = a.w & 0xFFFFu, r2g2 = b.x, b2 = b.y & 0xFFFFu, r3g3 = b.z, b3 = b.w & 0xFFFFu;
            const uint2 o0 = make_uint2(r0g0, b0 | (r1g1 << 16)), o1 = make_uint2((r1g1 >> 16) | (b1 << 16), r2g2), o2 = make_uint2(b2 | (r3g3 << 16), (r3g3 >> 16) | (b3 << 16));
            for (uint32_t p = 0; p < nRanks; p++) { uint2* d = reinterpret_cast<uint2*>(peers.output[p] + slotOffset + bandOffsetOut) + 3 * i; store_through(d, o0); store_through(d + 1, o1); store_through(d + 2, o2); }
        } else {
            const uint4 v = src[i];
            for (uint32_t p = 0; p < nRanks; p++) { uint2* d = reinterpret_cast<uint2*>(peers.output[p] + slotOffset + bandOffsetOut) + 2 * i; store_through(d, make_uint2(v.x, v.y)); store_through(d + 1, make_uint2(v.z, v.w)); }
        }
    }
}
// after k_peer_write has retired: "my band of `frame` has landed in your slot"
__global__ void k_signal_landed(PeerTable peers, uint32_t nRanks, uint32_t rank, uint32_t slot, uint32_t frame) {
    const uint32_t p = threadIdx.x;
    // (relaxed: the band's stores were write-through stores of an EARLIER kernel of this stream -- store_through above --, acknowledged by the peers' memory before
    // that kernel retired; nothing of the band is left in this GPU's caches for a release to publish)
    if (p < nRanks) __hip_atomic_store(&peers.flags[p]->landed[slot][rank], frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// at the head of a submit: "I am at frame `frame`: whatever you hold for me of frame - depth may be overwritten"
__global__ void k_signal_submitted(PeerTable peers, uint32_t nRanks, uint32_t rank, uint32_t frame) {
    const uint32_t p = threadIdx.x;
    if (p < nRanks) __hip_atomic_store(&peers.flags[p]->submitted[rank], frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (nothing to publish with it: it only says how far this rank has come)
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
    if (cfg->bandY0 % 8u || cfg->bandY1 % 8u || (cfg->bandY1 <= cfg->bandY0 && !cfg->frameHeight) || cfg->bandY1 < cfg->bandY0) return -1;
    if (cfg->frameHeight && (cfg->frameHeight % 8u || cfg->bandY1 > cfg->frameHeight)) return -1;
    if (cfg->transport > BRMI_TRANSPORT_RGB16F || (cfg->transport == BRMI_TRANSPORT_RGB16F && cfg->bytesPerPixel != 8)) return -1;
    if (cfg->path > BRMI_COMPOSE_PEER_WRITE || (cfg->path == BRMI_COMPOSE_PEER_WRITE && cfg->nRanks > kMaxRanks)) return -1;
    brmi_composer* c = new brmi_composer();
    c->cfg = *cfg;
    c->peerWrite = cfg->path == BRMI_COMPOSE_PEER_WRITE;
    const uint64_t tilesX = (cfg->width + 7u) / 8u, rowBytes = tilesX * 64u * cfg->bytesPerPixel;      // one 8-row tile row of the surface
    c->bandOffset = (uint64_t)(cfg->bandY0 / 8u) * rowBytes; c->bandBytes = (uint64_t)((cfg->bandY1 - cfg->bandY0) / 8u) * rowBytes;
    c->pixels = c->bandBytes / cfg->bytesPerPixel;
    c->stagingBytes = cfg->transport == BRMI_TRANSPORT_RGB16F ? c->pixels * 6u : c->bandBytes;
    c->outputBytes = c->stagingBytes * cfg->nRanks;
    c->rowBytes = rowBytes; c->rowBytesOut = cfg->transport == BRMI_TRANSPORT_RGB16F ? rowBytes / cfg->bytesPerPixel * 6u : rowBytes;
    if (cfg->frameHeight) {      // bands of any height: a staging buffer for up to the whole frame, the composed image IS the frame
        c->dynamic = true;
        c->stagingBytes = c->outputBytes = (uint64_t)(cfg->frameHeight / 8u) * c->rowBytesOut;
        c->bounds.assign(cfg->nRanks + 1u, 0u);      // until brmi_compose_set_bounds: only this rank's band is known
    }
    *out = c;
    CHECK_HIP(c, hipSetDevice(cfg->device));
    if (!c->peerWrite) {
        ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
        CHECK_NCCL(c, ncclCommInitRank(&c->comm, (int)cfg->nRanks, u, (int)cfg->rank));
    }
    // highest priority: from two GPUs on the gather bounds the frame rate, and with two frames in flight the render
    // streams keep every CU busy -- the collective's few workgroups must not queue behind them
    { int least = 0, greatest = 0; CHECK_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest)); CHECK_HIP(c, hipStreamCreateWithPriority(&c->collStream, hipStreamNonBlocking, greatest)); }
    c->staged.resize(cfg->depth); c->done.resize(cfg->depth); c->inFlight.assign(cfg->depth, false);
    for (uint32_t i = 0; i < cfg->depth; i++) { CHECK_HIP(c, hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming)); CHECK_HIP(c, hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming)); }
    return 0;
}

uint64_t brmi_compose_staging_bytes(const brmi_composer* c) { return c ? c->stagingBytes : 0; }
uint64_t brmi_compose_output_bytes(const brmi_composer* c) { return c ? c->outputBytes : 0; }

int brmi_compose_alloc_shared(brmi_composer* c) {
    if (!c) return -1;
    if (!c->peerWrite) return fail(c, -4, "brmi_compose_alloc_shared: the composer was created for the all-gather path");
    if (c->output) return fail(c, -4, "brmi_compose_alloc_shared: buffers are bound already");
    void* out = nullptr; void* fl = nullptr;
    CHECK_HIP(c, hipMalloc(&out, c->outputBytes * c->cfg.depth));
    // The flag words are written by OTHER GPUs (over hipIpc mappings) while a kernel on this GPU polls them: coarse-grained device memory is only
    // guaranteed coherent across agents at kernel boundaries, so a polling wave could keep reading a stale line from this GPU's own L2 until the
    // timeout.  Uncached (MTYPE_UC) device memory, as RCCL takes for its own flags: every access goes to memory.  The images stay ordinary
    // device memory on purpose: they are read by kernels launched AFTER the wait kernel has seen the landed words (a kernel boundary), and
    // uncached images would make every consumer read them at fabric speed.
    if (hipExtMallocWithFlags(&fl, sizeof(FlagBlock), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        CHECK_HIP(c, hipExtMallocWithFlags(&fl, sizeof(FlagBlock), hipDeviceMallocFinegrained));
    }
    CHECK_HIP(c, hipMemset(fl, 0, sizeof(FlagBlock)));
    CHECK_HIP(c, hipDeviceSynchronize());
    c->output = static_cast<uint8_t*>(out); c->flags = static_cast<FlagBlock*>(fl); c->ownsShared = true;
    c->staging = c->output;      // (unused on this path: the band goes straight from the surface into the images)
    c->peers.output[c->cfg.rank] = c->output; c->peers.flags[c->cfg.rank] = c->flags;
    return 0;
}

int brmi_compose_export(brmi_composer* c, uint8_t handle[BRMI_COMPOSE_HANDLE_BYTES]) {
    if (!c || !handle) return -1;
    if (!c->peerWrite || !c->output || !c->flags) return fail(c, -4, "brmi_compose_export: peer-write composer with shared buffers (brmi_compose_alloc_shared) needed");
    PeerHandle h{};
    CHECK_HIP(c, hipIpcGetMemHandle(&h.output, c->output));
    CHECK_HIP(c, hipIpcGetMemHandle(&h.flags, c->flags));
    h.outputBytes = c->outputBytes * c->cfg.depth; h.rank = c->cfg.rank;
    std::memset(handle, 0, BRMI_COMPOSE_HANDLE_BYTES); std::memcpy(handle, &h, sizeof(h));
    return 0;
}

int brmi_compose_import(brmi_composer* c, const uint8_t* handles, uint32_t count) {
    if (!c || !handles) return -1;
    if (!c->peerWrite || !c->output) return fail(c, -4, "brmi_compose_import: export first");
    if (count != c->cfg.nRanks) return fail(c, -1, "brmi_compose_import: %u handles for %u ranks", count, c->cfg.nRanks);
    for (uint32_t r = 0; r < count; r++) {
        if (r == c->cfg.rank) continue;
        PeerHandle h; std::memcpy(&h, handles + (size_t)r * BRMI_COMPOSE_HANDLE_BYTES, sizeof(h));
        if (h.rank != r || h.outputBytes != c->outputBytes * c->cfg.depth) return fail(c, -1, "brmi_compose_import: entry %u is rank %u with %llu B (expected %llu)", r, h.rank, (unsigned long long)h.outputBytes, (unsigned long long)(c->outputBytes * c->cfg.depth));
        void* po = nullptr; void* pf = nullptr;
        CHECK_HIP(c, hipIpcOpenMemHandle(&po, h.output, hipIpcMemLazyEnablePeerAccess)); c->opened.push_back(po);
        CHECK_HIP(c, hipIpcOpenMemHandle(&pf, h.flags, hipIpcMemLazyEnablePeerAccess)); c->opened.push_back(pf);
        c->peers.output[r] = static_cast<uint8_t*>(po); c->peers.flags[r] = static_cast<FlagBlock*>(pf);
    }
    c->imported = true;
    return 0;
}

int brmi_compose_last_wait_status(brmi_composer* c) {
    if (!c || !c->flags) return -1;
    uint32_t st = 0;
    CHECK_HIP(c, hipMemcpy(&st, &c->flags->status, 4, hipMemcpyDeviceToHost));
    return st ? fail(c, -6, "a wait for a peer's band of frame %u timed out", st) : 0;
}

int brmi_compose_bind(brmi_composer* c, void* staging, uint64_t stagingBytes, void* output, uint64_t outputBytes) {
    if (!c || !staging || !output) return -1;
    if (c->peerWrite) return fail(c, -4, "brmi_compose_bind: the peer-write path shares its buffers with other processes: brmi_compose_alloc_shared");
    if (stagingBytes < c->stagingBytes * c->cfg.depth || outputBytes < c->outputBytes * c->cfg.depth) return fail(c, -3, "brmi_compose_bind: %u buffers of %llu B (staging) and %llu B (output) are needed",
                                                                                                                      c->cfg.depth, (unsigned long long)c->stagingBytes, (unsigned long long)c->outputBytes);
    if ((reinterpret_cast<uintptr_t>(staging) | reinterpret_cast<uintptr_t>(output)) & 15u) return fail(c, -1, "brmi_compose_bind: buffers must be 16-byte aligned");
    c->staging = static_cast<uint8_t*>(staging); c->output = static_cast<uint8_t*>(output);
    return 0;
}

int brmi_compose_set_bounds(brmi_composer* c, const uint32_t* rowBounds) {
    if (!c || !rowBounds) return -1;
    if (!c->dynamic) return fail(c, -4, "brmi_compose_set_bounds: the composer was created without frameHeight (equal bands, fixed)");
    if (c->openRow) return fail(c, -4, "brmi_compose_set_bounds: a frame is open (brmi_compose_submit_rows up to row %u so far)", c->openRow);
    const uint32_t n = c->cfg.nRanks;
    if (rowBounds[0] != 0u || rowBounds[n] != c->cfg.frameHeight) return fail(c, -1, "brmi_compose_set_bounds: the bounds run from %u to %u, the frame from 0 to %u", rowBounds[0], rowBounds[n], c->cfg.frameHeight);
    for (uint32_t r = 0; r < n; r++) if (rowBounds[r] % 8u || rowBounds[r + 1] < rowBounds[r]) return fail(c, -1, "brmi_compose_set_bounds: bound %u = %u (ascending multiples of 8)", r, rowBounds[r]);
    c->bounds.assign(rowBounds, rowBounds + n + 1u);
    c->cfg.bandY0 = rowBounds[c->cfg.rank]; c->cfg.bandY1 = rowBounds[c->cfg.rank + 1u];
    c->bandOffset = (uint64_t)(c->cfg.bandY0 / 8u) * c->rowBytes; c->bandBytes = (uint64_t)((c->cfg.bandY1 - c->cfg.bandY0) / 8u) * c->rowBytes;
    c->pixels = c->bandBytes / c->cfg.bytesPerPixel;
    return 0;
}

int brmi_compose_balance_rows(const float* rankMs, const uint32_t* boundsIn, uint32_t nRanks, uint32_t frameHeight, uint32_t align, float damping, uint32_t minRows, float* rowCost, uint32_t* boundsOut) {
    if (!rankMs || !boundsIn || !boundsOut || !rowCost || nRanks == 0 || align == 0 || frameHeight % align || boundsIn[0] != 0u || boundsIn[nRanks] != frameHeight) return -1;
    if (!(damping > 0.0f) || damping > 1.0f) damping = 1.0f;
    minRows = std::max(minRows, align); minRows = (minRows + align - 1u) / align * align;
    if ((uint64_t)minRows * nRanks > frameHeight) return -1;
    const uint32_t blocks = frameHeight / align;
    for (uint32_t r = 0; r < nRanks; r++) if (boundsIn[r + 1] < boundsIn[r] || boundsIn[r] % align || !(rankMs[r] >= 0.0f)) return -1;
    // 1. the cost profile learns from this measurement: inside every band the profile is scaled so that it adds up to the band's measured time.  Frames measured under
    //    OTHER bounds shaped the profile inside the band; this one fixes its sum (iterative proportional fitting: a few partitions locate a horizon that one cannot)
    bool any = false;
    for (uint32_t b = 0; b < blocks; b++) any = any || rowCost[b] > 0.0f;
    if (!any) for (uint32_t b = 0; b < blocks; b++) rowCost[b] = 1.0f;
    double maxMs = 0.0;
    for (uint32_t r = 0; r < nRanks; r++) {
        const uint32_t lo = boundsIn[r] / align, hi = boundsIn[r + 1] / align;
        double sum = 0.0;
        for (uint32_t b = lo; b < hi; b++) sum += rowCost[b];
        if (hi > lo && sum > 0.0) { const float k = (float)(rankMs[r] / sum); for (uint32_t b = lo; b < hi; b++) rowCost[b] *= k; }
        else if (hi > lo) for (uint32_t b = lo; b < hi; b++) rowCost[b] = rankMs[r] / (float)(hi - lo);
        maxMs = std::max(maxMs, (double)rankMs[r]);
    }
    // 2. cut the profile into pieces of equal cost
    std::vector<double> cum(blocks + 1u, 0.0);
    for (uint32_t b = 0; b < blocks; b++) cum[b + 1u] = cum[b] + std::max(0.0f, rowCost[b]);
    const double total = cum[blocks];
    boundsOut[0] = 0u; boundsOut[nRanks] = frameHeight;
    uint32_t j = 0;
    for (uint32_t k = 1; k < nRanks; k++) {
        double ideal = (double)frameHeight * k / nRanks;
        if (total > 0.0) {
            const double want = total * k / nRanks;
            while (j + 1u < blocks && cum[j + 1u] < want) j++;
            const double c = cum[j + 1u] - cum[j];
            ideal = ((double)j + (c > 0.0 ? std::min(1.0, std::max(0.0, (want - cum[j]) / c)) : 0.0)) * align;
        }
        const double moved = boundsIn[k] + (double)damping * (ideal - boundsIn[k]);
        boundsOut[k] = (uint32_t)(std::min((double)frameHeight, std::max(0.0, moved)) / align + 0.5) * align;
    }
    // every band at least minRows high: push forwards, then backwards
    for (uint32_t k = 1; k <= nRanks; k++) if (boundsOut[k] < boundsOut[k - 1] + minRows) boundsOut[k] = boundsOut[k - 1] + minRows;
    boundsOut[nRanks] = frameHeight;
    for (uint32_t k = nRanks; k-- > 1; ) if (boundsOut[k] + minRows > boundsOut[k + 1]) boundsOut[k] = boundsOut[k + 1] - minRows;
    // 3. hysteresis: a partition the profile does not expect to beat the measured one by 3 % is not worth the chain strips and the occlusion history a moved band loses
    double maxPredicted = 0.0;
    for (uint32_t r = 0; r < nRanks; r++) maxPredicted = std::max(maxPredicted, cum[boundsOut[r + 1] / align] - cum[boundsOut[r] / align]);
    if (maxPredicted > 0.97 * maxMs) for (uint32_t k = 0; k <= nRanks; k++) boundsOut[k] = boundsIn[k];
    return 0;
}

int brmi_compose_submit(brmi_composer* c, const void* surface, brmi_compose_stream renderStream) {
    if (!c || !surface) return -1;
    if (!c->staging) return fail(c, -4, "brmi_compose_submit: call brmi_compose_bind first");
    hipStream_t rs = static_cast<hipStream_t>(renderStream);
    const uint32_t slot = (uint32_t)(c->frames % c->cfg.depth);
    if (c->peerWrite) {
        if (!c->imported && c->cfg.nRanks > 1) return fail(c, -4, "brmi_compose_submit: call brmi_compose_import first");
        const uint32_t frame = (uint32_t)(c->frames + 1u), n = c->cfg.nRanks;
        const unsigned long long ticks = (unsigned long long)(c->cfg.waitTimeoutMs ? c->cfg.waitTimeoutMs : 2000u) * 100000ull;
        const uint8_t* band = static_cast<const uint8_t*>(surface) + c->bandOffset;
        // 1. every peer learns that this rank is at `frame` (its copies of frame - depth are free); 2. this rank waits until every peer is
        // there too, i.e. until the slot it is about to overwrite in THEIR images is free; 3. the band; 4. "landed"
        hipLaunchKernelGGL(k_signal_submitted, dim3(1), dim3(64), 0, rs, c->peers, n, c->cfg.rank, frame);
        if (n > 1) hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, rs, c->flags->submitted, n, c->cfg.rank, frame, &c->flags->status, ticks);
        const uint64_t slotOffset = (uint64_t)slot * c->outputBytes, bandOut = c->bandOut();
        if (c->pixels == 0) { /* (a rank that owns no row of this frame still signals) */ }
        else if (c->cfg.transport == BRMI_TRANSPORT_RGB16F) {
            const uint64_t quads = c->pixels / 4u;
            hipLaunchKernelGGL(k_peer_write<true>, dim3((unsigned)std::min<uint64_t>(4096, (quads + 255) / 256)), dim3(256), 0, rs, reinterpret_cast<const uint4*>(band), c->peers, n, slotOffset, bandOut, quads);
        } else {
            const uint64_t vecs = c->bandBytes / 16u;
            hipLaunchKernelGGL(k_peer_write<false>, dim3((unsigned)std::min<uint64_t>(4096, (vecs + 255) / 256)), dim3(256), 0, rs, reinterpret_cast<const uint4*>(band), c->peers, n, slotOffset, bandOut, vecs);
        }
        hipLaunchKernelGGL(k_signal_landed, dim3(1), dim3(64), 0, rs, c->peers, n, c->cfg.rank, slot, frame);
        CHECK_HIP(c, hipGetLastError());
        c->frames++;
        return (int)slot;
    }
    uint8_t* st = c->staging + (uint64_t)slot * c->stagingBytes; uint8_t* dst = c->output + (uint64_t)slot * c->outputBytes;
    const uint8_t* band = static_cast<const uint8_t*>(surface) + c->bandOffset;
    if (c->inFlight[slot]) CHECK_HIP(c, hipStreamWaitEvent(rs, c->done[slot], 0));       // the slot's previous collective has read the staging buffer
    if (c->pixels == 0) { /* (dynamic bands: this rank owns no row of the frame; it still takes part in the collective) */ }
    else if (c->cfg.transport == BRMI_TRANSPORT_RGB16F) {
        const uint64_t quads = c->pixels / 4u;                                            // a band is whole 8x8 tiles
        hipLaunchKernelGGL(k_pack_rgb16f, dim3((unsigned)std::min<uint64_t>(4096, (quads + 255) / 256)), dim3(256), 0, rs, reinterpret_cast<const uint4*>(band), reinterpret_cast<uint2*>(st), quads);
        CHECK_HIP(c, hipGetLastError());
    } else CHECK_HIP(c, hipMemcpyAsync(st, band, c->bandBytes, hipMemcpyDeviceToDevice, rs));
    CHECK_HIP(c, hipEventRecord(c->staged[slot], rs));
    CHECK_HIP(c, hipStreamWaitEvent(c->collStream, c->staged[slot], 0));
    if (c->dynamic) {
        // bands of unequal height: one broadcast per rank, each with its own byte count, issued as ONE group (RCCL fuses a group into one launch; on the xGMI mesh every
        // root's band leaves over its own links, as in the all-gather)
        if (c->cfg.nRanks > 1 && c->bounds[c->cfg.nRanks] == 0u) return fail(c, -4, "brmi_compose_submit: call brmi_compose_set_bounds first (the composer was created with frameHeight)");
        CHECK_NCCL(c, ncclGroupStart());
        for (uint32_t r = 0; r < c->cfg.nRanks; r++) {
            const uint32_t y0 = c->cfg.nRanks > 1 ? c->bounds[r] : c->cfg.bandY0, y1 = c->cfg.nRanks > 1 ? c->bounds[r + 1] : c->cfg.bandY1;
            const uint64_t off = (uint64_t)(y0 / 8u) * c->rowBytesOut, count = (uint64_t)((y1 - y0) / 8u) * c->rowBytesOut;
            if (count == 0) continue;
            CHECK_NCCL(c, ncclBroadcast(r == c->cfg.rank ? (const void*)st : (const void*)(dst + off), dst + off, count, ncclUint8, (int)r, c->comm, c->collStream));
        }
        CHECK_NCCL(c, ncclGroupEnd());
    } else
    CHECK_NCCL(c, ncclAllGather(st, dst, c->stagingBytes, ncclUint8, c->comm, c->collStream));
    CHECK_HIP(c, hipEventRecord(c->done[slot], c->collStream));
    c->inFlight[slot] = true;
    c->frames++;
    return (int)slot;
}

int brmi_compose_submit_rows(brmi_composer* c, const void* surface, uint32_t row0, uint32_t row1, brmi_compose_stream renderStream) {
    if (!c || !surface) return -1;
    if (!c->peerWrite) return fail(c, -4, "brmi_compose_submit_rows: peer-write composers only (the all-gather moves whole bands: brmi_compose_submit)");
    if (!c->staging) return fail(c, -4, "brmi_compose_submit_rows: call brmi_compose_alloc_shared first");
    if (!c->imported && c->cfg.nRanks > 1) return fail(c, -4, "brmi_compose_submit_rows: call brmi_compose_import first");
    const uint32_t y0 = c->cfg.bandY0, y1 = c->cfg.bandY1;
    const uint32_t expect = c->openRow ? c->openRow : y0;
    if (row0 % 8u || row1 % 8u || row0 >= row1 || row0 != expect || row1 > y1)
        return fail(c, -1, "brmi_compose_submit_rows: rows [%u, %u) -- slabs are multiples of 8, ascending from %u and inside the band [%u, %u)", row0, row1, expect, y0, y1);
    hipStream_t rs = static_cast<hipStream_t>(renderStream), cs = c->collStream;
    const uint32_t slot = (uint32_t)(c->frames % c->cfg.depth), frame = (uint32_t)(c->frames + 1u), n = c->cfg.nRanks;
    const unsigned long long ticks = (unsigned long long)(c->cfg.waitTimeoutMs ? c->cfg.waitTimeoutMs : 2000u) * 100000ull;
    if (!c->slabReady) CHECK_HIP(c, hipEventCreateWithFlags(&c->slabReady, hipEventDisableTiming));
    if (!c->ownStores) CHECK_HIP(c, hipEventCreateWithFlags(&c->ownStores, hipEventDisableTiming));
    // the slab's rows are shaded by what is already enqueued on the render stream: the composer's stream follows it, the render stream goes on
    CHECK_HIP(c, hipEventRecord(c->slabReady, rs));
    CHECK_HIP(c, hipStreamWaitEvent(cs, c->slabReady, 0));
    if (row0 == y0) {      // the frame opens: peers learn that this rank is at `frame`; the slot it overwrites in THEIR images must be free
        hipLaunchKernelGGL(k_signal_submitted, dim3(1), dim3(64), 0, cs, c->peers, n, c->cfg.rank, frame);
        if (n > 1) hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, cs, c->flags->submitted, n, c->cfg.rank, frame, &c->flags->status, ticks);
    }
    const uint64_t tilesX = (c->cfg.width + 7u) / 8u, rowBytes = tilesX * 64u * c->cfg.bytesPerPixel;      // one 8-row tile row of the surface
    const uint64_t slabIn = (uint64_t)((row0 - y0) / 8u) * rowBytes, slabBytes = (uint64_t)((row1 - row0) / 8u) * rowBytes;
    const uint8_t* src = static_cast<const uint8_t*>(surface) + c->bandOffset + slabIn;
    const uint64_t slotOffset = (uint64_t)slot * c->outputBytes;
    if (c->cfg.transport == BRMI_TRANSPORT_RGB16F) {
        const uint64_t quads = slabBytes / c->cfg.bytesPerPixel / 4u, bandOut = c->bandOut() + slabIn / c->cfg.bytesPerPixel * 6u;
        hipLaunchKernelGGL(k_peer_write<true>, dim3((unsigned)std::min<uint64_t>(2048, (quads + 255) / 256)), dim3(256), 0, cs, reinterpret_cast<const uint4*>(src), c->peers, n, slotOffset, bandOut, quads);
    } else {
        const uint64_t vecs = slabBytes / 16u, bandOut = c->bandOut() + slabIn;
        hipLaunchKernelGGL(k_peer_write<false>, dim3((unsigned)std::min<uint64_t>(2048, (vecs + 255) / 256)), dim3(256), 0, cs, reinterpret_cast<const uint4*>(src), c->peers, n, slotOffset, bandOut, vecs);
    }
    {   // the composer's stream has read these rows of `surface` once this event has passed (brmi_compose_wait_source)
        brmi_composer::SourceRead* sr = nullptr;
        for (auto& e : c->sourceReads) if (e.surface == surface) sr = &e;
        if (!sr) {
            if (c->sourceReads.size() >= 16) return fail(c, -3, "brmi_compose_submit_rows: more than 16 different source surfaces");
            hipEvent_t ev = nullptr; CHECK_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->sourceReads.push_back({surface, ev}); sr = &c->sourceReads.back();
        }
        CHECK_HIP(c, hipEventRecord(sr->read, cs));
    }
    if (row1 == y1) {      // the frame closes
        hipLaunchKernelGGL(k_signal_landed, dim3(1), dim3(64), 0, cs, c->peers, n, c->cfg.rank, slot, frame);
        CHECK_HIP(c, hipEventRecord(c->ownStores, cs)); c->ownStoresRecorded = true;
        c->openRow = 0; c->frames++;
    } else c->openRow = row1;
    CHECK_HIP(c, hipGetLastError());
    return (int)slot;
}

int brmi_compose_wait_source(brmi_composer* c, const void* surface, brmi_compose_stream stream) {
    if (!c || !surface) return -1;
    // (brmi_compose_submit reads the surface on the caller's own render stream: stream order covers it, nothing was recorded, nothing to wait for)
    for (auto& e : c->sourceReads) if (e.surface == surface) CHECK_HIP(c, hipStreamWaitEvent(static_cast<hipStream_t>(stream), e.read, 0));
    return 0;
}

int brmi_compose_finish(brmi_composer* c, brmi_compose_stream stream, void** composed) {
    if (!c) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->peerWrite) {
        if (c->openRow) return fail(c, -4, "brmi_compose_finish: a frame is open (brmi_compose_submit_rows up to row %u so far)", c->openRow);
        if (c->ownStoresRecorded) CHECK_HIP(c, hipStreamWaitEvent(s, c->ownStores, 0));      // the rank's own slabs went through the composer's stream
        if (c->frames && c->cfg.nRanks > 1) {
            const uint32_t slot = (uint32_t)((c->frames - 1) % c->cfg.depth);
            const unsigned long long ticks = (unsigned long long)(c->cfg.waitTimeoutMs ? c->cfg.waitTimeoutMs : 2000u) * 100000ull;
            // (its own band was stored by a kernel already on the rank's render stream; the caller orders `stream` behind that one as with the all-gather)
            hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, s, c->flags->landed[slot], c->cfg.nRanks, c->cfg.rank, (uint32_t)c->frames, &c->flags->status, ticks);
            CHECK_HIP(c, hipGetLastError());
        }
        if (composed) *composed = c->frames ? c->output + (uint64_t)((c->frames - 1) % c->cfg.depth) * c->outputBytes : nullptr;
        return 0;
    }
    for (uint32_t i = 0; i < c->cfg.depth; i++) if (c->inFlight[i]) { CHECK_HIP(c, hipStreamWaitEvent(s, c->done[i], 0)); c->inFlight[i] = false; }
    if (composed) *composed = c->frames ? c->output + (uint64_t)((c->frames - 1) % c->cfg.depth) * c->outputBytes : nullptr;
    return 0;
}

void brmi_compose_destroy(brmi_composer* c) {
    if (!c) return;
    if (c->collStream) (void)hipStreamSynchronize(c->collStream);
    for (hipEvent_t e : c->staged) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
    if (c->slabReady) (void)hipEventDestroy(c->slabReady);
    if (c->ownStores) (void)hipEventDestroy(c->ownStores);
    for (auto& e : c->sourceReads) if (e.read) (void)hipEventDestroy(e.read);
    for (void* p : c->opened) (void)hipIpcCloseMemHandle(p);
    if (c->ownsShared) { (void)hipDeviceSynchronize(); (void)hipFree(c->output); (void)hipFree(c->flags); }
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->collStream) (void)hipStreamDestroy(c->collStream);
    delete c;
}

const char* brmi_compose_last_error(const brmi_composer* c) { return c ? c->err.c_str() : "null composer"; }

}  // extern "C"
