// brmi_raster.hip -- visibility-buffer clear, software rasteriser (K5) and depth copy (K6) for gfx950.
//
// Computes what SWRasterCluster does (BR/shaders/ClusterLOD/softwareRaster.hlsl:290-612): vertices
// of one meshlet to screen space, per-triangle edge-function setup, inclusive coverage at pixel
// centres with incrementally stepped barycentrics, 64-bit min of (depth | cluster | triangle).
// Scheduling is MI355X-first rather than the reference's one-128-thread-group-per-cluster indirect
// dispatch per raster bucket:
//   * k_raster: one wave64 per cluster (static round-robin; cluster count lives in HBM: no indirect
//     dispatch, no host read-back).  lane = triangle, two passes over a 128-triangle meshlet.  The
//     reference's WaveActiveAnyTrue(rectWidth > 4) vote (softwareRaster.hlsl:502) is evaluated per pass
//     over the lanes that survived setup, which is exactly the wave composition of a 128-thread group on
//     wave64 hardware.  The meshlet's screen-space vertices (<= 128 x 12 B) are staged once in LDS.
//     Triangles with a small clamped bounding box are walked by their lane with global 64-bit atomic-min.
//   * Sort-middle for everything larger: 64-bit global atomics are element-serialised in L2 (~100 G/s on
//     MI355X, measured), so a frame with 3x overdraw of large triangles spends its time there.  Larger
//     triangles are cut into records of <= 16 rows and appended to screen bins (256 px x 16 rows).
//     k_raster_bins is a pool of workgroups that take (bin, slice) items, longest first, from a plan the launch before wrote
//     (k_raster_overflow's last workgroup); a slice's records are counting-sorted by rows and row length (4, 8 or 16 lanes per
//     record), walked with LDS atomic-min into a 32 KB tile of keys, and the tile is merged into the visibility buffer once with
//     plain coalesced 64 B loads / stores (a bin one workgroup walks is exclusively owned; the slices of a larger bin park their
//     tiles and the last one folds them).  Phase 2 of a frame, while it draws little, skips the bins altogether.
//   * Per-pixel arithmetic is untouched by the re-scheduling: a lane that enters a row in the middle
//     (bin strip) steps the barycentrics pixel by pixel from the row start exactly as the serial loop does.
//   * the visibility surface is stored in 8x8 tiles (512 B = 4 cache lines), column-major inside the tile, so
//     lanes that own neighbouring rows hit the same cache line and every later full-screen pass touches
//     whole lines.
// Raster buckets (K4) collapse: with one PSO-free kernel there is nothing to sort by.
#include <algorithm>

#include "brmi_device.h"
#include "brmi_internal.h"
#include "brmi_texture.h"

namespace brmi {

// One record = up to BIN_ROWS rows of one triangle inside one bin band; it is appended to every bin strip its box overlaps.
struct BinRecord {           // 64 B
    uint32_t clusterIndex, triAndFlags;      // tri | useScanlineRanges << 8 | rowCount << 16
    int32_t  minX, rectWidth, rowStart;
    float    sb0, sb1, dx_b0, dx_b1, dy_b0, dy_b1, d0, d1, d2;   // barycentrics at (minX, rowStart), their steps, vertex depths
    uint32_t pad0, pad1;                      // pad0: strip (overflow queue only); pad1 != 0: alpha tested, its AlphaRecord sits at the same index of the side array
};
static_assert(sizeof(BinRecord) == 64, "one cache line");
// what the alpha test of a binned triangle needs besides the record (softwareRaster.hlsl:525-540): only written / read for
// clusters whose material is alpha tested, in scenes that have such materials
struct AlphaRecord { AlphaTri tri; uint32_t materialDataIndex, pad[2]; };     // 48 B
static_assert(sizeof(AlphaRecord) == 48, "AlphaRecord layout");
constexpr int BIN_W = 256, BIN_ROWS = 16;          // bin = 4096 keys = 32 KB of LDS
constexpr int BIN_W_SHIFT = 8, BIN_ROWS_SHIFT = 4;
// a bin's record counter has a 128 B line to itself: atomics on ONE cache line serialise at 50-90 per microsecond whatever their addresses, and
// the bins of a screen band (15 neighbours in one line, the horizon's among them) take thousands of slot reservations per frame
#ifndef BRMI_CHAIN_DIRTY_BLOCKS
#define BRMI_CHAIN_DIRTY_BLOCKS 1      // phase 2 records the 32 x 32 px blocks it may touch; the second depth-chain build redoes only those
#endif
constexpr uint32_t BIN_COUNT_STRIDE = BRMI_BIN_COUNT_STRIDE;
constexpr int BIN_WINDOW = 256;                     // bins a wave counts in LDS with its reservations held in registers (cells of its bin bounding box)
constexpr int COOP_ENTRIES = 64;                    // triangles with more bin entries than this are emitted by the whole wave ...
#ifndef BRMI_COOP_ENTRIES_TABLE
#define BRMI_COOP_ENTRIES_TABLE 512
#endif
constexpr int COOP_ENTRIES_TABLE = BRMI_COOP_ENTRIES_TABLE;             // ... or than this, when the launch has the wide count table (RasterArgs::tableCells > BIN_WINDOW)
constexpr uint32_t BIN_TABLE_MAX = 2048;            // cells of the wide table (dynamic LDS, 8 KB: every bin of a 4K frame or of a rank's 7680 x 1088 surface)

// Round 6: a triangle whose records are emitted by a workgroup of k_raster_wide instead of by the wave that set it up.  A triangle of a near surface reaches thousands of
// bins (a ground plane in the 8-GPU weak frame: 80 bin bands x 30 strips); one wave emitting them one band per lane was 200 us of a rank's frame and what the moving
// camera's phase 1 waited for.  `base` is the record of the triangle's first row INSIDE this GPU's rows (rowStart = that frame row, barycentrics
// stepped to it by the serial loop's additions); rows, bands and strips say where it goes.
struct WideTri { BinRecord base; int32_t yHi, band0, band1, strip0, strip1; uint32_t pad[3]; };      // 96 B
static_assert(sizeof(WideTri) == 96, "WideTri layout");
struct RasterArgs {
    WideTri* wideQueue; AlphaRecord* wideAlpha; uint32_t wideCapacity, wideCounter;      // wideQueue null: every triangle is emitted by its own wave; wideCounter: CNT_WIDE* of this launch
    uint32_t wideEntries;        // with the queue on: triangles of more bin entries than this are queued (the lane-by-lane emission keeps the smaller ones)
    uint32_t* wideFeedback;      // host-mapped word or null: the plan stores the launch's count of such triangles there (the host launches the wide pass by it)
    BinRecord* binRecords; uint32_t* binCounts; uint32_t binCapacity, binsX, binsY;
    BinRecord* overflow; uint32_t overflowPerStripe;     // 64 striped queues of records whose bin was full (pad0 = strip)
    AlphaRecord* binAlpha; AlphaRecord* overflowAlpha;   // side arrays of binRecords / overflow (alpha-tested scenes only)
    const ClusterUv* clusterUv;
    const AlphaMaterial* alphaMats;                      // per material: the alpha test's texture bindings resolved (k_frame_constants)
    const float* objConst;   // per object: MVP (16), objectToClip (16), modelViewZ (4)
    int bigTriArea;          // clamped-bbox pixels above which a triangle is binned
    int bigTriAreaDense; uint32_t denseClusterCount;     // ... on frames with at least this many visible clusters
    int bigTriAreaAlpha;     // the same for alpha-tested clusters: their pixels are far cheaper in a bin (LDS early-out, more lanes in flight)
    uint32_t binMinSlice;    // most records one workgroup walks alone: a bin with more is cut into slices (BRMI_BIN_MIN_SLICE)
    uint32_t binSharedSlice; // records per slice of such a bin (BRMI_BIN_SHARED_SLICE)
    // the plan of a k_raster_bins launch (plan_bins): header {itemCount, ticket}, per bin {records, first scratch tile, slices done}, the work items
    uint32_t tableCells;     // k_raster: words of dynamic LDS behind the launch (>= BIN_WINDOW): the LDS window a wave counts its records per bin in
    uint32_t* binPlan; uint32_t* binItems; unsigned long long* binScratch; uint32_t binScratchTiles, binItemCapacity;
    int debugFlags;          // experiments only (BRMI_RASTER_DEBUG): 1 = skip the direct walk, 2 = skip bin emission, 4 = skip the bin pass, 8 = direct walk without the atomic
    brmi_scene_buffers sc;
    const uint4* clusters; const ClusterSetup* setup;
    uint32_t* counters;
    uint8_t* chainDirty; uint32_t chainBlocksX;       // phase 2 only (else null): a byte per 32 x 32 px block its triangles may touch (byte 0: all, blocks from byte 4), for the second depth-chain build
    uint32_t firstCounter, countCounter;   // counter indices: first cluster (0xFFFFFFFF = 0) and cluster count
    const uint32_t* drawList;              // round 6: null = clusters first .. first + count of the visible list; else the `count` cluster indices to rasterise (draw list, late list)
    uint32_t* countFeedback;               // host-mapped word or null: the launch's cluster count, for the sizes of the frames that follow
    // round 6, k_raster<false, true> (the lean form): clusters it leaves to the general kernel -- a binned triangle, skinned vertices -- are named here
    uint32_t* generalList; uint32_t generalCounter; uint32_t* generalFeedback;
    // ... and its triangles large enough for the bins are queued (WideTri) for k_raster_emit; a full queue sends the cluster to the general list instead
    // The queue is 64 stripes (a wave appends to stripe blockIdx & 63: one head per 128 B line -- a single head serialises the launch's appends at ~100 per microsecond),
    // each `bigCapacity` entries and as many runs: a run is what one wave appended of one pass, {first entry, count}, and k_raster_emit takes a run per wave and step, so
    // the 64 triangles a wave emits are one cluster's, as in k_raster.  Head of stripe s: the 64-bit word at counters[bigCounter + 32 s], entries | runs << 32.
    WideTri* bigQueue; uint2* bigRuns; uint32_t bigCapacity, bigCounter, emitWideEntries;
    unsigned long long* vis;
    uint32_t visW, visH, tilesX, bandY0, bandY1;      // visW x visH: the FRAME (scissor clamp); bandY0 / bandY1: rows of the surface this GPU renders (records live in surface rows)
    uint32_t rowLo, rowHi;                            // frame rows k_raster looks at: the band, or the whole frame with the interleaved partition ...
    StripeMap stripes;                                // ... whose ownership test and frame row -> surface row mapping this is
    unsigned long long* debugStamps;                     // instrumented builds only
};

// `frameState` (may be null): the counters + survivor bitmasks block that the culling pass clears at the start of a frame; brmi_execute
// has this kernel clear it too, one launch instead of a kernel and a fill.
__global__ void __launch_bounds__(256) k_clear_vis(unsigned long long* vis, uint64_t n, uint4* frameState, uint64_t frameState16) {
    // 16 B per lane per store (n is a multiple of 64)
    ulonglong2* v2 = reinterpret_cast<ulonglong2*>(vis);
    const uint64_t n2 = n >> 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) v2[i] = make_ulonglong2(BRMI_VIS_EMPTY, BRMI_VIS_EMPTY);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frameState16; i += (uint64_t)gridDim.x * blockDim.x) frameState[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ClipScanlineRange (softwareRaster.hlsl:262-288).  The shader divides -value / step for a rising edge and value / -step for a falling one; IEEE
// division is sign-symmetric, so both are the same number: ONE correctly rounded division per edge whatever the lanes' signs (as two branches a
// wave whose lanes disagree about the sign ran both: six divisions per row, half of what a binned row of average length costs).
BRMI_DEV void clip_scanline(float value, float step, int& first, int& last, bool& has) {
    const float q = -value / step;                               // step == 0: not used
    const int c = to_int_sat(ceilf(q)), f = to_int_sat(floorf(q));
    if (step > 0.0f) first = first > c ? first : c;
    if (step < 0.0f) last = last < f ? last : f;
    has = has && (step != 0.0f || value >= 0.0f) && first <= last;
}

// Where a key goes: the visibility buffer (tiled, 64-bit atomic min in L2) or the LDS tile of a bin.
struct GlobalSink {
    static constexpr bool kPeekCheap = false;
    unsigned long long* vis; uint32_t tilesX; int dbg;
    BRMI_DEV void operator()(int px, int py, unsigned long long key) const {
        if (dbg & 8) { if (key == 0x1234567ull) vis[0] = key; return; }
        atomicMin(&vis[tiled_index((uint32_t)px, (uint32_t)py, tilesX)], key);
    }
    BRMI_DEV unsigned long long peek(int px, int py) const { return __builtin_nontemporal_load(&vis[tiled_index((uint32_t)px, (uint32_t)py, tilesX)]); }
};
struct LdsSink {
    static constexpr bool kPeekCheap = true;
    unsigned long long* tile; int x0, y0;       // tile[(px - x0) * BIN_ROWS + (py - y0)]
    BRMI_DEV void operator()(int px, int py, unsigned long long key) const { atomicMin(&tile[(px - x0) * BIN_ROWS + (py - y0)], key); }
    BRMI_DEV unsigned long long peek(int px, int py) const { return *(volatile const unsigned long long*)&tile[(px - x0) * BIN_ROWS + (py - y0)]; }
};

// The per-pixel alpha test: nothing for the plain kernels, SWAlphaTestFailed on the pixel's texcoord for an alpha-tested record.
// kEarlyZ: a key that cannot win the pixel's 64-bit min is dropped before the (expensive) test -- the test's outcome could not
// change the buffer either way.
struct NoAlpha { static constexpr bool kEarlyZ = false; BRMI_DEV bool operator()(float, float, float) const { return false; } };
struct TexAlpha {
    static constexpr bool kEarlyZ = true;
    const float* unorm; AlphaMaterial mat; AlphaTri tri;
    BRMI_DEV bool operator()(float b0, float b1, float b2) const { return alpha_test_failed(unorm, mat, pixel_texcoord(tri, b0, b1, b2)); }
};

// An alpha-tested pixel.  LDS tile: look first (a few cycles), sample only if the key can still win.  Visibility buffer in HBM: the
// look is as slow as the texel fetches, so both are requested together and the pixel costs one memory round trip, not two.
template <typename Sink, typename Alpha>
BRMI_DEV void emit_tested(const Sink& sink, const Alpha& alphaFails, int px, int py, unsigned long long key, float b0, float b1, float b2) {
    if (Sink::kPeekCheap) { if (key < sink.peek(px, py) && !alphaFails(b0, b1, b2)) sink(px, py, key); }
    else {
        const unsigned long long cur = sink.peek(px, py);
        const bool fails = alphaFails(b0, b1, b2);
        if (key < cur && !fails) sink(px, py, key);
    }
}

// n of the serial loop's steps `b0 += dx_b0; b1 += dx_b1` (a lane that enters a row in the middle): two dependent chains of additions in the
// loop's own order, eight steps per trip so that the trip's bookkeeping does not cost more than the additions
BRMI_DEV void step_barycentrics(float& b0, float& b1, float dx_b0, float dx_b1, int n) {
    for (; n >= 8; n -= 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) { b0 += dx_b0; b1 += dx_b1; }
    }
    for (; n > 0; n--) { b0 += dx_b0; b1 += dx_b1; }
}

// One scanline of one triangle (softwareRaster.hlsl:506-609), barycentrics at the row start given, restricted to
// pixels clipX0..clipX1.  The barycentrics are stepped pixel by pixel from the row's first pixel even when the walk
// starts further right, so every value is the one the serial loop produces.
template <typename Sink, typename Alpha>
BRMI_DEV void raster_row(const Sink& sink, const Alpha& alphaFails, int py, int minX, int rectWidth, bool useScanlineRanges, float sb0, float sb1,
                         float dx_b0, float dx_b1, float dx_b2, float d0, float d1, float d2, uint32_t clusterIndex, uint32_t t, int clipX0, int clipX1) {
    if (useScanlineRanges) {
        const float sb2 = 1.0f - sb0 - sb1;
        int firstOff = 0, lastOff = rectWidth - 1; bool has = true;
        clip_scanline(sb0, dx_b0, firstOff, lastOff, has);
        clip_scanline(sb1, dx_b1, firstOff, lastOff, has);
        clip_scanline(sb2, dx_b2, firstOff, lastOff, has);
        if (has) {
            float b0 = sb0 + (float)firstOff * dx_b0, b1 = sb1 + (float)firstOff * dx_b1;
            int x0 = minX + firstOff;
            const int x1 = min(minX + lastOff, clipX1);
            if (x0 < clipX0) { step_barycentrics(b0, b1, dx_b0, dx_b1, clipX0 - x0); x0 = clipX0; }
            for (int px = x0; px <= x1; px++) {
                const float b2 = 1.0f - b0 - b1;
                const float depth = b0 * d0 + b1 * d1 + b2 * d2;
                const unsigned long long key = (unsigned long long)pack_vis_key(depth, clusterIndex, t);
                if (Alpha::kEarlyZ) emit_tested(sink, alphaFails, px, py, key, b0, b1, b2);
                else sink(px, py, key);
                b0 += dx_b0; b1 += dx_b1;
            }
        }
    } else {
        float b0 = sb0, b1 = sb1;
        int x0 = minX;
        const int x1 = min(minX + rectWidth - 1, clipX1);
        if (x0 < clipX0) { step_barycentrics(b0, b1, dx_b0, dx_b1, clipX0 - x0); x0 = clipX0; }
        for (int px = x0; px <= x1; px++) {
            const float b2 = 1.0f - b0 - b1;
            if (b0 >= 0.0f && b1 >= 0.0f && b2 >= 0.0f) {
                const float depth = b0 * d0 + b1 * d1 + b2 * d2;
                const unsigned long long key = (unsigned long long)pack_vis_key(depth, clusterIndex, t);
                if (Alpha::kEarlyZ) emit_tested(sink, alphaFails, px, py, key, b0, b1, b2);
                else sink(px, py, key);
            }
            b0 += dx_b0; b1 += dx_b1;
        }
    }
}

// The prologue of raster_row on its own: where a clipped walk of one scanline starts (barycentrics stepped to pixel `px` exactly as the serial
// loop steps them), where it ends, and whether every pixel in between is covered (the analytic scanline range) or each one is tested.
struct SegWalk { float b0, b1; int px, x1; bool all; };
BRMI_DEV SegWalk seg_begin(int minX, int rectWidth, bool useScanlineRanges, float sb0, float sb1, float dx_b0, float dx_b1, float dx_b2, int clipX0, int clipX1) {
    SegWalk w; w.all = useScanlineRanges; w.b0 = sb0; w.b1 = sb1;
    int x0 = minX; w.x1 = min(minX + rectWidth - 1, clipX1);
    if (useScanlineRanges) {
        const float sb2 = 1.0f - sb0 - sb1;
        int firstOff = 0, lastOff = rectWidth - 1; bool has = true;
        clip_scanline(sb0, dx_b0, firstOff, lastOff, has);
        clip_scanline(sb1, dx_b1, firstOff, lastOff, has);
        clip_scanline(sb2, dx_b2, firstOff, lastOff, has);
        if (!has) { w.px = 0; w.x1 = -1; return w; }
        w.b0 = sb0 + (float)firstOff * dx_b0; w.b1 = sb1 + (float)firstOff * dx_b1;
        x0 = minX + firstOff; w.x1 = min(minX + lastOff, clipX1);
    }
    if (x0 < clipX0) { step_barycentrics(w.b0, w.b1, dx_b0, dx_b1, clipX0 - x0); x0 = clipX0; }
    w.px = x0;
    return w;
}

// Stores one record at a reserved slot of a bin; when the bin is full its rows are rasterised here with global atomics (counted).
template <typename Alpha>
BRMI_DEV void raster_record_global(const RasterArgs& a, const BinRecord& r, const Alpha& alpha, uint32_t strip, uint32_t firstRow, uint32_t rowStep) {
    const GlobalSink sink{a.vis, a.tilesX, 0};
    const uint32_t n = (r.triAndFlags >> 16) & 0xFFu;
    float sb0 = r.sb0, sb1 = r.sb1;
    uint32_t k = 0;
    for (uint32_t row = firstRow; row < n; row += rowStep) {
        for (; k < row; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
        const int py = r.rowStart + (int)row;
        if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1)
            raster_row(sink, alpha, py, r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), r.d0, r.d1, r.d2, r.clusterIndex, r.triAndFlags & 0x7Fu,
                       (int)(strip << BIN_W_SHIFT), (int)(strip << BIN_W_SHIFT) + BIN_W - 1);
    }
}
BRMI_DEV TexAlpha tex_alpha_of(const RasterArgs& a, const float* unorm, const AlphaRecord& ar) { return TexAlpha{unorm, a.alphaMats[ar.materialDataIndex], ar.tri}; }

// Stores one record at a reserved slot of a bin.  A full bin sends the record to the overflow queue of the wave's stripe
// (k_raster_overflow walks those row-parallel with global atomics); a full queue rasterises it right here.
// r.pad1 != 0: the record is alpha tested and its AlphaRecord `ar` travels with it.
BRMI_DEV void bin_store(const RasterArgs& a, const float* unorm, const BinRecord& r, const AlphaRecord& ar, uint32_t strip, uint32_t band, uint32_t slot) {
    const bool alpha = r.pad1 != 0u;
    const uint32_t bin = band * a.binsX + strip;
    if (slot < a.binCapacity) {
        a.binRecords[(size_t)bin * a.binCapacity + slot] = r;
        if (alpha) a.binAlpha[(size_t)bin * a.binCapacity + slot] = ar;
        return;
    }
    const uint32_t stripe = blockIdx.x & (CNT_STRIPE_COUNT - 1u);
    const uint32_t q = atomicAdd(&a.counters[CNT_STRIPES + stripe * CNT_STRIPE_WORDS + STRIPE_OVERFLOW], 1u);
    if (q < a.overflowPerStripe) {
        BinRecord o = r; o.pad0 = strip; a.overflow[(size_t)stripe * a.overflowPerStripe + q] = o;
        if (alpha) a.overflowAlpha[(size_t)stripe * a.overflowPerStripe + q] = ar;
        return;
    }
    if (alpha) raster_record_global(a, r, tex_alpha_of(a, unorm, ar), strip, 0u, 1u);
    else raster_record_global(a, r, NoAlpha{}, strip, 0u, 1u);
}
BRMI_DEV void bin_append(const RasterArgs& a, const float* unorm, const BinRecord& r, const AlphaRecord& ar, uint32_t strip, uint32_t band) {
    bin_store(a, unorm, r, ar, strip, band, atomicAdd(&a.binCounts[(size_t)(band * a.binsX + strip) * BIN_COUNT_STRIDE], 1u));
}

// three waves per SIMD (<= 168 VGPRs; the unconstrained kernel takes 181 and runs two): Bistro raster 122 -> 115 us, San Miguel 192 -> 182 us;
// four (<= 128 VGPRs) spills and loses it again
#ifndef BRMI_RASTER_WAVES
#define BRMI_RASTER_WAVES 3
#endif
// ALPHA: the scene has alpha-tested materials; clusters of such a material (BRMI_CS_ALPHA) also stage 1/w and the texcoord of their
// vertices and test every covered pixel.  Scenes without them run the plain instantiation (no extra registers or LDS).
#ifndef BRMI_RASTER_ALPHA_WAVES
#define BRMI_RASTER_ALPHA_WAVES 2
#endif
#ifndef BRMI_RASTER_SPLIT_FIRST
#define BRMI_RASTER_SPLIT_FIRST 2u
#endif
// LEAN (round 6): the kernel for frames of very many clusters of small triangles (the Zorah-class frame: 284 k clusters in the draw list, 62 M triangles for 33 M pixels).
// Such a launch is a queue of per-cluster round-trip chains (record -> vertices -> setup -> atomics), and what hides them is waves per SIMD: the general kernel holds
// 132 registers for paths these clusters never take (bin windows, the cooperative emission, alpha, skinning, the interleaved partition) and runs three.  The lean
// instantiation compiles those paths out and runs BRMI_RASTER_LEAN_WAVES; a cluster that needs one of them (a triangle large enough for the bins, skinned vertices)
// is put on `generalList` and drawn whole by a general launch behind this one -- keys are order-free and idempotent, so what the lean wave had already written of
// it is written again, nothing else.  Same arithmetic per vertex, triangle and pixel: the code below is the same code.
#ifndef BRMI_RASTER_LEAN_WAVES
#define BRMI_RASTER_LEAN_WAVES 6
#endif
template <bool ALPHA, bool LEAN = false>
__global__ void __launch_bounds__(64, LEAN ? BRMI_RASTER_LEAN_WAVES : (ALPHA ? BRMI_RASTER_ALPHA_WAVES : BRMI_RASTER_WAVES)) k_raster(RasterArgs a) {
    static_assert(!(ALPHA && LEAN), "the lean form has no alpha test");
    wave_prio<PRIO_RASTER>();
    // (one wave per workgroup: LDS hand-offs between its lanes need wave_lds_sync() only.  __syncthreads() also waits for every global store
    // and atomic the wave has in flight -- the record stores and the small boxes' atomic-mins: a memory round trip per hand-off.)
    __shared__ float sx[BRMI_MESHLET_MAX_VERTS], sy[BRMI_MESHLET_MAX_VERTS], sd[BRMI_MESHLET_MAX_VERTS];
    __shared__ float siw[ALPHA ? BRMI_MESHLET_MAX_VERTS : 1], su[ALPHA ? BRMI_MESHLET_MAX_VERTS : 1], sv[ALPHA ? BRMI_MESHLET_MAX_VERTS : 1];
    __shared__ float tpA[ALPHA ? 9 : 1][64];
    __shared__ float unormT[ALPHA ? 256 : 1];      // code / 255.0f
    if (ALPHA) { for (uint32_t i = threadIdx.x; i < 256u; i += 64u) unormT[i] = (float)i / 255.0f; __syncthreads(); }
    extern __shared__ uint32_t binBase[];          // a.tableCells words (launch_raster)
    __shared__ float tpF[9][64];
    __shared__ int tpI[4][64];
    __shared__ uint32_t rowOff[65];
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t lane0 = threadIdx.x;
#ifndef BRMI_RASTER_OPAQUE_LANE
#define BRMI_RASTER_OPAQUE_LANE 0
#endif
#if !BRMI_RASTER_OPAQUE_LANE
    const uint32_t lane = lane0;
#endif
    // wave-uniform by construction; said so to the compiler, which then fetches a cluster's records with scalar loads (one s_load per record
    // instead of a chain of vector loads with a wait after each: the fetch was 11-29 % of the kernel's wave-cycles, measured with phase stamps)
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.firstCounter == 0xFFFFFFFFu ? 0u : a.counters[a.firstCounter]));
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.counters[a.countCounter]);
    if (a.countFeedback && blockIdx.x == 0u && lane0 == 0u) __hip_atomic_store(a.countFeedback, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const GlobalSink gsink{a.vis, a.tilesX, a.debugFlags};
    // static round-robin over clusters: a shared queue head saturates at ~90 dequeues/us (MI355X_MICROARCH.md, row
    // "dequeue"), which is slower than the work itself once big triangles are handed off
    // Frames with far fewer clusters than the grid has waves (every BASELINE-class frame: ~2 k clusters on 1024 SIMDs) are bound by a
    // cluster's serial chain (fetch, setup, rows, bin reservation -- twice), not by throughput: there a cluster is dealt to 2, 4 or 8 waves.
    // Each wave transforms the vertices, takes ONE of the two 64-triangle passes (softwareRaster.hlsl:502 votes per wave of 64 triangles,
    // so the passes are independent), sets up all 64 triangles of it (the vote needs them) and emits only its share of them.  Keys are
    // order-free, so the result is the same.
    uint32_t split = 1u;
    // (round 4: the FIRST doubling is allowed up to twice the grid -- some waves then take two half-clusters --, the others up to the grid: a view of 6 - 8 k
    // clusters with large near triangles, where a cluster is 20 k bin records from one wave, takes two waves per cluster: raster 0.345 -> 0.32 ms at position
    // 20 of the bench's path; a Sponza-class frame of 1.5 k clusters keeps four per cluster, eight cost it 10 us.  A larger grid does the same for the
    // rasteriser alone, but 8,192 more waves that find nothing each wait for a slot beside the other frame's shading waves.)
    if (!LEAN && !(a.debugFlags & 0x40000000u)) { while (split < 8u && count * split * 2u <= (split == 1u ? BRMI_RASTER_SPLIT_FIRST * gridDim.x : gridDim.x)) split *= 2u; }
    const uint32_t parts = max(split >> 1, 1u), lanesPerPart = 64u / parts;      // shares of a pass
    const uint32_t items = count * split;
#ifdef BRMI_TILE_STAMPS
    unsigned long long kph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kprev = __builtin_amdgcn_s_memtime();
#define KSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); kph[k] += now_ - kprev; kprev = now_; } while (0)
#else
#define KSTAMP(k) do { } while (0)
#endif
    // Alpha-tested pixels of the small boxes (round 4, as in k_raster_bins<true>): the row walk only finds covered pixels and appends them (key,
    // texcoord, pixel, material) to this ring; whenever 64 are waiting the wave looks at the visibility buffer and samples for all of them at once --
    // every lane busy in the sampler instead of the few whose row has a covered pixel at that step.  The ring lives across batches and clusters; what
    // is left is tested when the wave has no more clusters.
    constexpr uint32_t RQ = 128, RQ_N = ALPHA ? RQ : 1;
    __shared__ unsigned long long rqKey[RQ_N];
    __shared__ float rqU[RQ_N], rqV[RQ_N];
    __shared__ uint32_t rqPix[RQ_N], rqMat[RQ_N];
    uint32_t rqHead = 0, rqTail = 0;      // wave-uniform
    auto rq_drain = [&](uint32_t n) {
        wave_lds_sync();
        if (lane0 < n) {
            const uint32_t e = (rqHead + lane0) & (RQ - 1u);
            const unsigned long long key = rqKey[e];
            const int px = (int)(rqPix[e] & 0xFFFFu), py = (int)(rqPix[e] >> 16);
            const AlphaMaterial m = a.alphaMats[rqMat[e]];
            // the look is as slow as the texel fetches: both are requested together and the pixel costs one memory round trip, not two
            const unsigned long long cur = gsink.peek(px, py);
            const bool fails = alpha_test_failed(unormT, m, f2{rqU[e], rqV[e]});
            if (key < cur && !fails) gsink(px, py, key);
        }
        rqHead += n;
        wave_lds_sync();
    };
    // (round 6: a frame that holds clusters back names the clusters to draw in a list -- written by an earlier launch, so through the scalar cache; the NEXT item's
    // entry is requested while this one is walked: one SGPR, unlike the 16 of the whole record that cost more than they saved in round 5)
    uint32_t listed = (a.drawList && blockIdx.x < items) ? kconst(a.drawList)[first + blockIdx.x / split] : 0u;
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
#if BRMI_RASTER_OPAQUE_LANE
        uint32_t lane = lane0; asm volatile("" : "+v"(lane));      // (what a cluster derives from the lane index is recomputed per cluster, not hoisted into registers and SGPR spill lanes)
#endif
        KSTAMP(7);
        const uint32_t c = item / split, sub = item % split;
        const uint32_t clusterIndex = a.drawList ? (uint32_t)__builtin_amdgcn_readfirstlane((int)listed) : first + c;
        if (a.drawList && item + gridDim.x < items) listed = kconst(a.drawList)[first + (item + gridDim.x) / split];
        const ClusterSetup cs = load_uniform(&a.setup[clusterIndex]);        // resolved by the compaction kernel: one hop instead of six
        bool toGeneral = LEAN && (cs.counts & BRMI_CS_SKINNED) != 0u;      // (wave-uniform)
        const uint32_t vertCount = cs.counts & 0xFFu, triCount = (cs.counts >> 8) & 0xFFu, posFormat = (cs.counts >> 16) & 0xFFu;
        const uint32_t passLo = split > 1u ? (sub / parts) * 64u : 0u, passHi = split > 1u ? min(passLo + 64u, triCount) : triCount;
        const uint32_t part = sub % parts;
        if (passLo + part * lanesPerPart >= triCount) continue;      // (wave-uniform) no triangle of this share exists
        const bool reverseWinding = ((cs.counts >> 24) & 1u) != 0u;
        const brmi_view_raster_info ri = load_uniform(&sc.viewRasterInfo[cs.viewId]);
        const float visWidth = (float)(ri.scissorMaxX - ri.scissorMinX), visHeight = (float)(ri.scissorMaxY - ri.scissorMinY);
        const float sMinXf = (float)ri.scissorMinX, sMinYf = (float)ri.scissorMinY;
        const auto oc = kconst(a.objConst + (size_t)cs.perObjectIndex * OBJ_CONST_FLOATS);
        m4 mvp;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) mvp.m[i][j] = oc[i * 4 + j];
        const f4 modelViewZ{oc[32], oc[33], oc[34], oc[35]};
        const uint8_t* posBase = cs.posBase;
        const uint8_t* triBase = cs.triBase;
        // the index bytes of both passes are requested before the vertex stage: behind the barrier they were a round trip of their own in
        // front of every pass's setup (15-27 % of the kernel's wave-cycles)
        uint32_t pi0[2] = {0u, 0u}, pi1[2] = {0u, 0u}, pi2[2] = {0u, 0u};
#pragma unroll
        for (uint32_t k = 0; k < 2u; k++) {
            const uint32_t t = passLo + k * 64u + lane;
            if (passLo + k * 64u < passHi && t < triCount) { pi0[k] = triBase[t * 3u]; pi1[k] = triBase[t * 3u + 1u]; pi2[k] = triBase[t * 3u + 2u]; }
        }
        // compute skinning, folded into the vertex fetch (softwareRaster.hlsl:349-360)
        const bool skinVerts = !LEAN && (cs.counts & (BRMI_CS_SKINNED | BRMI_CS_JOINTS)) == (BRMI_CS_SKINNED | BRMI_CS_JOINTS);
        const uint32_t skinSlot = skinVerts ? sc.perMeshInstance[cs.instanceIndex].skinningInstanceSlot : 0xFFFFFFFFu;
        const bool alphaCluster = ALPHA && (cs.counts & BRMI_CS_ALPHA) != 0u;
        ClusterUv cu{nullptr, nullptr};
        AlphaMaterial amat{};
        if (alphaCluster) { cu = a.clusterUv[clusterIndex]; amat = a.alphaMats[cs.materialDataIndex]; }

        if (a.debugFlags & 0x100) { const float probe_ = mvp.m[0][0] + modelViewZ.x + visWidth; if (probe_ == 1234.5f) a.counters[CNT_DROPPED_CLUSTERS] = 1u; }   // (instrumented runs: the cluster's records have arrived)
        KSTAMP(0);
        // vertex stage -> LDS (softwareRaster.hlsl:339-387)
        if (!toGeneral)
        for (uint32_t v = lane; v < vertCount; v += 64) {
            f3 lp{0.0f, 0.0f, 0.0f};
            if (posFormat == BRMI_POSITION_FORMAT_FLOAT3) {
                const float* pp = reinterpret_cast<const float*>(posBase + v * 12u);
                lp = f3{pp[0], pp[1], pp[2]};
            }
            if (skinVerts) {
                uint32_t joints[8]; float weights[8];
                load_skin_influences(cs.nrmBase + cs.jointDelta + v * 32u, (cs.counts & BRMI_CS_WEIGHTS) ? cs.nrmBase + cs.weightDelta + v * 32u : nullptr, joints, weights);
                lp = xyz(mul_point(lp, build_skin_matrix(sc.skinningMatrices, skinSlot, joints, weights)));
            }
            const f4 lp4{lp.x, lp.y, lp.z, 1.0f};
            const f4 clip = mul_vm(lp4, mvp);
            const float viewZ = dot4(lp4, modelViewZ);
            const float invW = 1.0f / clip.w;
            const float ndcx = clip.x * invW, ndcy = clip.y * invW;
            sx[v] = (ndcx + 1.0f) * 0.5f * visWidth + sMinXf;
            sy[v] = (1.0f - ndcy) * 0.5f * visHeight + sMinYf;
            sd[v] = -viewZ;
            if (alphaCluster) { const f2 uv = decode_uv(cu, v); siw[v] = invW; su[v] = uv.x; sv[v] = uv.y; }
        }
        wave_lds_sync();
        KSTAMP(1);

        // triangle stage: lane = triangle (softwareRaster.hlsl:416-611)
        for (uint32_t waveBase = passLo; waveBase < passHi && !toGeneral; waveBase += 64) {
            const uint32_t t = waveBase + lane;
            bool active = t < triCount;
            float d0 = 0, d1 = 0, d2 = 0, row_b0 = 0, row_b1 = 0, dx_b0 = 0, dx_b1 = 0, dy_b0 = 0, dy_b1 = 0;
            int minX = 0, minY = 0, maxX = -1, maxY = -1;
            AlphaRecord arec{};
            if (active) {
                const uint32_t pk = (waveBase - passLo) >> 6;
                uint32_t i0 = pk ? pi0[1] : pi0[0], i1 = pk ? pi1[1] : pi1[0], i2 = pk ? pi2[1] : pi2[0];
                if (reverseWinding) { const uint32_t tmp = i1; i1 = i2; i2 = tmp; }
                if (alphaCluster) { arec.tri = AlphaTri{siw[i0], siw[i1], siw[i2], f2{su[i0], sv[i0]}, f2{su[i1], sv[i1]}, f2{su[i2], sv[i2]}}; arec.materialDataIndex = cs.materialDataIndex; }
                const float s0x = sx[i0], s0y = sy[i0], s1x = sx[i1], s1y = sy[i1], s2x = sx[i2], s2y = sy[i2];
                d0 = sd[i0]; d1 = sd[i1]; d2 = sd[i2];
                if (d0 <= 0.0f || d1 <= 0.0f || d2 <= 0.0f) active = false;
                const float e01x = s1x - s0x, e01y = s1y - s0y, e02x = s2x - s0x, e02y = s2y - s0y;
                const float twiceArea = e01x * e02y - e01y * e02x;
                if (twiceArea >= 0.0f) active = false;
                if (active) {
                    const float invTwiceArea = -1.0f / twiceArea;
                    const float bbMinX = min2(min2(s0x, s1x), s2x), bbMinY = min2(min2(s0y, s1y), s2y);
                    const float bbMaxX = max2(max2(s0x, s1x), s2x), bbMaxY = max2(max2(s0y, s1y), s2y);
                    minX = to_int_sat(floorf(bbMinX)); minY = to_int_sat(floorf(bbMinY));
                    maxX = to_int_sat(floorf(bbMaxX)); maxY = to_int_sat(floorf(bbMaxY));
                    minX = max(minX, (int)ri.scissorMinX); minY = max(minY, (int)ri.scissorMinY);
                    maxX = min(maxX, (int)ri.scissorMaxX - 1); maxY = min(maxY, (int)ri.scissorMaxY - 1);
                    minX = max(minX, 0); minY = max(minY, 0);
                    maxX = min(maxX, (int)a.visW - 1); maxY = min(maxY, (int)a.visH - 1);
                    if (minX > maxX || minY > maxY) active = false;
                    else {
                        const float ox = (float)minX + 0.5f, oy = (float)minY + 0.5f;
                        const float e12x = s2x - s1x, e12y = s2y - s1y, e20x = s0x - s2x, e20y = s0y - s2y;
                        row_b0 = ((ox - s1x) * e12y - (oy - s1y) * e12x) * invTwiceArea;
                        row_b1 = ((ox - s2x) * e20y - (oy - s2y) * e20x) * invTwiceArea;
                        dx_b0 = e12y * invTwiceArea; dx_b1 = e20y * invTwiceArea;
                        dy_b0 = -e12x * invTwiceArea; dy_b1 = -e20x * invTwiceArea;
                    }
                }
            }
            const int rectWidth = maxX - minX + 1;
            if (!LEAN && a.chainDirty && active) {
                // round 5 (phase 2 only, wave-uniform): what phase 2 draws is small; the chain's second build then only redoes the 32 x 32 px blocks a phase-2 triangle's
                // box touches.  A byte per block, plain stores of 1 (every writer stores the same): no atomic (they serialise on a line), nothing to wait for.
                const int cy0 = max(minY, (int)a.rowLo), cy1 = min(maxY, (int)a.rowHi - 1);
                if (cy0 <= cy1) {
                    const int bx0 = minX >> 5, bx1 = maxX >> 5, by0 = cy0 >> 5, by1 = cy1 >> 5;
                    if ((bx1 - bx0 + 1) * (by1 - by0 + 1) > 16) a.chainDirty[0] = 1;
                    else for (int by = by0; by <= by1; by++) for (int bx = bx0; bx <= bx1; bx++) a.chainDirty[4u + (uint32_t)by * a.chainBlocksX + (uint32_t)bx] = 1;
                }
            }
            const bool useScanlineRanges = __any(active && rectWidth > 4);
            if (parts > 1u && lane / lanesPerPart != part) active = false;      // another wave's share of the pass
            const int rows = maxY - minY + 1;
            // frames of many small clusters bin from 32 px on (the row re-deal's global atomics are what their waves wait for: 34-35 % of the
            // kernel's wave-cycles, phase stamps); frames of few large clusters keep 64 (measured both ways, profiles/r03_experiments.md)
            const bool bigHere = active && rows * rectWidth > (alphaCluster ? a.bigTriAreaAlpha : (count >= a.denseClusterCount ? a.bigTriAreaDense : a.bigTriArea));
            const bool big = !LEAN && bigHere;
            // bins the box overlaps (rows clipped to this GPU's band)
            const int yLo = max(minY, (int)a.rowLo), yHi = min(maxY, (int)a.rowHi - 1);
            if (LEAN) {
                // the lean form sets a binned triangle up and hands it on: its records are emitted by k_raster_emit, from the queue (one slot reservation per wave and pass)
                const bool queue = bigHere && yLo <= yHi;
                const uint64_t qm = __ballot(queue);
                if (qm != 0ull) {
                    const int leader = __ffsll((unsigned long long)qm) - 1;
                    const uint32_t stripe = blockIdx.x & 63u, nq = (uint32_t)__popcll(qm);
                    unsigned long long head = 0ull;
                    if ((int)lane == leader) head = atomicAdd(reinterpret_cast<unsigned long long*>(&a.counters[a.bigCounter + stripe * 32u]), (1ull << 32) | (unsigned long long)nq);
                    const uint32_t first = (uint32_t)__shfl((int)(uint32_t)head, leader), run = (uint32_t)__shfl((int)(uint32_t)(head >> 32), leader);
                    const bool fits = first + nq <= a.bigCapacity && run < a.bigCapacity;      // (wave-uniform)
                    if ((int)lane == leader && run < a.bigCapacity) a.bigRuns[(size_t)stripe * a.bigCapacity + run] = fits ? uint2{first, nq} : uint2{0u, 0u};
                    if (queue && fits) {
                        const uint32_t slot = stripe * a.bigCapacity + first + lane_rank(qm);
                        float qb0 = row_b0, qb1 = row_b1;
                        for (int y = minY; y < yLo; y++) { qb0 += dy_b0; qb1 += dy_b1; }      // (the serial loop's additions down to this GPU's first row of the box)
                        WideTri w;
                        w.base.clusterIndex = clusterIndex; w.base.triAndFlags = t | (useScanlineRanges ? 0x100u : 0u); w.base.minX = minX; w.base.rectWidth = rectWidth; w.base.rowStart = yLo;
                        w.base.sb0 = qb0; w.base.sb1 = qb1; w.base.dx_b0 = dx_b0; w.base.dx_b1 = dx_b1; w.base.dy_b0 = dy_b0; w.base.dy_b1 = dy_b1; w.base.d0 = d0; w.base.d1 = d1; w.base.d2 = d2;
                        w.base.pad0 = 0u; w.base.pad1 = 0u;
                        w.yHi = yHi; w.band0 = yLo >> BIN_ROWS_SHIFT; w.band1 = yHi >> BIN_ROWS_SHIFT; w.strip0 = minX >> BIN_W_SHIFT; w.strip1 = maxX >> BIN_W_SHIFT; w.pad[0] = w.pad[1] = w.pad[2] = 0u;
                        a.bigQueue[slot] = w;
                    }
                    // (a full queue: the general launch draws the whole cluster; what is in the queue of it already is then drawn twice -- the same keys)
                    if (!fits) toGeneral = true;
                }
            }
            const int band0 = yLo >> BIN_ROWS_SHIFT, band1 = yHi >> BIN_ROWS_SHIFT, strip0 = minX >> BIN_W_SHIFT, strip1 = maxX >> BIN_W_SHIFT;
            // interleaved partition: a 16-row bin band of the frame is owned whole or not at all (chunks are multiples of 16 rows); an owned
            // one is bin band `vband` of this GPU's surfaces
            const bool striped = !LEAN && stripe_on(a.stripes);
            auto owns_band = [&](int band) { return !striped || stripe_owns(a.stripes, (uint32_t)band << BIN_ROWS_SHIFT); };
            auto vband = [&](int band) { return striped ? stripe_vrow(a.stripes, (uint32_t)band << BIN_ROWS_SHIFT) >> BIN_ROWS_SHIFT : (uint32_t)band; };
            const int nStrips = strip1 - strip0 + 1;
            // The bands this GPU owns of the box, as bin bands of its SURFACE (owned frame rows are consecutive surface rows, chunks are multiples of 16
            // rows).  Round 4: the interleaved partition counted and windowed the box's FRAME bands -- eight times the bands a rank owns of them on
            // the 8-GPU frame -- so most big triangles took the one-at-a-time whole-wave path or fell out of the LDS window (one global atomic per
            // record): a rank's k_raster ran 147 us where the 4K frame's takes 36 (tools/rank_balance.py, kernel stats of an emulated rank).
            int sband0 = band0, sband1 = band1;
            if (striped && big && yLo <= yHi) {
                const uint32_t fo = stripe_first_owned(a.stripes, (uint32_t)yLo), lo = stripe_last_owned(a.stripes, (uint32_t)yHi);
                if (lo == 0xFFFFFFFFu || fo > lo || fo > (uint32_t)yHi) { sband0 = 0; sband1 = -1; }
                else { sband0 = (int)(stripe_vrow(a.stripes, fo) >> BIN_ROWS_SHIFT); sband1 = (int)(stripe_vrow(a.stripes, lo) >> BIN_ROWS_SHIFT); }
            }
            const int entries = (big && yLo <= yHi) ? (sband1 - sband0 + 1) * nStrips : 0;
            const uint32_t flags = t | (useScanlineRanges ? 0x100u : 0u);
            // row start at the first row of this GPU's band: the serial loop's additions from the box top, made once per triangle with
            // all lanes stepping side by side (a lower band of a multi-GPU frame starts thousands of rows below the top of a large box)
            float band_b0 = row_b0, band_b1 = row_b1;
            if (entries > 0) for (int y = minY; y < yLo; y++) { band_b0 += dy_b0; band_b1 += dy_b1; }
            // A few bins per triangle: every lane appends its own records (below, after the small boxes).  A slot in a bin costs an atomic with
            // return on the bin's counter (~2 us round trip, and the triangles of a meshlet hit the same few bins), so the wave first counts
            // its records per bin in an LDS window over the bins it touches and reserves each bin's run with ONE global atomic -- requested
            // HERE, before the small boxes are walked, and used after them: the round trip was 44 % of this kernel's wave-cycles when the
            // wave sat through it (phase stamps, Bistro-class frame).
            // Round 4: views with large near triangles (a camera path through the street: raster 0.17 -> 0.43 ms, k_raster 36 -> 160 us per launch) put the union of a
            // wave's bin boxes beyond the 256-cell window, and every record then took a global atomic with return of its own, one after the other in
            // its lane.  With the wide table (up to 2048 cells: the whole 4K frame) such passes still count in LDS and reserve with one atomic per bin,
            // a batch of four in flight per lane; and since a lane's slots then come from LDS, triangles of up to 512 bin entries stay with their lane
            // instead of being emitted one at a time by the whole wave.
            // (round 6: with the wide pass behind this launch the limit is that pass's -- its emission is a workgroup's, a lane's is one record after the other)
            const int coopEntries = a.wideQueue ? (int)a.wideEntries : (a.tableCells > (uint32_t)BIN_WINDOW ? COOP_ENTRIES_TABLE : COOP_ENTRIES);
            const bool few = entries > 0 && entries <= coopEntries && !(a.debugFlags & 2);
            bool fewW = few;      // (a pass whose bins fit no window falls back to the 64-entry rule below)
            const bool anyFew = __any(few);
            int wb0 = 0, ws0 = 0, winW = 1, cells = 0; bool windowed = false, wideWindow = false;
            uint32_t resvBase[BIN_WINDOW / 64] = {};      // (the counts stay in the LDS window until the bases replace them: four registers less across the small boxes)
            if (anyFew) {
                int wb1 = few ? sband1 : -1, ws1 = few ? strip1 : -1;      // (the window is over surface bin bands)
                wb0 = few ? sband0 : 0x7FFFFFFF; ws0 = few ? strip0 : 0x7FFFFFFF;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    wb0 = min(wb0, __shfl_xor(wb0, o)); wb1 = max(wb1, __shfl_xor(wb1, o));
                    ws0 = min(ws0, __shfl_xor(ws0, o)); ws1 = max(ws1, __shfl_xor(ws1, o));
                }
                winW = ws1 - ws0 + 1; cells = (wb1 - wb0 + 1) * winW;
                windowed = cells <= (int)a.tableCells;      // wave-uniform
                wideWindow = windowed && cells > BIN_WINDOW;
                fewW = few && (windowed || entries <= COOP_ENTRIES);
                if (windowed) {
                    for (int cI = (int)lane; cI < cells; cI += 64) binBase[cI] = 0u;
                    wave_lds_sync();
                    if (few) for (int sb = sband0; sb <= sband1; sb++) for (int st = strip0; st <= strip1; st++) atomicAdd(&binBase[(sb - wb0) * winW + (st - ws0)], 1u);
                    wave_lds_sync();
                    if (!wideWindow)
#pragma unroll
                    for (int k = 0; k < BIN_WINDOW / 64; k++) {
                        const int cI = (int)lane + 64 * k;
                        const uint32_t cnt = cI < cells ? binBase[cI] : 0u;
                        if (cnt != 0u) resvBase[k] = atomicAdd(&a.binCounts[(size_t)((uint32_t)(wb0 + cI / winW) * a.binsX + (uint32_t)(ws0 + cI % winW)) * BIN_COUNT_STRIDE], cnt);
                    }
                }
            }
            KSTAMP(2);
            // Small boxes: global atomics.  lane = triangle leaves most lanes idle (culled triangles, boxes of very different
            // size), so the rows of the batch's small triangles are re-dealt to the lanes: an exclusive scan of the row counts,
            // the setup of every triangle parked in LDS, then lane k takes rows k, k + 64, ... of the concatenated row list.
#ifndef BRMI_RASTER_TINY
#define BRMI_RASTER_TINY 4
#endif
#ifndef BRMI_RASTER_TINY_RANGES
#define BRMI_RASTER_TINY_RANGES 0
#endif
            // Round 5: a pass whose small boxes are ALL at most BRMI_RASTER_TINY x BRMI_RASTER_TINY pixels (frames of sub-pixel triangles: the Zorah-class frame rasterises 62 M
            // triangles for 33 M pixels) is walked lane = triangle -- no prefix scan, no parking of thirteen values per triangle in LDS, no bisection per row task: the re-deal
            // exists to balance boxes of very different size, and these are all the same.  Same arithmetic per pixel (raster_row from the box's first row).
            const bool smallHere = active && !bigHere && !(a.debugFlags & 1) && yLo <= yHi;
            const bool tinyPass = BRMI_RASTER_TINY > 0 && !striped && !alphaCluster && (BRMI_RASTER_TINY_RANGES || !useScanlineRanges) && !__any(smallHere && (maxY - minY + 1) > BRMI_RASTER_TINY);
            if (tinyPass) {
                if (smallHere) {
                    float sb0 = row_b0, sb1 = row_b1;
                    for (int y = minY; y <= maxY; y++) {
                        if (y >= yLo && y <= yHi) raster_row(gsink, NoAlpha{}, y, minX, rectWidth, useScanlineRanges, sb0, sb1, dx_b0, dx_b1, -(dx_b0 + dx_b1), d0, d1, d2, clusterIndex, t, minX, minX + rectWidth - 1);
                        sb0 += dy_b0; sb1 += dy_b1;
                    }
                }
            } else
            {
                const bool small = smallHere;
                uint32_t myRows = small ? (uint32_t)(yHi - yLo + 1) : 0u;
                // interleaved partition: a box inside one chunk (nearly all small boxes) is owned whole or dropped here; one that straddles a
                // chunk boundary keeps its rows and the row tasks test each
                if (striped && myRows != 0u && (uint32_t)yLo / a.stripes.rows == (uint32_t)yHi / a.stripes.rows && !stripe_owns(a.stripes, (uint32_t)yLo)) myRows = 0u;
                uint32_t incl = myRows;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
                const uint32_t totalRows = (uint32_t)__shfl((int)incl, 63);
                if (totalRows != 0u) {
                    rowOff[lane] = incl - myRows;
                    if (lane == 63) rowOff[64] = totalRows;
                    tpF[0][lane] = row_b0; tpF[1][lane] = row_b1; tpF[2][lane] = dx_b0; tpF[3][lane] = dx_b1; tpF[4][lane] = dy_b0; tpF[5][lane] = dy_b1;
                    tpF[6][lane] = d0; tpF[7][lane] = d1; tpF[8][lane] = d2;
                    tpI[0][lane] = minX; tpI[1][lane] = rectWidth; tpI[2][lane] = minY; tpI[3][lane] = yLo;
                    if (alphaCluster) {
                        tpA[0][lane] = arec.tri.invW0; tpA[1][lane] = arec.tri.invW1; tpA[2][lane] = arec.tri.invW2;
                        tpA[3][lane] = arec.tri.uv0.x; tpA[4][lane] = arec.tri.uv0.y; tpA[5][lane] = arec.tri.uv1.x; tpA[6][lane] = arec.tri.uv1.y; tpA[7][lane] = arec.tri.uv2.x; tpA[8][lane] = arec.tri.uv2.y;
                    }
                    wave_lds_sync();
                    if (alphaCluster) {
                        for (uint32_t tb = 0; tb < totalRows; tb += 64) {
                            const uint32_t task = tb + lane;
                            bool on = task < totalRows;
                            uint32_t tri = 0;
#pragma unroll
                            for (uint32_t step = 32; step > 0; step >>= 1) if (on && rowOff[tri + step] <= task) tri += step;
                            const int t_minX = tpI[0][tri], t_w = tpI[1][tri], t_minY = tpI[2][tri];
                            const int py = on ? tpI[3][tri] + (int)(task - rowOff[tri]) : 0;
                            const float t_dx0 = tpF[2][tri], t_dx1 = tpF[3][tri], t_dy0 = tpF[4][tri], t_dy1 = tpF[5][tri];
                            float sb0 = tpF[0][tri], sb1 = tpF[1][tri];
                            if (striped && !stripe_owns(a.stripes, (uint32_t)py)) on = false;      // another GPU's row
                            if (on) for (int k = py - t_minY; k > 0; k--) { sb0 += t_dy0; sb1 += t_dy1; }      // the serial loop's row stepping
                            const int spy = striped ? (int)stripe_vrow(a.stripes, (uint32_t)py) : py;      // the row of this GPU's surface
                            SegWalk w{0.0f, 0.0f, 0, -1, useScanlineRanges};
                            if (on) w = seg_begin(t_minX, t_w, useScanlineRanges, sb0, sb1, t_dx0, t_dx1, -(t_dx0 + t_dx1), t_minX, t_minX + t_w - 1);
                            const AlphaTri at{tpA[0][tri], tpA[1][tri], tpA[2][tri], f2{tpA[3][tri], tpA[4][tri]}, f2{tpA[5][tri], tpA[6][tri]}, f2{tpA[7][tri], tpA[8][tri]}};
                            const float t_d0 = tpF[6][tri], t_d1 = tpF[7][tri], t_d2 = tpF[8][tri];
                            for (;;) {
                                const bool act = on && w.px <= w.x1;
                                if (!__any(act)) break;
                                const float b2 = 1.0f - w.b0 - w.b1;
                                const bool want = act && (w.all || (w.b0 >= 0.0f && w.b1 >= 0.0f && b2 >= 0.0f));
                                const unsigned long long m = __ballot(want);
                                if (want) {
                                    const float depth = w.b0 * t_d0 + w.b1 * t_d1 + b2 * t_d2;
                                    const uint32_t e = (rqTail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (RQ - 1u);
                                    const f2 uv = pixel_texcoord(at, w.b0, w.b1, b2);
                                    rqKey[e] = (unsigned long long)pack_vis_key(depth, clusterIndex, waveBase + tri); rqU[e] = uv.x; rqV[e] = uv.y;
                                    rqPix[e] = (uint32_t)w.px | ((uint32_t)spy << 16); rqMat[e] = cs.materialDataIndex;
                                }
                                rqTail += (uint32_t)__popcll(m);
                                if (act) { w.b0 += t_dx0; w.b1 += t_dx1; w.px++; }
                                if (rqTail - rqHead >= 64u) rq_drain(64u);
                            }
                        }
                    } else
                    for (uint32_t task = lane; task < totalRows; task += 64) {
                        uint32_t tri = 0;
#pragma unroll
                        for (uint32_t step = 32; step > 0; step >>= 1) if (rowOff[tri + step] <= task) tri += step;
                        const int t_minX = tpI[0][tri], t_w = tpI[1][tri], t_minY = tpI[2][tri];
                        const int py = tpI[3][tri] + (int)(task - rowOff[tri]);
                        const float t_dx0 = tpF[2][tri], t_dx1 = tpF[3][tri], t_dy0 = tpF[4][tri], t_dy1 = tpF[5][tri];
                        float sb0 = tpF[0][tri], sb1 = tpF[1][tri];
                        if (striped && !stripe_owns(a.stripes, (uint32_t)py)) continue;      // another GPU's row
                        for (int k = py - t_minY; k > 0; k--) { sb0 += t_dy0; sb1 += t_dy1; }      // the serial loop's row stepping
                        const int spy = striped ? (int)stripe_vrow(a.stripes, (uint32_t)py) : py;      // the row of this GPU's surface
                        raster_row(gsink, NoAlpha{}, spy, t_minX, t_w, useScanlineRanges, sb0, sb1, t_dx0, t_dx1, -(t_dx0 + t_dx1), tpF[6][tri], tpF[7][tri], tpF[8][tri], clusterIndex, waveBase + tri,
                                   t_minX, t_minX + t_w - 1);
                    }
                    wave_lds_sync();
                }
            }
            KSTAMP(3);
            // the records of triangles with a few bins: slots handed out from the runs reserved above
            if (anyFew) {
                if (windowed && !wideWindow) {
                    // the reserved runs have arrived (requested before the small boxes were walked)
#pragma unroll
                    for (int k = 0; k < BIN_WINDOW / 64; k++) { const int cI = (int)lane + 64 * k; if (cI < cells && binBase[cI] != 0u) binBase[cI] = resvBase[k]; }
                    wave_lds_sync();
                } else if (wideWindow) {
                    // the wide window: counts -> bases in place, four reservations in flight per lane and round
                    for (int c0 = 0; c0 < cells; c0 += 256) {
                        uint32_t cnt[4], base[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; cnt[k] = cI < cells ? binBase[cI] : 0u; }
#pragma unroll
                        for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; base[k] = cnt[k] != 0u ? atomicAdd(&a.binCounts[(size_t)((uint32_t)(wb0 + cI / winW) * a.binsX + (uint32_t)(ws0 + cI % winW)) * BIN_COUNT_STRIDE], cnt[k]) : 0u; }
#pragma unroll
                        for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; if (cnt[k] != 0u) binBase[cI] = base[k]; }
                    }
                    wave_lds_sync();
                }
                if (fewW) {
                    // step the row start band by band (the additions of the serial loop) and append one record per band and strip
                    float sb0 = band_b0, sb1 = band_b1;
                    int py = yLo;
                    while (py <= yHi) {
                        const int band = py >> BIN_ROWS_SHIFT;
                        const int n = min(((band + 1) << BIN_ROWS_SHIFT), yHi + 1) - py;
                        BinRecord r;
                        r.clusterIndex = clusterIndex; r.triAndFlags = flags | ((uint32_t)n << 16);
                        r.minX = minX; r.rectWidth = rectWidth; r.rowStart = striped ? (int)stripe_vrow(a.stripes, (uint32_t)py) : py;      // (an owned band's rows are consecutive surface rows)
                        r.sb0 = sb0; r.sb1 = sb1; r.dx_b0 = dx_b0; r.dx_b1 = dx_b1; r.dy_b0 = dy_b0; r.dy_b1 = dy_b1; r.d0 = d0; r.d1 = d1; r.d2 = d2; r.pad0 = 0; r.pad1 = alphaCluster ? 1u : 0u;
                        if (owns_band(band)) {
                            const uint32_t vb = vband(band);
                            for (int st = strip0; st <= strip1; st++) {
                                if (windowed) bin_store(a, unormT, r, arec, (uint32_t)st, vb, atomicAdd(&binBase[((int)vb - wb0) * winW + (st - ws0)], 1u));
                                else bin_append(a, unormT, r, arec, (uint32_t)st, vb);
                            }
                        }
                        for (int k = 0; k < n; k++) { sb0 += dy_b0; sb1 += dy_b1; }
                        py += n;
                    }
                }
                if (windowed) wave_lds_sync();    // binBase is reused by the next batch
            }
            if (anyFew && !windowed) KSTAMP(6); else KSTAMP(4);      // (instrumented builds: passes whose bins do not fit the LDS window reserve slot by slot)
            // many bins: the whole wave emits the triangle.  lane L owns bands band0 + L, band0 + L + 64, ...: it steps the row
            // start down to each of them (the same additions the serial loop makes) and appends the band's record to every strip.
            const bool isCoop = entries > 0 && !fewW && !(a.debugFlags & 2);
            uint64_t coop = __ballot(isCoop);
            const bool wideCand = entries > (int)a.wideEntries && !(a.debugFlags & 2);      // (with the queue on these are exactly the triangles `few` left out)
            const uint64_t candM = __ballot(wideCand);
            if (candM != 0ull) {
                // round 6: these go to the wide queue (one reservation per wave); what does not fit -- or all of them, when the host has not launched the wide pass for this
                // frame -- is emitted right here as before.  The count is kept either way: it is what the host decides by.
                uint32_t slot = 0u;
                if (lane == (uint32_t)__ffsll((unsigned long long)candM) - 1u) slot = atomicAdd(&a.counters[a.wideCounter], (uint32_t)__popcll(candM));
                slot = (uint32_t)__shfl((int)slot, __ffsll((unsigned long long)candM) - 1) + lane_rank(candM);
                const bool queued = wideCand && a.wideQueue != nullptr && slot < a.wideCapacity;
                if (queued) {
                    WideTri w;
                    w.base.clusterIndex = clusterIndex; w.base.triAndFlags = flags; w.base.minX = minX; w.base.rectWidth = rectWidth; w.base.rowStart = yLo;
                    w.base.sb0 = band_b0; w.base.sb1 = band_b1; w.base.dx_b0 = dx_b0; w.base.dx_b1 = dx_b1; w.base.dy_b0 = dy_b0; w.base.dy_b1 = dy_b1; w.base.d0 = d0; w.base.d1 = d1; w.base.d2 = d2;
                    w.base.pad0 = 0u; w.base.pad1 = alphaCluster ? 1u : 0u;
                    w.yHi = yHi; w.band0 = band0; w.band1 = band1; w.strip0 = strip0; w.strip1 = strip1; w.pad[0] = w.pad[1] = w.pad[2] = 0u;
                    a.wideQueue[slot] = w;
                    if (alphaCluster) { AlphaRecord ar = arec; ar.materialDataIndex = cs.materialDataIndex; a.wideAlpha[slot] = ar; }
                }
                coop = __ballot(isCoop && !queued);
            }
            while (coop != 0ull) {
                const int src = __ffsll((unsigned long long)coop) - 1;
                coop &= coop - 1ull;
                const float c_sb0 = __shfl(band_b0, src), c_sb1 = __shfl(band_b1, src), c_dx0 = __shfl(dx_b0, src), c_dx1 = __shfl(dx_b1, src);
                const float c_dy0 = __shfl(dy_b0, src), c_dy1 = __shfl(dy_b1, src), c_d0 = __shfl(d0, src), c_d1 = __shfl(d1, src), c_d2 = __shfl(d2, src);
                const int c_minX = __shfl(minX, src), c_w = __shfl(rectWidth, src), c_yLo = __shfl(yLo, src), c_yHi = __shfl(yHi, src);
                const int c_band0 = __shfl(band0, src), c_band1 = __shfl(band1, src), c_strip0 = __shfl(strip0, src), c_strip1 = __shfl(strip1, src);
                const uint32_t c_flags = (uint32_t)__shfl((int)flags, src);
                AlphaRecord c_arec{};
                if (alphaCluster) {
                    c_arec.tri = AlphaTri{__shfl(arec.tri.invW0, src), __shfl(arec.tri.invW1, src), __shfl(arec.tri.invW2, src), f2{__shfl(arec.tri.uv0.x, src), __shfl(arec.tri.uv0.y, src)},
                                          f2{__shfl(arec.tri.uv1.x, src), __shfl(arec.tri.uv1.y, src)}, f2{__shfl(arec.tri.uv2.x, src), __shfl(arec.tri.uv2.y, src)}};
                    c_arec.materialDataIndex = cs.materialDataIndex;
                }
                float sb0 = c_sb0, sb1 = c_sb1;
                int py = c_yLo;
                for (int band = c_band0 + (int)lane; band <= c_band1; band += 64) {
                    if (!owns_band(band)) continue;      // (the stepping below catches up from the last band this lane emitted)
                    const int start = max(band << BIN_ROWS_SHIFT, c_yLo);
                    for (; py < start; py++) { sb0 += c_dy0; sb1 += c_dy1; }
                    const int n = min(((band + 1) << BIN_ROWS_SHIFT), c_yHi + 1) - start;
                    BinRecord r;
                    r.clusterIndex = clusterIndex; r.triAndFlags = c_flags | ((uint32_t)n << 16);
                    r.minX = c_minX; r.rectWidth = c_w; r.rowStart = striped ? (int)stripe_vrow(a.stripes, (uint32_t)start) : start;
                    const uint32_t vb = vband(band);
                    r.sb0 = sb0; r.sb1 = sb1; r.dx_b0 = c_dx0; r.dx_b1 = c_dx1; r.dy_b0 = c_dy0; r.dy_b1 = c_dy1; r.d0 = c_d0; r.d1 = c_d1; r.d2 = c_d2; r.pad0 = 0; r.pad1 = alphaCluster ? 1u : 0u;
                    // all the band's bin slots are requested before the first one is used: the atomics overlap instead of costing one
                    // round trip per strip (a full-width triangle touches 15-30 strips)
                    for (int st0 = c_strip0; st0 <= c_strip1; st0 += 8) {
                        uint32_t slots[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) slots[k] = (st0 + k <= c_strip1) ? atomicAdd(&a.binCounts[(size_t)(vb * a.binsX + (uint32_t)(st0 + k)) * BIN_COUNT_STRIDE], 1u) : 0u;
#pragma unroll
                        for (int k = 0; k < 8; k++) if (st0 + k <= c_strip1) bin_store(a, unormT, r, c_arec, (uint32_t)(st0 + k), vb, slots[k]);
                    }
                }
            }
            KSTAMP(5);
        }
        if (LEAN && toGeneral && lane == 0u) a.generalList[atomicAdd(&a.counters[a.generalCounter], 1u)] = clusterIndex;
        wave_lds_sync();   // LDS is reused by the next cluster
    }
    if (ALPHA && rqTail != rqHead) rq_drain(rqTail - rqHead);
#ifdef BRMI_TILE_STAMPS
    if (lane0 < 8u && (a.debugFlags & 0x100)) { unsigned long long v = 0; for (int k = 0; k < 8; k++) if (lane0 == (uint32_t)k) v = kph[k]; atomicAdd(a.debugStamps + 16u + lane0, v); }
    if (lane0 == 0u && (a.debugFlags & 0x100)) {      // the launch's longest wave: its cycles per phase (slots 40 .. 47), kept by a 64-bit max on its total (slot 48)
        unsigned long long tot = 0; for (int k = 0; k < 8; k++) tot += kph[k];
        const unsigned long long before = atomicMax(a.debugStamps + 48u, tot);
        if (tot > before) for (int k = 0; k < 8; k++) a.debugStamps[40 + k] = kph[k];      // (racy between near-equal waves: a diagnostic)
    }
#endif
}

// The lean rasteriser's binned triangles (RasterArgs::bigQueue), 64 queue entries per wave and step: the record emission of k_raster on its own -- per-bin counts in
// the LDS window, one reservation per bin and wave, slots handed out from LDS; triangles of more entries than the window rule allows are emitted by the whole wave, or
// go on to the wide queue.  Consecutive entries are triangles of one cluster (a wave of k_raster<false, true> appends a pass's triangles together), so a wave's bins
// are as few as they were there.  Records, and so keys, as k_raster writes them.
__global__ void __launch_bounds__(64) k_raster_emit(RasterArgs a) {
    wave_prio<PRIO_RASTER>();
    extern __shared__ uint32_t binBase[];          // a.tableCells words
    const uint32_t lane = threadIdx.x;
    const uint32_t stripe = blockIdx.x & 63u;
    const uint32_t nRuns = min((uint32_t)__builtin_amdgcn_readfirstlane((int)a.counters[a.bigCounter + stripe * 32u + 1u]), a.bigCapacity);
    const AlphaRecord noAlpha{};
    for (uint32_t run = blockIdx.x >> 6; run < nRuns; run += gridDim.x >> 6) {
        const uint2 rd = load_uniform(&a.bigRuns[(size_t)stripe * a.bigCapacity + run]);
        const bool have = lane < rd.y;
        WideTri w{};
        if (have) w = a.bigQueue[(size_t)stripe * a.bigCapacity + rd.x + lane];
        w.pad[0] = w.pad[1] = w.pad[2] = 0u;      // (what the writer stored there; said here so that the three words are not carried -- in scratch -- to the wide queue's store)
        const int nStrips = w.strip1 - w.strip0 + 1;
        int entries = have ? (w.band1 - w.band0 + 1) * nStrips : 0;
#ifdef BRMI_EXPERIMENTS
        if (a.debugFlags & 0x200) { if (w.base.sb0 == 1234.5f && w.yHi == -77) a.counters[CNT_DROPPED_CLUSTERS] = 1u; entries = 0; }      // (timing runs: the entries have arrived, nothing is emitted)
#endif
        // (the wide pass takes this kernel's triangles from 16 entries on, not 128: a lane's records are emitted one after the other and the launch lasts as long as its longest
        // lane -- 144 us at 128, 73 at 16, 61 at 8 for the Zorah-class frame's 573 k queued triangles, k_raster_wide 22 us either way; profiles/r06_experiments.md)
        const int coopEntries = a.wideQueue ? (int)a.emitWideEntries : (a.tableCells > (uint32_t)BIN_WINDOW ? COOP_ENTRIES_TABLE : COOP_ENTRIES);
        const bool few = entries > 0 && entries <= coopEntries;
        bool fewW = few;
        int wb0 = 0, ws0 = 0, winW = 1, cells = 0; bool windowed = false;
        if (__any(few)) {
            int wb1 = few ? w.band1 : -1, ws1 = few ? w.strip1 : -1;
            wb0 = few ? w.band0 : 0x7FFFFFFF; ws0 = few ? w.strip0 : 0x7FFFFFFF;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                wb0 = min(wb0, __shfl_xor(wb0, o)); wb1 = max(wb1, __shfl_xor(wb1, o));
                ws0 = min(ws0, __shfl_xor(ws0, o)); ws1 = max(ws1, __shfl_xor(ws1, o));
            }
            winW = ws1 - ws0 + 1; cells = (wb1 - wb0 + 1) * winW;
            windowed = cells <= (int)a.tableCells;      // wave-uniform
            fewW = few && (windowed || entries <= COOP_ENTRIES);
            if (windowed) {
                for (int cI = (int)lane; cI < cells; cI += 64) binBase[cI] = 0u;
                wave_lds_sync();
                if (few) for (int sb = w.band0; sb <= w.band1; sb++) for (int st = w.strip0; st <= w.strip1; st++) atomicAdd(&binBase[(sb - wb0) * winW + (st - ws0)], 1u);
                wave_lds_sync();
                // counts -> bases in place, four reservations in flight per lane and round
                for (int c0 = 0; c0 < cells; c0 += 256) {
                    uint32_t cnt[4], rb[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; cnt[k] = cI < cells ? binBase[cI] : 0u; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; rb[k] = cnt[k] != 0u ? atomicAdd(&a.binCounts[(size_t)((uint32_t)(wb0 + cI / winW) * a.binsX + (uint32_t)(ws0 + cI % winW)) * BIN_COUNT_STRIDE], cnt[k]) : 0u; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int cI = c0 + (int)lane + 64 * k; if (cnt[k] != 0u) binBase[cI] = rb[k]; }
                }
                wave_lds_sync();
            }
#ifdef BRMI_EXPERIMENTS
            if (a.debugFlags & 0x400) { if (windowed) wave_lds_sync(); continue; }      // (timing runs: counted and reserved, no record stored)
#endif
            if (fewW) {
                // step the row start band by band (the additions of the serial loop) and append one record per band and strip
                float sb0 = w.base.sb0, sb1 = w.base.sb1;
                int py = w.base.rowStart;
                while (py <= w.yHi) {
                    const int band = py >> BIN_ROWS_SHIFT;
                    const int rows = min(((band + 1) << BIN_ROWS_SHIFT), w.yHi + 1) - py;
                    BinRecord r = w.base;
                    r.triAndFlags = w.base.triAndFlags | ((uint32_t)rows << 16); r.rowStart = py; r.sb0 = sb0; r.sb1 = sb1;
                    for (int st = w.strip0; st <= w.strip1; st++) {
                        if (windowed) bin_store(a, nullptr, r, noAlpha, (uint32_t)st, (uint32_t)band, atomicAdd(&binBase[(band - wb0) * winW + (st - ws0)], 1u));
                        else bin_append(a, nullptr, r, noAlpha, (uint32_t)st, (uint32_t)band);
                    }
                    for (int k = 0; k < rows; k++) { sb0 += w.base.dy_b0; sb1 += w.base.dy_b1; }
                    py += rows;
                }
            }
            if (windowed) wave_lds_sync();    // binBase is reused by the next step
        }
        // many bins: the wide queue takes the triangle, or the whole wave emits it (lane L owns bands band0 + L, band0 + L + 64, ...)
        const bool isCoop = entries > 0 && !fewW;
        uint64_t coop = __ballot(isCoop);
        const bool wideCand = entries > (int)a.emitWideEntries;
        const uint64_t candM = __ballot(wideCand);
        if (candM != 0ull) {
            const int leader = __ffsll((unsigned long long)candM) - 1;
            uint32_t slot = 0u;
            if ((int)lane == leader) slot = atomicAdd(&a.counters[a.wideCounter], (uint32_t)__popcll(candM));
            slot = (uint32_t)__shfl((int)slot, leader) + lane_rank(candM);
            const bool queued = wideCand && a.wideQueue != nullptr && slot < a.wideCapacity;
            if (queued) a.wideQueue[slot] = w;
            coop = __ballot(isCoop && !queued);
        }
        while (coop != 0ull) {
            const int src = __ffsll((unsigned long long)coop) - 1;
            coop &= coop - 1ull;
            BinRecord r;
            r.clusterIndex = (uint32_t)__shfl((int)w.base.clusterIndex, src); r.minX = __shfl(w.base.minX, src); r.rectWidth = __shfl(w.base.rectWidth, src);
            r.dx_b0 = __shfl(w.base.dx_b0, src); r.dx_b1 = __shfl(w.base.dx_b1, src); r.dy_b0 = __shfl(w.base.dy_b0, src); r.dy_b1 = __shfl(w.base.dy_b1, src);
            r.d0 = __shfl(w.base.d0, src); r.d1 = __shfl(w.base.d1, src); r.d2 = __shfl(w.base.d2, src); r.pad0 = 0u; r.pad1 = 0u;
            const uint32_t c_flags = (uint32_t)__shfl((int)w.base.triAndFlags, src);
            const int c_yLo = __shfl(w.base.rowStart, src), c_yHi = __shfl(w.yHi, src), c_band0 = __shfl(w.band0, src), c_band1 = __shfl(w.band1, src);
            const int c_strip0 = __shfl(w.strip0, src), c_strip1 = __shfl(w.strip1, src);
            float sb0 = __shfl(w.base.sb0, src), sb1 = __shfl(w.base.sb1, src);
            int py = c_yLo;
            for (int band = c_band0 + (int)lane; band <= c_band1; band += 64) {
                const int start = max(band << BIN_ROWS_SHIFT, c_yLo);
                for (; py < start; py++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
                const int rows = min(((band + 1) << BIN_ROWS_SHIFT), c_yHi + 1) - start;
                r.triAndFlags = c_flags | ((uint32_t)rows << 16); r.rowStart = start; r.sb0 = sb0; r.sb1 = sb1;
                for (int st0 = c_strip0; st0 <= c_strip1; st0 += 8) {
                    uint32_t slots[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) slots[k] = (st0 + k <= c_strip1) ? atomicAdd(&a.binCounts[(size_t)((uint32_t)band * a.binsX + (uint32_t)(st0 + k)) * BIN_COUNT_STRIDE], 1u) : 0u;
#pragma unroll
                    for (int k = 0; k < 8; k++) if (st0 + k <= c_strip1) bin_store(a, nullptr, r, noAlpha, (uint32_t)(st0 + k), (uint32_t)band, slots[k]);
                }
            }
        }
    }
}

// The queued triangles (WideTri): eight single-wave workgroups per triangle take its bin bands in turn -- each steps the row start down to its bands with the serial loop's
// additions, exactly as the emitting wave of k_raster does (brmi_raster.hip: "many bins") -- and the lanes of a wave the strips of the band: one slot reservation and one
// 64 B store per lane, four bands' reservations in flight.  Runs between k_raster and the plan (which reads the bins' final counts).
// (Single-wave workgroups: the first form, a 512-thread workgroup per triangle, was as fast alone and cost the Zorah-class frame 0.2 ms IN FLIGHT -- eight waves that have to
// start together on one CU wait for the other frame's shading waves to retire, on the stream the next frame waits for: 2.40 against 2.20 ms, profiles/r06_experiments.md.)
constexpr uint32_t WIDE_SHARES = 8;
template <bool ALPHA>
__global__ void __launch_bounds__(64) k_raster_wide(RasterArgs a) {
    wave_prio<PRIO_RASTER>();
    __shared__ float unormT[ALPHA ? 256 : 1];
    if (ALPHA) { for (uint32_t i = threadIdx.x; i < 256u; i += 64u) unormT[i] = (float)i / 255.0f; __syncthreads(); }
    const uint32_t n = min(a.counters[a.wideCounter], a.wideCapacity);
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t waves = WIDE_SHARES;
    const bool striped = stripe_on(a.stripes);
    for (uint32_t item = blockIdx.x; item < n * WIDE_SHARES; item += gridDim.x) {
        const uint32_t e = item / WIDE_SHARES, wave = item % WIDE_SHARES;
        const WideTri w = load_uniform(&a.wideQueue[e]);
        AlphaRecord arec{};
        const bool alpha = ALPHA && w.base.pad1 != 0u;
        if (alpha) arec = a.wideAlpha[e];
        float sb0 = w.base.sb0, sb1 = w.base.sb1;
        int py = w.base.rowStart;
        // this wave's bands, four at a time: records first (the stepping), then the four bands' slot reservations side by side, then the stores
        for (int bandBase = w.band0 + (int)wave; bandBase <= w.band1; bandBase += 4 * (int)waves) {
            // (only what differs from band to band is kept per band; the record is put together at the store)
            float bsb0[4], bsb1[4]; int brow[4]; uint32_t bflags[4], vb[4]; bool own[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int band = bandBase + k * (int)waves;
                own[k] = band <= w.band1 && (!striped || stripe_owns(a.stripes, (uint32_t)band << BIN_ROWS_SHIFT));
                vb[k] = 0u; bsb0[k] = 0.0f; bsb1[k] = 0.0f; brow[k] = 0; bflags[k] = 0u;
                if (own[k]) {      // (a band this GPU does not own is skipped; the stepping catches up from the last band this wave emitted)
                    const int start = max(band << BIN_ROWS_SHIFT, w.base.rowStart);
                    for (; py < start; py++) { sb0 += w.base.dy_b0; sb1 += w.base.dy_b1; }
                    const int rows = min(((band + 1) << BIN_ROWS_SHIFT), w.yHi + 1) - start;
                    bflags[k] = w.base.triAndFlags | ((uint32_t)rows << 16);
                    brow[k] = striped ? (int)stripe_vrow(a.stripes, (uint32_t)start) : start;
                    bsb0[k] = sb0; bsb1[k] = sb1;
                    vb[k] = striped ? stripe_vrow(a.stripes, (uint32_t)band << BIN_ROWS_SHIFT) >> BIN_ROWS_SHIFT : (uint32_t)band;
                }
            }
            for (int st0 = w.strip0; st0 <= w.strip1; st0 += 64) {
                const int st = st0 + (int)lane;
                uint32_t slots[4];
#pragma unroll
                for (int k = 0; k < 4; k++) slots[k] = (own[k] && st <= w.strip1) ? atomicAdd(&a.binCounts[(size_t)(vb[k] * a.binsX + (uint32_t)st) * BIN_COUNT_STRIDE], 1u) : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) if (own[k] && st <= w.strip1) {
                    BinRecord r = w.base;
                    r.triAndFlags = bflags[k]; r.rowStart = brow[k]; r.sb0 = bsb0[k]; r.sb1 = bsb1[k];
                    bin_store(a, unormT, r, arec, (uint32_t)st, vb[k], slots[k]);
                }
            }
        }
    }
}

// One workgroup per bin: the bin's records are walked one lane per row (16 records at a time) with LDS atomic-min into a tile
// of keys; the tile is then merged into the visibility buffer in 64 B pieces (8 vertically adjacent pixels of a tile column).
#ifndef BRMI_BIN_THREADS
#define BRMI_BIN_THREADS 512
#endif
#ifndef BRMI_BIN_PRIORITY
#define BRMI_BIN_PRIORITY 0
#endif
#ifndef BRMI_BIN_SORT
#define BRMI_BIN_SORT 1
#endif
#ifndef BRMI_BIN_SORT_MIN
#define BRMI_BIN_SORT_MIN 96u
#endif
constexpr uint32_t BIN_ORDER_CAP = 1024;       // longest slice the walk order is built for (longer ones: BRMI_BIN_CAPACITY above 8192) walk in arrival order
// The alpha-tested variant is bound by the latency of its per-pixel texel fetches: it wants as many resident workgroups as the
// register file and the LDS allow.  BRMI_ALPHA_LIST sizes two LDS arrays (the task lists' prefix sums, 6 B per record); with the 32 KB tile, the
// rings of waiting pixels (BRMI_ALPHA_RING x 20 B per wave) and the unorm table a workgroup is 50 KB at 1024 records and 64-entry rings: three
// workgroups per CU = six waves per SIMD at 80 VGPRs (round 4; 67 KB / two workgroups / four waves at 2048 records and 128 entries).
#ifndef BRMI_BIN_ALPHA_WAVES
#define BRMI_BIN_ALPHA_WAVES 6
#endif
#ifndef BRMI_ALPHA_LIST
#define BRMI_ALPHA_LIST 1024      // (round 4: 2048 until the ring of waiting pixels made LDS the limit; BRMI_BIN_MIN_SLICE's default is 1024 anyway)
#endif
template <bool ALPHA>
__global__ void __launch_bounds__(BRMI_BIN_THREADS, ALPHA ? BRMI_BIN_ALPHA_WAVES : 1) k_raster_bins(RasterArgs a) {
    wave_prio<PRIO_BINS>();
    __shared__ unsigned long long tile[BIN_W * BIN_ROWS];
    __shared__ float unormT[ALPHA ? 256 : 1];
#ifndef BRMI_ALPHA_SEG_SHIFT
#define BRMI_ALPHA_SEG_SHIFT 4
#endif
    constexpr int ALPHA_SEG_SHIFT = BRMI_ALPHA_SEG_SHIFT;     // pixels per task = 1 << shift
    constexpr uint32_t ALPHA_LIST = BRMI_ALPHA_LIST;      // alpha-tested records a bin hands to the task pass (later ones take the row path)
    __shared__ uint16_t rowStart[ALPHA ? ALPHA_LIST + 1 : 1];       // exclusive prefix of the opaque records' row counts (<= 16 rows each: the total fits 16 bits)
    __shared__ uint32_t taskStart[ALPHA ? ALPHA_LIST + 1 : 1];      // exclusive prefix of the listed records' task counts
    __shared__ uint32_t scanPart[ALPHA ? 2 * BRMI_BIN_THREADS / 64 : 1];      // the waves' sums of the task scan
    __shared__ uint32_t taskNext, rowNext;
#ifndef BRMI_ALPHA_COMPACT
#define BRMI_ALPHA_COMPACT 1
#endif
    // pixels of alpha-tested records that are covered and can still win their key, waiting for the test: a ring per wave (see the task pass)
// (64 entries: with the 1024-record task lists the workgroup needs 50 KB of LDS and 80 VGPRs -- three workgroups per CU, six waves per SIMD, instead of two and four
    // with 128 entries and 2048 records: San-Miguel-class raster stage 0.265 -> 0.245 ms, frame in flight 0.867 -> 0.838.  A step that would overflow the ring tests what
    // waits first.)
#ifndef BRMI_ALPHA_RING
#define BRMI_ALPHA_RING 64
#endif
    constexpr uint32_t AQ = BRMI_ALPHA_RING, AQ_WAVES = (ALPHA && BRMI_ALPHA_COMPACT) ? BRMI_BIN_THREADS / 64 : 1, AQ_N = (ALPHA && BRMI_ALPHA_COMPACT) ? AQ : 1;
    __shared__ unsigned long long qKey[AQ_WAVES][AQ_N];
    __shared__ float qU[AQ_WAVES][AQ_N], qV[AQ_WAVES][AQ_N];
    __shared__ uint32_t qMeta[AQ_WAVES][AQ_N];
    __shared__ uint16_t order[ALPHA ? 1 : BIN_ORDER_CAP];       // the slice's records in walk order (opaque scenes)
    __shared__ uint32_t classCount[18], classBase[19];
    // The launch is a pool of workgroups that take work items -- (bin, slice) pairs, longest first -- from the list plan_bins wrote (the launch
    // before: k_raster_overflow).  A bin with up to binMinSlice records is one item and keeps the plain read-modify-write merge (it owns its
    // pixels).  A bin with more (the near field of a dense frame: thousands of slivers on one strip of ground; one workgroup walked such a bin
    // for 400 us while 250 CUs idled) is cut into slices; every slice parks its tile of keys in a scratch tile, and the slice that finishes
    // last folds the others' tiles into its own and merges once, plainly -- no per-key atomics (as atomic-min merges, eight keys of a 64 B line
    // from several slices at a time, slices shorter than 1024 records LOST: raster 0.22 -> 0.25 -> 0.29 ms at 512 / 256).
    // Why a list: a grid of (bins x slices) workgroups started in grid order; a slice of 1024 records that started 30 us into the launch ended
    // it at 126 us while the balanced load was 56 us, and 14,000 of the 16,200 workgroups found nothing (tools/bins_timeline.py).
    __shared__ uint32_t curItem, doneBefore;
    const uint32_t itemCount = min(a.binPlan[0], a.binItemCapacity), nBins = a.binsX * a.binsY;
    const uint32_t* binN = a.binPlan + 16, * binSlot = binN + nBins; uint32_t* binDone = a.binPlan + 16 + 2u * nBins;
#ifdef BRMI_TILE_STAMPS
    unsigned long long bph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bprev = __builtin_amdgcn_s_memtime();
    unsigned long long wWait = 0, wWalk = 0, wPrev = 0;
#define BSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); bph[k] += now_ - bprev; bprev = now_; } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif
    if (ALPHA) for (uint32_t i = threadIdx.x; i < 256u; i += BRMI_BIN_THREADS) unormT[i] = (float)i / 255.0f;
    // Items are handed out by ticket, in list order.  An atomic with return on one address serves ~90 per microsecond, so a thousand workgroups
    // asking at once would wait up to 11 us for their first item: the first half of the pool starts on the item of its own index instead.  (Not
    // the whole pool: workgroups that find no room on a CU at first -- the other frame's shading pass is resident -- would sit on the long items
    // of their index until the resident ones run out of tickets, and end the launch with them.)
    const uint32_t staticItems = gridDim.x / 2u;
    uint32_t itemIndex = blockIdx.x;
    if (blockIdx.x >= staticItems) {
        if (threadIdx.x == 0) curItem = staticItems + atomicAdd(&a.binPlan[1], 1u);
        __syncthreads();
        itemIndex = curItem;
    }
    for (; itemIndex < itemCount; ) {
    // the thread's index as the compiler cannot see through: what an item derives from it (tile addresses, scan predicates, merge items) is then
    // recomputed per item with a few VALU instructions instead of being hoisted out of this loop into registers that the pixel loops need (the
    // alpha variant spilled ten such loop invariants)
    uint32_t tid = threadIdx.x;
    if (ALPHA) asm volatile("" : "+v"(tid));
#ifdef BRMI_TILE_STAMPS
    const unsigned long long wgStart = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same clock on every CU: a timeline of the launch's items
#endif
    const uint32_t item = a.binItems[itemIndex];
    const uint32_t bin = item & 0xFFFFu, slice = (item >> 16) & 0xFFu, sliceCount = (item >> 24) + 1u;
    const uint32_t strip = bin % a.binsX, band = bin / a.binsX;
    const uint32_t nAll = binN[bin];
    // the slices of a bin share its records evenly (a multiple of the 32 records one step walks)
    const uint32_t sliceSize = sliceCount == 1u ? nAll : ((nAll + sliceCount - 1u) / sliceCount + 31u) & ~31u;
    const bool shared = sliceCount > 1u;                // other workgroups walk records of this bin too
    const uint32_t first = slice * sliceSize;
    const uint32_t n = max(first, min(nAll, first + sliceSize));      // (the plan's slices are never empty: sliceSize <= binSharedSlice, a multiple of 32)
    if (ALPHA && tid == 0) { taskNext = 0u; rowNext = 0u; }
    for (uint32_t i = tid; i < BIN_W * BIN_ROWS; i += BRMI_BIN_THREADS) tile[i] = BRMI_VIS_EMPTY;
    __syncthreads();
    BSTAMP(0);
    const int x0 = (int)(strip << BIN_W_SHIFT), y0 = (int)(band << BIN_ROWS_SHIFT);
    const LdsSink sink{tile, x0, y0};
    const BinRecord* recs = a.binRecords + (size_t)bin * a.binCapacity;
    // ---- order of the walk.  A wave walks its records in lockstep: a step lasts as long as its longest row, and a record with r rows keeps
    // r of its lanes busy.  The records of a bin arrive in no order (mean 6 rows, widths from 2 to 256 px: timeline and model in
    // profiles/r03_experiments.md -- the slowest wave of a 1024-record slice stepped 1,300 pixels where 300 would do with every lane busy, and
    // the launch ends with those slices).  So the slice is counting-sorted by (rows <= 4 / <= 8 / <= 16, log2 of the pixels a row steps): a
    // record gets 4, 8 or 16 lanes, and the records a wave takes together are about equally wide.  Keys are order-free (64-bit min).
    constexpr uint32_t WIDTH_CLASSES = 6, SORT_CLASSES = 3 * WIDTH_CLASSES;
    const uint32_t m = n - first;
    const bool sorted = BRMI_BIN_SORT && !ALPHA && m >= BRMI_BIN_SORT_MIN && m <= BIN_ORDER_CAP;
    if (sorted) {
        if (tid < SORT_CLASSES) classCount[tid] = 0u;
        __syncthreads();
        uint32_t key[BIN_ORDER_CAP / BRMI_BIN_THREADS], slot[BIN_ORDER_CAP / BRMI_BIN_THREADS];
#pragma unroll
        for (uint32_t k = 0; k < BIN_ORDER_CAP / BRMI_BIN_THREADS; k++) {
            const uint32_t i = tid + k * BRMI_BIN_THREADS;
            key[k] = 0xFFFFFFFFu;
            if (i < m) {
                const BinRecord* r = recs + first + i;
                const uint32_t rows = (r->triAndFlags >> 16) & 0xFFu;
                const int steps = min(r->minX + r->rectWidth, x0 + BIN_W) - r->minX;          // pixels a row's lane steps and walks
                const uint32_t wc = steps <= 8 ? 0u : min(WIDTH_CLASSES - 1u, 29u - (uint32_t)__clz((uint32_t)(steps - 1)));      // <= 8, 16, 32, 64, 128, more
                key[k] = (rows <= 4u ? 0u : rows <= 8u ? 1u : 2u) * WIDTH_CLASSES + wc;
                slot[k] = atomicAdd(&classCount[key[k]], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t c = 0; c < SORT_CLASSES; c++) { classBase[c] = run; run += classCount[c]; }
            classBase[SORT_CLASSES] = run;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < BIN_ORDER_CAP / BRMI_BIN_THREADS; k++) if (key[k] != 0xFFFFFFFFu) order[classBase[key[k]] + slot[k]] = (uint16_t)(tid + k * BRMI_BIN_THREADS);
        __syncthreads();
    }
    // the record of the NEXT step is requested before this step's rows are walked: the records come from HBM (the launch before wrote 86 MiB of
    // them on a dense frame), a step's load is a round trip of its own, and a slice is 32 steps -- the record walk was 75 % of this kernel's
    // wave-cycles (phase stamps).  (Lane = record with the rows re-dealt to the lanes -- an owner byte per row in LDS, the record's fields
    // fetched from the owning lane with ds_bpermute -- walks a dense frame's 3-6-row records 8 % faster on its own, but needs 78 VGPRs
    // instead of 46: beside three k_shade waves per SIMD the kernel then finds no room, and the frame with two in flight got 18 % slower.)
    if (!ALPHA)
    for (uint32_t rc = sorted ? 0u : 2u; rc < 3u; rc++) {
    const uint32_t sh = 2u + rc;                                          // 4, 8 or 16 lanes per record
    const uint32_t sub = tid >> sh, row = tid & ((1u << sh) - 1u), per = BRMI_BIN_THREADS >> sh;
    const uint32_t cs = sorted ? classBase[rc * WIDTH_CLASSES] : 0u, ce = sorted ? classBase[(rc + 1u) * WIDTH_CLASSES] : m;
    auto record_at = [&](uint32_t idx) { return first + (sorted ? (uint32_t)order[idx] : idx); };
    BinRecord pending{}; uint32_t riPending = 0;
    if (cs + sub < ce) { riPending = record_at(cs + sub); pending = recs[riPending]; }
#ifdef BRMI_TILE_STAMPS
    wPrev = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t base = cs; base < ce; base += per) {
        const uint32_t idx = base + sub;
        const BinRecord r = pending;
#ifdef BRMI_TILE_STAMPS
        { uint32_t probe_ = r.clusterIndex; asm volatile("" :: "v"(probe_)); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); wWait += now_ - wPrev; wPrev = now_; }     // the record has arrived
#endif
        if (idx + per < ce) { riPending = record_at(idx + per); pending = recs[riPending]; }
        if (idx >= ce) continue;
        const uint32_t rows = (r.triAndFlags >> 16) & 0xFFu;
        if (row < rows) {
            float sb0 = r.sb0, sb1 = r.sb1;
            for (uint32_t k = 0; k < row; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
            const int py = r.rowStart + (int)row;
            if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1)
                raster_row(sink, NoAlpha{}, py, r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), r.d0, r.d1, r.d2, r.clusterIndex, r.triAndFlags & 0x7Fu,
                           x0, x0 + BIN_W - 1);
        }
#ifdef BRMI_TILE_STAMPS
        { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); wWalk += now_ - wPrev; wPrev = now_; }
#endif
    }
    }   // row classes
    BSTAMP(1);
    if (ALPHA) {
        // Scenes with alpha-tested materials (round 4).  A slice holds at most ALPHA_LIST records here (the plan's slices; launch_raster), in arrival order:
        // opaque and tested triangles mixed, two floor triangles that span the bin beside hundreds of 10-pixel slivers.  One look at every record's
        // header gives two task lists -- the ROWS of the opaque records, and the 16-pixel SEGMENTS of the tested records' rows (a tested pixel
        // costs a texcoord, dependent texel fetches and the filter) -- as exclusive prefix sums in LDS; a task finds its record by bisection and steps
        // its barycentrics from the record's start like every clipped walk, so the keys are the serial loop's.  The waves take tasks 64 at a time
        // from two counters, rows first (more keys in the tile for the tested pixels to lose against untested), segments after, with no barrier in
        // between.  (Before: sixteen lanes per record whatever its rows, the tested records' lanes idle, a barrier, then the segments in a fixed
        // deal: 19 % + 15 % + 10 % of the kernel's wave-cycles went to the record walk and the two waits, San-Miguel-class frame.)
        const uint32_t listed = min(m, ALPHA_LIST);
        if (tid == 0 && m > ALPHA_LIST) atomicAdd(&a.counters[CNT_DROPPED_RECORDS], m - ALPHA_LIST);      // impossible by construction; counted anyway
        constexpr uint32_t PER = ALPHA_LIST / BRMI_BIN_THREADS;
        uint32_t mineA[PER], mineO[PER]; uint32_t sumA = 0, sumO = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t j = tid * PER + k;
            mineA[k] = 0u; mineO[k] = 0u;
            if (j < listed) {
                const BinRecord* r = recs + first + j;
                const uint32_t rows = (r->triAndFlags >> 16) & 0xFFu;
                const int bx0 = max(r->minX, x0), bx1 = min(r->minX + r->rectWidth - 1, x0 + BIN_W - 1);
                if (r->pad1 != 0u) mineA[k] = bx1 < bx0 ? 0u : rows * (uint32_t)(((bx1 - bx0) >> ALPHA_SEG_SHIFT) + 1);
                else mineO[k] = bx1 < bx0 ? 0u : rows;
            }
            sumA += mineA[k]; sumO += mineO[k];
        }
        const uint32_t laneQ = tid & 63u, waveQ = tid >> 6;
        uint32_t inclA = sumA, inclO = sumO;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t va = (uint32_t)__shfl_up((int)inclA, o), vo = (uint32_t)__shfl_up((int)inclO, o);
            if (laneQ >= (uint32_t)o) { inclA += va; inclO += vo; }
        }
        if (laneQ == 63u) { scanPart[waveQ] = inclA; scanPart[BRMI_BIN_THREADS / 64 + waveQ] = inclO; }
        __syncthreads();
        uint32_t runA = inclA - sumA, runO = inclO - sumO;
        for (uint32_t w = 0; w < waveQ; w++) { runA += scanPart[w]; runO += scanPart[BRMI_BIN_THREADS / 64 + w]; }
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) { const uint32_t j = tid * PER + k; if (j <= listed) { taskStart[j] = runA; rowStart[j] = (uint16_t)runO; } runA += mineA[k]; runO += mineO[k]; }
        if (tid == BRMI_BIN_THREADS - 1u && listed == ALPHA_LIST) { taskStart[ALPHA_LIST] = runA; rowStart[ALPHA_LIST] = (uint16_t)runO; }      // j never reaches ALPHA_LIST in the loop above
        __syncthreads();
        const uint32_t total = listed ? taskStart[listed] : 0u, totalRows = listed ? rowStart[listed] : 0u;
        BSTAMP(2);
        // ---- rows of the opaque records
        for (;;) {
            uint32_t tb = 0;
            if (laneQ == 0u) tb = atomicAdd(&rowNext, 64u);
            tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
            if (tb >= totalRows) break;
            const uint32_t task = tb + laneQ;
            if (task < totalRows) {
                uint32_t j = 0;
#pragma unroll
                for (uint32_t step = ALPHA_LIST / 2; step > 0; step >>= 1) if (j + step <= listed && rowStart[j + step] <= task) j += step;
                const BinRecord r = recs[first + j];
                const uint32_t row = task - rowStart[j];
                float sb0 = r.sb0, sb1 = r.sb1;
                for (uint32_t k = 0; k < row; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
                const int py = r.rowStart + (int)row;
                if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1)
                    raster_row(sink, NoAlpha{}, py, r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), r.d0, r.d1, r.d2, r.clusterIndex, r.triAndFlags & 0x7Fu,
                               x0, x0 + BIN_W - 1);
            }
        }
        BSTAMP(1);
#if BRMI_ALPHA_COMPACT
        // Round 4.  A segment's lane used to test its pixels where it found them: of the 64 lanes of a wave a third had a covered pixel whose key could
        // still win at any one step, and the wave ran the sampler (texcoord -> texel addresses -> dependent fetches -> filter: the bulk of this pass,
        // which was 53 % of the kernel's wave-cycles on the San-Miguel-class frame) for them alone, eight times per task.  Now the walk only FINDS
        // such pixels -- coverage, key, a look at the tile -- and appends them (key, texcoord, tile cell, material) to a ring of its wave in LDS
        // (ballot + prefix count: no atomics); whenever 64 are waiting the wave tests them with every lane busy.  The keys and the test are the same;
        // only the order in which keys reach the tile's 64-bit min changes.
        {
            const uint32_t lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
            uint32_t qHead = 0, qTail = 0;      // wave-uniform
            auto drain = [&](uint32_t n) {
                wave_lds_sync();
                if (lane < n) {
                    const uint32_t e = (qHead + lane) & (AQ - 1u);
                    const unsigned long long key = qKey[wave][e];
                    const uint32_t meta = qMeta[wave][e], cell = meta & 0xFFFu;
                    if (key < *(volatile const unsigned long long*)&tile[cell]) {        // (another pixel may have taken the cell since)
                        const AlphaMaterial m = a.alphaMats[meta >> 12];
                        if (!alpha_test_failed(unormT, m, f2{qU[wave][e], qV[wave][e]})) atomicMin(&tile[cell], key);
                    }
                }
                qHead += n;
                wave_lds_sync();
                BSTAMP(6);      // (instrumented builds: the sampling of 64 waiting pixels)
            };
            // (tasks are taken 64 at a time from a counter of the workgroup: segments cost anything from nothing to sixteen sampled pixels, and with a
            // fixed deal the waves waited 14 % of the kernel's wave-cycles for the slowest one at the barrier behind this pass)
            for (;;) {
                uint32_t tb = 0;
                if (lane == 0u) tb = atomicAdd(&taskNext, 64u);
                tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
                if (tb >= total) break;
                const uint32_t task = tb + lane;
                bool on = task < total;
                SegWalk w{0.0f, 0.0f, 0, -1, false};
                float dx0 = 0, dx1 = 0, d0 = 0, d1 = 0, d2 = 0; int py = 0; uint32_t cluster = 0, tri = 0, mat = 0;
                AlphaTri at{};
                if (on) {
                    uint32_t j = 0;
#pragma unroll
                    for (uint32_t step = ALPHA_LIST / 2; step > 0; step >>= 1) if (j + step <= listed && taskStart[j + step] <= task) j += step;
                    const uint32_t ri = first + j;
                    const BinRecord r = recs[ri];
                    const AlphaRecord ar = a.binAlpha[(size_t)bin * a.binCapacity + ri];
                    const int bx0 = max(r.minX, x0), bx1 = min(r.minX + r.rectWidth - 1, x0 + BIN_W - 1);
                    const uint32_t nseg = (uint32_t)(((bx1 - bx0) >> ALPHA_SEG_SHIFT) + 1), local = task - taskStart[j];
                    const uint32_t trow = local / nseg, tseg = local - trow * nseg;
                    const int sx0 = bx0 + (int)(tseg << ALPHA_SEG_SHIFT), sx1 = min(sx0 + (1 << ALPHA_SEG_SHIFT) - 1, bx1);
                    float sb0 = r.sb0, sb1 = r.sb1;
                    for (uint32_t k = 0; k < trow; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
                    py = r.rowStart + (int)trow;
                    on = (uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1;
                    if (on) w = seg_begin(r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), sx0, sx1);
                    dx0 = r.dx_b0; dx1 = r.dx_b1; d0 = r.d0; d1 = r.d1; d2 = r.d2; cluster = r.clusterIndex; tri = r.triAndFlags & 0x7Fu; mat = ar.materialDataIndex; at = ar.tri;
                }
                { uint32_t probe_ = cluster + mat; asm volatile("" :: "v"(probe_)); }
                BSTAMP(7);      // (instrumented builds: a task's record has arrived and its segment is set up)
#pragma nounroll
                for (int k = 0; k < (1 << ALPHA_SEG_SHIFT); k++) {
                    const bool act = on && w.px <= w.x1;
                    if (!__any(act)) break;
                    const float b2 = 1.0f - w.b0 - w.b1;
                    const bool cov = w.all || (w.b0 >= 0.0f && w.b1 >= 0.0f && b2 >= 0.0f);
                    const float depth = w.b0 * d0 + w.b1 * d1 + b2 * d2;
                    const unsigned long long key = (unsigned long long)pack_vis_key(depth, cluster, tri);
                    const uint32_t cell = (uint32_t)((w.px - x0) * BIN_ROWS + (py - y0)) & 0xFFFu;
                    const bool want = act && cov && key < *(volatile const unsigned long long*)&tile[cell];
                    const unsigned long long m = __ballot(want);
                    if (AQ < 128u && qTail - qHead + (uint32_t)__popcll(m) > AQ) { BSTAMP(3); drain(qTail - qHead); }      // (a 64-entry ring: make room first)
                    if (want) {
                        const uint32_t e = (qTail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (AQ - 1u);
                        const f2 uv = pixel_texcoord(at, w.b0, w.b1, b2);
                        qKey[wave][e] = key; qU[wave][e] = uv.x; qV[wave][e] = uv.y; qMeta[wave][e] = cell | (mat << 12);
                    }
                    qTail += (uint32_t)__popcll(m);
                    if (act) { w.b0 += dx0; w.b1 += dx1; w.px++; }
                    if (qTail - qHead >= 64u) { BSTAMP(3); drain(64u); }
                }
                BSTAMP(3);
            }
            if (qTail != qHead) drain(qTail - qHead);
        }
#endif
        BSTAMP(3);
    }
    __syncthreads();
    BSTAMP(4);
    // ---- a slice of a shared bin: park the tile; the slice that arrives last folds the others in and goes on to merge
    bool atomicMerge = false, fold = true;
    if (shared) {
        const uint32_t slot = binSlot[bin];
        if (slot == 0xFFFFFFFFu) atomicMerge = true;      // more shared bins than scratch tiles (counted by the plan): the keys go to L2 one by one
        else {
            // agent-scope stores / loads (write through / read around the XCD's L2: the folding workgroup may sit on another XCD)
            unsigned long long* mine = a.binScratch + ((size_t)slot + slice) * (BIN_W * BIN_ROWS);
            for (uint32_t i = tid; i < BIN_W * BIN_ROWS; i += BRMI_BIN_THREADS) __hip_atomic_store(&mine[i], tile[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every wave's stores have left (vmcnt), the workgroup has met, then the counter: the order the hand-off needs.  (Agent-scope
            // release / acquire fences instead write back and invalidate the XCD's whole L2 per slice: raster 0.22 -> 0.45 ms.)
            // This is the fence-free form MI355X_MICROARCH.md lists as measured on gfx950 ("Hand-offs measured with sc1 loads in place of the
            // acquire", first row): EVERY store of the handed-off bytes is write-through (sc1), every storing wave drains vmcnt, ONE lane adds to an
            // agent-scope counter behind the workgroup's barrier, the consumer is the workgroup whose add returned last, its other waves load
            // behind a barrier that lane joins, and EVERY load of the bytes is an sc1 load to registers.  Not an architectural guarantee of the
            // HIP memory model; tests/test_parity_gpu.py forces shared bins (BRMI_BIN_MIN_SLICE=64) on full frames against the oracle.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) doneBefore = __hip_atomic_fetch_add(&binDone[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            fold = doneBefore + 1u == sliceCount;          // else another slice folds and merges
            if (fold) {
            // (a thread's eight keys of a tile are requested together: the fold is a memory round trip per tile, not per key)
            const unsigned long long* tiles = a.binScratch + (size_t)slot * (BIN_W * BIN_ROWS);
            unsigned long long best[BIN_W * BIN_ROWS / BRMI_BIN_THREADS];
#pragma unroll
            for (uint32_t k = 0; k < BIN_W * BIN_ROWS / BRMI_BIN_THREADS; k++) best[k] = tile[tid + k * BRMI_BIN_THREADS];
            for (uint32_t z = 0; z < sliceCount; z++) {
                if (z == slice) continue;
                unsigned long long o[BIN_W * BIN_ROWS / BRMI_BIN_THREADS];
#pragma unroll
                for (uint32_t k = 0; k < BIN_W * BIN_ROWS / BRMI_BIN_THREADS; k++) o[k] = __hip_atomic_load(&tiles[(size_t)z * (BIN_W * BIN_ROWS) + tid + k * BRMI_BIN_THREADS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (uint32_t k = 0; k < BIN_W * BIN_ROWS / BRMI_BIN_THREADS; k++) best[k] = o[k] < best[k] ? o[k] : best[k];
            }
#pragma unroll
            for (uint32_t k = 0; k < BIN_W * BIN_ROWS / BRMI_BIN_THREADS; k++) tile[tid + k * BRMI_BIN_THREADS] = best[k];
            if (tid == 0) __hip_atomic_store(&binDone[bin], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            }
        }
    }
    // merge: item = (column x, upper / lower 8 rows) = 8 keys = 64 B, contiguous in the tile and in the 8x8-tiled surface
    for (uint32_t item2 = tid; fold && item2 < BIN_W * 2; item2 += BRMI_BIN_THREADS) {
        const uint32_t half = item2 >> 8 /* BIN_W items per half */, xl = item2 & (BIN_W - 1);
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(&tile[xl * BIN_ROWS + half * 8u]);
        ulonglong2 k[4];
        bool any = false;
#pragma unroll
        for (int q = 0; q < 4; q++) { k[q] = src[q]; any = any || k[q].x != BRMI_VIS_EMPTY || k[q].y != BRMI_VIS_EMPTY; }
        if (!any) continue;      // untouched (this also covers columns / rows beyond the target size)
        const uint32_t px = (uint32_t)x0 + xl, py = (uint32_t)y0 + half * 8u;
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(&a.vis[(((py >> 3) * a.tilesX + (px >> 3)) << 6) | ((px & 7u) << 3)]);
        if (atomicMerge) {
            unsigned long long* d1 = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
            for (int q = 0; q < 4; q++) { if (k[q].x != BRMI_VIS_EMPTY) atomicMin(&d1[2 * q], k[q].x); if (k[q].y != BRMI_VIS_EMPTY) atomicMin(&d1[2 * q + 1], k[q].y); }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ulonglong2 g = dst[q];
            const ulonglong2 m2 = make_ulonglong2(g.x < k[q].x ? g.x : k[q].x, g.y < k[q].y ? g.y : k[q].y);
            if (m2.x != g.x || m2.y != g.y) dst[q] = m2;
        }
    }
    BSTAMP(5);
#ifdef BRMI_TILE_STAMPS
    if (tid == 0 && (a.debugFlags & 0x400)) {       // timeline: one entry per work item
        unsigned long long* w = a.debugStamps + 64u + 4u * (size_t)itemIndex;
        w[0] = wgStart; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = (unsigned long long)(n - first) | ((unsigned long long)item << 32); w[3] = wWalk;
    }
#endif
    __syncthreads();                                    // the merge has read the tile (and everyone has read curItem)
    if (tid == 0) curItem = staticItems + atomicAdd(&a.binPlan[1], 1u);
    __syncthreads();
    itemIndex = curItem;
    }   // work items
#ifdef BRMI_TILE_STAMPS
    if ((threadIdx.x & 63u) < 8u && (a.debugFlags & 0x200)) { unsigned long long v = 0; for (int k = 0; k < 8; k++) if ((threadIdx.x & 63u) == (uint32_t)k) v = bph[k]; atomicAdd(a.debugStamps + 32u + (threadIdx.x & 63u), v); }
#endif
}

// The plan of the k_raster_bins launch that follows (the last workgroup of k_raster_overflow): every bin's record count is taken (and cleared
// for the next frame), bins with more than binMinSlice records are cut into slices of at most binSharedSlice and get scratch tiles, and the
// (bin, slice) items are counting-sorted by length, longest first -- the pool of workgroups then ends its launch with short items.
BRMI_DEV void plan_bins(const RasterArgs& a) {
    __shared__ uint32_t classCount[16], classBase[16], tileRun;
    const uint32_t nBinsAll = a.binsX * a.binsY;
    uint32_t* binN = a.binPlan + 16, * binSlot = binN + nBinsAll, * binDone = binSlot + nBinsAll;
    // (round 6: a GPU that renders a row band of the frame only ever fills the bins of the band's bin rows -- the 8-GPU weak frame has 16,320 bins, a rank's band 2,280)
    const uint32_t firstBin = a.stripes.count > 1u ? 0u : min((a.rowLo >> BIN_ROWS_SHIFT) * a.binsX, nBinsAll);
    const uint32_t nBins = a.stripes.count > 1u ? nBinsAll : min(((a.rowHi + BIN_ROWS - 1u) >> BIN_ROWS_SHIFT) * a.binsX, nBinsAll);
    // LDS hand-offs between the block's waves: the LDS operations have to be done, not the global stores (which only this thread reads
    // again, if at all) -- __syncthreads() would wait for those too, a memory round trip per barrier on the next launch's critical path
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (threadIdx.x < 16u) classCount[threadIdx.x] = 0u;
    if (threadIdx.x == 0) tileRun = 0u;
    lds_barrier();
    auto slices_of = [&](uint32_t n) { return n == 0u ? 0u : n <= a.binMinSlice ? 1u : min((n + a.binSharedSlice - 1u) / a.binSharedSlice, 256u); };
    auto class_of = [&](uint32_t n, uint32_t sc) { const uint32_t len = (n + sc - 1u) / sc; return min(15u, 31u - (uint32_t)__clz(len)); };      // log2 of the slice length, 16 classes (a slice of exactly 65536 records -- BRMI_BIN_CAPACITY = BRMI_BIN_MIN_SLICE = 65536 -- shares the last)
    constexpr uint32_t K = 8;
    const bool oneChunk = nBins - firstBin <= K * blockDim.x;      // (every frame up to 8K: a thread keeps its bins' counts in registers between the passes)
    uint32_t n[K];
    for (uint32_t b0 = firstBin; b0 < nBins; b0 += K * blockDim.x) {
#pragma unroll
        for (uint32_t k = 0; k < K; k++) { const uint32_t b = b0 + k * blockDim.x + threadIdx.x; n[k] = b < nBins ? min(a.binCounts[(size_t)b * BIN_COUNT_STRIDE], a.binCapacity) : 0u; }      // eight loads in flight
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t b = b0 + k * blockDim.x + threadIdx.x;
            if (b >= nBins) continue;
            const uint32_t sc = slices_of(n[k]);
            a.binCounts[(size_t)b * BIN_COUNT_STRIDE] = 0u; binN[b] = n[k]; binDone[b] = 0u;
            uint32_t slot = 0xFFFFFFFFu;
            if (sc > 1u) {      // (any order: a tile range per shared bin)
                // (an LDS compare-and-swap loop that reserves only what fits -- so that one oversized bin does not push the bins behind it onto the atomic merge -- made this
                // block, which the next launch waits for, twice as slow: plan 16 -> 34 us on the Bistro-class frame.  Demand beyond the scratch tiles is reported: binPlan[2].)
                const uint32_t base = atomicAdd(&tileRun, sc); if (base + sc <= a.binScratchTiles) slot = base;
            }
            binSlot[b] = slot;
            if (sc != 0u) atomicAdd(&classCount[class_of(n[k], sc)], sc);
        }
    }
    lds_barrier();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int c = 15; c >= 0; c--) { classBase[c] = run; run += classCount[c]; } a.binPlan[0] = run; a.binPlan[1] = 0u; a.binPlan[2] = tileRun; }
    lds_barrier();
    for (uint32_t b0 = firstBin; b0 < nBins; b0 += K * blockDim.x) {
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t b = b0 + k * blockDim.x + threadIdx.x;
            if (b >= nBins) continue;
            const uint32_t nb = oneChunk ? n[k] : binN[b], sc = slices_of(nb);
            if (sc == 0u) continue;
            const uint32_t at = atomicAdd(&classBase[class_of(nb, sc)], sc);
            // (binItemCapacity is the worst case -- every bin cut into slices of 32 -- up to 2^22 items: only frames beyond 16K x 16K can run out, and say so)
            if (at + sc > a.binItemCapacity) atomicAdd(&a.counters[CNT_DROPPED_RECORDS], nb);
            for (uint32_t z = 0; z < sc; z++) if (at + z < a.binItemCapacity) a.binItems[at + z] = b | (z << 16) | ((sc - 1u) << 24);
        }
    }
}

// Records that did not fit their bin: four per wave64, one lane per row, global 64-bit atomics.  Runs BEFORE k_raster_bins (whose plain
// read-modify-write merges then see these keys like k_raster's own); its last workgroup writes that launch's plan.  The queue lengths are
// cleared with the frame's counters and, between the two raster phases, by k_seed_phase2.
// THREADS: 256, or 1024 for surfaces of more than 4096 bins (an 8K frame has 8,100: the plan is ONE workgroup's work and was 51 us of the Zorah-class frame's chain)
template <bool ALPHA, uint32_t THREADS = 256>
__global__ void __launch_bounds__(THREADS) k_raster_overflow(RasterArgs a) {
    wave_prio<PRIO_BINS>();
#ifdef BRMI_TILE_STAMPS
    if (blockIdx.x == 0u) {        // (instrumented builds: how long the plan takes, in 10 ns units; slots 40 / 41 of the stamp words)
        const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();
        plan_bins(a);
        __syncthreads();
        if (threadIdx.x == 0 && (a.debugFlags & 0x400)) { atomicAdd(a.debugStamps + 40u, __builtin_amdgcn_s_memrealtime() - t0_); atomicAdd(a.debugStamps + 41u, 1ull); }
        return;
    }
#endif
    if (blockIdx.x == 0u) {       // (the first workgroup: it is what the next launch waits for, so it should not queue behind the walkers)
        if (a.wideFeedback && threadIdx.x == 0u) __hip_atomic_store(a.wideFeedback, a.counters[a.wideCounter], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        plan_bins(a); return;
    }
    // a wave = four records at a time, one lane per row
    const uint32_t lane = threadIdx.x & 63u, sub = lane >> 4, row = lane & 15u;
    const uint32_t walker = (blockIdx.x - 1u) * (THREADS / 64u) + (threadIdx.x >> 6), walkers = (gridDim.x - 1u) * (THREADS / 64u);
    __shared__ float unormT[ALPHA ? 256 : 1];
    if (ALPHA) { if (threadIdx.x < 256u) unormT[threadIdx.x] = (float)threadIdx.x / 255.0f; __syncthreads(); }
    // lane = stripe: all 64 queue lengths with one load; nearly every frame has none
    const uint32_t mine = min(a.counters[CNT_STRIPES + lane * CNT_STRIPE_WORDS + STRIPE_OVERFLOW], a.overflowPerStripe);
    uint64_t busy = __ballot(mine != 0u);
    while (busy != 0ull) {
        const uint32_t stripe = (uint32_t)__ffsll((unsigned long long)busy) - 1u;
        busy &= busy - 1ull;
        const uint32_t n = (uint32_t)__shfl((int)mine, (int)stripe);
        for (uint32_t base = walker * 4u; base < n; base += walkers * 4u) {
            const uint32_t ri = base + sub;
            if (ri >= n) continue;
            const BinRecord r = a.overflow[(size_t)stripe * a.overflowPerStripe + ri];
            if (ALPHA && r.pad1 != 0u) raster_record_global(a, r, tex_alpha_of(a, unormT, a.overflowAlpha[(size_t)stripe * a.overflowPerStripe + ri]), r.pad0, row, 16u);
            else raster_record_global(a, r, NoAlpha{}, r.pad0, row, 16u);
        }
    }
}


__global__ void __launch_bounds__(256) k_depth_copy(const unsigned long long* vis, float* depth, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = vis[i];
        depth[i] = (k == BRMI_VIS_EMPTY) ? as_f32(BRMI_DEPTH_EMPTY_BITS) : as_f32(((uint32_t)(k >> BRMI_VIS_META_BITS)) << 1);
    }
}

// ---- Round 6: the draw list's RE-TEST -----------------------------------------------------------------------------------------------------
// The reference rasterises every cluster of the visible list (SoftwareRasterizeClustersPass1, CLodExtension.cpp:1920-2088); four in five of them own no pixel of the
// frame -- hidden behind nearer clusters the occlusion test's sphere-against-four-texels let through (profiles/r05_experiments.md, r06_experiments.md).  A cluster
// that cannot win a pixel of the phase-1 depth image need not be drawn for the image to be the reference's: the 64-bit min only ever lowers a key.  So phase 1
// draws the clusters its culling PREDICTED visible (the draw list), builds the depth chain from the keys that leaves, and then this kernel looks at every held
// cluster once: the box of its vertices (MeshletBox) through the frame's own object-to-clip matrix -- the rasteriser's arithmetic for the corners -- gives a pixel
// rectangle that contains every pixel any triangle of the cluster can touch and a depth no vertex is nearer than; when every texel of the chain over that
// rectangle (a texel = the FARTHEST key depth of its pixels, "empty" where one has no key) is nearer, the cluster is skipped for good, else it goes to the late
// list and the late pass draws it before anything reads the phase-1 depth.  The phase-1 keys are then exactly those of drawing everything: phase 2, the chain
// (rebuilt where the late pass drew) and every later stage see the reference's frame -- the whole parity suite runs with this on.  The prediction may be anything;
// THIS test is what has to be conservative: box_behind_chain (brmi_internal.h) states how.
// A lane per held cluster: the shape of the cluster cull (eight corner transforms and up to maxTexels^2 independent 4 B loads per lane).
struct RetestArgs {
    const HeldRecord* held; const MeshletBox* boxes; const float* objConst; const brmi_view_raster_info* viewRasterInfo;
    HzbDesc hzb; uint32_t visW, visH, capacity, maxTexels, bandY0, bandY1;      // bandY0 .. bandY1: the rows this GPU renders (the rectangle's clamp)
    uint32_t* counters; uint32_t* lateList; uint32_t* heldFeedback;      // (host-mapped word or null: the held count, for the grids of the frames that follow)
};
__global__ void __launch_bounds__(256) k_retest_held(RetestArgs a) {
    wave_prio<PRIO_RASTER>();
    const uint32_t n = min(a.counters[CNT_HELD1], a.capacity);
    if (a.heldFeedback && blockIdx.x == 0u && threadIdx.x == 0u) __hip_atomic_store(a.heldFeedback, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const brmi_view_raster_info ri = a.viewRasterInfo[0];      // (single view: ClusterSetup::viewId is the main camera's)
    const BoxViewport vp{(float)(ri.scissorMaxX - ri.scissorMinX), (float)(ri.scissorMaxY - ri.scissorMinY), (float)ri.scissorMinX, (float)ri.scissorMinY,
                         max((int)ri.scissorMinX, 0), max(max((int)ri.scissorMinY, 0), (int)a.bandY0), min((int)ri.scissorMaxX - 1, (int)a.visW - 1), min(min((int)ri.scissorMaxY - 1, (int)a.visH - 1), (int)a.bandY1 - 1)};
    unsigned long long lateVT = 0ull;
    const uint32_t rounded = (n + 63u) & ~63u;      // wave-uniform trip count (wave_append inside)
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < rounded; g += gridDim.x * blockDim.x) {
        const bool on = g < n;
        HeldRecord hr{0u, 0u, 0u, 0u};
        bool late = false;
        if (on) {
            hr = a.held[g];
            const MeshletBox bx = a.boxes[hr.boxIndex];
            const float* oc = a.objConst + (size_t)hr.perObjectIndex * OBJ_CONST_FLOATS;
            late = !box_behind_chain(a.hzb, bx, oc, oc + 32, vp, a.maxTexels);
        }
        // the late list: one atomic per wave
        const bool append = on && late;
        const uint32_t slot = wave_append(&a.counters[CNT_LATE1], append);
        if (append) { a.lateList[slot] = hr.clusterIndex; lateVT += (unsigned long long)(hr.vertsTris & 0xFFFFu) | ((unsigned long long)(hr.vertsTris >> 16) << 32); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lateVT += (unsigned long long)__shfl_xor((long long)lateVT, o);
    if ((threadIdx.x & 63u) == 0u && lateVT != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(&a.counters[CNT_DRAWN_VT]), lateVT);
}

int launch_clear(brmi_pass* p, hipStream_t s) {
    // inside brmi_execute the culling pass follows immediately: take its frame clear along (frameClearBytes is a multiple of 256)
    uint4* frameState = p->fuseFrameClear ? p->wsPtr<uint4>(p->ws.counters) : nullptr;
    hipLaunchKernelGGL(k_clear_vis, dim3(2048), dim3(256), 0, s, static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel, p->bandPixelCount,
                       frameState, frameState ? p->ws.frameClearBytes / 16 : 0ull);
    p->frameStateCleared = p->fuseFrameClear;
    BRMI_LAUNCH_CHECK(p, "k_clear_vis");
    return BRMI_OK;
}

static bool stripe_count_on(const brmi_pass* p) { return p->stripes.count > 1u; }

int launch_raster(brmi_pass* p, uint32_t phase, hipStream_t s) {
    if (phase != 1 && phase != 2) return fail(p, BRMI_ERR_INVALID, "brmi_raster: phase %u (1 or 2)", phase);
    if (phase == 2 && !p->cfg.enableOcclusionCulling) return fail(p, BRMI_ERR_STATE, "brmi_raster: phase 2 needs a pass created with enableOcclusionCulling");
    RasterArgs a;
    a.sc = p->scene; a.clusters = static_cast<const uint4*>(p->res[BRMI_RES_VISIBLE_CLUSTERS]); a.counters = p->counters();
    a.setup = p->wsPtr<ClusterSetup>(p->ws.clusterSetup);
    a.firstCounter = 0xFFFFFFFFu; a.countCounter = CNT_VISIBLE;
    a.drawList = nullptr; a.countFeedback = nullptr; a.generalList = nullptr; a.generalCounter = CNT_GENERAL1; a.generalFeedback = nullptr; a.bigQueue = nullptr; a.bigRuns = nullptr; a.bigCapacity = 0u; a.bigCounter = CNT_BIG1; a.emitWideEntries = p->leanWideEntries;
    // round 6: a frame whose culling held clusters back (launch_cull) rasterises its draw list here, re-tests the held clusters against the keys that leaves, and
    // draws the ones it cannot prove hidden in a late pass -- all inside this stage, before anything reads the phase-1 depth (k_retest_held)
    const bool hold = phase == 1 && p->holdThisFrame;
    if (hold) { a.drawList = p->wsPtr<uint32_t>(p->ws.drawList); a.countCounter = CNT_DRAW1; }
    // round 6: triangles that reach very many bins are queued for k_raster_wide -- launched while the frames before had such triangles (host-mapped word 7: phase 1's
    // count, stored by the plan; a frame or two old, and either way the same records)
    const uint32_t lastWide = p->phase2FeedbackHost ? reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost)[7] : 0u;
    const bool wideOn = p->wideCapacity != 0u && lastWide >= std::max(1u, p->wideMinTriangles);
    a.wideQueue = wideOn ? p->wsPtr<WideTri>(p->ws.wideQueue) : nullptr; a.wideAlpha = p->wsPtr<AlphaRecord>(p->ws.wideAlpha); a.wideCapacity = p->wideCapacity;
    a.wideCounter = phase == 2 ? (uint32_t)CNT_WIDE2 : (uint32_t)CNT_WIDE1; a.wideEntries = p->wideEntries;
    a.wideFeedback = (phase == 1 && p->phase2FeedbackDev) ? p->phase2FeedbackDev + 7 : nullptr;
    const dim3 wgrid(std::max(64u, std::min(8192u, lastWide * 8u)));      // eight single-wave workgroups per queued triangle (WIDE_SHARES)
    // (the interleaved partition's surface rows are not the frame rows the boxes are in: there the second build redoes everything)
    a.chainDirty = (BRMI_CHAIN_DIRTY_BLOCKS && phase == 2 && p->stripes.count <= 1u) ? p->wsPtr<uint8_t>(p->ws.chainDirty) : nullptr; a.chainBlocksX = (p->cfg.width + 31u) / 32u;
    p->chainDirtyTracked = a.chainDirty != nullptr;
    if (phase == 2) { a.firstCounter = CNT_VISIBLE; a.countCounter = CNT_VISIBLE2; }   // clusters [visible1, visible1 + visible2)
    a.vis = static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]);
    a.visW = p->cfg.width; a.visH = p->frameHeight(); a.tilesX = p->tilesX; a.bandY0 = p->bandY0; a.bandY1 = p->bandY1;
    a.stripes = p->stripes; a.rowLo = stripe_count_on(p) ? 0u : p->bandY0; a.rowHi = stripe_count_on(p) ? p->frameHeight() : p->bandY1;
    a.binRecords = p->wsPtr<BinRecord>(p->ws.binRecords); a.binCounts = p->wsPtr<uint32_t>(p->ws.binCounts);
    a.binCapacity = p->binCapacity; a.binsX = p->binsX; a.binsY = p->binsY;
    a.overflow = p->wsPtr<BinRecord>(p->ws.binOverflow); a.overflowPerStripe = p->binOverflowPerStripe;
    a.objConst = p->wsPtr<float>(p->ws.objConst);
    a.bigTriArea = p->bigTriArea; a.bigTriAreaAlpha = p->bigTriAreaAlpha; a.debugFlags = p->rasterDebug;
    a.bigTriAreaDense = p->bigTriAreaDense; a.denseClusterCount = p->denseClusterCount;
    // alpha-tested scenes: a slice's alpha records all go through the task list of k_raster_bins<true> (BRMI_ALPHA_LIST entries), so no slice is longer than that
    a.binMinSlice = p->sceneHasAlphaTest ? std::min(p->binMinSlice, (uint32_t)BRMI_ALPHA_LIST) : p->binMinSlice;
    a.binSharedSlice = std::max(32u, std::min(p->binSharedSlice, a.binMinSlice) & ~31u);        // a multiple of the 32 records a step walks: no slice of the plan is empty
    a.binPlan = p->wsPtr<uint32_t>(p->ws.binPlan); a.binItems = p->wsPtr<uint32_t>(p->ws.binItems); a.binScratch = p->wsPtr<unsigned long long>(p->ws.binScratch);
    a.binScratchTiles = p->binScratchTiles; a.binItemCapacity = p->binItemCapacity;
    // the LDS window k_raster counts its records per bin in: every bin of the surface when that is at most 2048 cells (8 KB), else 2048 (BRMI_BIN_TABLE: 256 = the round-3 window only)
    static const uint32_t tableEnv = (uint32_t)std::max(256l, std::min(2048l, experiment("bin_table", BIN_TABLE_MAX)));
    a.tableCells = std::max<uint32_t>(BIN_WINDOW, std::min<uint32_t>(tableEnv, (p->binsX * p->binsY + 63u) & ~63u));
    a.binAlpha = p->wsPtr<AlphaRecord>(p->ws.binAlpha); a.overflowAlpha = p->wsPtr<AlphaRecord>(p->ws.overflowAlpha);
    a.clusterUv = p->wsPtr<ClusterUv>(p->ws.clusterUv);
    a.alphaMats = p->wsPtr<AlphaMaterial>(p->ws.alphaMats);
    // (k_raster_bins<true> packs the material index of a waiting pixel with its 12-bit tile cell into one word)
    if (p->sceneHasAlphaTest && p->scene.materialCount > (1u << 20)) return fail(p, BRMI_ERR_INVALID, "brmi_raster: %u materials in a scene with alpha-tested ones (at most %u)", p->scene.materialCount, 1u << 20);
    if (p->sceneHasAlphaTest) if (int rc = ensure_frame_constants(p, s)) return rc;
    a.debugStamps = p->wsPtr<unsigned long long>(p->ws.debugStamps);
    // the pool of k_raster_bins: four 512-thread workgroups per CU is what the LDS holds; phase 2 rarely has an item at all
    // Round 4: phase-2 launches are sized by what the host last saw phase 2 draw (the host-mapped word of the ranking kernel, read without a wait: a frame
    // or two old; every kernel here strides its grid or takes items by ticket, so any size gives the same keys).  Beside another frame's shading half a
    // wave that finds nothing still has to find a slot -- 200 VGPRs for k_raster<true>, 66 KB of LDS for a k_raster_bins<true> workgroup -- and the three
    // phase-2 launches of a still camera cost the San-Miguel-class frame ~150 us of its geometry chain in flight (kernel stats: k_raster<true> 61 us,
    // k_raster_overflow<true> 53, k_raster_bins<true> 150 per launch on average, phase 1 and 2 alike).
    static const bool sizeByHint = experiment("phase2_sized", 1) != 0;
    uint32_t hint2 = 0xFFFFFFFFu;
    if (phase == 2 && sizeByHint && p->phase2FeedbackHost) hint2 = *reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost);
    auto pow2_at_least = [](uint32_t v) { uint32_t r = 1; while (r < v && r < (1u << 30)) r <<= 1; return r; };
    const bool sized2 = hint2 < 128u;
    const dim3 bgrid(phase == 2 ? (sized2 ? std::max(16u, std::min(256u, pow2_at_least(hint2 * 2u))) : std::min(p->binGrid, 256u)) : p->binGrid);
    // Phase 2 draws what phase 1's stale depth chain hid: nothing with a still camera, tens of clusters with a slowly moving one, a thousand
    // with a fast one.  While the last count the host has seen (a host-mapped word the phase-2 ranking kernel stores, read here without any
    // wait: a frame or two old) is small, the triangles all take the row re-deal with global atomics -- no records, so no plan and no bins
    // launch: one launch instead of three on a chain of small launches that each wait for slots while another frame shades.  Above the
    // limit the three launches pay (1,000 clusters: raster2 0.068 against 0.121 ms).  Either way the same keys.
    // (not in scenes with alpha-tested materials: a tested pixel costs three times as much through the global chain as in a bin, and the
    // San-Miguel-class camera path got 9 % slower with 19 phase-2 clusters per frame)
    const bool direct2 = phase == 2 && !p->sceneHasAlphaTest && p->phase2FeedbackHost && *reinterpret_cast<volatile uint32_t*>(p->phase2FeedbackHost) <= p->phase2DirectMax && p->phase2DirectMax != 0u;
    if (direct2) a.bigTriArea = a.bigTriAreaAlpha = a.bigTriAreaDense = 0x3FFFFFFF;
    // phase 2 rarely has more than a handful of clusters: 2048 workgroups (the kernel strides; two waves per SIMD) start and retire a little
    // faster than 8192 that find nothing (-3 us per frame)
    static const uint32_t grid2 = (uint32_t)std::max(64l, experiment("raster_grid2", 2048));
    const dim3 rgrid(phase == 2 ? std::min(p->rasterGrid, sized2 ? std::max(128u, std::min(grid2, pow2_at_least(hint2 * 16u))) : grid2) : p->rasterGrid);
    const dim3 ogrid(phase == 2 && sized2 ? std::max(2u, std::min(129u, hint2 / 4u + 2u)) : 129u);      // (block 0 plans the bins launch; the others walk the overflow queues)
    if (p->sceneHasAlphaTest) {
        hipLaunchKernelGGL(k_raster<true>, rgrid, dim3(64), a.tableCells * 4u, s, a);
        if (!direct2 && wideOn) hipLaunchKernelGGL(k_raster_wide<true>, wgrid, dim3(64), 0, s, a);
        if (!direct2) hipLaunchKernelGGL(k_raster_overflow<true>, ogrid, dim3(256), 0, s, a);
        if (!direct2 && !(p->rasterDebug & 4)) hipLaunchKernelGGL(k_raster_bins<true>, bgrid, dim3(BRMI_BIN_THREADS), 0, s, a);
    } else {
        // round 6: the lean form for frames of very many clusters (k_raster<false, true>), the general launch behind it for what it leaves.  Decided from what the host last
        // saw of such launches (host-mapped words 8 / 9: the main launch's cluster count, the general launch's; a frame or two old -- either way the same keys).
        volatile uint32_t* fb = p->phase2FeedbackHost;
        bool lean = false;
        if (phase == 1 && fb && p->leanMinClusters != 0u && p->stripes.count <= 1u) {
            const uint32_t lastCount = fb[8], lastGeneral = fb[9];
            if (p->leanActive) {
                if (lastCount < p->leanMinClusters) p->leanActive = false;
                else if ((uint64_t)lastGeneral * 100u > (uint64_t)lastCount * p->leanMaxGeneralPct) { p->leanActive = false; p->leanRetryIn = 64u; }
            } else if (p->leanRetryIn != 0u) p->leanRetryIn--;
            else if (lastCount >= p->leanMinClusters) { p->leanActive = true; fb[9] = 0u; }
            lean = p->leanActive;
            a.countFeedback = p->phase2FeedbackDev + 8;
        }
        if (phase == 1) p->leanLastLaunch = lean;
        if (lean) {
            a.generalList = p->wsPtr<uint32_t>(p->ws.generalList); a.generalCounter = CNT_GENERAL1; a.generalFeedback = p->phase2FeedbackDev + 9;
            a.bigQueue = p->wsPtr<WideTri>(p->ws.bigQueue); a.bigRuns = p->wsPtr<uint2>(p->ws.bigRuns); a.bigCapacity = p->leanQueue / 64u; a.bigCounter = CNT_BIG1;
            hipLaunchKernelGGL((k_raster<false, true>), dim3(p->leanGrid), dim3(64), 0, s, a);
            hipLaunchKernelGGL(k_raster_emit, dim3(p->leanEmitGrid), dim3(64), a.tableCells * 4u, s, a);
            RasterArgs g = a;
            g.drawList = a.generalList; g.firstCounter = 0xFFFFFFFFu; g.countCounter = CNT_GENERAL1; g.countFeedback = p->phase2FeedbackDev + 9;
            const dim3 ggrid(std::min(p->rasterGrid, std::max(128u, pow2_at_least(std::min((uint32_t)fb[9], 1u << 20) * 16u))));
            hipLaunchKernelGGL(k_raster<false>, ggrid, dim3(64), g.tableCells * 4u, s, g);
        } else
        hipLaunchKernelGGL(k_raster<false>, rgrid, dim3(64), a.tableCells * 4u, s, a);
        if (!direct2 && wideOn) hipLaunchKernelGGL(k_raster_wide<false>, wgrid, dim3(64), 0, s, a);
        if (!direct2) {
            if (p->binsX * p->binsY > 4096u) hipLaunchKernelGGL((k_raster_overflow<false, 1024>), dim3((ogrid.x - 1u + 3u) / 4u + 1u), dim3(1024), 0, s, a);
            else hipLaunchKernelGGL(k_raster_overflow<false>, ogrid, dim3(256), 0, s, a);
        }
        if (!direct2 && !(p->rasterDebug & 4)) hipLaunchKernelGGL(k_raster_bins<false>, bgrid, dim3(BRMI_BIN_THREADS), 0, s, a);
    }
    BRMI_LAUNCH_CHECK(p, "k_raster");
    if (hold) {
        // the chain of the keys the draw list left (LinearDepthCopyPass1 + LinearDepthDownsamplePass1's work, done here; the frame's own build after this stage then only
        // redoes the blocks the late pass touched: launch_hzb)
        p->chainBuiltInRaster = false;
        if (int rc = launch_hzb(p, s, true, false)) return rc;
        volatile uint32_t* fb = p->phase2FeedbackHost;      // words 5 / 6: the late and the held count of the frames before (no wait; any value gives the same keys)
        const uint32_t lastLate = fb ? fb[5] : 0u, lastHeld = fb ? fb[6] : 0xFFFFFFFFu;
        RetestArgs r;
        r.held = p->wsPtr<HeldRecord>(p->ws.heldRecords); r.boxes = p->wsPtr<MeshletBox>(p->ws.meshletBoxes); r.objConst = a.objConst; r.viewRasterInfo = p->scene.viewRasterInfo;
        r.hzb = p->hzbDesc(); r.visW = a.visW; r.visH = a.visH; r.capacity = p->cfg.maxVisibleClusters; r.maxTexels = p->retestMaxTexels; r.bandY0 = p->bandY0; r.bandY1 = p->bandY1;
        r.counters = p->counters(); r.lateList = p->wsPtr<uint32_t>(p->ws.lateList); r.heldFeedback = p->phase2FeedbackDev ? p->phase2FeedbackDev + 6 : nullptr;
        const uint32_t tgrid = std::max(16u, std::min(4096u, pow2_at_least(std::min(lastHeld, p->cfg.maxVisibleClusters) / 256u + 1u)));
        hipLaunchKernelGGL(k_retest_held, dim3(tgrid), dim3(256), 0, s, r);
        RasterArgs l = a;
        l.drawList = r.lateList; l.countCounter = CNT_LATE1; l.countFeedback = p->phase2FeedbackDev ? p->phase2FeedbackDev + 5 : nullptr;
        l.wideCounter = CNT_WIDE1B; l.wideFeedback = nullptr;
        // what the late pass draws changes the depth under it: it records the 32 x 32 px blocks its triangles may touch, like phase 2
        l.chainDirty = BRMI_CHAIN_DIRTY_BLOCKS ? p->wsPtr<uint8_t>(p->ws.chainDirty) : nullptr; l.chainBlocksX = (p->cfg.width + 31u) / 32u;
        p->chainDirtyTracked = l.chainDirty != nullptr; p->chainBuiltInRaster = true;
        // few late clusters (a still or slowly moving camera: the prediction and the re-test ask the same question of nearly the same depth): every triangle through the
        // row re-deal with global atomics, ONE launch, as the small phase 2; many: records, plan and bins once more
        const bool directLate = lastLate <= (p->sceneHasAlphaTest ? 0u : p->lateDirectMax);
        if (directLate) l.bigTriArea = l.bigTriAreaAlpha = l.bigTriAreaDense = 0x3FFFFFFF;
        const dim3 lgrid(std::min(p->rasterGrid, std::max(128u, pow2_at_least(std::min(lastLate, 1u << 20) * 16u))));
        if (p->sceneHasAlphaTest) {
            hipLaunchKernelGGL(k_raster<true>, lgrid, dim3(64), l.tableCells * 4u, s, l);
            if (!directLate && wideOn) hipLaunchKernelGGL(k_raster_wide<true>, wgrid, dim3(64), 0, s, l);
            if (!directLate) { hipLaunchKernelGGL(k_raster_overflow<true>, dim3(129), dim3(256), 0, s, l); hipLaunchKernelGGL(k_raster_bins<true>, dim3(p->binGrid), dim3(BRMI_BIN_THREADS), 0, s, l); }
        } else {
            hipLaunchKernelGGL(k_raster<false>, lgrid, dim3(64), l.tableCells * 4u, s, l);
            if (!directLate && wideOn) hipLaunchKernelGGL(k_raster_wide<false>, wgrid, dim3(64), 0, s, l);
            if (!directLate) {
                if (p->binsX * p->binsY > 4096u) hipLaunchKernelGGL((k_raster_overflow<false, 1024>), dim3(33), dim3(1024), 0, s, l);
                else hipLaunchKernelGGL(k_raster_overflow<false>, dim3(129), dim3(256), 0, s, l);
                hipLaunchKernelGGL(k_raster_bins<false>, dim3(p->binGrid), dim3(BRMI_BIN_THREADS), 0, s, l);
            }
        }
        BRMI_LAUNCH_CHECK(p, "late raster pass");
    }
    return BRMI_OK;
}

int launch_depth_copy(brmi_pass* p, hipStream_t s) {
    hipLaunchKernelGGL(k_depth_copy, dim3(2048), dim3(256), 0, s, static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel,
                       static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]) + p->bandFirstPixel, p->bandPixelCount);
    BRMI_LAUNCH_CHECK(p, "k_depth_copy");
    return BRMI_OK;
}

}  // namespace brmi
