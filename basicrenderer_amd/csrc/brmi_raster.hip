// brmi_raster.hip -- visibility-buffer clear, software rasteriser (K5) and depth copy (K6) for gfx950.
//
// Computes what SWRasterCluster does (BR/shaders/ClusterLOD/softwareRaster.hlsl:290-612): vertices
// of one meshlet to screen space, per-triangle edge-function setup, inclusive coverage at pixel
// centres with incrementally stepped barycentrics, 64-bit min of (depth | cluster | triangle).
// Scheduling is MI355X-first rather than the reference's one-128-thread-group-per-cluster indirect
// dispatch per raster bucket:
//   * persistent single-wave workgroups pull cluster indices from a device-side queue counter
//     (cluster count lives in HBM; no indirect dispatch, no host read-back, natural load balance
//     over clusters of very different pixel area);
//   * wave64: lane = triangle, two passes over a 128-triangle meshlet.  The reference's
//     WaveActiveAnyTrue(rectWidth > 4) vote (softwareRaster.hlsl:502) is evaluated per pass over the
//     lanes that survived setup, which is exactly the wave composition of a 128-thread group on
//     wave64 hardware;
//   * the meshlet's screen-space vertices (<=128 x 12 B) are staged once in LDS;
//   * the visibility surface is stored in 8x8 tiles (512 B = 4 cache lines) so the atomics of a
//     meshlet footprint, and every later full-screen pass, touch whole lines.
// Raster buckets (K4) collapse: with one PSO-free kernel there is nothing to sort by.
#include "brmi_device.h"
#include "brmi_internal.h"

namespace brmi {

// A triangle whose clamped bounding box exceeds BIG_TRI_AREA pixels is not walked by its lane: the lane
// emits one record per 64-row chunk (carrying the exactly stepped scanline start of the chunk) and
// k_raster_big walks those rows one lane per row.  Same arithmetic per pixel, different lane mapping.
struct BigTriRecord {        // 64 B
    uint32_t clusterIndex, triAndFlags;      // tri | useScanlineRanges << 8 | rowCount << 16
    int32_t  minX, rectWidth, rowStart;
    float    sb0, sb1, dx_b0, dx_b1, dy_b0, dy_b1, d0, d1, d2;
    uint32_t pad0, pad1;
};
constexpr int RASTER_SEG = 256;        // longest run of pixels one lane walks in the row-parallel kernel
constexpr int REC_ROWS = 16;           // rows per big-triangle record: k_raster_big runs four records per wave64

struct RasterArgs {
    BigTriRecord* bigTris; uint32_t bigTriCapacity;
    const float* objConst;   // per object: MVP (16), objectToClip (16), modelViewZ (4)
    int bigTriArea;          // clamped-bbox pixels above which a triangle goes to the row-parallel kernel
    brmi_scene_buffers sc;
    const uint4* clusters;
    uint32_t* counters;
    uint32_t firstCounter, countCounter;   // counter indices: first cluster (0xFFFFFFFF = 0) and cluster count
    uint32_t* queue;                        // work-queue head (zeroed before launch)
    unsigned long long* vis;
    uint32_t visW, visH, tilesX, bandY0, bandY1;
};

__global__ void __launch_bounds__(256) k_clear_vis(unsigned long long* vis, uint64_t n) {
    // 16 B per lane per store (n is a multiple of 64)
    ulonglong2* v2 = reinterpret_cast<ulonglong2*>(vis);
    const uint64_t n2 = n >> 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) v2[i] = make_ulonglong2(BRMI_VIS_EMPTY, BRMI_VIS_EMPTY);
}

BRMI_DEV void clip_scanline(float value, float step, int& first, int& last, bool& has) {
    if (!has) return;
    if (step > 0.0f) { const int c = to_int_sat(ceilf(-value / step)); first = first > c ? first : c; }
    else if (step < 0.0f) { const int f = to_int_sat(floorf(value / -step)); last = last < f ? last : f; }
    else has = value >= 0.0f;
    has = has && first <= last;
}

// Visibility write.  MODE 0 is the product path (64-bit atomic min).  MODE 3 reads the key first and skips the
// atomic when it cannot win (the stored key only ever decreases, so a stale read is merely conservative).
// MODEs 1 (plain store) and 2 (no write) exist for bandwidth experiments only and give wrong images.
template <int MODE>
BRMI_DEV void emit_key(unsigned long long* addr, unsigned long long key, unsigned long long& sink) {
    if (MODE == 0) atomicMin(addr, key);
    else if (MODE == 1) *addr = key;
    else if (MODE == 2) sink ^= key;
    else { if (key < __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(addr, key); }
}

// One scanline of one triangle (softwareRaster.hlsl:506-609), barycentrics at the row start given.
// `seg` >= 0 restricts the walk to the seg-th run of RASTER_SEG pixels after the row's first covered pixel;
// the barycentrics are still stepped pixel by pixel from the row start, so every value is the one the
// serial loop produces.  Surfaces are 8x8 tiles stored column-major inside the tile (8 vertically adjacent
// pixels are one contiguous 64 B run: lanes that own neighbouring rows hit the same cache line).
template <int MODE>
BRMI_DEV void raster_row(unsigned long long& sink, unsigned long long* vis, uint32_t tilesX, int py, int minX, int rectWidth, bool useScanlineRanges, float sb0, float sb1,
                         float dx_b0, float dx_b1, float dx_b2, float d0, float d1, float d2, uint32_t clusterIndex, uint32_t t, int seg) {
    const uint32_t rowBase = (((uint32_t)py >> 3) * tilesX << 6) | ((uint32_t)py & 7u);
    if (useScanlineRanges) {
        const float sb2 = 1.0f - sb0 - sb1;
        int firstOff = 0, lastOff = rectWidth - 1; bool has = true;
        clip_scanline(sb0, dx_b0, firstOff, lastOff, has);
        clip_scanline(sb1, dx_b1, firstOff, lastOff, has);
        clip_scanline(sb2, dx_b2, firstOff, lastOff, has);
        if (has) {
            float b0 = sb0 + (float)firstOff * dx_b0, b1 = sb1 + (float)firstOff * dx_b1;
            int x0 = minX + firstOff, x1 = minX + lastOff;
            if (seg >= 0) {
                const int skip = seg * RASTER_SEG;
                if (skip > lastOff - firstOff) return;
                for (int k = 0; k < skip; k++) { b0 += dx_b0; b1 += dx_b1; }
                x0 += skip; x1 = min(x1, x0 + RASTER_SEG - 1);
            }
            for (int px = x0; px <= x1; px++) {
                const float b2 = 1.0f - b0 - b1;
                const float depth = b0 * d0 + b1 * d1 + b2 * d2;
                emit_key<MODE>(&vis[rowBase + (((uint32_t)px >> 3) << 6) + (((uint32_t)px & 7u) << 3)], (unsigned long long)pack_vis_key(depth, clusterIndex, t), sink);
                b0 += dx_b0; b1 += dx_b1;
            }
        }
    } else {
        if (seg > 0) return;      // narrow boxes (width <= 4) are never segmented
        float b0 = sb0, b1 = sb1;
        for (int px = minX; px < minX + rectWidth; px++) {
            const float b2 = 1.0f - b0 - b1;
            if (b0 >= 0.0f && b1 >= 0.0f && b2 >= 0.0f) {
                const float depth = b0 * d0 + b1 * d1 + b2 * d2;
                emit_key<MODE>(&vis[rowBase + (((uint32_t)px >> 3) << 6) + (((uint32_t)px & 7u) << 3)], (unsigned long long)pack_vis_key(depth, clusterIndex, t), sink);
            }
            b0 += dx_b0; b1 += dx_b1;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) k_raster(RasterArgs a) {
    unsigned long long sink = 0;
    __shared__ float sx[BRMI_MESHLET_MAX_VERTS], sy[BRMI_MESHLET_MAX_VERTS], sd[BRMI_MESHLET_MAX_VERTS];
    const brmi_scene_buffers& sc = a.sc;
    const uint32_t lane = threadIdx.x;
    const uint32_t first = a.firstCounter == 0xFFFFFFFFu ? 0u : a.counters[a.firstCounter];
    const uint32_t count = a.counters[a.countCounter];
    // static round-robin over clusters: a shared queue head saturates at ~90 dequeues/us (MI355X_MICROARCH.md, row
    // "dequeue"), which is slower than the work itself once big triangles are handed off
    for (uint32_t c = blockIdx.x; c < count; c += gridDim.x) {
        const uint32_t clusterIndex = first + c;
        const uint4 pc = a.clusters[clusterIndex];
        const uint32_t viewID = vc_view(pc), instanceID = vc_instance(pc), localMeshlet = vc_meshlet(pc);
        const uint8_t* slab = sc.slabs[vc_slab(pc)];
        const uint32_t pageOff = vc_page_offset(pc);
        const brmi_page_header* hdr = reinterpret_cast<const brmi_page_header*>(slab + pageOff);
        const brmi_meshlet_descriptor* desc = reinterpret_cast<const brmi_meshlet_descriptor*>(slab + pageOff + hdr->descriptorOffset + localMeshlet * 64u);
        const uint32_t vertCount = min((desc->bitsAndVertexCount >> 24) & 0xFFu, BRMI_MESHLET_MAX_VERTS);
        const uint32_t triCount = min(desc->triangleCountAndRefinedGroup & 0xFFFFu, BRMI_MESHLET_MAX_TRIS);
        const brmi_per_mesh_instance* meshInst = sc.perMeshInstance + instanceID;
        const brmi_per_object* obj = sc.perObject + meshInst->perObjectBufferIndex;
        const brmi_view_raster_info ri = sc.viewRasterInfo[viewID];
        const float visWidth = (float)(ri.scissorMaxX - ri.scissorMinX), visHeight = (float)(ri.scissorMaxY - ri.scissorMinY);
        const float sMinXf = (float)ri.scissorMinX, sMinYf = (float)ri.scissorMinY;
        const float* oc = a.objConst + (size_t)meshInst->perObjectBufferIndex * 36u;
        const m4 mvp = load_m4(oc);
        const f4 modelViewZ{oc[32], oc[33], oc[34], oc[35]};
        const uint32_t posFormat = hdr->compressedPositionQuantExp;
        const uint8_t* posBase = slab + pageOff + hdr->positionBitstreamOffset + desc->positionBitOffset;
        const uint8_t* triBase = slab + pageOff + hdr->triangleStreamOffset + desc->triangleByteOffset;
        const bool reverseWinding = (obj->objectFlags & BRMI_OBJECT_FLAG_REVERSE_WINDING) != 0;

        // vertex stage -> LDS (softwareRaster.hlsl:339-387)
        for (uint32_t v = lane; v < vertCount; v += 64) {
            f3 lp{0.0f, 0.0f, 0.0f};
            if (posFormat == BRMI_POSITION_FORMAT_FLOAT3) {
                const float* pp = reinterpret_cast<const float*>(posBase + v * 12u);
                lp = f3{pp[0], pp[1], pp[2]};
            }
            const f4 lp4{lp.x, lp.y, lp.z, 1.0f};
            const f4 clip = mul_vm(lp4, mvp);
            const float viewZ = dot4(lp4, modelViewZ);
            const float invW = 1.0f / clip.w;
            const float ndcx = clip.x * invW, ndcy = clip.y * invW;
            sx[v] = (ndcx + 1.0f) * 0.5f * visWidth + sMinXf;
            sy[v] = (1.0f - ndcy) * 0.5f * visHeight + sMinYf;
            sd[v] = -viewZ;
        }
        __syncthreads();

        // triangle stage: lane = triangle (softwareRaster.hlsl:416-611)
        for (uint32_t waveBase = 0; waveBase < triCount; waveBase += 64) {
            const uint32_t t = waveBase + lane;
            bool active = t < triCount;
            float d0 = 0, d1 = 0, d2 = 0, row_b0 = 0, row_b1 = 0, dx_b0 = 0, dx_b1 = 0, dy_b0 = 0, dy_b1 = 0;
            int minX = 0, minY = 0, maxX = -1, maxY = -1;
            if (active) {
                uint32_t i0 = triBase[t * 3u], i1 = triBase[t * 3u + 1u], i2 = triBase[t * 3u + 2u];
                if (reverseWinding) { const uint32_t tmp = i1; i1 = i2; i2 = tmp; }
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
            const bool useScanlineRanges = __any(active && rectWidth > 4);
            // classify: big triangles are handed to k_raster_big as 16-row records; records of boxes wider than
            // RASTER_SEG go to the "wide" end of the queue (walked by a whole wave, 4 lanes per row)
            const int rows = maxY - minY + 1;
            bool big = active && rows * rectWidth > a.bigTriArea;
            const bool wide = rectWidth > RASTER_SEG;
            uint32_t nrec = 0;
            if (big) {
                for (int py = minY; py <= maxY; py += REC_ROWS) {
                    const int n = min(REC_ROWS, maxY - py + 1);
                    if ((uint32_t)(py + n) > a.bandY0 && (uint32_t)py < a.bandY1) nrec++;
                }
            }
            // one reservation per wave and per queue end
            uint32_t inclN = (big && !wide) ? nrec : 0u, inclW = (big && wide) ? nrec : 0u;
            const uint32_t mineN = inclN, mineW = inclW;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t vn = (uint32_t)__shfl_up((int)inclN, o), vw = (uint32_t)__shfl_up((int)inclW, o);
                if (lane >= (uint32_t)o) { inclN += vn; inclW += vw; }
            }
            const uint32_t totalN = (uint32_t)__shfl((int)inclN, 63), totalW = (uint32_t)__shfl((int)inclW, 63);
            uint32_t slot = 0;
            if (totalN + totalW != 0) {
                uint32_t baseN = 0, baseW = 0;
                if (lane == 0) { baseN = atomicAdd(&a.counters[CNT_BIG_TRIS], totalN); baseW = atomicAdd(&a.counters[CNT_BIG_TRIS_WIDE], totalW); }
                baseN = (uint32_t)__shfl((int)baseN, 0); baseW = (uint32_t)__shfl((int)baseW, 0);
                if ((uint64_t)baseN + totalN + baseW + totalW > a.bigTriCapacity) {      // queue full: give the slots back and walk everything here
                    if (lane == 0) { atomicSub(&a.counters[CNT_BIG_TRIS], totalN); atomicSub(&a.counters[CNT_BIG_TRIS_WIDE], totalW); }
                    big = false;
                }
                // narrow records grow from the bottom, wide records from the top of the same array
                slot = wide ? (a.bigTriCapacity - 1u - (baseW + inclW - mineW)) : (baseN + inclN - mineN);
            }
            if (active) {
                const float dx_b2 = -(dx_b0 + dx_b1);
                float sb0 = row_b0, sb1 = row_b1;
                if (big) {
                    for (int py = minY; py <= maxY; py += REC_ROWS) {
                        const int n = min(REC_ROWS, maxY - py + 1);
                        if ((uint32_t)(py + n) > a.bandY0 && (uint32_t)py < a.bandY1) {
                            BigTriRecord r;
                            r.clusterIndex = clusterIndex;
                            r.minX = minX; r.rectWidth = rectWidth; r.rowStart = py;
                            r.sb0 = sb0; r.sb1 = sb1; r.dx_b0 = dx_b0; r.dx_b1 = dx_b1; r.dy_b0 = dy_b0; r.dy_b1 = dy_b1; r.d0 = d0; r.d1 = d1; r.d2 = d2; r.pad0 = 0; r.pad1 = 0;
                            r.triAndFlags = t | (useScanlineRanges ? 0x100u : 0u) | ((uint32_t)n << 16);
                            a.bigTris[slot] = r;
                            slot = wide ? slot - 1u : slot + 1u;
                        }
                        for (int k = 0; k < n; k++) { sb0 += dy_b0; sb1 += dy_b1; }
                    }
                } else {
                    for (int py = minY; py <= maxY; py++) {
                        if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1)
                            raster_row<MODE>(sink, a.vis, a.tilesX, py, minX, rectWidth, useScanlineRanges, sb0, sb1, dx_b0, dx_b1, dx_b2, d0, d1, d2, clusterIndex, t, -1);
                        sb0 += dy_b0; sb1 += dy_b1;
                    }
                }
            }
        }
        __syncthreads();   // LDS is reused by the next cluster
    }
    if (MODE == 2 && sink == 0x123456789ull) a.vis[0] = sink;
}

// Row-parallel walk of the queued 16-row records; records are taken round-robin (no shared queue head).
//   narrow records (box width <= RASTER_SEG): four per wave64, 16 lanes = 16 rows each;
//   wide records: one per wave64, 4 lanes per row, lane `phase` walks pixel runs phase, phase+4, ... of RASTER_SEG.
// Every lane steps to its row / run exactly as the serial loop would.
template <int MODE>
__global__ void __launch_bounds__(64) k_raster_big(RasterArgs a) {
    unsigned long long sink = 0;
    const uint32_t lane = threadIdx.x, sub = lane >> 4, row = lane & 15u;
    const uint32_t nNarrow = min(a.counters[CNT_BIG_TRIS], a.bigTriCapacity);
    const uint32_t nWide = min(a.counters[CNT_BIG_TRIS_WIDE], a.bigTriCapacity - nNarrow);
    for (uint32_t base = blockIdx.x * 4u; base < nNarrow; base += gridDim.x * 4u) {
        const uint32_t ri = base + sub;
        if (ri >= nNarrow) continue;
        const BigTriRecord r = a.bigTris[ri];
        const uint32_t n = (r.triAndFlags >> 16) & 0xFFu;
        if (row < n) {
            float sb0 = r.sb0, sb1 = r.sb1;
            for (uint32_t k = 0; k < row; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
            const int py = r.rowStart + (int)row;
            if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1)
                raster_row<MODE>(sink, a.vis, a.tilesX, py, r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), r.d0, r.d1, r.d2,
                                 r.clusterIndex, r.triAndFlags & 0x7Fu, -1);
        }
    }
    for (uint32_t wi = blockIdx.x; wi < nWide; wi += gridDim.x) {
        const BigTriRecord r = a.bigTris[a.bigTriCapacity - 1u - wi];
        const uint32_t n = (r.triAndFlags >> 16) & 0xFFu;
        if (row < n) {
            float sb0 = r.sb0, sb1 = r.sb1;
            for (uint32_t k = 0; k < row; k++) { sb0 += r.dy_b0; sb1 += r.dy_b1; }
            const int py = r.rowStart + (int)row;
            if ((uint32_t)py >= a.bandY0 && (uint32_t)py < a.bandY1) {
                const int segs = (r.rectWidth + RASTER_SEG - 1) / RASTER_SEG;
                for (int sg = (int)sub; sg < segs; sg += 4)
                    raster_row<MODE>(sink, a.vis, a.tilesX, py, r.minX, r.rectWidth, (r.triAndFlags & 0x100u) != 0, sb0, sb1, r.dx_b0, r.dx_b1, -(r.dx_b0 + r.dx_b1), r.d0, r.d1, r.d2,
                                     r.clusterIndex, r.triAndFlags & 0x7Fu, sg);
            }
        }
    }
    if (MODE == 2 && sink == 0x123456789ull) a.vis[0] = sink;
}

// K6: linear depth from the visibility key (gbuffer.hlsl:114-161); one lane per pixel, tile order
__global__ void __launch_bounds__(256) k_depth_copy(const unsigned long long* vis, float* depth, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = vis[i];
        depth[i] = (k == BRMI_VIS_EMPTY) ? as_f32(BRMI_DEPTH_EMPTY_BITS) : as_f32(((uint32_t)(k >> BRMI_VIS_META_BITS)) << 1);
    }
}

int launch_clear(brmi_pass* p, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_vis, dim3(2048), dim3(256), 0, s, static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel, p->bandPixelCount);
    BRMI_LAUNCH_CHECK(p, "k_clear_vis");
    return BRMI_OK;
}

int launch_raster(brmi_pass* p, uint32_t phase, hipStream_t s) {
    if (phase != 1 && phase != 2) return fail(p, BRMI_ERR_INVALID, "brmi_raster: phase %u (1 or 2)", phase);
    if (phase == 2 && !p->cfg.enableOcclusionCulling) return fail(p, BRMI_ERR_STATE, "brmi_raster: phase 2 needs a pass created with enableOcclusionCulling");
    RasterArgs a;
    a.sc = p->scene; a.clusters = static_cast<const uint4*>(p->res[BRMI_RES_VISIBLE_CLUSTERS]); a.counters = p->counters();
    a.firstCounter = 0xFFFFFFFFu; a.countCounter = CNT_VISIBLE;
    if (phase == 2) {   // clusters [visible1, visible1 + visible2); the big-triangle queue starts empty again
        a.firstCounter = CNT_VISIBLE; a.countCounter = CNT_VISIBLE2;
        static_assert(CNT_BIG_TRIS_WIDE == CNT_BIG_TRIS + 1, "queue counters are cleared together");
        BRMI_HIP(p, hipMemsetAsync(p->counters() + CNT_BIG_TRIS, 0, 2 * sizeof(uint32_t), s));
    }
    a.queue = p->counters() + CNT_WORDS;   // one spare word after the counters block
    a.vis = static_cast<unsigned long long*>(p->res[BRMI_RES_VISIBILITY]);
    a.visW = p->cfg.width; a.visH = p->cfg.height; a.tilesX = p->tilesX; a.bandY0 = p->bandY0; a.bandY1 = p->bandY1;
    a.bigTris = p->wsPtr<BigTriRecord>(p->ws.bigTris); a.bigTriCapacity = p->bigTriCapacity;
    a.objConst = p->wsPtr<float>(p->ws.objConst);
    a.bigTriArea = p->bigTriArea;
    switch (p->rasterMode) {
#define BRMI_RASTER_LAUNCH(M) case M: hipLaunchKernelGGL(k_raster<M>, dim3(256 * 16), dim3(64), 0, s, a); hipLaunchKernelGGL(k_raster_big<M>, dim3(256 * 32), dim3(64), 0, s, a); break;
        BRMI_RASTER_LAUNCH(1) BRMI_RASTER_LAUNCH(2) BRMI_RASTER_LAUNCH(3)
        default: hipLaunchKernelGGL(k_raster<0>, dim3(256 * 16), dim3(64), 0, s, a); hipLaunchKernelGGL(k_raster_big<0>, dim3(256 * 32), dim3(64), 0, s, a); break;
#undef BRMI_RASTER_LAUNCH
    }
    BRMI_LAUNCH_CHECK(p, "k_raster");
    return BRMI_OK;
}

int launch_depth_copy(brmi_pass* p, hipStream_t s) {
    hipLaunchKernelGGL(k_depth_copy, dim3(2048), dim3(256), 0, s, static_cast<const unsigned long long*>(p->res[BRMI_RES_VISIBILITY]) + p->bandFirstPixel,
                       static_cast<float*>(p->res[BRMI_RES_LINEAR_DEPTH]) + p->bandFirstPixel, p->bandPixelCount);
    BRMI_LAUNCH_CHECK(p, "k_depth_copy");
    return BRMI_OK;
}

}  // namespace brmi
