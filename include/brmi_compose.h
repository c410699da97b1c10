/*
 * brmi_compose.h -- C ABI of libbrmi_compose.so: composition of the screen-tile partition over the GPUs of one node.
 *
 * SURVEY.md section 8(e) / BASELINE.json north_star: the frame is split into row bands, every GPU renders its band with libbrmi.so and the
 * lit HDR bands are composed with ONE RCCL all-gather over xGMI so that every rank holds the whole image.  The reference has no
 * multi-GPU path; the interface follows the pass interface of include/brmi.h (a stream in place of the command list, caller-owned
 * device memory, status codes + last-error string).  This library is the only place that links RCCL; libbrmi.so does not.
 *
 * Pipelining: submit() copies the band out of the lit target (which the next frame overwrites) into one of `depth` staging
 * buffers on the render stream and starts the all-gather on the composer's own stream; the collective of frame k overlaps the
 * rendering of frame k + 1.  A staging / output buffer is reused only after its collective has finished (stream-ordered, no host
 * synchronisation).  finish() makes a stream wait for everything in flight and names the newest composed image.
 */
#ifndef BRMI_COMPOSE_H
#define BRMI_COMPOSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRMI_COMPOSE_ID_BYTES 128u          /* sizeof(ncclUniqueId) */

typedef enum brmi_compose_transport {
    BRMI_TRANSPORT_SURFACE = 0,             /* the band's bytes as they are in the tiled surface */
    BRMI_TRANSPORT_RGB16F  = 1              /* RGBA16F surface only: the three colour channels, 6 B per pixel (the lit target's alpha is the constant 1) */
} brmi_compose_transport;

/* How the bands travel.
 *   ALLGATHER   one ncclAllGather per frame on the composer's own stream (RCCL builds rings over the xGMI mesh).
 *   PEER_WRITE  no collective and no RCCL: every rank maps every other rank's output buffers (hipIpcMemHandle, exchanged once by the host
 *               like the unique id) and a kernel on the render stream stores its band straight into each peer's image, at the band's
 *               place -- N - 1 independent store streams over N - 1 different xGMI links, no staging copy, no ring hops (DESIGN.md
 *               section 6).  A frame number per (slot, writer) says a band has landed; a writer waits for the reader's own submit of the
 *               frame before it overwrites the slot (the image of frame f stays valid until the rank submits frame f + depth, as with
 *               the all-gather).  Setup: create -> alloc_shared (or bind memory that is the base of a hipMalloc allocation) -> export ->
 *               [host moves the handles] -> import -> submit / finish as before. */
typedef enum brmi_compose_path { BRMI_COMPOSE_ALLGATHER = 0, BRMI_COMPOSE_PEER_WRITE = 1 } brmi_compose_path;
#define BRMI_COMPOSE_HANDLE_BYTES 160u      /* two hipIpcMemHandle_t (output buffers, flag words) + the buffer size + the rank */

typedef struct brmi_compose_config {
    uint32_t structSize;                    /* sizeof(brmi_compose_config) */
    uint32_t width;                         /* frame width in pixels (the tiled surface is ceil(width / 8) tiles wide) */
    uint32_t bandY0, bandY1;                /* rows this rank owns; multiples of 8, the same height on every rank */
    uint32_t bytesPerPixel;                 /* of the surface (8 for the RGBA16F lit target) */
    uint32_t transport;                     /* brmi_compose_transport */
    uint32_t depth;                         /* staging / output buffers in flight (>= 1; 2 overlaps one frame) */
    uint32_t rank, nRanks;
    int32_t  device;                        /* HIP device of this rank */
    uint32_t path;                          /* brmi_compose_path (0 = the all-gather) */
    uint32_t waitTimeoutMs;                 /* PEER_WRITE: a wait for a peer's flag gives up after this long and latches an error (0 = 2000 ms) */
    /* Round 6, cost-balanced contiguous regions (brmi_set_band): frameHeight > 0 = the ranks' bands partition a frame of that many rows (a multiple of 8), need not be
     * of one height and may move from frame to frame (brmi_compose_set_bounds).  The composed image is then THE FRAME in transport form, every band at its own rows;
     * a staging buffer holds up to the whole frame.  bandY0 / bandY1 are the first frame's band.  ALLGATHER path: the bands travel as one group of ncclBroadcast
     * calls, one per rank, each with that rank's byte count (an all-gather needs equal counts); PEER_WRITE path: stores at the band's place, as before. */
    uint32_t frameHeight;
    uint32_t reserved[3];
} brmi_compose_config;

typedef struct brmi_composer brmi_composer;
typedef void* brmi_compose_stream;          /* hipStream_t */

/* Rank 0 creates the id; the host hands the bytes to every rank over any channel it has (MPI, a socket, torch.distributed). */
int brmi_compose_unique_id(uint8_t id[BRMI_COMPOSE_ID_BYTES]);
/* Collective: every rank calls it with the same id (ncclCommInitRank). */
int brmi_compose_create(const brmi_compose_config* cfg, const uint8_t id[BRMI_COMPOSE_ID_BYTES], brmi_composer** out);
/* Bytes of ONE staging buffer (this rank's band in transport form) and of ONE output buffer (all bands); the caller binds
 * `depth` of each, contiguous. */
uint64_t brmi_compose_staging_bytes(const brmi_composer* c);
uint64_t brmi_compose_output_bytes(const brmi_composer* c);
int brmi_compose_bind(brmi_composer* c, void* staging, uint64_t stagingBytes, void* output, uint64_t outputBytes);
/* PEER_WRITE only.  alloc_shared: `depth` output buffers + the flag words as allocations of their own (what hipIpcGetMemHandle wants), owned
 * by the composer and bound; export: this rank's handles; import: every rank's handles (nRanks x BRMI_COMPOSE_HANDLE_BYTES, rank order; the
 * rank's own entry is ignored).  last_wait_status: 0, or -6 once a wait for a peer timed out (the composed images are not to be trusted). */
int brmi_compose_alloc_shared(brmi_composer* c);
int brmi_compose_export(brmi_composer* c, uint8_t handle[BRMI_COMPOSE_HANDLE_BYTES]);
int brmi_compose_import(brmi_composer* c, const uint8_t* handles, uint32_t count);
int brmi_compose_last_wait_status(brmi_composer* c);
/* Composers with frameHeight > 0: the partition of the frames submitted from now on -- rows [rowBounds[r], rowBounds[r + 1]) belong to rank r; nRanks + 1 ascending
 * multiples of 8 from 0 to frameHeight, the same array on every rank (a rank may own no row at all).  Not while a frame is open (brmi_compose_submit_rows). */
int brmi_compose_set_bounds(brmi_composer* c, const uint32_t* rowBounds);
/* Where the boundaries should go.  `rowCost` (frameHeight / align floats, in / out, all zero before the first call) is the host's running estimate of how a frame's
 * cost is spread over the rows: each call first scales it inside every band [boundsIn[r], boundsIn[r + 1]) so that the band adds up to the time `rankMs[r]` the rank
 * just measured (the shape inside a band comes from frames measured under other bounds: a few partitions locate a horizon one cannot), then `boundsOut` cuts it into
 * pieces of equal cost, moved `damping` (0 .. 1] of the way there from `boundsIn`, rounded to multiples of `align` rows (8, or 16 = the raster bin), every band at least
 * `minRows` high; bounds the estimate does not expect to beat the measured ones by 3 % are left as they are.  Pure host arithmetic: every rank evaluates it on the same
 * numbers (one small all-gather of the times per rebalance) and gets the same partition.  Returns 0, or -1 for bad arguments. */
int brmi_compose_balance_rows(const float* rankMs, const uint32_t* boundsIn, uint32_t nRanks, uint32_t frameHeight, uint32_t align, float damping, uint32_t minRows, float* rowCost, uint32_t* boundsOut);
/* `surface`: base of the tiled surface (the whole frame's allocation).  Returns the buffer slot used (>= 0) or a negative status. */
int brmi_compose_submit(brmi_composer* c, const void* surface, brmi_compose_stream renderStream);
/* PEER_WRITE only -- the composition overlapped with the frame's own shading ("overlap gather of finished tiles with shading of later tiles via a
 * second stream").  A frame is submitted as slabs of rows [row0, row1) of the rank's band (frame rows, multiples of 8, ascending, together covering
 * [bandY0, bandY1)): call it after the launches that shade those rows have been enqueued on `renderStream` (brmi_shade_rows).  An event orders the
 * slab's stores -- on the composer's own stream, into every rank's image -- behind them, so they travel while the render stream shades the next
 * slab.  The first slab of a frame opens it (the "submitted" signal and the wait for the peers' slots), the last one closes it ("landed") and
 * returns the slot; earlier slabs return the slot too.  brmi_compose_finish orders a consumer behind the rank's own stores as well. */
int brmi_compose_submit_rows(brmi_composer* c, const void* surface, uint32_t row0, uint32_t row1, brmi_compose_stream renderStream);
/* brmi_compose_submit_rows reads `surface` on the COMPOSER's stream, behind the render stream, which does not wait for it.  Before anything writes
 * that surface again (the pass's next frame: with N frames in flight, N frames later) the writing stream must be ordered behind those reads:
 * this call makes `stream` wait for the newest brmi_compose_submit_rows read of `surface` (no-op for a surface that was never handed over in rows;
 * brmi_compose_submit reads on the render stream itself and needs nothing).  Without it a lagging peer -- the frame opens with a wait for every
 * peer's slot -- lets a later frame's shading overwrite rows that have not been copied yet: torn images, no error. */
int brmi_compose_wait_source(brmi_composer* c, const void* surface, brmi_compose_stream stream);
/* `stream` waits for every collective in flight; *composed = the output buffer of the newest frame (NULL before the first submit). */
int brmi_compose_finish(brmi_composer* c, brmi_compose_stream stream, void** composed);
void brmi_compose_destroy(brmi_composer* c);
const char* brmi_compose_last_error(const brmi_composer* c);

#ifdef __cplusplus
}
#endif
#endif /* BRMI_COMPOSE_H */
