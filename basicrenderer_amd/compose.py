"""Screen-tile partition of a frame over the GPUs of one node and its RCCL composition.

Geometry, materials and lights are replicated; rank r owns the row band [r*Hb, (r+1)*Hb) of the frame
(Hb a multiple of the 8-row surface tile, so a band is one contiguous byte range of every tiled surface).
The lit HDR bands are composed with ONE all-gather (equal counts) so that every rank ends up with the
full image -- the only data-path collective of the path.
"""
TILE = 8
BAND_ROWS = 1080          # rows per rank for N > 1 (7680 x 1080 = one 4K frame worth of pixels)


STRIPE_BAND_ROWS = 1088   # rows per rank with the interleaved partition: chunks are multiples of the 16-row raster bin, and 1080 has no such divisor
STRIPE_ROWS = 64          # default chunk height: the fastest of 16 / 32 / 64 / 272 / 544 at N = 8 (profiles/r03_rank_balance.md)


def frame_size(n_gpus, partition="bands"):
    """Weak scaling: every rank shades one 4K frame's worth of pixels (8,294,400 with row bands, 8,355,840 with the interleaved
    partition).  N = 1 is the 4K frame of BASELINE.json."""
    if n_gpus == 1:
        return (3840, 2160)
    return (7680, (STRIPE_BAND_ROWS if partition == "stripes" else BAND_ROWS) * n_gpus)


def strong_frame(n_gpus):
    """Strong scaling (BASELINE.json configs[3]: "4K, screen-tile partition across 2/4/8"): THE 4K frame split N ways by the interleaved partition.
    Chunks are multiples of the 16-row raster bin and every rank owns the same number of them, so the frame is 3840 x 2176 -- 2160 rows padded
    to the next multiple of 16 N for N = 2, 4, 8 (0.7 % more pixels, stated in the bench line) -- in chunks of 128 / N rows (64, 32, 16)."""
    if n_gpus == 1:
        return (3840, 2176), 0
    height = -(-2160 // (16 * n_gpus)) * (16 * n_gpus)
    for rows in (64, 48, 32, 16):
        if height % (rows * n_gpus) == 0:
            return (3840, height), rows
    raise ValueError(f"no chunk height for {n_gpus} GPUs")


def stripe_frame_rows(rank, n_gpus, height, rows=STRIPE_ROWS):
    """Rows of the frame a rank owns under the interleaved partition, in the order of its compact surfaces (include/brmi.h,
    brmi_config::stripe*): chunks of `rows` rows, `n_gpus` chunks to a group, one chunk per group and rank, the order inside a group
    running back and forth from group to group."""
    import numpy as np
    if rows % 16 or height % (rows * n_gpus):
        raise ValueError(f"height {height} does not split into groups of {n_gpus} chunks of {rows} rows (a multiple of 16)")
    v = np.arange(height // n_gpus)
    g = v // rows
    slot = np.where(g % 2 == 1, n_gpus - 1 - rank, rank)
    return (g * n_gpus + slot) * rows + v % rows


def band_of(rank, n_gpus, height):
    if n_gpus == 1:
        return (0, height)
    rows = height // n_gpus
    if rows % TILE or rows * n_gpus != height:
        raise ValueError(f"height {height} does not split into {n_gpus} bands of whole 8-row tiles")
    return (rank * rows, (rank + 1) * rows)


def equal_bounds(n_gpus, height, align=16):
    """Row bounds [0, b1, ..., height] of n contiguous bands of (nearly) equal height, multiples of `align`: where a balanced partition starts."""
    b = [min(height, int(round(height * k / n_gpus / align)) * align) for k in range(n_gpus + 1)]
    b[0], b[-1] = 0, height
    return b


class RowBalancer:
    """Cost-balanced contiguous regions (brmi_compose_balance_rows, libbrmi_compose.so): keeps the running per-row cost estimate and the current bounds; `update(rank_ms)`
    folds one measurement in and returns the new bounds.  Deterministic host arithmetic: every rank holds one and feeds it the same numbers."""

    def __init__(self, n_ranks, height, align=16, damping=1.0, min_rows=32, bounds=None):
        import numpy as np
        if height % align:
            raise ValueError(f"{height} rows are not a multiple of {align}")
        self.n, self.height, self.align, self.damping, self.min_rows = n_ranks, height, align, damping, min_rows
        self.bounds = list(bounds) if bounds is not None else equal_bounds(n_ranks, height, align)
        self.row_cost = np.zeros(height // align, dtype=np.float32)
        self.best = None      # (slowest rank's time, bounds) of the best partition measured so far

    def settle(self):
        """The best partition any update() measured (the estimate keeps moving boundaries by a bin row or two around the optimum; the slowest rank's time is what counts):
        makes it the current one and returns it."""
        if self.best is not None:
            self.bounds = list(self.best[1])
        return self.bounds

    def update(self, rank_ms):
        import ctypes as C
        from . import capi
        n = self.n
        if self.best is None or max(rank_ms) < self.best[0]:
            self.best = (max(rank_ms), list(self.bounds))
        ms = (C.c_float * n)(*[float(x) for x in rank_ms])
        bin_ = (C.c_uint32 * (n + 1))(*[int(x) for x in self.bounds])
        out = (C.c_uint32 * (n + 1))()
        rc = capi.compose_lib().brmi_compose_balance_rows(ms, bin_, n, int(self.height), int(self.align), C.c_float(self.damping), int(self.min_rows),
                                                           self.row_cost.ctypes.data_as(C.POINTER(C.c_float)), out)
        if rc != 0:
            raise ValueError(f"brmi_compose_balance_rows failed ({rc}): times {list(rank_ms)}, bounds {self.bounds}, height {self.height}")
        self.bounds = [int(x) for x in out]
        return self.bounds


def band_byte_range(band, width, bytes_per_pixel):
    """Byte range of a row band inside a tiled (8x8) surface."""
    tiles_x = (width + TILE - 1) // TILE
    if band[0] % TILE or band[1] % TILE:
        raise ValueError("band must be tile aligned")
    row_bytes = tiles_x * TILE * TILE * bytes_per_pixel
    return (band[0] // TILE) * row_bytes, (band[1] // TILE) * row_bytes


def compose_bands(surface_u8, band, width, bytes_per_pixel, out=None, group=None):
    """All-gather the caller's band of a tiled surface (flat uint8 tensor); returns the composed surface."""
    import torch
    import torch.distributed as dist
    lo, hi = band_byte_range(band, width, bytes_per_pixel)
    mine = surface_u8[lo:hi]
    n = dist.get_world_size(group)
    if out is None:
        out = torch.empty((hi - lo) * n, dtype=torch.uint8, device=surface_u8.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out


def rgb_of(surface_u8):
    """The three colour channels of a (tiled) RGBA16F byte surface as a strided [pixels, 3] int16 view (fp16 bit patterns)."""
    import torch
    return surface_u8.view(torch.int16).view(-1, 4)[:, :3]


class BandComposer:
    """Pipelined composition for a frame loop: the all-gather of frame k runs on the collective stream while frame k+1 is rendered.

    xGMI is point to point (7 links per GPU), so an all-gather of 66 MB bands to 8 ranks takes about as long as rendering a band;
    serialised behind the frame it would halve the throughput.  `submit()` copies the caller's band to one of `depth` staging
    buffers (the HDR target is overwritten by the next frame) and starts the all-gather asynchronously; a staging buffer is
    reused only after its collective has finished.  `finish()` waits for everything and returns the last composed surface.

    `transport="rgb16f"` (RGBA16F lit target only) gathers the three colour channels and delivers the composed image as RGB16F, 6 B
    per pixel instead of 8: the lit target's alpha is the constant 1.0 of an opaque frame, and the gather -- not the rendering -- is
    what bounds the frame rate from two GPUs on.  `out[i]` is then a [pixels, 3] int16 tensor of fp16 bit
    patterns in the same (tiled, band after band) pixel order; the staging copy drops the alpha on the way.
    """

    def __init__(self, surface_u8, band, width, bytes_per_pixel, depth=2, group=None, transport="surface"):
        import torch
        import torch.distributed as dist
        if transport not in ("surface", "rgb16f") or (transport == "rgb16f" and bytes_per_pixel != 8):
            raise ValueError("transport must be 'surface' or, for an RGBA16F surface, 'rgb16f'")
        lo, hi = band_byte_range(band, width, bytes_per_pixel)
        self.dist, self.group, self.depth, self.transport = dist, group, depth, transport
        self.byte_range, self.surface_ptr = (lo, hi), surface_u8.data_ptr()
        n = dist.get_world_size(group)
        if transport == "rgb16f":
            self.src = rgb_of(surface_u8[lo:hi])                                       # strided view: copy_ compacts it
            pixels = (hi - lo) // 8
            self.stage = [torch.empty((pixels, 3), dtype=torch.int16, device=surface_u8.device) for _ in range(depth)]
            self.out = [torch.empty((pixels * n, 3), dtype=torch.int16, device=surface_u8.device) for _ in range(depth)]
        else:
            self.src = surface_u8[lo:hi]
            self.stage = [torch.empty_like(self.src) for _ in range(depth)]
            self.out = [torch.empty((hi - lo) * n, dtype=torch.uint8, device=surface_u8.device) for _ in range(depth)]
        self.work = [None] * depth
        self.frames = 0

    def submit(self, surface_u8=None):
        """`surface_u8`: the frame's surface when it is not the one given at construction (two passes in flight alternate their targets)."""
        i = self.frames % self.depth
        if self.work[i] is not None:
            self.work[i].wait()
        src = self.src
        if surface_u8 is not None and surface_u8.data_ptr() != self.surface_ptr:
            lo, hi = self.byte_range
            src = rgb_of(surface_u8[lo:hi]) if self.transport == "rgb16f" else surface_u8[lo:hi]
        self.stage[i].copy_(src)
        import torch
        # the collective sees bytes (RCCL has no 16-bit integer type, and nothing is reduced)
        self.work[i] = self.dist.all_gather_into_tensor(self.out[i].view(-1).view(torch.uint8), self.stage[i].view(-1).view(torch.uint8), group=self.group, async_op=True)
        self.frames += 1
        return i

    def finish(self):
        for w in self.work:
            if w is not None:
                w.wait()
        self.work = [None] * self.depth
        return self.out[(self.frames - 1) % self.depth] if self.frames else None


def compose_unequal_bands(surface_u8, bounds, width, bytes_per_pixel, out=None, group=None):
    """Bands of unequal height (cost-balanced regions: `bounds` = [0, b1, ..., height], rank r owns rows [bounds[r], bounds[r + 1])) composed into THE FRAME: one
    broadcast per rank with that rank's byte count, straight to the band's place -- what libbrmi_compose.so issues as one RCCL group (brmi_compose_set_bounds), here over
    torch.distributed (gloo on CPU in the tests).  Returns the composed tiled surface (every rank holds all of it)."""
    import torch
    import torch.distributed as dist
    n, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(bounds) != n + 1:
        raise ValueError(f"{len(bounds)} bounds for {n} ranks")
    total = band_byte_range((0, bounds[-1]), width, bytes_per_pixel)[1]
    if out is None:
        out = torch.empty(total, dtype=torch.uint8, device=surface_u8.device)
    work = []
    for r in range(n):
        lo, hi = band_byte_range((bounds[r], bounds[r + 1]), width, bytes_per_pixel)
        if hi == lo:
            continue
        if r == rank:
            out[lo:hi].copy_(surface_u8[lo:hi])
        work.append(dist.broadcast(out[lo:hi], src=dist.get_global_rank(group, r) if group is not None else r, group=group, async_op=True))
    for w in work:
        w.wait()
    return out


class NativeBandComposer:
    """BandComposer's pipeline behind the C ABI of libbrmi_compose.so (include/brmi_compose.h): the staging copy, the RCCL all-gather on
    the composer's own stream and the event ordering all live in C++; this class only owns the buffers (the library allocates nothing)
    and hands the ncclUniqueId from rank 0 to the others over torch.distributed.  `out[i]` / finish() as BandComposer."""

    def __init__(self, surface_u8, band, width, bytes_per_pixel, depth=2, group=None, transport="surface", rank=None, world=None, frame_height=0):
        """frame_height > 0: bands of unequal, moving height (set_bounds before a frame's submit); the composed image is then the whole frame in transport form."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import capi
        self.torch, self.C, self.lib = torch, C, capi.compose_lib()
        dev = surface_u8.device
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if self.lib.brmi_compose_unique_id(buf) != 0:
                raise RuntimeError("brmi_compose_unique_id failed")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            ident = ident.to(dev) if dist.get_backend(group) == "nccl" else ident
            dist.broadcast(ident, src=0, group=group)
        ident = bytes(ident.cpu().tolist())
        cfg = capi.ComposeConfig()
        cfg.structSize = C.sizeof(capi.ComposeConfig)
        cfg.width, cfg.bandY0, cfg.bandY1, cfg.bytesPerPixel = width, band[0], band[1], bytes_per_pixel
        cfg.transport = {"surface": 0, "rgb16f": 1}[transport]
        cfg.depth, cfg.rank, cfg.nRanks, cfg.device = depth, rank, world, dev.index or 0
        cfg.frameHeight = frame_height
        self._h = capi.vp()
        rc = self.lib.brmi_compose_create(C.byref(cfg), ident, C.byref(self._h))
        if rc != 0:
            msg = self.lib.brmi_compose_last_error(self._h).decode() if self._h else ""
            raise RuntimeError(f"brmi_compose_create failed ({rc}): {msg}")
        sb, ob = self.lib.brmi_compose_staging_bytes(self._h), self.lib.brmi_compose_output_bytes(self._h)
        self.stage = torch.empty(sb * depth, dtype=torch.uint8, device=dev)
        self.outbuf = torch.empty(ob * depth, dtype=torch.uint8, device=dev)
        self._check(self.lib.brmi_compose_bind(self._h, self.stage.data_ptr(), self.stage.numel(), self.outbuf.data_ptr(), self.outbuf.numel()), "brmi_compose_bind")
        view = (lambda t: t.view(torch.int16).view(-1, 3)) if transport == "rgb16f" else (lambda t: t)
        self.out = [view(self.outbuf[i * ob:(i + 1) * ob]) for i in range(depth)]
        self.surface, self.depth, self.frames, self.dev = surface_u8, depth, 0, dev

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.brmi_compose_last_error(self._h).decode()}")
        return rc

    def _stream(self):
        return self.C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def set_bounds(self, bounds):
        """The partition of the frames submitted from now on (brmi_compose_set_bounds; composers made with frame_height)."""
        arr = (self.C.c_uint32 * len(bounds))(*[int(b) for b in bounds])
        self._check(self.lib.brmi_compose_set_bounds(self._h, arr), "brmi_compose_set_bounds")

    def submit(self, surface_u8=None):
        slot = self._check(self.lib.brmi_compose_submit(self._h, (self.surface if surface_u8 is None else surface_u8).data_ptr(), self._stream()), "brmi_compose_submit")
        self.frames += 1
        return slot

    def finish(self):
        ptr = self.C.c_void_p()
        self._check(self.lib.brmi_compose_finish(self._h, self._stream(), self.C.byref(ptr)), "brmi_compose_finish")
        return self.out[(self.frames - 1) % self.depth] if self.frames else None

    def close(self):
        if self._h:
            self.torch.cuda.synchronize(self.dev)
            self.lib.brmi_compose_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevicePointer:
    """A device allocation owned by somebody else (here: libbrmi_compose.so's shared buffers) as a torch tensor, through the array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


class PeerBandComposer:
    """The same pipeline without a collective (include/brmi_compose.h, BRMI_COMPOSE_PEER_WRITE): every rank maps every other rank's output
    buffers (hipIpcMemHandle) and a kernel on the render stream stores the band into all of them; flag words say when a band has landed.
    `exchange(handle_bytes) -> [handle bytes of every rank, in rank order]` is the host's channel for the one-time handle exchange (the
    default uses torch.distributed.all_gather_object; a test with two processes and no process group passes its own).
    `out[i]` / finish() as BandComposer; finish() makes the current stream wait for the peers' bands of the newest frame."""

    def __init__(self, surface_u8, band, width, bytes_per_pixel, depth=2, group=None, transport="surface", rank=None, world=None, exchange=None, timeout_ms=2000, frame_height=0):
        import ctypes as C
        import torch
        from . import capi
        self.torch, self.C, self.lib = torch, C, capi.compose_lib()
        dev = surface_u8.device
        if rank is None or world is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        cfg = capi.ComposeConfig()
        cfg.structSize = C.sizeof(capi.ComposeConfig)
        cfg.width, cfg.bandY0, cfg.bandY1, cfg.bytesPerPixel = width, band[0], band[1], bytes_per_pixel
        cfg.transport = {"surface": 0, "rgb16f": 1}[transport]
        cfg.depth, cfg.rank, cfg.nRanks, cfg.device = depth, rank, world, dev.index or 0
        cfg.path, cfg.waitTimeoutMs = 1, timeout_ms
        cfg.frameHeight = frame_height
        self._h = capi.vp()
        rc = self.lib.brmi_compose_create(C.byref(cfg), bytes(128), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"brmi_compose_create (peer write) failed ({rc})")
        self._check(self.lib.brmi_compose_alloc_shared(self._h), "brmi_compose_alloc_shared")
        mine = C.create_string_buffer(capi.COMPOSE_HANDLE_BYTES)
        self._check(self.lib.brmi_compose_export(self._h, mine), "brmi_compose_export")
        if exchange is None:
            import torch.distributed as dist

            def exchange(b):
                got = [None] * world
                dist.all_gather_object(got, b, group=group)
                return got
        handles = exchange(mine.raw) if world > 1 else [mine.raw]
        self._check(self.lib.brmi_compose_import(self._h, b"".join(handles), world), "brmi_compose_import")
        ob = self.lib.brmi_compose_output_bytes(self._h)
        self.surface, self.depth, self.frames, self.dev, self._ob, self.transport = surface_u8, depth, 0, dev, ob, transport
        self._band, self._rank = tuple(band), rank
        self.out = [None] * depth

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.brmi_compose_last_error(self._h).decode()}")
        return rc

    def _stream(self):
        return self.C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def submit(self, surface_u8=None):
        slot = self._check(self.lib.brmi_compose_submit(self._h, (self.surface if surface_u8 is None else surface_u8).data_ptr(), self._stream()), "brmi_compose_submit")
        self.frames += 1
        return slot

    def set_bounds(self, bounds):
        """The partition of the frames submitted from now on (brmi_compose_set_bounds; composers made with frame_height): this rank's band moves with it."""
        arr = (self.C.c_uint32 * len(bounds))(*[int(b) for b in bounds])
        self._check(self.lib.brmi_compose_set_bounds(self._h, arr), "brmi_compose_set_bounds")
        self._band = (int(bounds[self._rank]), int(bounds[self._rank + 1]))      # (submit_rows counts a frame when a slab ends at the band's last row)

    def submit_rows(self, row0, row1, surface_u8=None, stream_ptr=None):
        """One slab of the frame: rows [row0, row1) of the band, whose shading is already enqueued on the current stream; the stores travel on the
        composer's own stream while the current stream goes on shading (brmi_compose_submit_rows).  The slab that ends at the band's last row
        closes the frame."""
        slot = self._check(self.lib.brmi_compose_submit_rows(self._h, (self.surface if surface_u8 is None else surface_u8).data_ptr(), self.C.c_uint32(row0), self.C.c_uint32(row1),
                                                               self._stream() if stream_ptr is None else self.C.c_void_p(stream_ptr)), "brmi_compose_submit_rows")
        if row1 == self._band[1]:
            self.frames += 1
        return slot

    def wait_source(self, surface_u8=None, stream_ptr=None):
        """Before `surface` is written again (the pass's next frame): the stream (default: the current one) waits for the composer's reads of the rows
        submit_rows handed over (brmi_compose_wait_source)."""
        self._check(self.lib.brmi_compose_wait_source(self._h, (self.surface if surface_u8 is None else surface_u8).data_ptr(),
                                                        self._stream() if stream_ptr is None else self.C.c_void_p(stream_ptr)), "brmi_compose_wait_source")

    def finish(self):
        ptr = self.C.c_void_p()
        self._check(self.lib.brmi_compose_finish(self._h, self._stream(), self.C.byref(ptr)), "brmi_compose_finish")
        if not self.frames:
            return None
        t = self.torch.as_tensor(_DevicePointer(ptr.value, self._ob), device=self.dev)
        return t.view(self.torch.int16).view(-1, 3) if self.transport == "rgb16f" else t

    def slot_image(self, slot):
        """The composed image in buffer `slot` (0 .. depth - 1) as finish() shapes it: frame f lives in slot f % depth until frame f + depth is submitted."""
        ptr = self.C.c_void_p()
        self._check(self.lib.brmi_compose_finish(self._h, self._stream(), self.C.byref(ptr)), "brmi_compose_finish")
        base = ptr.value - ((self.frames - 1) % self.depth) * self._ob
        t = self.torch.as_tensor(_DevicePointer(base + slot * self._ob, self._ob), device=self.dev)
        return t.view(self.torch.int16).view(-1, 3) if self.transport == "rgb16f" else t

    def wait_status(self):
        """0, or raises once a wait for a peer timed out (synchronises)."""
        self.torch.cuda.synchronize(self.dev)
        return self._check(self.lib.brmi_compose_last_wait_status(self._h), "peer wait")

    def close(self):
        if self._h:
            self.torch.cuda.synchronize(self.dev)
            self.lib.brmi_compose_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
