"""Harness over libbrmi.so: owns the HBM resources a render graph would own and drives the pass.

PyTorch is plumbing here (device memory, streams, torch.distributed); every kernel is in
libbrmi.so behind the C ABI of include/brmi.h.  There is no CPU fallback: if the HIP library is
missing or no GPU is present this module raises.
"""
import ctypes as C

import numpy as np

from . import capi


class BrmiError(RuntimeError):
    pass


def detile(flat, width, height, tile=8):
    """Tiled 8x8 storage, column-major inside a tile (numpy, leading dim = padded pixels) -> linear [H, W, ...]."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    rest = flat.shape[1:]
    a = flat.reshape((ty, tx, tile, tile) + rest)            # [tileY, tileX, xInTile, yInTile]
    a = np.transpose(a, (0, 3, 1, 2) + tuple(range(4, 4 + len(rest)))).reshape((ty * tile, tx * tile) + rest)
    return np.ascontiguousarray(a[:height, :width])


def tile(img, tile=8):
    """Linear [H, W, ...] -> tiled 8x8 storage (column-major inside a tile), padded to whole tiles; inverse of detile."""
    height, width = img.shape[:2]
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    rest = img.shape[2:]
    pad = np.zeros((ty * tile, tx * tile) + rest, dtype=img.dtype)
    pad[:height, :width] = img
    a = pad.reshape((ty, tile, tx, tile) + rest)                      # [tileY, yInTile, tileX, xInTile]
    a = np.transpose(a, (0, 2, 3, 1) + tuple(range(4, 4 + len(rest))))
    return np.ascontiguousarray(a).reshape((ty * tx * tile * tile,) + rest)


class VisibilityRenderer:
    """One brmi_pass (= CLodExtension + VisUtil + light clustering + deferred shading of one view)."""

    def __init__(self, scene, device="cuda:0", max_clusters=None, occlusion=False, stats=False, band=(0, 0), stripes=None, **cfg_over):
        """stripes = (rows, count, index): the interleaved screen partition (brmi_config::stripe*): this pass owns the chunks of `rows` rows
        with index = `index` (mod `count`) of the scene's frame and renders them into compact surfaces of height / count rows; read-backs
        return those compact images, `frame_rows()` says which rows of the frame they are."""
        import torch
        if not torch.cuda.is_available():
            raise BrmiError("no GPU: libbrmi.so needs an MI355X (there is no CPU fallback)")
        self.torch = torch
        self.lib = capi.brmi_lib()
        self.scene, self.device = scene, torch.device(device)
        torch.cuda.set_device(self.device)
        self.W, self.H = scene.width, scene.height
        self.stripes = stripes
        if stripes is not None:
            rows, count, index = stripes
            if scene.height % (rows * count):
                raise ValueError(f"{scene.height} rows do not split into chunks of {rows} rows for {count} GPUs")
            self.H = scene.height // count          # the compact surfaces
        cfg = capi.Config()
        self.lib.brmi_default_config(C.byref(cfg), self.W, self.H)
        if stripes is not None:
            cfg.stripeRows, cfg.stripeCount, cfg.stripeIndex, cfg.fullHeight = stripes[0], stripes[1], stripes[2], scene.height
        est = max(4096, scene.stats.get("instancedMeshlets") or 2 * scene.stats["meshletsTotal"] * max(1, scene.stats["instances"]) // max(1, scene.stats["meshes"]))      # every (instance, meshlet) pair: nothing more can be visible
        cfg.maxVisibleClusters = int(max_clusters or min(1 << 24, max(1 << 16, est)))
        cfg.maxTraversalRecords = cfg.maxVisibleClusters
        cfg.enableOcclusionCulling = 1 if occlusion else 0
        cfg.collectPassStatistics = 1 if stats else 0
        cfg.bandY0, cfg.bandY1 = band
        for k, v in cfg_over.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = capi.vp()
        self._check(self.lib.brmi_create(C.byref(cfg), C.byref(self._h)), "brmi_create", use_pass=False)
        self.sb, self._scene_keep = scene.device_buffers(self.device)
        self.device_arrays = dict(scene.device_arrays)      # this pass's own copies (two passes in flight each have their camera buffers)
        self._check(self.lib.brmi_set_scene(self._h, C.byref(self.sb)), "brmi_set_scene")
        self.descs = {}

        def cb(_user, d):
            d = d.contents
            self.descs[d.id] = dict(name=d.name.decode(), bytes=d.bytes, width=d.width, height=d.height, bpp=d.bytesPerPixel)

        self._cb = capi.DECLARE_CB(cb)
        self._check(self.lib.brmi_declare(self._h, self._cb, None), "brmi_declare")
        self.res = {}
        binds = (capi.ResourceBinding * len(self.descs))()
        for i, (rid, d) in enumerate(sorted(self.descs.items())):
            t = torch.zeros(max(16, int(d["bytes"])), dtype=torch.uint8, device=self.device)
            self.res[rid] = t
            binds[i].id, binds[i].ptr, binds[i].bytes = rid, t.data_ptr(), t.numel()
        self.stream = torch.cuda.current_stream(self.device)
        self._check(self.lib.brmi_setup(self._h, binds, len(self.descs), self._s()), "brmi_setup")
        self.update()

    # ------------------------------------------------------------------------------------------
    def _s(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc, what, use_pass=True):
        if rc != 0:
            msg = self.lib.brmi_last_error(self._h).decode() if use_pass and self._h else ""
            raise BrmiError(f"{what} failed ({rc}): {msg}")

    def frame_rows(self):
        """Row of the scene's frame behind every row of this pass's surfaces (the identity without `stripes`)."""
        if self.stripes is None:
            return np.arange(self.H)
        from . import compose
        rows, count, index = self.stripes
        return compose.stripe_frame_rows(index, count, self.scene.height, rows)

    def set_camera_from(self, other, frame_index=0):
        """Next frame's camera: copy `other`'s camera / culling-camera buffers (same geometry, another camera step) into
        the device buffers the pass reads -- what the reference's CameraManager does between frames -- then brmi_update."""
        for name in ("cameras", "cullingCameras"):
            self.device_arrays[name].copy_(self.torch.from_numpy(other.arrays[name]).to(self.device))
        self._cam_scene = other
        self._cam_bytes = None
        self.update(frame_index)

    def set_camera_device(self, cameras_dev, culling_cameras_dev, cameras_host, frame_index=0):
        """Next frame's camera from buffers already in HBM (uint8 tensors with the bytes of Scene.camera_at): two device-to-device copies on the
        current stream into this pass's camera buffers, then brmi_update with the host copy of the same camera."""
        self.device_arrays["cameras"].copy_(cameras_dev, non_blocking=True)
        self.device_arrays["cullingCameras"].copy_(culling_cameras_dev, non_blocking=True)
        upd = capi.FrameUpdate(cameras_host.ctypes.data, self.scene.per_frame_host().ctypes.data, frame_index)
        self._check(self.lib.brmi_update(self._h, C.byref(upd), self._s()), "brmi_update")
        self._cam_bytes = cameras_host      # update() of a later frame repeats this camera

    def set_band(self, y0, y1):
        """Rows [y0, y1) this pass renders from its next frame on (brmi_set_band; passes created with dynamicBand=1).  update() before the frame, as always."""
        self._check(self.lib.brmi_set_band(self._h, capi.u32(int(y0)), capi.u32(int(y1))), "brmi_set_band")
        self.band = (int(y0), int(y1))

    def update(self, frame_index=0):
        if getattr(self, "_cam_bytes", None) is not None:
            upd = capi.FrameUpdate(self._cam_bytes.ctypes.data, self.scene.per_frame_host().ctypes.data, frame_index)
            self._check(self.lib.brmi_update(self._h, C.byref(upd), self._s()), "brmi_update")
            return
        src = getattr(self, "_cam_scene", self.scene)
        cam, pf = src.camera_host(), src.per_frame_host()
        upd = capi.FrameUpdate(cam.ctypes.data, pf.ctypes.data, frame_index)
        self._check(self.lib.brmi_update(self._h, C.byref(upd), self._s()), "brmi_update")

    def execute(self, shading_stream=None):
        """The whole frame on the current stream; with `shading_stream` (a torch stream) the resolve + shading half goes there."""
        if shading_stream is None:
            self._check(self.lib.brmi_execute(self._h, self._s()), "brmi_execute")
        else:
            self._check(self.lib.brmi_execute_split(self._h, self._s(), C.c_void_p(shading_stream.cuda_stream)), "brmi_execute_split")
        err, self._slab_error = getattr(self, "_slab_error", None), None
        if err is not None:
            raise BrmiError(f"the shade-slab hook raised during execute: {err!r}") from err

    def stage(self, name, *args):
        fn = getattr(self.lib, "brmi_" + name)
        self._check(fn(self._h, *[capi.u32(a) for a in args], self._s()), "brmi_" + name)
        err, self._slab_error = getattr(self, "_slab_error", None), None
        if err is not None:
            raise BrmiError(f"the shade-slab hook raised during brmi_{name}: {err!r}") from err

    def counters(self):
        c = capi.Counters()
        self._check(self.lib.brmi_read_counters(self._h, C.byref(c), self._s()), "brmi_read_counters")
        return c

    def stage_times(self):
        ms = (capi.f32 * len(capi.STAGE_NAMES))()
        self._check(self.lib.brmi_stage_times(self._h, ms), "brmi_stage_times")
        return dict(zip(capi.STAGE_NAMES, [float(x) for x in ms]))

    def set_timed_stages(self, names=None):
        """Record HIP events only around the named stages (None = all)."""
        mask = 0xFFFFFFFF if names is None else sum(1 << capi.STAGE_NAMES.index(n) for n in names)
        self._check(self.lib.brmi_set_timed_stages(self._h, mask), "brmi_set_timed_stages")

    def algorithmic_bytes(self):
        per = (capi.u64 * len(capi.STAGE_NAMES))()
        total = capi.u64()
        self._check(self.lib.brmi_algorithmic_bytes(self._h, per, C.byref(total)), "brmi_algorithmic_bytes")
        return dict(zip(capi.STAGE_NAMES, [int(x) for x in per])), int(total.value)

    def held_clusters(self):
        """(held, late): indices into visible_clusters() of the last frame's phase-1 clusters the culling held back, and of those the late pass drew (brmi_debug_read_held)."""
        nh, nl = capi.u32(), capi.u32()
        self._check(self.lib.brmi_debug_read_held(self._h, None, 0, C.byref(nh), None, 0, C.byref(nl)), "brmi_debug_read_held")
        held, late = np.zeros(max(1, nh.value), dtype=np.uint32), np.zeros(max(1, nl.value), dtype=np.uint32)
        self._check(self.lib.brmi_debug_read_held(self._h, held.ctypes.data_as(C.POINTER(capi.u32)), len(held), C.byref(nh), late.ctypes.data_as(C.POINTER(capi.u32)), len(late), C.byref(nl)), "brmi_debug_read_held")
        return held[: nh.value], late[: nl.value]

    def lean_clusters(self):
        """(1 if the last frame's phase-1 main launch was the lean rasteriser, clusters it left to the general launch, triangles queued for k_raster_emit, runs of
        them) -- brmi_debug_lean_clusters."""
        out = (capi.u32 * 4)()
        self._check(self.lib.brmi_debug_lean_clusters(self._h, out), "brmi_debug_lean_clusters")
        return tuple(int(v) for v in out)

    def wide_triangles(self):
        """(phase-1 draw pass, late pass, phase 2) counts of the last frame's triangles queued for the workgroup-wide record emission (brmi_debug_wide_triangles)."""
        out = (capi.u32 * 3)()
        self._check(self.lib.brmi_debug_wide_triangles(self._h, out), "brmi_debug_wide_triangles")
        return tuple(int(x) for x in out)

    def algorithmic_bytes_launched(self):
        """The same for the kernel variants the frame launched (brmi_algorithmic_bytes_launched)."""
        per = (capi.u64 * len(capi.STAGE_NAMES))()
        total = capi.u64()
        self._check(self.lib.brmi_algorithmic_bytes_launched(self._h, per, C.byref(total)), "brmi_algorithmic_bytes_launched")
        return dict(zip(capi.STAGE_NAMES, [int(x) for x in per])), int(total.value)

    # -- read-back (linear layout) ---------------------------------------------------------------
    def _img(self, rid, dtype, comps=1):
        self.torch.cuda.synchronize(self.device)
        raw = self.res[capi.RES[rid]].cpu().numpy()
        a = raw.view(dtype)
        n = a.size // comps
        a = a.reshape((n, comps)) if comps > 1 else a.reshape((n,))
        return detile(a, self.W, self.H)

    def visibility(self):
        return self._img("VISIBILITY", np.uint64)

    def depth(self):
        return self._img("LINEAR_DEPTH", np.float32)

    def hdr(self):
        return self._img("HDR_COLOR", np.uint64)

    def gbuffer(self):
        return dict(normals=self._img("GBUF_NORMALS", np.float32, 4), albedo=self._img("GBUF_ALBEDO", np.uint32),
                    coat=self._img("GBUF_COAT", np.uint64), emissive=self._img("GBUF_EMISSIVE", np.uint64),
                    fuzz=self._img("GBUF_FUZZ", np.uint64), mr=self._img("GBUF_METALLIC_ROUGHNESS", np.uint32),
                    motion=self._img("GBUF_MOTION_VECTORS", np.uint32))

    def visible_clusters(self):
        c = self.counters()
        n = c.visibleClusters + c.visibleClustersPhase2
        self.torch.cuda.synchronize(self.device)
        raw = self.res[capi.RES["VISIBLE_CLUSTERS"]][: n * 16].cpu().numpy()
        return raw.view(np.uint32).reshape(n, 4)

    def invalidate_hzb(self):
        self._check(self.lib.brmi_invalidate_hzb(self._h), "brmi_invalidate_hzb")

    def set_shade_slabs(self, slabs, callback=None):
        """brmi_set_shade_slabs: the deferred shading in `slabs` row slabs; `callback(row0, row1, stream_ptr)` is called on the host after each slab's
        launches (the hook for PeerBandComposer.submit_rows).  slabs <= 1 or no callback: one launch over the band."""
        if callback is None or slabs <= 1:
            self._slab_cb = None
            self._check(self.lib.brmi_set_shade_slabs(self._h, capi.u32(0), None, None), "brmi_set_shade_slabs")
            return
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)

        def trampoline(user, r0, r1, stream):
            # ctypes swallows an exception raised inside a callback (it only prints it): keep the first one and re-raise it when the call into the
            # library that ran the hook has returned (execute), so that a failed hand-over of a slab does not leave a half-composed frame unnoticed
            try:
                callback(int(r0), int(r1), stream)
            except BaseException as e:      # noqa: BLE001
                if self._slab_error is None:
                    self._slab_error = e
        self._slab_error = None
        self._slab_cb = proto(trampoline)      # kept alive with the pass
        self._check(self.lib.brmi_set_shade_slabs(self._h, capi.u32(slabs), C.cast(self._slab_cb, C.c_void_p), None), "brmi_set_shade_slabs")

    def set_history_source(self, other):
        """Frames in flight: phase 1 tests against the depth chain `other` built for the frame before (None unlinks)."""
        self._check(self.lib.brmi_set_history_source(self._h, other._h if other is not None else None), "brmi_set_history_source")

    def hzb_mips(self):
        """Mips >= 1 of the linear-depth chain as row-major arrays (mip 0 is depth() padded to a power of two)."""
        self.torch.cuda.synchronize(self.device)
        raw = self.res[capi.RES["HZB"]].cpu().numpy().view(np.float32)
        pw, ph = 1 << (self.W - 1).bit_length(), 1 << (self.H - 1).bit_length()
        out, off, mip = [], 0, 1
        while pw > 1 or ph > 1:
            pw, ph = max(1, pw >> 1), max(1, ph >> 1)
            out.append(raw[off: off + pw * ph].reshape(ph, pw))
            off += pw * ph
            mip += 1
        return out

    def light_clusters(self):
        self.torch.cuda.synchronize(self.device)
        c = self.res[capi.RES["LIGHT_CLUSTERS"]].cpu().numpy().view(np.uint32)
        p = self.res[capi.RES["LIGHT_PAGES"]].cpu().numpy().view(np.uint32)
        return c[: (c.size // 12) * 12].reshape(-1, 12), p[: (p.size // 14) * 14].reshape(-1, 14)

    def hdr_tensor(self):
        """The tiled HDR target as a torch uint8 view (for RCCL composition)."""
        return self.res[capi.RES["HDR_COLOR"]]

    def close(self):
        if self._h:
            self.torch.cuda.synchronize(self.device)
            self.lib.brmi_destroy(self._h)
            self._h = None
            # every byte the pass used is caller-owned: drop the resource tensors and the uploaded scene with the pass
            self.res = {}
            self._scene_keep = []
            self.device_arrays = {}
            if hasattr(self.scene, "device_arrays"):
                self.scene.device_arrays = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
