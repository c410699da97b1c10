"""N > 1 path on CPU: two processes (gloo, 127.0.0.1) each produce their row band, the bands are all-gathered
with the same helper bench.py uses over RCCL, and the composed surface must equal the single-process frame."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import orc
    from basicrenderer_amd import Scene, compose
    from basicrenderer_amd.renderer import detile, tile
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = Scene("tiny", W, H, point_lights=4, seed=2)            # every rank builds the same (replicated) scene
    band = compose.band_of(rank, world, H)
    f = orc.OracleFrame(sc, threads=2)                            # the CPU checker stands in for the GPU pass here
    f.cull(); f.raster(band=band); f.depth_copy(); f.gbuffer(band=band); f.light_cluster(); f.shade(band=band)
    surface = torch.from_numpy(tile(f.hdr).view(np.uint8).copy())
    composed = compose.compose_bands(surface, band, W, 8)
    # the pipelined composer bench.py uses: three frames whose band content changes, every composed frame must be complete
    live = surface.clone()
    composer = compose.BandComposer(live, band, W, 8, depth=2)
    lo, hi = compose.band_byte_range(band, W, 8)
    for frame in range(3):
        live[lo:hi] = surface[lo:hi] ^ frame                      # "render" frame k into the target
        slot = composer.submit()
        live[lo:hi] = 0xEE                                         # the next frame overwrites the target while the gather is in flight
        composer.work[slot].wait()
        want = compose.compose_bands(surface ^ frame, band, W, 8)
        assert torch.equal(composer.out[slot], want), f"pipelined composition of frame {frame}"
    assert torch.equal(composer.finish(), compose.compose_bands(surface ^ 2, band, W, 8))
    # colour-only transport (bench.py's choice for N > 1): the composed RGB16F image is the colour channels of the composed surface
    rgb = compose.BandComposer(live, band, W, 8, depth=2, transport="rgb16f")
    for frame in range(3):
        live[lo:hi] = surface[lo:hi] ^ frame
        slot = rgb.submit()
        live[lo:hi] = 0xEE
        rgb.work[slot].wait()
        want = compose.rgb_of(compose.compose_bands(surface ^ frame, band, W, 8))
        assert rgb.out[slot].shape == want.shape and torch.equal(rgb.out[slot], want), f"rgb transport, frame {frame}"
    assert torch.equal(rgb.finish(), compose.rgb_of(compose.compose_bands(surface ^ 2, band, W, 8)))
    # two frames in flight (bench.py's default): the frames alternate between the targets of two passes, one composer gathers both
    targets = [surface.clone(), surface.clone()]
    for transport in ("surface", "rgb16f"):
        both = compose.BandComposer(targets[0], band, W, 8, depth=2, transport=transport)
        for frame in range(4):
            tgt = targets[frame & 1]
            tgt[lo:hi] = surface[lo:hi] ^ (frame + 7)
            slot = both.submit(tgt)
            both.work[slot].wait()
            want = compose.compose_bands(surface ^ (frame + 7), band, W, 8)
            assert torch.equal(both.out[slot], compose.rgb_of(want) if transport == "rgb16f" else want), f"alternating targets, {transport}, frame {frame}"
        both.finish()
    # the interleaved partition (bench.py's default for N > 1): every rank holds its rows in a COMPACT surface (chunks of 16 rows, back and
    # forth inside the groups); the composer gathers the compact surfaces rank after rank, and stripe_frame_rows says where the rows go
    full_frame = orc.OracleFrame(sc, threads=2).run().hdr
    mine = compose.stripe_frame_rows(rank, world, H, 16)
    compact = torch.from_numpy(tile(np.ascontiguousarray(full_frame[mine])).view(np.uint8).copy())
    gathered = compose.compose_bands(compact, (0, H // world), W, 8)
    per_rank = gathered.numel() // world
    rebuilt = np.zeros_like(full_frame)
    for rk in range(world):
        part = gathered[rk * per_rank:(rk + 1) * per_rank].numpy().view(np.uint64)
        rebuilt[compose.stripe_frame_rows(rk, world, H, 16)] = detile(part, W, H // world)
    assert np.array_equal(rebuilt, full_frame), "interleaved partition: the gathered compact surfaces do not rebuild the frame"
    # round 6, cost-balanced contiguous regions: bands of UNEQUAL height (rank 0 owns 40 of the 64 rows), every rank renders its own and the frame is composed with one
    # broadcast per rank and byte count -- the call sequence libbrmi_compose.so issues as one RCCL group (brmi_compose_set_bounds)
    bounds = [0, 40, 64] if world == 2 else compose.equal_bounds(world, H, 8)
    mine_band = (bounds[rank], bounds[rank + 1])
    fu = orc.OracleFrame(sc, threads=2)
    fu.cull(); fu.raster(band=mine_band); fu.depth_copy(); fu.gbuffer(band=mine_band); fu.light_cluster(); fu.shade(band=mine_band)
    mine_surface = torch.from_numpy(tile(fu.hdr).view(np.uint8).copy())
    whole = compose.compose_unequal_bands(mine_surface, bounds, W, 8)
    assert np.array_equal(detile(whole.numpy().view(np.uint64), W, H), full_frame), "unequal bands do not compose to the frame"
    # ... and the balancer every rank runs on the all-gathered times gives every rank the same partition
    times = torch.tensor([0.5 + rank], dtype=torch.float64)
    every = [torch.zeros_like(times) for _ in range(world)]
    dist.all_gather(every, times)
    bal = compose.RowBalancer(world, H, align=8, min_rows=8, bounds=bounds)
    newb = torch.tensor(bal.update([float(x.item()) for x in every]), dtype=torch.int64)
    ref = newb.clone(); dist.broadcast(ref, src=0)
    assert torch.equal(newb, ref) and newb[0] == 0 and newb[-1] == H and (newb[1:] > newb[:-1]).all()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # the max-over-ranks timing reduction of bench.py
    assert t.item() == world
    if rank == 0:
        np.save(out_path, composed.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_composition_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from basicrenderer_amd import Scene
    from basicrenderer_amd.renderer import detile
    W, H, world = 128, 64, 2
    out = str(tmp_path / "composed.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, out), nprocs=world, join=True)
    composed = np.load(out).view(np.uint64)
    full = orc.OracleFrame(Scene("tiny", W, H, point_lights=4, seed=2)).run()
    assert np.array_equal(detile(composed, W, H), full.hdr)
    assert (full.hdr != 0).any()


def test_row_balancer_finds_the_horizon_in_a_few_partitions():
    """brmi_compose_balance_rows (libbrmi_compose.so, host arithmetic): a frame whose cost sits on a horizon of a few hundred rows (the San-Miguel-class 8-GPU frame: as
    equal bands the ranks hold 171 / 483 / 2,013 / 19,420 / 4,465 / 93 / 35 / 21 visible clusters).  Fed with the band times of a synthetic cost profile -- a fixed cost
    per rank, a floor per row, a sharp ridge and a shoulder -- the bounds reach max / min <= 1.25 within four updates and stay there; bounds are multiples of the
    alignment, ascending, at least min_rows apart; bad arguments are refused."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    from basicrenderer_amd import capi, compose
    H, n = 8704, 8
    dens = np.ones(H) * 0.2 / 1088
    dens[3300:3700] += 2.5 / 400
    dens[3700:4600] += 0.8 / 900
    measure = lambda b: [float(dens[b[r]:b[r + 1]].sum() + 0.05) for r in range(n)]
    bal = compose.RowBalancer(n, H)
    assert bal.bounds == [1088 * k for k in range(9)]
    first = measure(bal.bounds)
    assert max(first) / min(first) > 10
    ratios = []
    for _ in range(8):
        ms = measure(bal.bounds)
        ratios.append(max(ms) / min(ms))
        b = bal.update(ms)
        assert b[0] == 0 and b[-1] == H and all(x % 16 == 0 for x in b) and all(b[k + 1] - b[k] >= 32 for k in range(n))
    assert max(ratios[4:]) <= 1.4 and min(ratios) <= 1.36, ratios          # (the 16-row step on the ridge is 0.1 ms: the last few percent are quantisation)
    assert max(measure(bal.bounds)) < 0.25 * max(first)
    # a frame that is already balanced is left alone (hysteresis)
    flat = compose.RowBalancer(4, 4352)
    assert flat.update([1.0, 1.01, 0.99, 1.0]) == [0, 1088, 2176, 3264, 4352]
    lib = capi.compose_lib()
    ms, bi, bo, cost = (C.c_float * 2)(1.0, 1.0), (C.c_uint32 * 3)(0, 32, 64), (C.c_uint32 * 3)(), (C.c_float * 4)()
    assert lib.brmi_compose_balance_rows(ms, bi, 2, 64, 16, C.c_float(1.0), 16, cost, bo) == 0
    assert lib.brmi_compose_balance_rows(ms, bi, 2, 60, 16, C.c_float(1.0), 16, cost, bo) == -1        # height not a multiple of the alignment
    assert lib.brmi_compose_balance_rows(ms, bi, 2, 64, 16, C.c_float(1.0), 48, cost, bo) == -1        # two bands of 48 rows do not fit 64


def test_interleaved_partition_is_a_partition_of_the_rows():
    """stripe_frame_rows (the Python statement of brmi_config::stripe*): over the ranks every row of the frame appears exactly once, a rank's rows
    come in whole chunks, and the chunk a rank owns inside a group runs back and forth from group to group."""
    sys.path.insert(0, ROOT)
    from basicrenderer_amd import compose
    strong = [(n,) + (compose.strong_frame(n)[0][1], compose.strong_frame(n)[1]) for n in (2, 4, 8)]      # bench.py's strong leg: 3840 x 2176 in chunks of 64 / 32 / 16 rows
    assert strong == [(2, 2176, 64), (4, 2176, 32), (8, 2176, 16)] and compose.strong_frame(1) == ((3840, 2176), 0)
    for n, H, rows in [(8, 1088 * 8, 64), (4, 1088 * 4, 16), (2, 64, 16), (3, 288, 32)] + strong:
        seen = np.zeros(H, dtype=int)
        for r in range(n):
            fr = compose.stripe_frame_rows(r, n, H, rows)
            assert len(fr) == H // n
            seen[fr] += 1
            chunks = fr.reshape(-1, rows)
            assert (np.diff(chunks, axis=1) == 1).all() and (chunks[:, 0] % rows == 0).all()
            slots = (chunks[:, 0] // rows) % n
            assert (slots[0::2] == r).all() and (slots[1::2] == n - 1 - r).all()
        assert (seen == 1).all()
    for bad in ((8, 1080 * 8, 64), (2, 64, 8)):
        try:
            compose.stripe_frame_rows(0, *bad)
            raise AssertionError("accepted a chunk height that does not fit")
        except ValueError:
            pass
