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
    from basicrenderer_amd.renderer import tile
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
