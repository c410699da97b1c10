"""One rank of the multi-process peer-write composition tests (tests/test_parity_gpu.py).  Started as a fresh child process per rank (no exec
after GPU initialisation): all ranks live on cuda:0, exchange their hipIpcMemHandles through files in a scratch directory and compose
`frames` frames of a synthetic RGBA16F surface whose bytes depend on (rank, frame).  Writes the composed image of the last frame to
<dir>/composed_<rank>.npy; the parent checks it against the bytes both ranks must have contributed."""
import os
import sys
import time

import numpy as np


def surface_bytes(rank, frame, nbytes):
    """The rank's whole tiled surface for a frame (deterministic, different per rank and frame)."""
    rng = np.random.Generator(np.random.PCG64(1234 + 1000 * frame + rank))
    return rng.integers(0, 256, nbytes, dtype=np.uint8)


def moving_bounds(frame, world, rows):
    """Round 6: the partition of frame `frame` under cost-balanced bands -- unequal heights that move from frame to frame (multiples of 8, every band at least 8 rows)."""
    b = [0] + [k * (rows // world) + 8 * ((frame + 2 * k) % 3 - 1) for k in range(1, world)] + [rows]
    return b


def main():
    root, scratch, rank, world, transport, frames = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
    slabs = int(sys.argv[7]) if len(sys.argv) > 7 else 0          # > 0: every frame goes through brmi_compose_submit_rows in this many slabs of rows
    pipelined = len(sys.argv) > 8 and sys.argv[8] == "pipelined"   # no host wait between frames: the surface is rewritten on the render stream behind brmi_compose_wait_source
    balanced = len(sys.argv) > 8 and sys.argv[8] == "balanced"     # bands of unequal, moving height at their own rows of the composed frame (brmi_compose_set_bounds)
    sys.path.insert(0, root)
    import torch
    from basicrenderer_amd import compose
    dev = torch.device("cuda:0")
    W, rows = 256, 32 * world
    band = (rank * 32, (rank + 1) * 32)
    nbytes = (W // 8) * (rows // 8) * 64 * 8
    surf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)

    def exchange(mine):
        with open(os.path.join(scratch, f"handle_{rank}.tmp"), "wb") as f:
            f.write(mine)
        os.replace(os.path.join(scratch, f"handle_{rank}.tmp"), os.path.join(scratch, f"handle_{rank}.bin"))
        out = []
        for r in range(world):
            path, t0 = os.path.join(scratch, f"handle_{r}.bin"), time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 60:
                    raise RuntimeError(f"rank {r} never exported its handles")
                time.sleep(0.01)
            out.append(open(path, "rb").read())
        return out

    comp = compose.PeerBandComposer(surf, band, W, 8, depth=2, transport=transport, rank=rank, world=world, exchange=exchange, timeout_ms=20000, frame_height=rows if balanced else 0)
    last = None
    if balanced:
        for f in range(frames):
            bounds = moving_bounds(f, world, rows)
            comp.set_bounds(bounds)
            surf.copy_(torch.from_numpy(surface_bytes(rank, f, nbytes)).to(dev))
            if rank == 1 and f == 1:
                time.sleep(0.3)
            y0, y1 = bounds[rank], bounds[rank + 1]
            if slabs and y1 - y0 >= 16:      # two slabs of rows on the composer's stream
                mid = (y0 + (y1 - y0) // 2) // 8 * 8
                comp.submit_rows(y0, mid); comp.submit_rows(mid, y1)
            else:
                comp.submit()
            last = comp.finish()
            torch.cuda.synchronize()
        comp.wait_status()
        np.save(os.path.join(scratch, f"composed_{rank}.npy"), last.cpu().numpy())
        frames = 0
    if pipelined and not balanced:
        # Frames in flight as a renderer has them: every frame's bytes are in HBM already, the "shading" of a frame is a device copy into THE surface on the
        # render stream, the slabs go to the composer's stream, and nothing waits on the host.  One rank is late, so the others' composer streams sit in the
        # wait for its slot while their render streams run on: only brmi_compose_wait_source keeps the next frame's shading off rows that are not copied yet.
        staged = [torch.from_numpy(surface_bytes(rank, f, nbytes)).to(dev) for f in range(frames)]
        torch.cuda.synchronize()
        step = 32 // slabs
        for f in range(frames):
            if rank == 1 and f == 1:
                time.sleep(0.5)
            if not os.environ.get("BRMI_TEST_SKIP_WAIT_SOURCE"):
                comp.wait_source()
            surf.copy_(staged[f], non_blocking=True)
            for k in range(slabs):
                comp.submit_rows(band[0] + k * step, band[0] + (k + 1) * step)
        last = comp.finish()
        torch.cuda.synchronize()
        comp.wait_status()
        np.save(os.path.join(scratch, f"composed_{rank}.npy"), last.cpu().numpy())
        np.save(os.path.join(scratch, f"composed_prev_{rank}.npy"), comp.slot_image((frames - 2) % 2).cpu().numpy())      # the frame before the last is still in the other slot
        frames = 0
    for f in range(frames):
        surf.copy_(torch.from_numpy(surface_bytes(rank, f, nbytes)).to(dev))
        if rank == 1 and f == 1:
            time.sleep(0.3)      # one rank late: the other one's wait kernels really wait
        if slabs:
            # the slabs of a frame one by one, as a renderer that shades its band in row slabs would hand them over (the stores travel on the composer's stream)
            step = 32 // slabs
            for k in range(slabs):
                comp.submit_rows(band[0] + k * step, band[0] + (k + 1) * step)
        else:
            comp.submit()
        last = comp.finish()
        torch.cuda.synchronize()
    if not pipelined and not balanced:
        comp.wait_status()
        np.save(os.path.join(scratch, f"composed_{rank}.npy"), last.cpu().numpy())
    # both ranks keep their buffers mapped until the other one is done reading
    open(os.path.join(scratch, f"done_{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(scratch, f"done_{r}")) for r in range(world)) and time.time() - t0 < 60:
        time.sleep(0.01)
    comp.close()


if __name__ == "__main__":
    main()
