"""Worker of test_rccl_composition_of_unequal_bands_with_two_ranks (tests/test_parity_gpu.py): two ranks, one GPU each, launched by torch.distributed.run.
Every rank owns a band of its own height of a 256 x 128 frame, libbrmi_compose.so composes the frame with one RCCL group of broadcasts (brmi_compose_set_bounds), the
bounds move between frames, and every rank checks the composed frame against what both ranks were known to hold (the surfaces are seeded by rank and frame)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from basicrenderer_amd import compose
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group("nccl")
    W, H = 256, 128
    nbytes = H // 8 * (W // 8) * 64 * 8

    def surface_of(r, frame):
        g = torch.Generator().manual_seed(1000 * frame + r)
        return torch.randint(0, 255, (nbytes,), dtype=torch.uint8, generator=g)

    for transport in ("surface", "rgb16f"):
        live = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        c = compose.NativeBandComposer(live, (0, 8), W, 8, depth=2, transport=transport, frame_height=H)
        for frame, bounds in enumerate(([0, 48, H], [0, 104, H], [0, 8, H])):
            c.set_bounds(bounds)
            live.copy_(surface_of(rank, frame))
            c.submit()
            got = c.finish()
            torch.cuda.synchronize()
            want = torch.empty(nbytes, dtype=torch.uint8)
            for r in range(world):
                lo, hi = compose.band_byte_range((bounds[r], bounds[r + 1]), W, 8)
                want[lo:hi] = surface_of(r, frame)[lo:hi]
            want = want.cuda()
            assert torch.equal(got, want if transport == "surface" else compose.rgb_of(want)), f"rank {rank}, {transport}, frame {frame}: the composed frame differs"
        c.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("rccl unequal bands ok")


if __name__ == "__main__":
    main()
