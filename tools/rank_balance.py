#!/usr/bin/env python3
"""tools/rank_balance.py -- how evenly the screen partition of `bench.py --gpus N` loads the ranks, measured on ONE GPU.

Every rank of the N-GPU frame (7680 x 1080 N, weak scaling) is rendered in turn on this GPU, without the composition, in the bench's
arrangement (two frames in flight), and its ms per frame is reported.  `--chunks K` gives a rank K contiguous chunks instead of one band --
rank r owns chunks r, r + N, ..., each 1080 / K rows -- rendered by K pass rings side by side on their own stream pairs: the interleaved
assignment of SURVEY.md 8(e) at the granularity that keeps the geometry half proportional to 1 / N.  Prints one JSON
line: per-rank times, max / min, and the visible clusters each rank rasterises.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--stripe-rows", type=int, default=0, help="> 0: the in-kernel interleaved partition (brmi_config::stripe*) with chunks of this many rows, one pass ring per rank, "
                                                                "on a frame of 1088 rows per rank (1080 has no multiple of 16 among its divisors)")
    ap.add_argument("--workload", default="bistro")
    ap.add_argument("--leg", default=None, choices=["weak", "strong"],
                    help="round 5: one of bench.py --gpus N's two legs (multi_gpu_legs) with its frame and chunk height -- weak: 7680 x 1088 N in chunks of 64 rows; strong: the "
                         "4K frame, 3840 x 2176, in chunks of 128 / N rows -- and the one-GPU frame it is measured against, timed in the same call (`n1_ms`)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--frames-in-flight", type=int, default=2)
    ap.add_argument("--only", type=int, nargs="*", help="ranks to measure (default: all)")
    ap.add_argument("--partition", default=None, choices=["balanced"],
                    help="round 6: cost-balanced contiguous regions (brmi_set_band + brmi_compose_balance_rows) on the leg's frame -- ONE ring of passes with dynamicBand renders "
                         "the ranks' bands in turn, the balancer moves the bounds after every round of measurements (--balance-rounds), the last round is the table")
    ap.add_argument("--balance-rounds", type=int, default=6)
    args = ap.parse_args()
    import torch
    import bench
    from basicrenderer_amd import Scene, compose
    from basicrenderer_amd.renderer import VisibilityRenderer
    dev = torch.device("cuda:0")
    n, K, fif = args.ranks, args.chunks, args.frames_in_flight
    W, H = compose.frame_size(n)
    ref_frame = None
    if args.leg == "weak":
        (W, H), args.stripe_rows, ref_frame = compose.frame_size(n, "stripes"), args.stripe_rows or 64, compose.frame_size(1)
    elif args.leg == "strong":
        (W, H), args.stripe_rows = compose.strong_frame(n)
        ref_frame = (W, H)
    elif args.stripe_rows:
        H = 1088 * n
    preset, kw, features = bench.WORKLOADS[args.workload]
    scene = Scene(preset, W, H, point_lights=bench.LIGHTS[args.workload], directional=True, material_features=features, **kw)
    rows = H // (n * K)
    if not args.stripe_rows and (rows % 8 or rows * n * K != H):
        raise SystemExit(f"{H} rows do not split into {n} x {K} chunks of whole 8-row tiles")
    # stream pairs made once (HIP maps streams onto a few hardware queues): one geometry stream and `fif` shading streams per chunk
    geo = [torch.cuda.Stream(dev, priority=-1) for _ in range(K)]
    shade = [[torch.cuda.Stream(dev, priority=0) for _ in range(fif)] for _ in range(K)]
    out = {"ranks": n, "stripe_rows": args.stripe_rows, "chunks_per_rank": K, "rows_per_chunk": rows, "frame": [W, H], "workload": args.workload, "frames_in_flight": fif, "per_rank": []}
    if args.leg:
        out["leg"] = args.leg
    if ref_frame is not None:
        # the one-GPU frame the leg is measured against (bench.py: n1_reference), same arrangement, same call
        ref_scene = scene if ref_frame == (W, H) else Scene(preset, ref_frame[0], ref_frame[1], point_lights=bench.LIGHTS[args.workload], directional=True, material_features=features, **kw)
        ring = [VisibilityRenderer(ref_scene, device=dev, stats=(j == 0), occlusion=True) for j in range(fif)]
        for j in range(fif):
            ring[j].set_history_source(ring[(j - 1) % fif])
        for i in range(10 + args.steps):
            if i == 10:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            p = ring[i % fif]
            with torch.cuda.stream(geo[0]):
                p.update(); p.execute(shade[0][i % fif])
        torch.cuda.synchronize()
        out["n1_ms"], out["n1_frame"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4), list(ref_frame)
        for p in ring:
            p.close()
        del ring, ref_scene
        torch.cuda.empty_cache()
    if args.partition == "balanced":
        # one ring for every rank and round: the band moves (brmi_set_band), the surfaces are the frame's
        ring = [VisibilityRenderer(scene, device=dev, stats=(j == 0), band=(0, H // n // 16 * 16), occlusion=True, dynamicBand=1) for j in range(fif)]
        for j in range(fif):
            ring[j].set_history_source(ring[(j - 1) % fif])
        bal = compose.RowBalancer(n, H, align=16, min_rows=32)
        out["partition"], out["rounds"] = "balanced", []
        for rnd in range(args.balance_rounds):
            bounds, per = list(bal.bounds), []
            for r in range(n):
                for p in ring:
                    p.set_band(bounds[r], bounds[r + 1])
                steps = args.steps if rnd == args.balance_rounds - 1 else max(20, args.steps // 3)
                for i in range(10 + steps):
                    if i == 10:
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                    p = ring[i % fif]
                    with torch.cuda.stream(geo[0]):
                        p.update(); p.execute(shade[0][i % fif])
                torch.cuda.synchronize()
                c = ring[0].counters()
                per.append({"rank": r, "rows": [bounds[r], bounds[r + 1]], "ms_per_frame": round((time.perf_counter() - t0) / steps * 1e3, 4), "visible_clusters": int(c.visibleClusters), "held": int(c.reserved[1])})
            t = [x["ms_per_frame"] for x in per]
            out["rounds"].append({"bounds": bounds, "ms": t, "max_over_min": round(max(t) / min(t), 3)})
            out["per_rank"] = per
            if rnd + 2 < args.balance_rounds:
                bal.update(t)
            elif rnd + 2 == args.balance_rounds:
                bal.update(t); bal.settle()      # the last round measures the best partition found
        for p in ring:
            p.close()
    for r in ([] if args.partition == "balanced" else (args.only if args.only else range(n))):
        rings = []
        for k in range(K):
            c = r + k * n
            band = (c * rows, (c + 1) * rows)
            if args.stripe_rows:
                ring = [VisibilityRenderer(scene, device=dev, stats=(j == 0), stripes=(args.stripe_rows, n, r), occlusion=True) for j in range(fif)]
            else:
                ring = [VisibilityRenderer(scene, device=dev, stats=(j == 0), band=band, occlusion=True) for j in range(fif)]
            for j in range(fif):
                ring[j].set_history_source(ring[(j - 1) % fif])
            rings.append(ring)

        def frame(i):
            for k, ring in enumerate(rings):
                p = ring[i % fif]
                with torch.cuda.stream(geo[k]):
                    p.update()
                    p.execute(shade[k][i % fif])

        for i in range(10):
            frame(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        vis = sum(int(ring[0].counters().visibleClusters) for ring in rings)
        out["per_rank"].append({"rank": r, "ms_per_frame": round(ms, 4), "visible_clusters": vis})
        for ring in rings:
            for p in ring:
                p.close()
        torch.cuda.empty_cache()
    t = [x["ms_per_frame"] for x in out["per_rank"]]
    out["max_over_min"] = round(max(t) / min(t), 3)
    out["max_ms"], out["min_ms"], out["mean_ms"] = max(t), min(t), round(sum(t) / len(t), 4)
    if "n1_ms" in out:
        # render-side efficiency of the slowest rank: weak = one rank's share of pixels per second against the one-GPU frame's; strong = speed-up / N
        px_rank, px_ref = W * H / n, out["n1_frame"][0] * out["n1_frame"][1]
        out["render_side_efficiency"] = round((px_rank / max(t)) / (px_ref / out["n1_ms"]), 4) if args.leg == "weak" else round(out["n1_ms"] / max(t) / n, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
