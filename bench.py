#!/usr/bin/env python3
"""bench.py -- shaded Mpixels/s of the visibility-buffer + clustered-resolve path on N MI355X.

One "step" = one pass of the hot path over one synthetic frame already resident in HBM:
clear -> cull -> software raster -> G-buffer(+depth) -> light clustering -> OpenPBR shade
(-> RCCL all-gather of the HDR bands when N > 1).

N = 1 : BASELINE.json configs[2], "Bistro 4K, meshlet cull + vis-buffer raster, 256 point lights" (3840x2160) -- the frame the
        north star's one numeric target is stated on (>= 60 fps).  The same run then measures configs[1] ("Sponza 4K,
        visibility-buffer + clustered resolve, 1 dir + 64 point lights") and reports it as `configs1` inside the line.
N > 1 : BASELINE.json configs[3] (San-Miguel-class scene, 30 % alpha-tested + texture-sampled materials), two legs in one line (`weak`, `strong`; multi_gpu_legs), each
        with the one-GPU frame it is measured against, rendered inside the same job.  The top-level fields are the weak leg's:
        weak scaling by screen tile: the frame is 7680 x 1088 N (7680 x 1080 N with --partition bands), geometry replicated; every rank shades
        one 4K frame's worth of pixels -- by default the chunks of 64 rows the interleaved partition deals it (brmi_config::stripe*: one chunk
        of every group of N, compact surfaces; the slowest of 8 ranks takes 0.78 ms per frame against 1.62 ms with contiguous bands, measured
        rank by rank on one GPU) -- and the HDR shares are composed so that every rank holds all of them (one RCCL all-gather per frame, or
        --composer peer: stores into hipIpc-mapped images, no collective).  Without a launcher around it (no WORLD_SIZE) `--gpus N` starts its
        own N ranks; it never reports a 1-GPU number for an N-GPU request.  `--emulate-rank R` renders one rank's share alone on one GPU.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (largest mean stage time):
algorithmic bytes of SURVEY.md 8(d) / mean launch duration from HIP events on the execute stream.
`cpu_baseline` times the CPU oracle (a port of the reference HLSL; test infrastructure) on one
frame of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# workload -> (scene preset, Scene keyword arguments, default material feature bits)
#   bistro        BASELINE.json configs[2] as SURVEY.md 8(d) states it: ~3.0 M triangles in ~150 meshes, ~2,000 instances, every mesh through the
#                 library's cluster-LOD builder; surfaces carry relief at every scale (reliefSlope) so that the builder's errors keep fine levels alive
#   bistro_r2     the frame rounds 1-2 timed under that name: 3.0 M INSTANCED triangles, quadtree DAGs, smooth surfaces (2 k visible clusters)
#   bistro_dense  the r2 street with 20 x the triangle budget and fractal relief: > 20 k visible clusters, > 150 k meshlets tested per frame
#   san_miguel    configs[3]: 30 % of the materials alpha tested, most of them texture sampled (material features 24), as SURVEY.md 8(d) config 4 says
WORKLOADS = {"sponza": ("sponza", {}, 0),
             "bistro": ("bistro", dict(unique_budget=True, lod_builder="own", relief_slope=1.5), 0),
             "bistro_r2": ("bistro", {}, 0),
             "san_miguel": ("san_miguel", {}, 24),
             "bistro_dense": ("bistro", dict(size_scale=20.0, detail=96.0), 0),
             # zorah       configs[4] on ONE GPU: 8K, 100 k instances of two 459 k-triangle meshes (39 G instanced triangles), 1 % of the instances skinned;
             #             ~670 k visible clusters out of ~1 M meshlets tested -- the LOD-select / traversal regime the reference is built for (README.md:11)
             "zorah": ("zorah", dict(skinned_fraction=0.01), 0),
             # bistro_skinned  the headline frame with 30 % of its instances skinned (compute skinning folded into cull bounds, the rasteriser's vertex stage
             #             and the G-buffer pass's vertex fetch -- four joints per vertex): what the skinned paths cost under the driver's clock
             "bistro_skinned": ("bistro", dict(unique_budget=True, lod_builder="own", relief_slope=1.5, skinned_fraction=0.3), 0)}
LIGHTS = {"sponza": 64, "bistro": 256, "bistro_r2": 256, "san_miguel": 256, "bistro_dense": 256, "zorah": 64, "bistro_skinned": 256}
FRAME_SIZE = {"zorah": (7680, 4320)}      # N = 1 frame of a workload that is not the 4K one
BASELINE_CONFIG = {"sponza": "configs[1]", "bistro": "configs[2]", "bistro_r2": "configs[2], the instanced-budget frame of rounds 1-2", "san_miguel": "configs[3]",
                   "bistro_dense": "configs[2], dense geometry", "zorah": "configs[4] on one GPU",
                   "bistro_skinned": "configs[2], 30 % of the instances skinned"}
PATH_STEP = float(os.environ.get("BRMI_BENCH_PATH_STEP", "0.02"))        # --camera-path: position on the preset's camera path advances by this much per frame (one unit = 0.35 m sideways, 0.6 m ahead, 4 degrees)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_PEAK_WAVE_INSTS = 1.2288e12   # wave64 VALU instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md, cycle constants)
# ... and what tools/valu_issue_probe.hip measured on the box (profiles/r03_valu_issue_probe.txt): independent v_fma / v_mul / v_mov streams at 2-4 waves
# per SIMD issue one wave64 instruction per 2.3-2.5 cycles per SIMD at the 2.2-2.4 GHz the chip holds (0.93-1.06 T/s); three-source selects /
# medians one per 4.2 cycles, v_rcp / v_exp / v_sqrt one per 8.2 cycles (0.30 T/s)
VALU_PEAK_MEASURED = 0.96e12
# profiles/<tag>_traffic.json (tools/profile.sh), keyed by (workload, material feature bits): traffic of another configuration is not this one's
PROFILE_TAG = {("sponza", 0): "r06_sponza4k", ("bistro", 0): "r06_bistro4k", ("bistro_r2", 0): "r02_bistro4k", ("san_miguel", 0): "r02_sanmiguel4k", ("bistro_dense", 0): "r06_bistro4k_dense",
               ("san_miguel", 24): "r06_sanmiguel4k", ("zorah", 0): "r06_zorah8k", ("sponza", 136): "r02_sponza4k_parallax"}
PROFILE_FALLBACK = {"r06_sponza4k": "r05_sponza4k", "r06_bistro4k": "r05_bistro4k", "r06_bistro4k_dense": "r05_bistro4k_dense", "r06_sanmiguel4k": "r05_sanmiguel4k", "r06_zorah8k": "r05_zorah8k"}      # until the round's profiles are committed
_STREAM_CACHE = {}
DOMINANT_KERNEL = {"raster": "k_raster", "gbuffer": "k_gbuffer", "shade": "k_shade", "cull": "k_traverse+k_cull_clusters", "clear": "k_clear_vis",
                   "light_cluster": "k_light_clustering", "depth_copy": "k_depth_copy", "hzb": "k_hzb_head", "cull2": "k_traverse+k_cull_clusters", "raster2": "k_raster"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS),
                    help="default: bistro for N = 1 (BASELINE.json configs[2], the north star's target frame), san_miguel for N > 1 (configs[3]: \"San-Miguel 4K, screen-tile partition "
                         "across 2/4/8\").  sponza = configs[1]; zorah = configs[4] on one GPU; "
                         "bistro_dense: the Bistro-class street with 20x the triangle budget and fractal relief, so that the 1 px LOD test keeps "
                         "pixel-sized triangles: > 20 k visible clusters, > 150 k meshlets tested per 4K frame (SURVEY.md 8 a-3's regime)")
    ap.add_argument("--no-second", action="store_true", help="N = 1 only: do not add the configs[1] (Sponza) measurement as `configs1` to the line")
    ap.add_argument("--no-dense", action="store_true", help="N = 1 only: do not add the dense-geometry measurement (bistro_dense) as `dense` to the line")
    ap.add_argument("--no-skinned", action="store_true", help="N = 1 only: do not add the skinned leg (the headline frame with 30 %% of its instances skinned) as `skinned` to the line")
    ap.add_argument("--no-third", action="store_true", help="N = 1 only: do not add the configs[3] measurement (San-Miguel-class frame, 30 %% alpha-tested and texture-sampled materials, "
                                                             "one GPU) as `configs3` to the line")
    ap.add_argument("--no-fourth", action="store_true", help="N = 1 only: do not add the configs[4] measurement (Zorah-class 8K frame: 100 k instances, 1 %% skinned, ~670 k visible "
                                                              "clusters, one GPU) as `configs4` to the line")
    ap.add_argument("--camera-path-fast", type=float, default=0.1,
                    help="N = 1 only: path units per frame of a second, faster camera-path leg (`path_fast`: hundreds of phase-2 clusters per frame); 0 skips it")
    ap.add_argument("--camera-path", type=int, default=200,
                    help="N = 1 only: after the static-camera region, K frames along the preset's camera path (a new camera every frame, so that phase 2 of the "
                         "occlusion chain has work) reported as `path`; 0 skips it")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps frames each; the median is reported.  0 = 5 when --steps < 200, else 1")
    ap.add_argument("--occlusion", type=int, default=1, choices=[0, 1],
                    help="2-phase HZB occlusion culling (reference default: on, BR/include/Renderer.h:220); timed frames are steady state")
    ap.add_argument("--lod-builder", default=None, choices=["quadtree", "own"],
                    help="own: mesh LOD DAGs from the library's cluster-LOD builder (irregular meshlets, ~384-cluster groups) instead of the generator's quadtree; default: the workload's (bistro: own)")
    ap.add_argument("--material-features", type=int, default=None,
                    help="default: the workload's (san_miguel: 24, else 0).  scene generator feature bits (brmi_scene.h): 1 coat, 2 fuzz, 4 mirrored instances, 8 texture-sampled materials, 16 alpha-tested materials, 32 vertex colours, 64 OpenPBR layer textures, 128 parallax; 0 = BASELINE.json's constant-factor configuration")
    ap.add_argument("--partition", default="auto", choices=["auto", "stripes", "bands", "balanced"],
                    help="N > 1: how the frame is split.  stripes: the interleaved partition of SURVEY.md 8(e) -- chunks of --stripe-rows rows, one "
                         "chunk of every group of N per rank, compact surfaces (brmi_config::stripe*).  balanced (round 6): contiguous bands whose boundaries follow the ranks' "
                         "measured frame times (brmi_set_band + brmi_compose_balance_rows; --balance-rounds rounds of --balance-frames frames before the timed region, then fixed): a "
                         "cluster is set up by one rank and the band test drops the rest of the hierarchy.  bands: equal contiguous bands of 1080 rows.  auto (default): balanced -- "
                         "what the rank-by-rank emulation on one GPU measured as the better of the two on both legs (profiles/r06_rank_balance.md)")
    ap.add_argument("--balance-rounds", type=int, default=6)
    ap.add_argument("--balance-frames", type=int, default=24)
    ap.add_argument("--bounds", default=None, help="--partition balanced with --emulate-rank: the row bounds to render with (comma separated, N + 1 values), e.g. the last round of tools/rank_balance.py")
    ap.add_argument("--stripe-rows", type=int, default=64, help="chunk height of the interleaved partition (a multiple of 16 that divides 1088: 16, 32, 64, 272, 544)")
    ap.add_argument("--transport", default="rgb16f", choices=["rgb16f", "surface"],
                    help="what the band composition gathers: the colour channels as RGB16F (default; the composed image has no alpha plane) or the RGBA16F surface bytes")
    ap.add_argument("--composer", default="native", choices=["native", "peer", "torch"],
                    help="who composes the bands: libbrmi_compose.so with one RCCL all-gather per frame (native; default), libbrmi_compose.so's peer-write path "
                         "(peer: hipIpc-mapped output buffers, every rank stores its band into every other rank's image, no collective), or torch.distributed")
    ap.add_argument("--compose-slabs", type=int, default=0,
                    help="--composer peer: the deferred shading runs in this many row slabs and every slab's rows are handed to the composer as soon as its launches "
                         "are enqueued (brmi_set_shade_slabs + brmi_compose_submit_rows): the stores travel on the composer's stream while the next slab is shaded")
    ap.add_argument("--force-compose", action="store_true", help="run the RCCL band composition even with one rank (checks the collective path on a single GPU)")
    ap.add_argument("--frames-in-flight", type=int, default=3, choices=[1, 2, 3, 4, 5, 6],
                    help="3 (default: the reference's numFramesInFlight default, Renderer.h:110) or 2: that many passes with their own resources render the frames "
                         "in turn on a geometry stream and a shading stream (brmi_set_history_source + brmi_execute_split) -- frame k+1's culling and rasterisation, "
                         "which are latency-bound and leave most of the chip idle, overlap frame k's G-buffer and shading; with two passes a pass's geometry half waits for "
                         "its own shading half of two frames ago, which the third pass removes (round 4: Bistro-class 0.535 -> 0.510 ms, Sponza-class 0.412 -> 0.387, "
                         "the dense and San-Miguel-class frames unchanged).  1: one pass, one stream, frames back to back.  "
                         "Every frame does all of its work either way; "
                         "the roofline block is measured on serial frames (a kernel's duration while it shares the CUs with another frame is not its own)")
    ap.add_argument("--keep-uniform-layer-planes", type=int, default=0, choices=[0, 1],
                    help="brmi_config::keepUniformLayerPlanes: 1 = the coat / fuzz G-buffer planes of a scene whose materials all store the same word are filled "
                         "once and not rewritten (G-buffer kernel 93 -> 84 us).  Default 0: every frame writes every plane")
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="testing on ONE GPU: render rank R's share of the --gpus N frame alone (same partition, frame size and passes as the N-GPU run; "
                         "no process group; the composer, if forced, gathers this rank's share only).  The line says so and is not an N-GPU number")
    ap.add_argument("--camera-at", type=float, default=None,
                    help="experiments: the static camera of every measured leg sits at this position of the preset's camera path (path units; default: the preset's camera)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=float, default=1.0, help="fraction of the frame height the CPU baseline renders")
    ap.add_argument("--cpu-scale-1thread", type=float, default=0.25,
                    help="fraction of the frame height the single-thread CPU baseline renders (BASELINE.md 4(a): 1 thread and all cores; culling covers the whole frame either way)")
    ap.add_argument("--legs", default="weak,strong", help="N > 1: which legs to run (weak, strong, or both)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="N > 1, checks only: all N ranks render on cuda:0 of a one-GPU box -- torch.distributed over gloo (CPU tensors), the peer-write composer between the "
                         "processes (RCCL refuses two ranks on one device).  Runs the whole N-rank control flow of this file (balancing rounds, composition, the reductions, "
                         "the line); its numbers are N processes sharing one GPU, not an N-GPU measurement, and the line says so")
    args = ap.parse_args()
    if args.gpus < 1:
        fail_line(args, "--gpus must be >= 1")
    args.workload_given = args.workload is not None
    if args.workload is None:
        args.workload = "bistro" if args.gpus == 1 else "san_miguel"
    if args.emulate_rank is not None:
        if not (0 <= args.emulate_rank < args.gpus):
            fail_line(args, f"--emulate-rank {args.emulate_rank} of --gpus {args.gpus}")
        import torch
        torch.cuda.set_device(0)
        out = multi_gpu_legs(args, args.gpus, args.emulate_rank, 0, emulated=True)
        out["emulated"] = f"rank {args.emulate_rank} of {args.gpus} alone on one GPU: value counts all {args.gpus} ranks' pixels over THIS rank's time; not an N-GPU measurement"
        import ctypes
        ctypes.CDLL(None).fflush(None)          # (RCCL's version banner sits in the C stdout buffer under --force-compose: the JSON stays the last line)
        print(json.dumps(out), flush=True)
        return
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # No launcher around us: start the N ranks ourselves -- as a CHILD torchrun, before this process has touched the GPU -- and
        # exit with its code.  (Never an exec from a process that initialised HIP, and never a silent fall-back to one rank.)
        sys.exit(self_launch(args))
    if world_env is not None and int(world_env) != args.gpus:
        fail_line(args, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")

    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() < args.gpus and not args.shared_gpu:
        fail_line(args, f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.shared_gpu:
        local_rank = 0
        if args.composer == "native" and not os.environ.get("BRMI_BENCH_KEEP_COMPOSER"):      # (kept: RCCL refuses the second rank of a device and the run shows its fall-back to peer writes)
            args.composer = "peer"
    if world > 1 or args.force_compose:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.shared_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus
    torch.cuda.set_device(local_rank)

    if world > 1:
        out = multi_gpu_legs(args, world, rank, local_rank, emulated=False)
        if out is not None and args.shared_gpu:
            out["shared_gpu"] = f"all {world} ranks rendered on ONE GPU (gloo + the peer-write composer between the processes): a check of the N-rank control flow, not an N-GPU measurement"
    else:
        out = measure(args, args.workload, world, rank, local_rank, cpu=(world == 1 and not args.no_cpu_baseline))
    if world == 1 and not args.no_second and args.workload != "sponza":
        # the metric string's own wording ("vis-buffer+resolve") is configs[1]: measured by the same run, reported beside the target frame
        second = measure(args, "sponza", world, rank, local_rank, cpu=False, path=False)
        if out is not None:
            out["configs1"] = {k: second[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_minmax", "config", "roofline", "stage_ms") if k in second}
    if world == 1 and not args.no_third and args.workload != "san_miguel":
        # configs[3]'s frame as SURVEY.md 8(d) config 4 states it -- 30 % alpha-tested, texture-sampled materials -- on one GPU, under the same clock
        third = measure(args, "san_miguel", world, rank, local_rank, cpu=False, path="path")       # (its camera-path leg too: phase 2 of the alpha-tested scene)
        if out is not None:
            out["configs3"] = {k: third[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_minmax", "config", "roofline", "stage_ms", "serial_frame_ms", "path") if k in third}
    if world == 1 and not args.no_dense and args.workload != "bistro_dense":
        # SURVEY.md 8 a-3 / a-5's regime (tens of thousands of visible clusters, pixel-sized triangles) under the same clock as the headline
        dense = measure(args, "bistro_dense", world, rank, local_rank, cpu=False, path=False)
        if out is not None:
            out["dense"] = {k: dense[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_minmax", "config", "roofline", "stage_ms", "serial_frame_ms") if k in dense}
    if world == 1 and not args.no_skinned and args.workload != "bistro_skinned":
        # row a-10 under the same clock: the headline frame with 30 % of its instances skinned
        sk = measure(args, "bistro_skinned", world, rank, local_rank, cpu=False, path=False)
        if out is not None:
            out["skinned"] = {k: sk[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_minmax", "config", "stage_ms", "serial_frame_ms") if k in sk}
    if world == 1 and not args.no_fourth and args.workload != "zorah":
        # configs[4]'s frame on one GPU: 8K, 100 k instances, ~670 k visible clusters -- the LOD-select regime (rows a-2 / a-3 / a-10) under the same clock
        fourth = measure(args, "zorah", world, rank, local_rank, cpu=False, path=False)
        if out is not None:
            out["configs4"] = {k: fourth[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_minmax", "config", "roofline", "stage_ms", "serial_frame_ms") if k in fourth}
    result_line = json.dumps(out) if out is not None else None
    if world > 1 or args.force_compose:
        # RCCL writes a version banner to the C stdout buffer; pushed out here, on every rank, so that rank 0's JSON is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()                       # every rank has flushed before rank 0 prints
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


def multi_gpu_legs(args, n, rank, local_rank, emulated):
    """--gpus N > 1.  Legs under one clock, and the one-GPU frame each of them is measured against, made inside the same job (which scenes: `plan` below):
      weak    every rank shades one 4K frame's worth of pixels: the frame is 7680 x 1088 N, interleaved partition in chunks of --stripe-rows rows; the
              reference is the scene's 3840 x 2160 frame on one GPU (rank 0's; every rank renders it, each on its own GPU)
      strong  THE 4K frame (3840 x 2176: 2160 rows padded to a multiple of 16 N) split N ways in chunks of 128 / N rows; the reference is that frame on one GPU
    efficiency_vs_n1 = (value_N / N) / value_1 with value = pixels dispatched per second by all ranks.  The top-level fields of the line are the weak leg's
    (`scaling`: weak -- what the north star's 0.9 is stated on)."""
    from basicrenderer_amd import compose
    legs = [x for x in args.legs.split(",") if x in ("weak", "strong")] or ["weak"]
    # Round 6: without --workload the line's own value is the HEADLINE scene's weak leg -- `bench.py --gpus 1` reports that scene (configs[2]), so value(N) / N / value(1) across
    # the driver's runs compares one scene with itself -- and configs[3]'s scene (San-Miguel-class, what BASELINE.json names for the partition) runs both legs beside it, under
    # `configs3_weak` / `configs3_strong`, each with its own one-GPU reference.  With --workload: that scene's legs, `weak` / `strong`, as before.
    plan = [(leg, leg, args.workload) for leg in legs] if args.workload_given else [("weak", "weak", "bistro")] + [("configs3_" + leg, leg, "san_miguel") for leg in legs]
    result = {}
    for key, leg, wl in plan:
        if leg == "weak":
            part = args.partition if args.partition != "auto" else "balanced"
            frame, rows = compose.frame_size(n, "bands" if part == "bands" else "stripes"), args.stripe_rows
            ref_frame = compose.frame_size(1)          # the N = 1 problem of weak scaling: the scene's 4K frame (8.29 Mpixel; a rank of the N-GPU frame shades 8.36 M)
        else:
            part = args.partition if args.partition != "auto" else "balanced"
            frame, rows = compose.strong_frame(n)
            ref_frame = frame
        got = measure(args, wl, n, rank, local_rank, cpu=False, path=False, emulated=emulated, frame=frame, stripe_rows=rows, partition=part)
        ref = measure(args, wl, 1, 0, local_rank, cpu=False, path=False, emulated=True, frame=ref_frame, solo=True)      # every rank, on its own GPU; rank 0's is reported
        if got is not None:
            keep = ("value", "unit", "ms_per_step", "ms_per_step_minmax", "rank_ms_per_step", "config", "stage_ms", "serial_frame_ms", "host_issue_ms_per_step")
            entry = {k: got[k] for k in keep if k in got}
            entry["n1_reference"] = {"value": ref["value"], "ms_per_step": ref["ms_per_step"], "frame": list(ref_frame), "pixels": ref_frame[0] * ref_frame[1]}
            entry["efficiency_vs_n1"] = round(got["value"] / n / ref["value"], 4)
            entry["leg"] = leg
            result[key] = (got, entry)
    if not result:
        return None
    first = plan[0][0]
    out = result[first][0]
    out["scaling"] = plan[0][1]
    for key, (_, entry) in result.items():
        out[key] = entry
    return out


def measure(args, workload, n, rank, local_rank, cpu, path=True, emulated=False, frame=None, stripe_rows=None, solo=False, partition=None):
    """K timed steps of one workload on this rank's GPU (all ranks call it together); rank 0 returns the result object.
    frame: (W, H) instead of the workload's default; stripe_rows: chunk height of the interleaved partition instead of --stripe-rows; solo: a one-GPU
    measurement inside an N-GPU job (no process group is touched; every rank returns its own result)."""
    import torch
    import torch.distributed as dist
    from basicrenderer_amd import Scene, compose
    from basicrenderer_amd.renderer import VisibilityRenderer
    dev = torch.device(f"cuda:{local_rank}")
    cdev = torch.device("cpu") if args.shared_gpu else dev      # where the tensors of the process group's collectives live (gloo: host memory)
    lights = LIGHTS[workload]
    preset, scene_kw, default_features = WORKLOADS[workload]
    scene_kw = dict(scene_kw)
    features = default_features if args.material_features is None else args.material_features
    if args.lod_builder is not None:
        scene_kw["lod_builder"] = args.lod_builder
    lod_builder = scene_kw.get("lod_builder", "quadtree")
    multi = n > 1 and not emulated and not solo         # a process group exists
    partition = partition or (args.partition if args.partition != "auto" else "stripes")
    striped = n > 1 and partition == "stripes"
    balanced = n > 1 and partition == "balanced"
    stripe_rows = stripe_rows or args.stripe_rows
    W, H = compose.frame_size(n, "bands" if partition == "bands" else "stripes")
    if n == 1 and workload in FRAME_SIZE:
        W, H = FRAME_SIZE[workload]
    if frame is not None:
        W, H = frame
    if striped:
        compose.stripe_frame_rows(rank, n, H, stripe_rows)      # (raises on a chunk height that does not fit)
        band = (0, H // n)                                             # the rank's compact surfaces hold its rows only: the composer takes all of them
        part = dict(stripes=(stripe_rows, n, rank))
    elif balanced:
        # cost-balanced contiguous regions: full-frame surfaces, a band that moves (brmi_set_band); the first frames render equal bands
        balancer = compose.RowBalancer(n, H, align=16, min_rows=32, bounds=[int(x) for x in args.bounds.split(",")] if args.bounds else None)
        band = (balancer.bounds[rank], balancer.bounds[rank + 1])
        part = dict(band=band, dynamicBand=1)
    else:
        band = compose.band_of(rank, n, H)
        part = dict(band=band)
    scene = Scene(preset, W, H, point_lights=lights, directional=True, material_features=features, **scene_kw)
    r = VisibilityRenderer(scene, device=dev, stats=True, occlusion=bool(args.occlusion), keepUniformLayerPlanes=args.keep_uniform_layer_planes, **part)
    fif = args.frames_in_flight
    passes, streams, shade_streams = [r], [torch.cuda.current_stream(dev)], [None, None]
    if fif >= 2:
        # the second pass has its own resources and scene upload (its camera buffers are its own); phase 1 of each tests against the chain
        # the other built for the frame before
        for _ in range(fif - 1):
            passes.append(VisibilityRenderer(scene, device=dev, stats=False, occlusion=bool(args.occlusion), keepUniformLayerPlanes=args.keep_uniform_layer_planes, **part))
        if args.occlusion:
            for k in range(fif):
                passes[k].set_history_source(passes[(k - 1) % fif])        # the pass that renders the frame before
        mode = os.environ.get("BRMI_BENCH_STREAMS", "split")
        if mode == "split":
            # the passes share a geometry stream (higher priority: its launches are latency-bound and want CU slots the moment they are ready)
            # the streams are made once per process and shared by every workload measured in it: HIP maps streams onto a few hardware queues,
            # and a second set of streams created for the third workload of a run landed on the queues of the first (geometry and shading
            # halves serialised: the dense frame took 1.84 ms in flight against 0.97 ms measured on its own)
            key = (str(dev), fif, "split")
            if key not in _STREAM_CACHE:
                gp, sp = (int(v) for v in os.environ.get("BRMI_BENCH_PRIORITIES", "-1,0").split(","))     # (experiments)
                geometry, shading = torch.cuda.Stream(dev, priority=gp), torch.cuda.Stream(dev, priority=sp)
                # a shading stream per pass: a frame's pixel pass may start on the tail of the frame before's k_shade (-1 % against one shared stream)
                _STREAM_CACHE[key] = ([geometry] * fif, [shading] + [torch.cuda.Stream(dev, priority=sp) for _ in range(fif - 1)])
            streams, shade_streams = _STREAM_CACHE[key]
        else:
            key = (str(dev), fif, "whole")
            if key not in _STREAM_CACHE:
                _STREAM_CACHE[key] = ([torch.cuda.Stream(dev) for _ in range(fif)], [None] * fif)        # one stream per pass, whole frames
            streams, shade_streams = _STREAM_CACHE[key]

    if args.camera_at is not None:
        cam_at = scene.camera_at(args.camera_at)
        for q in passes:
            q.set_camera_device(torch.from_numpy(cam_at[0]).to(dev), torch.from_numpy(cam_at[1]).to(dev), cam_at[0])
        torch.cuda.synchronize()
    hdr = r.hdr_tensor()
    # all-gather of frame k overlaps the rendering of frame k + 1; the colour channels travel (RGB16F, 6 B/px): the lit target's alpha is constant
    composer, composer_used = None, args.composer
    balance_frozen = False
    if multi or (args.force_compose and not solo):
        composer_used = args.composer
        if args.composer in ("native", "peer"):
            # libbrmi_compose.so issues the composition itself.  Whether it can is decided WITHOUT a collective first -- load the library, make
            # a unique id (RCCL path) -- and agreed on by all ranks (one all_reduce): ncclCommInitRank is collective, so a rank that cannot
            # even load the library would leave the others blocked inside it.  Creation proper follows only when every rank said yes; any
            # failure there (peer path: mapping a peer's buffers) is again agreed on before the timed region, and all ranks fall back to the
            # torch.distributed composer together.
            import ctypes as C
            from basicrenderer_amd import capi
            failed, why = 0, ""
            try:
                lib = capi.compose_lib()
                if args.composer == "native" and rank == 0 and lib.brmi_compose_unique_id((C.c_uint8 * 128)()) != 0:
                    raise RuntimeError("brmi_compose_unique_id failed")
            except Exception as e:      # noqa: BLE001
                failed, why = 1, f"{type(e).__name__}: {e}"

            def agree(flag):
                if multi:
                    t = torch.tensor([flag], device=cdev, dtype=torch.int32)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    return int(t.item())
                return flag
            failed = agree(failed)
            # (round 6: the RCCL composer has never run with more than one rank on this pool -- RCCL refuses two ranks on one device -- so a failure of it on the driver's
            # 8-GPU node must not cost the run: the peer-write composer is tried next, and then torch.distributed's all-gather of EQUAL bands, the balancing rounds skipped)
            kinds = [args.composer] + (["peer"] if args.composer == "native" and multi else [])
            whys = [why] if why else []
            for kind in ([] if failed else kinds):
                failed = 0
                try:
                    composer = (compose.PeerBandComposer if kind == "peer" else compose.NativeBandComposer)(hdr, band, W, 8, transport=args.transport, **(dict(rank=0, world=1) if emulated else {}),
                                                                                                             **(dict(frame_height=H) if balanced else {}),
                                                                                                             **(dict(timeout_ms=int(os.environ.get("BRMI_BENCH_PEER_TIMEOUT_MS", "30000"))) if args.shared_gpu and kind == "peer" else {}))
                    if balanced and not emulated:
                        composer.set_bounds(balancer.bounds)
                except Exception as e:      # noqa: BLE001
                    failed = 1; whys.append(f"{kind}: {type(e).__name__}: {e}")
                failed = agree(failed)
                if not failed:
                    composer_used = kind if kind == args.composer else f"{kind} ({args.composer} composer failed" + (f": {whys[-1]}" if whys else " on another rank") + ")"
                    break
                if composer is not None:
                    composer.close()
                    composer = None
            if failed:
                why = "; ".join(whys) or "it failed on another rank"
                print(f"bench.py rank {rank}: libbrmi_compose.so's composers failed ({why}); torch.distributed all-gather of equal bands instead", file=sys.stderr, flush=True)
                if balanced:
                    balance_frozen = True      # equal bands: the all-gather needs equal shares (the balancer's first bounds are the equal split)
                composer = compose.BandComposer(hdr, band, W, 8, transport=args.transport)
                composer_used = "torch (" + args.composer + " composer failed: " + why + ")"
        elif balanced:
            fail_line(args, "--partition balanced composes bands of unequal height: --composer native or peer")
        else:
            composer = compose.BandComposer(hdr, band, W, 8, transport=args.transport)

    slabbed = bool(composer is not None and args.compose_slabs > 1 and hasattr(composer, "submit_rows"))
    if slabbed:
        for q in passes:
            q.set_shade_slabs(args.compose_slabs, lambda r0, r1, stream, q=q: composer.submit_rows(r0, r1, q.hdr_tensor(), stream_ptr=stream))
    frame_no = [0]
    compose_on = [True]

    def step(serial=False):
        k = 0 if serial else frame_no[0] % len(passes)
        p = passes[k]
        with torch.cuda.stream(streams[k]):
            p.update()                  # the per-frame Update phase (camera / per-frame constants), as the reference's passes run it every frame
            if slabbed and not os.environ.get("BRMI_BENCH_SKIP_WAIT_SOURCE"):                 # the composer's stream may still be reading this pass's HDR rows of its previous frame: the stream that shades waits for those reads
                composer.wait_source(p.hdr_tensor(), stream_ptr=(streams[k] if serial or shade_streams[k] is None else shade_streams[k]).cuda_stream)
            p.execute(None if serial else shade_streams[k])
        if composer and not slabbed and compose_on[0]:
            with torch.cuda.stream(streams[k] if serial or shade_streams[k] is None else shade_streams[k]):
                composer.submit(p.hdr_tensor())
        frame_no[0] += 1

    def serial_frames(count):
        """`count` frames of pass 0 alone with nothing else on the GPU (per-stage HIP events mean a kernel's own duration only then)."""
        torch.cuda.synchronize()
        t_serial = time.perf_counter()
        for _ in range(count):
            step(serial=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t_serial) / count * 1e3

    for _ in range(args.warmup):
        step()
    balance_log = None
    if balanced and not args.bounds and not balance_frozen:
        # The partition finds its bounds before the clock starts: a round = --balance-frames frames of every rank's current band WITHOUT the composition (a rank's own
        # time, not its wait for the others), the times all-gathered, brmi_compose_balance_rows on every rank (same numbers, same bounds), brmi_set_band on the ring's
        # passes and brmi_compose_set_bounds.  Then the bounds stay (a renderer would repeat a round every second or so).
        if slabbed:
            fail_line(args, "--partition balanced with --compose-slabs: not supported (the slabs follow a fixed band)")
        if composer:
            composer.finish()
        compose_on[0] = False
        balance_log = []
        for rnd in range(max(0, args.balance_rounds)):
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.balance_frames):
                step()
            torch.cuda.synchronize()
            mine = torch.tensor([(time.perf_counter() - t0) / args.balance_frames * 1e3], dtype=torch.float64, device=cdev)
            if multi:
                every = [torch.zeros_like(mine) for _ in range(n)]
                dist.all_gather(every, mine)
                times = [float(x.item()) for x in every]
            else:
                times = [float(mine.item())] * n      # (one emulated rank cannot know the others' times: pass --bounds)
            balance_log.append({"bounds": list(balancer.bounds), "ms": [round(x, 4) for x in times]})
            balancer.update(times)
            if rnd + 1 == max(0, args.balance_rounds):
                balancer.settle()      # the timed region runs the best partition measured
            band = (balancer.bounds[rank], balancer.bounds[rank + 1])
            for q in passes:
                q.set_band(*band)
            if composer and not emulated:
                composer.set_bounds(balancer.bounds)
        compose_on[0] = True
        for _ in range(len(passes) + 2):      # the moved bands' first frames (no depth history in the rows a band gained)
            step()
    # Ten untimed frames after the warm-up are timed stage by stage; the timed region keeps HIP events only around the dominant stage (an
    # event pair is a barrier on the stream, ten pairs per frame cost ~5 %), whose mean launch duration feeds `roofline`.
    if composer:
        composer.finish()
    torch.cuda.synchronize()
    r.stage_times()                       # drop the warm-up window (first-frame effects)
    serial_frames(10)                     # untimed: per-stage profile of the steady state, all stages
    warm_ms = r.stage_times()
    dom_stage = dominant_stage(workload, features, warm_ms)
    longest_stage = max(warm_ms, key=lambda k: warm_ms[k]) if warm_ms else dom_stage      # this run's stage timers (whole stages: the raster stage is three kernels, two phases)
    r.set_timed_stages([dom_stage])
    # The timed region: EXACTLY --steps frames between a barrier + synchronize on both sides, MAX over ranks.  A short region (the driver's
    # 20 steps are 9 ms) is one noisy sample, so it is repeated (--repeats; 5 when --steps < 200) and the MEDIAN region is reported, the
    # spread beside it (`ms_per_step_minmax`); `steps` stays the per-region count.
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps < 200 else 1)
    regions, issue_s, own = [], [], []
    for _ in range(repeats):
        frame_no[0] = 0
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        issue_s.append(time.perf_counter() - t0)      # the host has issued every launch of the region (the GPU is still at work)
        if composer:
            composer.finish()               # the last frames' collectives are inside the timed region
        torch.cuda.synchronize()
        t_done = time.perf_counter()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        own.append(float(t_done - t0))
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append(float(t.item()))
    dt = sorted(regions)[len(regions) // 2]
    # every rank's own time to finish its frames (before the closing barrier), median region: how evenly the partition loads the ranks
    rank_ms = [sorted(own)[len(own) // 2] / args.steps * 1e3]
    if multi:
        mine = torch.tensor(rank_ms, dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(n)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) for x in every]

    if composer is not None and hasattr(composer, "wait_status"):
        # peer-write composition: a wait for a peer's flag that timed out latches a status and the frame is composed anyway -- such a run is not a measurement
        try:
            composer.wait_status()
        except Exception as e:      # noqa: BLE001
            fail_line(args, f"peer-write composer: {e}")
    stage_ms = r.stage_times()          # mean over the last timed steps (HIP events on the execute stream); dominant stage only
    in_flight_ms = None
    if fif >= 2:
        # the dominant kernel shared the CUs with the other pass's frame during the timed region: its own duration comes from serial frames
        in_flight_ms = stage_ms[dom_stage]
        serial_ms = serial_frames(min(args.steps, 200))
        stage_ms = r.stage_times()
    if warm_ms:
        stage_ms = {k: (stage_ms[k] if k == dom_stage else warm_ms[k]) for k in warm_ms}
    per_stage_bytes, total_bytes = r.algorithmic_bytes()
    launched_bytes, _ = r.algorithmic_bytes_launched()
    c = r.counters()
    path_out, path_fast_out = None, None
    if path and n == 1 and args.camera_path > 0 and args.occlusion:
        path_out = camera_path(args, scene, passes, streams, shade_streams, r, dev, PATH_STEP)
        if args.camera_path_fast > 0 and path != "path":
            path_fast_out = camera_path(args, scene, passes, streams, shade_streams, r, dev, args.camera_path_fast)
    out = None
    if rank == 0 or emulated or solo:
        shaded = W * H if balanced else W * (band[1] - band[0]) * n            # pixels dispatched per step, all ranks
        ms_per_step = dt / args.steps * 1e3
        value = shaded / 1e6 / (dt / args.steps)
        dom = dom_stage if stage_ms.get(dom_stage, 0.0) > 0 else max(stage_ms, key=lambda k: stage_ms[k])      # the stage the events stayed around (two stages within noise of each other must not swap here)
        dom_s = stage_ms[dom] * 1e-3
        achieved = per_stage_bytes[dom] / dom_s / 1e9 if dom_s > 0 else 0.0
        frame_gbs = total_bytes / (dt / args.steps) / 1e9
        # From the committed rocprofv3 passes of this same command (tools/profile.sh): HBM-side bytes of the dominant kernel per launch
        # (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) and its VALU instruction count (SQ_INSTS_VALU).  null when no profile of this
        # workload / kernel is committed.
        traffic, valu = None, None
        try:
            tag = PROFILE_TAG.get((workload, features), "none")
            if not os.path.exists(os.path.join(ROOT, "profiles", tag + "_traffic.json")):
                tag = PROFILE_FALLBACK.get(tag, tag)
            tj = json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))
            # the stage's kernels; template instantiations ("k_shade<0>", "k_raster<true>") are matched by base name.  A one-kernel stage reports
            # the instantiation that takes the most time per launch; the raster stage is three kernels launched once per occlusion phase, and
            # its traffic is their sum over the frame's launches (the committed averages are per launch, over both phases' launches)
            stage_kernels = {"raster": ["k_raster", "k_raster_emit", "k_raster_wide", "k_raster_bins", "k_raster_overflow"]}.get(dom, [DOMINANT_KERNEL.get(dom, dom).split("<")[0]])
            traffic_sum, wi, found = 0, 0, False
            for base in stage_kernels:
                # (instantiations the steady state launches: a variant the first frames of a run took -- the in-place G-buffer kernel while the pass had no depth chain to cull against --
                # has a handful of calls in the trace and is not this frame's kernel)
                cands = [k for k in tj if k.split("<")[0] == base and tj[k].get("calls_per_frame", 1.0) >= 0.5]
                if n == 1 and cands:
                    rec = tj[max(cands, key=lambda k: tj[k].get("avg_us", 0.0))]
                    # launches per frame of THIS kernel in the committed --kernel-trace --stats run (Calls / frames; tools/pmc_report.py): since phase 2
                    # rasterises directly while it is small, k_raster runs twice per frame but k_raster_bins / k_raster_overflow once
                    per_frame = rec.get("calls_per_frame") or (2 if (dom == "raster" and args.occlusion) else 1)
                    traffic_sum += rec["hbm_bytes_per_launch"] * per_frame
                    wi += rec.get("valu_wave_insts_per_launch", 0) * per_frame
                    found = True
            if found:
                traffic = int(traffic_sum)
                if wi:
                    # the microarchitecture guide's issue peak: one wave64 VALU instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz
                    valu = {"wave_insts": int(wi), "insts_per_px": round(wi * 64.0 / (W * (band[1] - band[0])), 1), "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS,
                            "frac": round(wi / dom_s / VALU_PEAK_WAVE_INSTS, 5), "peak_measured_wave_insts_per_s": VALU_PEAK_MEASURED,
                            "frac_of_measured_peak": round(wi / dom_s / VALU_PEAK_MEASURED, 5),
                            "source": "SQ_INSTS_VALU of the committed profile / this run's launch time; measured peak: tools/valu_issue_probe.hip (full-rate instructions)"}
        except (OSError, ValueError, KeyError):
            pass
        hbm_frac = achieved / HBM_PEAK_GBS
        # what the counters say the kernel MOVED (committed profile) over this run's launch time, and the bytes the launched variant is obliged to move (44 B / px of
        # k_shade<0> when no material has a coat or fuzz layer, against 8(d)'s 60): `frac` keeps 8(d)'s definition, these two are the honest utilisation beside it
        frac_traffic = round(traffic / dom_s / 1e9 / HBM_PEAK_GBS, 5) if (traffic and dom_s > 0) else None
        frac_launched = round(launched_bytes[dom] / dom_s / 1e9 / HBM_PEAK_GBS, 5) if dom_s > 0 else None
        nearer = "valu" if (valu and valu["frac"] > max(hbm_frac, frac_traffic or 0.0)) else "hbm"
        out = {
            "metric": "shaded Mpixels/s @4K (vis-buffer+resolve)" if (W, H) not in FRAME_SIZE.values() else f"shaded Mpixels/s @{W}x{H} (vis-buffer+resolve)", "value": round(value, 2), "unit": "Mpixels/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "repeats": repeats, "ms_per_step_minmax": [round(min(regions) / args.steps * 1e3, 4), round(max(regions) / args.steps * 1e3, 4)],
            "host_issue_ms_per_step": round(sorted(issue_s)[len(issue_s) // 2] / args.steps * 1e3, 4),      # the host thread's time to issue a frame's launches (it runs ahead of the GPU when this is below ms_per_step)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}-class procedural frame, {W}x{H}, 1 directional + {lights} point lights, "
                                   f"{scene.stats['instancedTriangles']} instanced tris, {scene.stats['instances']} instances"
                                   + f" ({scene.stats['uniqueTriangles']} in {scene.stats['meshes']} meshes)"
                                   + (", LOD DAGs from the library's cluster-LOD builder" if lod_builder == "own" else "")
                                   + (f", relief slope {scene_kw['relief_slope']}" if scene_kw.get("relief_slope") else "")
                                   + (f", material features {features} (8 = texture-sampled, 16 = alpha-tested materials)" if features else "")
                                   + ((f", interleaved partition: chunks of {stripe_rows} rows, one per group of {n} and rank, {band[1]} rows per rank in compact surfaces" if striped else (f", {n} cost-balanced contiguous bands (this rank: rows {band[0]} - {band[1]})" if balanced else f", {n} row bands of {band[1] - band[0]} rows")) + f" + composition of HDR on every rank ({args.transport}, pipelined one frame deep, {'libbrmi_compose.so: one RCCL all-gather per frame' if composer_used == 'native' else ('libbrmi_compose.so: peer writes over hipIpc-mapped images, no collective' if composer_used == 'peer' else ('torch.distributed' if composer_used == 'torch' else composer_used))})" if composer else ""),
                       "baseline_config": BASELINE_CONFIG[workload],
                       "fps": round(1e3 / ms_per_step, 1),
                       "pixels_per_gpu": W * (band[1] - band[0]), "visible_clusters_rank0": int(c.visibleClusters),
                       "occlusion_culling": bool(args.occlusion), "visible_clusters_phase2_rank0": int(c.visibleClustersPhase2),
                       "meshlets_tested_rank0": int(c.meshletsTested), "partition": (f"interleaved chunks of {stripe_rows} rows x{n}" if striped else ((f"cost-balanced contiguous bands x{n}" + (" (left at the equal split: composer fallback)" if balance_frozen else "")) if balanced else f"row bands x{n}")) if n > 1 else "single GPU",
                       "frames_in_flight": fif},
            "roofline": {"bound": nearer, "bound_note": "the ceiling the kernel sits nearer to: VALU issue (roofline.valu.frac, SQ_INSTS_VALU of the committed profile) or HBM (frac / frac_traffic); "
                                                        "achieved / peak / unit / frac are the HBM figures of SURVEY.md 8(d) either way",
                         "kernel_stage": dom, "longest_stage_this_run": longest_stage, "kernel_stage_is_longest_stage": bool(dom == longest_stage),
                         "kernel": {"raster": "k_raster (+ k_raster_emit, k_raster_wide) + k_raster_bins (+ k_raster_overflow), both occlusion phases"}.get(dom, DOMINANT_KERNEL.get(dom, dom)), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(hbm_frac, 5), "traffic": traffic, "frac_traffic": frac_traffic, "valu": valu,
                         "algorithmic_bytes_per_launch": int(per_stage_bytes[dom]), "algorithmic_bytes_launched_variant": int(launched_bytes[dom]), "frac_launched_variant": frac_launched,
                         "launch_ms": round(stage_ms[dom], 4),
                         "whole_frame": {"algorithmic_bytes": int(total_bytes), "achieved_GBps": round(frame_gbs, 2), "frac": round(frame_gbs / HBM_PEAK_GBS, 5)}},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items() if v > 0},
            "stage_ms_note": (f"'{dom}' from HIP events inside the timed region; the other stages from 10 untimed frames before it" if fif == 1 else
                              f"serial frames of one pass (nothing else on the GPU): '{dom}' over {min(args.steps, 200)} frames after the timed region, the other stages over 10 frames "
                              f"before it; inside the timed region, sharing the CUs with the other pass's frame, '{dom}' took {in_flight_ms:.4f} ms per launch"),
        }
        if balance_log is not None:
            out["balance_rounds"] = balance_log
            out["bounds"] = list(balancer.bounds)
        if n > 1:
            out["rank_ms_per_step"] = {"max": round(max(rank_ms), 4), "min": round(min(rank_ms), 4), "ranks": len(rank_ms),
                                       "note": "each rank's own time to finish the region's frames, composition included (emulated: this rank alone)"}
        if fif >= 2:
            out["roofline"]["launch_ms_in_flight"] = round(in_flight_ms, 4)
            out["serial_frame_ms"] = round(serial_ms, 4)      # one pass, one stream, frames back to back (what --frames-in-flight 1 times)
        if path_out is not None:
            out["path"] = path_out
        if path_fast_out is not None:
            out["path_fast"] = path_fast_out
        if cpu:
            out["cpu_baseline"] = cpu_baseline(scene, args.cpu_scale)
            out["cpu_baseline_1thread"] = cpu_baseline(scene, args.cpu_scale_1thread, threads=1, budget_s=6.0, max_frames=2)
            out["configs0"] = cpu_forward_baseline()
    if composer is not None:
        # the leg's composer goes before the next leg makes its own (the RCCL path: every rank destroys its communicator at the same point of the same sequence)
        composer.finish()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()      # every rank has finished its frames: nobody polls, writes or waits on a peer's image / flag block any more when the first rank frees its own
        if hasattr(composer, "close"):
            composer.close()
        if multi:
            dist.barrier()
    for p in passes:
        p.close()
    return out


KERNEL_STAGE = (("k_shade", "shade"), ("k_gbuffer", "gbuffer"), ("k_resolve_setup", "gbuffer"), ("k_raster", "raster"), ("k_cull", "cull"), ("k_scan", "cull"), ("k_scatter", "cull"),
                ("k_hzb", "hzb"), ("k_clear_vis", "clear"), ("k_lc_", "light_cluster"))


def dominant_stage(workload, features, stage_ms):
    """The stage of the DOMINANT KERNEL: the kernel with the longest average launch in the committed kernel trace of this workload (profiles/<tag>_kernel_stats.csv,
    rocprofv3 --kernel-trace --stats of this command with --frames-in-flight 1).  The stage timers bracket whole stages, and the raster stage -- three kernels, two
    occlusion phases -- can outlast the shading stage by a few microseconds without any of its kernels being the longest (round 5, Bistro-class: raster 0.170 ms =
    k_raster 2 x 0.036 + plan 0.013 + k_raster_bins 0.085, shade = k_shade 0.165).  Without a committed trace: the longest stage."""
    import csv
    longest = max(stage_ms, key=lambda k: stage_ms[k])
    try:
        tag = PROFILE_TAG.get((workload, features), "none")
        path = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", PROFILE_FALLBACK.get(tag, tag) + "_kernel_stats.csv")
        best, best_ns = None, 0.0
        rows = list(csv.DictReader(open(path)))
        frames = max([int(r["Calls"]) for r in rows if "k_frame_constants" in r["Name"]] or [1])
        for row in rows:
            name = row["Name"].replace("void ", "").replace("brmi::", "")
            stage = next((st for prefix, st in KERNEL_STAGE if name.startswith(prefix)), None)
            if int(row["Calls"]) * 2 < frames:
                continue      # a variant only the first frames of the trace launched (no depth chain yet: more clusters, the in-place G-buffer kernel) is not the steady state's kernel
            if stage is not None and float(row["AverageNs"]) > best_ns and stage_ms.get(stage, 0.0) > 0.0:
                best, best_ns = stage, float(row["AverageNs"])
        return best or longest
    except (OSError, KeyError, ValueError):
        return longest


def camera_path(args, scene, passes, streams, shade_streams, r, dev, path_step):
    """K frames along the preset's camera path, a new camera every frame (what the reference's CameraManager does between frames): the previous
    frame's depth chain no longer matches, phase 1 rejects what it should not, and phase 2 of the occlusion chain (replay, cull, rasterise,
    second chain build) has work -- the static-camera region times it finding nothing.  Same arrangement as the timed region (frames in flight);
    the cameras are resident in HBM before the clock starts, each frame copies its own into the pass's camera buffers on the geometry stream."""
    import numpy as np
    import torch
    K = args.camera_path
    cams = [scene.camera_at(path_step * (k + 1), path_step * k) for k in range(K)]
    cam_dev = [(torch.from_numpy(c).to(dev), torch.from_numpy(cc).to(dev)) for c, cc in cams]
    fif = len(passes)

    def frame(k, serial=False):
        i = 0 if serial else k % fif
        p = passes[i]
        with torch.cuda.stream(streams[i]):
            p.set_camera_device(cam_dev[k][0], cam_dev[k][1], cams[k][0])
            p.execute(None if serial else shade_streams[i])

    for k in range(min(K, 8)):          # leave the static camera: the first frames of a path see the largest jump
        frame(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        frame(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same path once more, one pass, with a counter read-back per frame (untimed): how much work phase 2 had, and the stage times
    r.set_timed_stages(None)
    r.stage_times()
    phase1, phase2, replayed = [], [], []
    if fif >= 2:
        passes[0].set_history_source(None)      # one pass alone tests against its own chain of the frame before
    for k in range(K):
        frame(k, serial=True)
        if k >= K - 32:
            c = r.counters()
            phase1.append(int(c.visibleClusters)); phase2.append(int(c.visibleClustersPhase2)); replayed.append(int(c.replayNodes) + int(c.replayMeshlets))
    stage_ms = r.stage_times()
    if fif >= 2:
        passes[0].set_history_source(passes[-1])
    # back to the static camera for whatever is measured next
    base = scene.camera_at(0.0)
    for i, p in enumerate(passes):
        with torch.cuda.stream(streams[i]):
            p.set_camera_device(torch.from_numpy(base[0]).to(dev), torch.from_numpy(base[1]).to(dev), base[0])
    torch.cuda.synchronize()
    px = scene.width * scene.height
    return {"frames": K, "ms_per_step": round(dt / K * 1e3, 4), "value": round(px / 1e6 / (dt / K), 2), "unit": "Mpixels/s",
            "camera": f"{path_step} path units per frame ({0.35 * path_step:.4f} m sideways, {0.6 * path_step:.4f} m ahead, {0.07 * path_step * 57.2958:.3f} degrees of yaw)",
            "visible_clusters_phase1_mean": round(float(np.mean(phase1)), 1), "visible_clusters_phase2_mean": round(float(np.mean(phase2)), 1),
            "replayed_records_mean": round(float(np.mean(replayed)), 1), "frames_in_flight": fif,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items() if v > 0},
            "stage_ms_note": "serial frames of one pass over the same path (last 32 frames); ms_per_step is the in-flight arrangement of the timed region"}


def fail_line(args, why):
    """An N-GPU request that cannot be honoured is an error line and a non-zero exit, never a 1-GPU number."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": "shaded Mpixels/s @4K (vis-buffer+resolve)", "value": None, "unit": "Mpixels/s", "n_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "error": why}), flush=True)
    else:
        print(f"bench.py rank {os.environ.get('RANK')}: {why}", file=sys.stderr, flush=True)      # (rank 0 owns stdout's one line)
    sys.exit(2)


def self_launch(args):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(scene, scale, threads=None, budget_s=10.0, max_frames=16):
    """The CPU oracle (port of the reference HLSL) on the same frame: all host cores, or `threads`; whole frames (or the stated share of the rows) for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    cores = threads or orc.effective_cores()
    f = orc.OracleFrame(scene, threads=cores)
    H = scene.height
    rows = max(8, int(H * scale) // 8 * 8)
    band = (0, rows) if rows < H else (0, 0)
    # whole frames of the same workload until ~10 s of wall time have been spent (at least 2, at most 16 frames)
    times = []
    while len(times) < min(2, max_frames) or (sum(times) < budget_s and len(times) < max_frames):
        if hasattr(f, "vis"):
            del f.vis
        t0 = time.perf_counter()
        f.cull()
        f.raster(band=band)
        f.depth_copy()
        f.gbuffer(band=band)
        f.light_cluster()
        f.shade(band=band)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    px = scene.width * (rows if rows < H else H)
    return {"value": round(px / 1e6 / dt, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} frames ({sum(times):.1f} s), median {dt:.2f} s per frame of rows [0,{rows if rows < H else H}) of {scene.width}x{H} ({px} px); "
                      f"cull, raster, G-buffer, light clustering, shade; " + ("one thread" if cores == 1 else "OpenMP over clusters / scanlines")}


def cpu_forward_baseline():
    """BASELINE.json configs[0] -- "Sponza static frame, 1080p, 1 directional light, forward PBR -- CPU scalar raster reference (no GPU)": the
    CPU oracle's forward entry (lighting from the unquantised material inputs, shaders.hlsl:221-229) on all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from basicrenderer_amd import Scene
    sc = Scene("sponza", 1920, 1080, point_lights=0, directional=True)
    cores = orc.effective_cores()
    f = orc.OracleFrame(sc, threads=cores)
    times = []
    while len(times) < 2 or (sum(times) < 4.0 and len(times) < 16):
        if hasattr(f, "vis"):
            del f.vis
        t0 = time.perf_counter()
        f.cull(); f.raster(); f.depth_copy(); f.gbuffer(forward=True); f.light_cluster(); f.shade(forward=True)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(1920 * 1080 / 1e6 / dt, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port", "baseline_config": "configs[0]",
            "sample": f"{len(times)} frames ({sum(times):.1f} s), median {dt:.3f} s per 1920x1080 frame, 1 directional light, forward PBR: cull, raster, material resolve, forward lighting"}


if __name__ == "__main__":
    main()
