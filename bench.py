#!/usr/bin/env python
"""Headline benchmark: Mrays/s (primary + shadow) at 3840x2160 on the 1M-cell mixed-primitive scene.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full frame through the hot path (primary rays, closest hit, one shadow ray per light per
hit, Blinn-Phong, gamma) with the scene already resident in HBM.  With N > 1 the SAME frame is split into
64x64 tiles dealt round-robin over the ranks (scene replicated), each rank renders its tiles, the tiles are
gathered to rank 0 over RCCL (rt_gather_tiles of the C ABI: one send per peer, grouped receives on rank 0) and
scattered into the frame there; that gather + assembly is inside the timed region, so per-N values are strong
scaling of one frame.

Ray count per frame = W*H primary + (#primary hits) * (#lights) shadow rays, exactly as the reference casts
them (s2899221/Code/raytracer.cpp:210-240).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
FP32_PEAK_TFLOPS = 157.3       # MI355X FP32 vector peak
L1_PATH_PEAK_GBS = 64 * 256 * 2.4          # bytes per clock and CU x CUs x GHz
NODE_BYTES, PRIM_BYTES = 32, 64
# algorithmic flops: a slab test (6 sub, 6 mul, 12 min/max, 6 compares), a primitive test by kind (cube / cylinder: two
# 3x3 matrix-vector products and slabs or a quadratic; ellipsoid; plane / triangle)
NODE_FLOPS = 30
PRIM_FLOPS = {0: 120, 1: 75, 2: 120, 3: 45, 4: 75}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(scene, prims, lights, camd, budget_s, threads):
    """The CPU restatement (oracle, "port") walking the same flattened BVH on a strided pixel sample of the same
    frame, all requested host threads (OpenMP over rows).  Returns the bench JSON object and the sampled counts."""
    from oracle import bind
    O = bind.Oracle()
    osc = O.scene(prims, lights)
    nodes, slots, n_slots, n_loose, radius = scene.bvh()
    bvh = O.bvh(nodes, slots, n_slots, n_loose, radius, scene.wide_bvh())
    cam = bind.camera_from_dict(O, camd)
    W, H = camd["res"]
    threads = threads or min(os.cpu_count() or 1, 16)
    # calibrate on a coarse grid, then pick the stride that fills the budget
    t0 = time.perf_counter()
    _, _, counts, hits = O.bvh_render(osc, bvh, cam, step=16, threads=threads)
    dt = time.perf_counter() - t0
    n_cal = ((W + 15) // 16) * ((H + 15) // 16)
    per_px = dt / max(n_cal, 1)
    step = max(1, int(np.ceil(np.sqrt(W * H * per_px / max(budget_s, 1e-3)))))
    t0 = time.perf_counter()
    _, _, counts, hits = O.bvh_render(osc, bvh, cam, step=step, threads=threads)
    dt = time.perf_counter() - t0
    n_px = ((W + step - 1) // step) * ((H + step - 1) // step)
    rays = n_px + hits * len(lights)
    # the same walk on one thread, and the reference's own algorithm (every primitive against every ray, strict '<') on
    # one thread: a few seconds each, on coarser samples of the same frame
    step1 = max(step, int(np.ceil(np.sqrt(W * H * per_px * threads / 3.0))))
    t0 = time.perf_counter()
    _, _, _, hits1 = O.bvh_render(osc, bvh, cam, step=step1, threads=1)
    dt1 = time.perf_counter() - t0
    rays1 = ((W + step1 - 1) // step1) * ((H + step1 - 1) // step1) + hits1 * len(lights)
    n_brute = max(4, int(3.0 / (150e-9 * max(len(prims), 1))))
    rng = np.random.default_rng(1)
    pix = np.stack([rng.integers(0, W, n_brute) + 0.5, rng.integers(0, H, n_brute) + 0.5], axis=1)
    brute_rays = O.pixel_rays(cam, pix)
    t0 = time.perf_counter()
    O.trace_rays(osc, brute_rays, threads=1)
    dtb = time.perf_counter() - t0
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "single_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "sample": "every %dth pixel, %d rays, %.1f s" % (step1, rays1, dt1)},
        "brute_force_single_thread": {"value": n_brute / dtb / 1e6, "unit": "Mrays/s",
                                      "sample": "%d primary rays against all %d primitives (the reference's one_pass_intersection), %.1f s"
                                                % (n_brute, len(prims), dtb)},
        "sample": "every %dth pixel in x and y of the same frame (%d primary + %d shadow rays, %.1f s), "
                  "FP64 C restatement through the same flattened BVH, OpenMP x%d, CPU: %s"
                  % (step, n_px, hits * len(lights), dt, threads, cpu_model()),
        "nodes_per_primary_ray": counts[0] / max(n_px, 1), "prims_per_primary_ray": counts[1] / max(n_px, 1),
        "nodes_per_shadow_ray": counts[2] / max(hits * len(lights), 1), "prims_per_shadow_ray": counts[3] / max(hits * len(lights), 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-side", type=int, default=1000, help="cells per side of the primitive grid (1000 -> 1M cells)")
    ap.add_argument("--res", default="3840x2160")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--max-leaf", type=int, default=0)
    ap.add_argument("--max-scatters", type=int, default=0)
    ap.add_argument("--device-build", action="store_true", help="build the hierarchy on the GPU (Morton order, top-down split)")
    ap.add_argument("--instanced", action="store_true",
                    help="the two-level workload instead: one 10 008-triangle mesh placed --n-side x --n-side times (100 -> 100 M triangles)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 disables it)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import torch
    from raytracer_amd import capi
    from raytracer_amd import scenes as S

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (RT_BENCH_SHARE_GPU=1): every rank uses GPU 0 and the tiles travel
    # over gloo through host memory.  It checks the flow, not the speed; the driver never sets it.
    rehearsal = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H = (int(v) for v in args.res.lower().split("x"))
    if args.instanced:
        if args.device_build:
            raise SystemExit("--instanced scenes are built on the host")
        prims, instances, lights, camd = S.instanced_field(n_side=args.n_side if args.n_side != 1000 else 100)
        camd = dict(camd, res=(W, H))
        t0 = time.perf_counter()
        scene = capi.Scene.create_instanced(prims, instances, lights, max_leaf=args.max_leaf)
    else:
        half = args.n_side / 2.0
        prims = S.mixed_field(n_side=args.n_side, seed=0x100000)
        lights = S.mixed_field_lights(half_extent=half, height=120.0 * half / 500.0, intensity=4000.0 * (half / 500.0) ** 2)
        camd = S.mixed_field_camera(half_extent=half, res=(W, H))
        t0 = time.perf_counter()
        scene = capi.Scene.create(prims, lights, max_leaf=args.max_leaf, on_device=args.device_build)
    build_s = time.perf_counter() - t0
    info = scene.info()
    cam = capi.make_camera(**camd)

    ctx = capi.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.upload(scene)
    if args.device_build:
        ctx.upload(scene)          # the first build also loads the builder's code objects: report a build in steady state

    tiling = capi.Tiling(W, H, args.tile, world)
    if world == 1:
        frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
        def step():
            ctx.render_rect_dev(cam, (0, 0, W, H), frame.data_ptr(), None, args.max_scatters, 0.0)
    else:
        frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda") if rank == 0 else None
        if rehearsal:
            from raytracer_amd import tiles
            tg = tiles.TileGather(torch, dist, W, H, args.tile, world, rank, "cuda")
            def step():
                ctx.render_tiles_dev(cam, tiling, rank, tg.local.data_ptr(), args.max_scatters, 0.0)
                gathered = tg.gather()
                if rank == 0:
                    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
        else:
            # the C ABI's own exchange: rank 0 names the group, the id travels over torch.distributed, every rank joins
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            comm = capi.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
            local = torch.zeros(capi.tiles_per_rank(tiling) * args.tile * args.tile * 3, dtype=torch.uint8, device="cuda")
            def step():
                ctx.render_tiles_dev(cam, tiling, rank, local.data_ptr(), args.max_scatters, 0.0)
                comm.gather_tiles(tiling, local.data_ptr(), frame.data_ptr() if rank == 0 else None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ray counts and algorithmic traffic of one frame: one untimed pass through the counting kernels
    ctx.set_counting(True)
    ctx.reset_counters()
    step()
    torch.cuda.synchronize()
    c = ctx.counters()
    ctx.set_counting(False)
    keys = sorted(c)
    tot = torch.tensor([c[k] for k in keys], dtype=torch.int64, device="cuda")
    if dist is not None:
        dist.all_reduce(tot)
    total = dict(zip(keys, (int(v) for v in tot.tolist())))
    rays_per_frame = total["primary_rays"] + total["shadow_rays"] + total["bounce_rays"]
    # algorithmic bytes of this rank's launches, per kernel: 32 B per node record fetched, 64 B per primitive record
    # fetched, 8 B written per primary ray, 8 B read + 1 B written per shadow ray
    bytes_primary = NODE_BYTES * c["nodes_primary"] + PRIM_BYTES * c["prims_primary"] + 8 * c["primary_rays"]
    bytes_shadow = NODE_BYTES * c["nodes_shadow"] + PRIM_BYTES * c["prims_shadow"] + 9 * c["shadow_rays"]

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_timing(True)          # HIP events around every launch, on the stream the kernels run on
    ctx.timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = ctx.timing()
    ctx.set_timing(False)
    # the shadow pass dominates (about 80 % of the frame): one launch per batch of up to 8 lights and per bounce level
    wide = args.device_build or scene.n_wide() > 0
    kname, k_ms, k_bytes = "k_rays<1,false,0,%s,%s> (shadow pass)" % ("true" if args.instanced else "false", "true" if wide else "false"), tm["shadow_ms"] / max(tm["shadow_launches"], 1), \
        bytes_shadow / max(tm["shadow_launches"] // args.steps, 1)
    by_kind = [int(v) for v in info.n_by_kind]
    prim_flops = sum(PRIM_FLOPS.get(k, 0) * n for k, n in enumerate(by_kind)) / max(sum(by_kind), 1)
    k_flops = (NODE_FLOPS * c["nodes_shadow"] + prim_flops * c["prims_shadow"]) / max(tm["shadow_launches"] // args.steps, 1)
    stage_ms = {k: tm[k] / args.steps for k in ("primary_ms", "shadow_ms", "resolve_ms", "bounce_ms")}
    t = torch.tensor([elapsed, k_ms], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, k_ms_max = (float(v) for v in t.tolist())

    if rank == 0:
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        # HBM-side bytes per launch of the same kernel on the same workload, from committed rocprofv3 PMC passes
        # (profiles/make_traffic.py); null when no measurement matches this run
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")
        if world == 1 and args.max_scatters == 0 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj["workload"]["n_side"] == args.n_side and tj["workload"]["rays_per_frame"] == rays_per_frame:
                traffic = tj["traffic_bytes_per_launch"]
        frame_bytes = bytes_primary + bytes_shadow
        device_ms = sum(stage_ms.values())
        out = {
            "metric": "Mrays/sec (primary+shadow) at %dx%d" % (W, H),
            "value": rays_per_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("%d instances of a %d-triangle mesh (%.1f M triangles, two-level hierarchy), %dx%d, %d mirror bounces, shadow rays to %d point lights"
                             % (info.n_instances, info.n_prims, info.n_instanced_prims / 1e6, W, H, args.max_scatters, len(lights))) if args.instanced else
                            "1M-cell mixed primitives (tri/sphere/box + ground plane), %dx%d, primary + shadow rays to %d point lights"
                            % (W, H, len(lights)),
                "n_side": args.n_side, "n_prims": int(info.n_instanced_prims if args.instanced else info.n_prims), "bvh_nodes": int(info.n_nodes), "bvh_depth": int(info.depth),
                "wide_nodes": (2 * int(info.n_slots) if ctx.build_time_ms() > 0 else scene.n_wide()),
                "max_scatters": args.max_scatters, "tile": args.tile if world > 1 else None,
                "partition": ("image tiles round-robin over %d ranks, scene replicated, rt_gather_tiles (RCCL send / grouped recv) to rank 0" % world
                              + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else "")) if world > 1 else "single GPU",
                "rays_per_frame": rays_per_frame, "primary_rays": total["primary_rays"], "primary_hits": total["primary_hits"],
                "shadow_rays": total["shadow_rays"], "bounce_rays": total["bounce_rays"], "scene_build_s": round(build_s, 3),
                "hierarchy": "Morton order cut top-down at the highest differing key bit, built on the GPU in %.2f ms" % ctx.build_time_ms() if args.device_build
                             else ("binned SAH on the host: one tree over the instances, one per mesh" if args.instanced else "binned SAH, built on the host"),
                "arithmetic": "FP32 padded-box traversal through 4-wide nodes, FP64 leaf tests and shading",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # what actually crossed the fabric, per second: the algorithmic figure above counts every node and record a
                # ray fetches, and most of those fetches are served by L1 / L2, so it can exceed the HBM peak
                "hbm_side_achieved": traffic / (k_ms * 1e-3) / 1e9 if traffic else None,
                "hbm_side_frac": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                # the secondary bound: algorithmic flops of the same launch against the FP32 vector peak
                "fp32": {"achieved": k_flops / (k_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": k_flops / (k_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "flops_per_launch": k_flops},
                # the path every load takes into a CU's L1 (texture addresser): 64 bytes per clock and CU (the CDNA3 figure,
                # assumed unchanged), 256 CUs at 2.4 GHz - the counters show this path about four fifths busy
                "l1_path": {"achieved": achieved, "peak": L1_PATH_PEAK_GBS, "unit": "GB/s", "frac": achieved / L1_PATH_PEAK_GBS},
                # what a frame must read and write at least once: every node and record, 3 bytes per pixel
                "compulsory_bytes_per_frame": NODE_BYTES * int(info.n_nodes) + PRIM_BYTES * int(info.n_slots + info.n_loose) + 3 * W * H,
                "kernel": kname, "kernel_ms": k_ms, "kernel_ms_max_over_ranks": k_ms_max,
                "algorithmic_bytes_per_launch": k_bytes,
                "stage_ms_per_frame": stage_ms,
                "frame_algorithmic_bytes": frame_bytes,
                "frame_achieved_GBps": frame_bytes / (device_ms * 1e-3) / 1e9 if device_ms > 0 else None,
                "nodes_per_primary_ray": c["nodes_primary"] / max(c["primary_rays"], 1),
                "prims_per_primary_ray": c["prims_primary"] / max(c["primary_rays"], 1),
                "nodes_per_shadow_ray": c["nodes_shadow"] / max(c["shadow_rays"], 1),
                "prims_per_shadow_ray": c["prims_shadow"] / max(c["shadow_rays"], 1),
            },
        }
        if world == 1 and args.cpu_seconds > 0 and not args.device_build and not args.instanced:
            out["cpu_baseline"] = cpu_baseline(scene, prims, lights, camd, args.cpu_seconds, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
