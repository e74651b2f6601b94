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
NODE_BYTES, PRIM_BYTES = 32, 64
WIDE_NODE_BYTES = 4 * NODE_BYTES           # a step of the walk fetches one wide node: four boxes
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


def kernel_source_sha16():
    """Names the kernels a measurement belongs to: a hash over the device sources and the headers they include."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("raytracer_amd/csrc/rt_kernels.hip", "raytracer_amd/csrc/rt_device.h", "raytracer_amd/csrc/rt_internal.h", "include/rt_hip.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def mean_sd(xs):
    xs = [float(x) for x in xs]
    m = sum(xs) / len(xs)
    return m, (sum((x - m) ** 2 for x in xs) / (len(xs) - 1)) ** 0.5 if len(xs) > 1 else 0.0


# The real reference, unmodified, g++ -O2, one thread (it has none), ray loop only, primary rays, `time_driver <scene>
# <camera> 3` in the development container (8-vCPU Xeon): the reference does not travel to the GPU box, so these are
# quoted, not re-measured.  Mrays/s, seconds +- sigma over 3 trials.
CPU_REFERENCE_SURVEY = {
    "what": "unmodified reference (s2899221/Code, tests/raytracer/time_driver.cpp), g++ 11.4 -O2, 1 thread, primary rays only, "
            "measured in the development container (8-vCPU Intel Xeon), 3 trials each",
    "Test2_141_prims_129600_rays": {"brute_force_Mrays_s": 0.050, "brute_force_s": [2.581, 0.021], "bvh_Mrays_s": 0.386, "bvh_s": [0.336, 0.006]},
    "200_rand_200_prims_230400_rays": {"brute_force_Mrays_s": 0.129, "brute_force_s": [1.785, 0.146], "bvh_Mrays_s": 0.355, "bvh_s": [0.649, 0.018]},
    "Test1_21_prims": {"brute_force_Mrays_s": 1.08, "bvh_Mrays_s": 0.81},
    "toy_4_prims": {"brute_force_Mrays_s": 1.96, "bvh_Mrays_s": 0.76},
}


def cpu_baseline(scene, prims, lights, camd, budget_s, threads, trials=3):
    """The CPU restatement (oracle, "port") walking the same flattened BVH on a strided pixel sample of the same
    frame, all requested host threads (OpenMP over rows); every leg `trials` times, mean and sigma as the reference's
    timing driver reports them (tests/raytracer/time_driver.cpp:11-22).  Returns the bench JSON object."""
    from oracle import bind
    O = bind.Oracle()
    osc = O.scene(prims, lights)
    nodes, slots, n_slots, n_loose, radius = scene.bvh()
    bvh = O.bvh(nodes, slots, n_slots, n_loose, radius, scene.wide_bvh(), scene.loose_boxes())
    cam = bind.camera_from_dict(O, camd)
    W, H = camd["res"]
    threads = threads or min(os.cpu_count() or 1, 16)
    # calibrate on a coarse grid, then pick the stride that fills a third of the budget per trial
    t0 = time.perf_counter()
    _, _, counts, hits = O.bvh_render(osc, bvh, cam, step=16, threads=threads)
    dt = time.perf_counter() - t0
    n_cal = ((W + 15) // 16) * ((H + 15) // 16)
    per_px = dt / max(n_cal, 1)
    step = max(1, int(np.round(np.sqrt(W * H * per_px * trials / max(0.6 * budget_s, 1e-3)))))
    n_px = ((W + step - 1) // step) * ((H + step - 1) // step)
    times = []
    for _ in range(trials):
        t0 = time.perf_counter()
        _, _, counts, hits = O.bvh_render(osc, bvh, cam, step=step, threads=threads)
        times.append(time.perf_counter() - t0)
    rays = n_px + hits * len(lights)
    dt, dt_sd = mean_sd(times)
    # the same walk on one thread, and the reference's own algorithm (every primitive against every ray, strict '<') on
    # one thread: about a second per trial each, on coarser samples of the same frame
    step1 = max(step, int(np.ceil(np.sqrt(W * H * per_px * threads * trials / max(0.2 * budget_s, 1e-3)))))
    times1 = []
    for _ in range(trials):
        t0 = time.perf_counter()
        _, _, _, hits1 = O.bvh_render(osc, bvh, cam, step=step1, threads=1)
        times1.append(time.perf_counter() - t0)
    dt1, dt1_sd = mean_sd(times1)
    rays1 = ((W + step1 - 1) // step1) * ((H + step1 - 1) // step1) + hits1 * len(lights)
    n_brute = max(4, int(0.2 * budget_s / trials / (150e-9 * max(len(prims), 1))))
    rng = np.random.default_rng(1)
    pix = np.stack([rng.integers(0, W, n_brute) + 0.5, rng.integers(0, H, n_brute) + 0.5], axis=1)
    brute_rays = O.pixel_rays(cam, pix)
    timesb = []
    for _ in range(trials):
        t0 = time.perf_counter()
        O.trace_rays(osc, brute_rays, threads=1)
        timesb.append(time.perf_counter() - t0)
    dtb, dtb_sd = mean_sd(timesb)
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "trials": trials,
        "seconds_mean": dt, "seconds_sd": dt_sd,
        "single_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "trials": trials, "seconds_mean": dt1, "seconds_sd": dt1_sd,
                          "sample": "every %dth pixel, %d rays" % (step1, rays1)},
        "brute_force_single_thread": {"value": n_brute / dtb / 1e6, "unit": "Mrays/s", "trials": trials, "seconds_mean": dtb, "seconds_sd": dtb_sd,
                                      "sample": "%d primary rays against all %d primitives (the reference's one_pass_intersection)"
                                                % (n_brute, len(prims))},
        "sample": "every %dth pixel in x and y of the same frame (%d primary + %d shadow rays, %.1f s per trial), "
                  "FP64 C restatement through the same flattened BVH, OpenMP x%d, CPU: %s"
                  % (step, n_px, hits * len(lights), dt, threads, cpu_model()),
        "nodes_per_primary_ray": counts[0] / max(n_px, 1), "prims_per_primary_ray": counts[1] / max(n_px, 1),
        "nodes_per_shadow_ray": counts[2] / max(hits * len(lights), 1), "prims_per_shadow_ray": counts[3] / max(hits * len(lights), 1),
        "cpu_reference_survey": CPU_REFERENCE_SURVEY,
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
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget (0 disables it)")
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
    # N > 1: an event between a rank's render and the exchange, so that the line shows compute and gather apart
    splits = []
    def split_mark():
        if splits is not None and len(splits) < 4096:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            splits.append(e)
    gather_path = "rt_gather_tiles (RCCL send / grouped recv)"
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
                split_mark()
                gathered = tg.gather()
                if rank == 0:
                    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
        else:
            # the C ABI's own exchange: rank 0 names the group, the id travels over torch.distributed, every rank joins
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            comm, why = None, ""
            try:
                comm = capi.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
            except Exception as e:                       # e.g. RCCL not loadable under the name the library looks for
                why = str(e)
            # every rank must take the same road: if the group could not be formed anywhere, all fall back to the same
            # exchange through torch.distributed (RCCL underneath as well), and the line says so
            agreed = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0:
                if comm is not None:
                    comm.close()
                gather_path = "torch.distributed.gather (rt_comm_create failed: %s)" % (why or "on another rank")
                from raytracer_amd import tiles
                tg = tiles.TileGather(torch, dist, W, H, args.tile, world, rank, "cuda")
                def step():
                    ctx.render_tiles_dev(cam, tiling, rank, tg.local.data_ptr(), args.max_scatters, 0.0)
                    split_mark()
                    gathered = tg.gather()
                    if rank == 0:
                        ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
            else:
                local = torch.zeros(capi.tiles_per_rank(tiling) * args.tile * args.tile * 3, dtype=torch.uint8, device="cuda")
                def step():
                    ctx.render_tiles_dev(cam, tiling, rank, local.data_ptr(), args.max_scatters, 0.0)
                    split_mark()
                    comm.gather_tiles(tiling, local.data_ptr(), frame.data_ptr() if rank == 0 else None)
                # one untimed frame through either exchange: what rt_gather_tiles assembles on rank 0 must be what the
                # gather through torch.distributed assembles; if not (or if it raises), every rank takes the latter
                from raytracer_amd import tiles
                tg = tiles.TileGather(torch, dist, W, H, args.tile, world, rank, "cuda")
                def step_td():
                    ctx.render_tiles_dev(cam, tiling, rank, tg.local.data_ptr(), args.max_scatters, 0.0)
                    split_mark()
                    gathered = tg.gather()
                    if rank == 0:
                        ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
                same, why = 1, ""
                try:
                    step()
                    torch.cuda.synchronize()
                    first = frame.clone() if rank == 0 else None
                    step_td()
                    torch.cuda.synchronize()
                    if rank == 0 and not torch.equal(first, frame):
                        same, why = 0, "its frame differs from the one gathered through torch.distributed"
                except Exception as e:
                    same, why = 0, str(e)
                agreed = torch.tensor([same], dtype=torch.int32, device="cuda")
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
                del splits[:]
                if int(agreed.item()) == 0:
                    gather_path = "torch.distributed.gather (rt_gather_tiles set aside: %s)" % (why or "on another rank")
                    step = step_td
                else:
                    gather_path += ", frame checked against a gather through torch.distributed"
                    del tg

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

    # what the vector-memory path of this GPU can deliver with the walk's own load sequence and nothing behind it (a few
    # milliseconds, untimed): every lane its own node of an L1-resident table / one node per wave / plain streaming
    probe = None
    if rank == 0:
        probe = {"node_per_lane_l1": max(ctx.probe_fetch_rate(0, 16 << 10, 4096) for _ in range(3)),
                 "node_per_lane_l2": max(ctx.probe_fetch_rate(0, 16 << 20, 2048) for _ in range(3)),
                 "node_per_wave_l1": max(ctx.probe_fetch_rate(1, 16 << 10, 4096) for _ in range(3)),
                 "stream_hbm_16B_loads": max(ctx.probe_fetch_rate(2, 1 << 30, 2048) for _ in range(3))}

    for _ in range(args.warmup):
        step()
    fence()
    del splits[:]
    ctx.set_timing(True)          # HIP events around every launch, on the stream the kernels run on
    ctx.timing()
    # one more pair of events per step, on the same stream, for the spread over the steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record(stream)
    for k in range(args.steps):
        step()
        marks[k + 1].record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    # per step on this rank: render = step start -> split, exchange (gather + assembly, and on rank 0 the wait for the
    # slowest peer) = split -> step end
    render_ms = [marks[k].elapsed_time(splits[k]) for k in range(args.steps)] if len(splits) == args.steps else None
    gather_ms = [splits[k].elapsed_time(marks[k + 1]) for k in range(args.steps)] if len(splits) == args.steps else None
    tm = ctx.timing()
    ctx.set_timing(False)
    # the shadow pass dominates (about 70 % of the frame): one launch per batch of up to 8 lights and per bounce level
    wide = args.device_build or scene.n_wide() > 0
    kname, k_ms, k_bytes = "k_rays<1,false,0,%s,%s> (shadow pass)" % ("true" if args.instanced else "false", "true" if wide else "false"), tm["shadow_ms"] / max(tm["shadow_launches"], 1), \
        bytes_shadow / max(tm["shadow_launches"] // args.steps, 1)
    by_kind = [int(v) for v in info.n_by_kind]
    prim_flops = sum(PRIM_FLOPS.get(k, 0) * n for k, n in enumerate(by_kind)) / max(sum(by_kind), 1)
    k_flops = (NODE_FLOPS * c["nodes_shadow"] + prim_flops * c["prims_shadow"]) / max(tm["shadow_launches"] // args.steps, 1)
    stage_ms = {k: tm[k] / args.steps for k in ("entry_ms", "primary_ms", "shadow_ms", "resolve_ms", "bounce_ms")}
    t = torch.tensor([elapsed, k_ms], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, k_ms_max = (float(v) for v in t.tolist())
    # what every rank did: rays of its share, mean render and exchange time per step
    mine = torch.tensor([c["primary_rays"] + c["shadow_rays"] + c["bounce_rays"],
                         sum(render_ms) / len(render_ms) if render_ms else elapsed / args.steps * 1e3,
                         sum(gather_ms) / len(gather_ms) if gather_ms else 0.0], dtype=torch.float64, device="cuda")
    per_rank = [mine.clone() for _ in range(world)]
    if dist is not None:
        dist.all_gather(per_rank, mine)
    per_rank = [[float(v) for v in r.tolist()] for r in per_rank]

    if rank == 0:
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        # HBM-side bytes per launch of the same kernel on the same workload, from committed rocprofv3 PMC passes
        # (profiles/make_traffic.py; counters cannot be read from inside this process); null when no measurement
        # matches this run's workload
        traffic, traffic_src = None, None
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        for name in sorted((f for f in os.listdir(pdir) if f.endswith("_traffic.json")), reverse=True):
            tj = json.load(open(os.path.join(pdir, name)))
            # a measurement belongs to one set of kernels on one workload: the file must carry this build's source hash
            if (world == 1 and tj["workload"]["max_scatters"] == args.max_scatters and tj["workload"]["n_side"] == args.n_side
                    and bool(tj["workload"].get("instanced")) == args.instanced and bool(tj["workload"].get("device_build")) == args.device_build
                    and tj["workload"]["rays_per_frame"] == rays_per_frame and tj.get("res", "3840x2160") == "%dx%d" % (W, H)
                    and tj.get("algorithmic_bytes_per_launch") == k_bytes and tj.get("kernel_source_sha16") == kernel_source_sha16()):
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/" + name
                break
        # the ceiling of the walk's node fetches: wide nodes per second with every lane on its own L1-resident node
        node_peak = probe["node_per_lane_l1"] * WIDE_NODE_BYTES / 1e9
        # what the walks read from: binary and wide nodes, primitive records, materials
        scene_bytes = 32 * int(info.n_nodes) + 128 * int(scene.n_wide()) + 160 * int(info.n_slots + info.n_loose) + 32 * int(info.n_prims)
        walked_bytes = (128 * int(scene.n_wide()) if scene.n_wide() else 32 * int(info.n_nodes)) + 160 * int(info.n_slots + info.n_loose)
        ms_mean, ms_sd = mean_sd(step_ms)
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
                            "%s-cell mixed primitives (tri/sphere/box + ground plane), %dx%d, primary + shadow rays to %d point lights"
                            % ("%dM" % round(args.n_side ** 2 / 1e6) if args.n_side >= 1000 else "%dk" % round(args.n_side ** 2 / 1e3), W, H, len(lights)),
                "n_side": args.n_side, "n_prims": int(info.n_instanced_prims if args.instanced else info.n_prims), "bvh_nodes": int(info.n_nodes), "bvh_depth": int(info.depth),
                "wide_nodes": (2 * int(info.n_slots) if ctx.build_time_ms() > 0 else scene.n_wide()),
                "max_scatters": args.max_scatters, "instanced": args.instanced, "device_build": args.device_build, "tile": args.tile if world > 1 else None,
                "partition": ("image tiles round-robin over %d ranks, scene replicated, %s to rank 0" % (world, gather_path)
                              + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else "")) if world > 1 else "single GPU",
                "rays_per_frame": rays_per_frame, "primary_rays": total["primary_rays"], "primary_hits": total["primary_hits"],
                "shadow_rays": total["shadow_rays"], "bounce_rays": total["bounce_rays"], "scene_build_s": round(build_s, 3),
                "hierarchy": "Morton order cut top-down at the highest differing key bit, built on the GPU in %.2f ms" % ctx.build_time_ms() if args.device_build
                             else ("binned SAH on the host: one tree over the instances, one per mesh" if args.instanced else "binned SAH, built on the host"),
                "arithmetic": "FP32 padded-box traversal through 4-wide nodes, FP64 leaf tests and shading",
                "kernel_source_sha16": kernel_source_sha16(),
                "rays_per_frame_by_rank": [int(r[0]) for r in per_rank],
            },
            "ms_per_step_device": {"mean": ms_mean, "sd": ms_sd, "min": min(step_ms), "max": max(step_ms), "n": len(step_ms),
                                   "what": "HIP events around each step on the kernels' stream (rank 0)"},
            "roofline": {
                # Bound: HBM by the survey's accounting.  achieved = ALGORITHMIC bytes of the dominant launch (32 B per box
                # tested, 64 B per primitive record, 9 B per shadow ray) / its duration from HIP events on the kernels' stream;
                # peak = 8 TB/s.  Neighbouring rays share their nodes, so most of those bytes are served by L1 / L2 (and, for a scene
                # the size of n_side 1000's 0.3 GB, the Infinity Cache) and the quotient can exceed 1: `served_by_caches` says how
                # much, and `hbm_side` gives what actually crossed the fabric (PMC counters).  --n-side 4000 (4.7 GB of nodes and
                # records) is the case far beyond every cache.
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                # what the walks read from (wide nodes and primitive records) against the 256 MiB Infinity Cache, and the share of
                # the algorithmic bytes that never crossed the fabric (served by L1 / L2: neighbouring rays share their nodes)
                "cache_resident": bool(walked_bytes < (256 << 20)), "walked_bytes": walked_bytes, "scene_bytes": scene_bytes,
                "served_by_caches": (1.0 - traffic / k_bytes) if traffic and k_bytes else None,
                # the same launch against HBM itself: what crossed the fabric (PMC counters) per second
                "hbm_side": {"achieved": traffic / (k_ms * 1e-3) / 1e9 if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None},
                # the walk's own load sequence with nothing behind it, measured in this run (rt_probe_fetch_rate: every lane
                # its own 128-byte node, table resident in L1): what the vector-memory path of a CU can deliver for this pattern
                "l1_probe": {"achieved": achieved, "peak": node_peak, "unit": "GB/s", "frac": achieved / node_peak,
                             "peak_source": "measured in this run: rt_probe_fetch_rate(pattern 0, 16 KiB table) x 128 B per wide node"},
                # the secondary bound: algorithmic flops of the same launch against the FP32 vector peak
                "fp32": {"achieved": k_flops / (k_ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": k_flops / (k_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "flops_per_launch": k_flops},
                "probe_GBps": {"node_per_lane_l1": probe["node_per_lane_l1"] * WIDE_NODE_BYTES / 1e9,
                               "node_per_lane_l2": probe["node_per_lane_l2"] * WIDE_NODE_BYTES / 1e9,
                               "node_per_wave_l1": probe["node_per_wave_l1"] * WIDE_NODE_BYTES / 1e9,
                               "stream_hbm": probe["stream_hbm_16B_loads"] * 16 / 1e9},
                # what a frame must read and write at least once: every node and record, 3 bytes per pixel
                "compulsory_bytes_per_frame": NODE_BYTES * int(info.n_nodes) + PRIM_BYTES * int(info.n_slots + info.n_loose) + 3 * W * H,
                "kernel": kname, "kernel_ms": k_ms, "kernel_ms_max_over_ranks": k_ms_max,
                "algorithmic_bytes_per_launch": k_bytes,
                "stage_ms_per_frame": stage_ms,
                # N > 1, per rank: its render (all stages of its share) and the exchange that follows (gather + assembly;
                # on rank 0 this includes waiting for the slowest peer), HIP events on each rank's stream, mean per step
                "render_ms_by_rank": [r[1] for r in per_rank] if world > 1 else None,
                "gather_ms_by_rank": [r[2] for r in per_rank] if world > 1 else None,
                "gather_ms": per_rank[0][2] if world > 1 else None,
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
