"""Time one rank's share of a tiled 4K frame for 1, 2, 4, 8 ranks on one GPU (no gather): how far from 1/N is it?
Run from the repository root on the GPU box: PYTHONPATH=. python profiles/tile_share.py"""
import time

import sys

import torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").lower().split("x"))
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
for tile in (64,):
    for n in (1, 2, 4, 8):
        tiling = capi.Tiling(W, H, tile, n)
        per = capi.tiles_per_rank(tiling) * tile * tile * 3
        buf = torch.empty(per, dtype=torch.uint8, device="cuda")
        worst = 0.0
        for r in range(n):
            for _ in range(3):
                ctx.render_tiles_dev(cam, tiling, r, buf.data_ptr(), 0, 0.0)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.render_tiles_dev(cam, tiling, r, buf.data_ptr(), 0, 0.0)
            ctx.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / 20 * 1e3)
        print("tile %d ranks %d: slowest rank %.3f ms  (x%d = %.2f ms)" % (tile, n, worst, n, worst * n), flush=True)
