"""Stage times (HIP events) of one rank's share of the tiled 4K frame, for 1 and 8 ranks: which launch loses when the share is small?
Run from the repository root on the GPU box: PYTHONPATH=. python profiles/tile_share_stages.py"""
import sys

import torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").lower().split("x"))
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
ctx.set_timing(True)
for n in (1, 2, 4, 8):
    tiling = capi.Tiling(W, H, 64, n)
    buf = torch.empty(capi.tiles_per_rank(tiling) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
    ctx.synchronize(); ctx.timing()
    for _ in range(20):
        ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
    t = ctx.timing()
    print("ranks %d: entry searches %.3f closest hits + surface %.3f shadow %.3f resolve %.3f ms per frame" % (n, t["entry_ms"] / 20, t["primary_ms"] / 20, t["shadow_ms"] / 20, t["resolve_ms"] / 20), flush=True)
