import os, sys, time, torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
ctx.set_timing(True)
for n in (1, 8):
    tiling = capi.Tiling(W, H, 64, n)
    buf = torch.empty(capi.tiles_per_rank(tiling) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
    ctx.synchronize(); ctx.timing()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
    ctx.synchronize()
    el = (time.perf_counter() - t0) / 20 * 1e3
    t = ctx.timing()
    print("%s div %s ranks %d: %.3f ms  primary %.3f shadow %.3f resolve %.3f" % (os.environ.get("RT_HIP_LIB", "base"), os.environ.get("RT_WAVE_DIV", "8"), n, el, t["primary_ms"] / 20, t["shadow_ms"] / 20, t["resolve_ms"] / 20), flush=True)
