import os, torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
ctx.set_timing(True)
buf = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
def report(name, f):
    for _ in range(3): f()
    ctx.synchronize(); ctx.timing()
    for _ in range(20): f()
    t = ctx.timing()
    print("%s %-22s entry %.3f primary %.3f shadow %.3f resolve %.3f  sum %.3f ms" % (os.environ.get("RT_HIP_LIB", "base")[-10:], name, t["entry_ms"]/20, t["primary_ms"]/20, t["shadow_ms"]/20, t["resolve_ms"]/20,
          (t["entry_ms"]+t["primary_ms"]+t["shadow_ms"]+t["resolve_ms"])/20), flush=True)
tiling = capi.Tiling(W, H, 64, 8)
report("tiles 64, rank 0 of 8", lambda: ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0))
report("full frame", lambda: ctx.render_rect_dev(cam, (0, 0, W, H), buf.data_ptr(), None, 0, 0.0))
