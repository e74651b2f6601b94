"""Is a rank's 1/8 share slow because it is small, or because its tiles are scattered?  Stage times of (a) a contiguous band of
the 4K frame with 1/8 of the pixels (the middle rows), (b) the 1/8 share as 64x64 tiles dealt round-robin."""
import torch
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
    print("%-28s entry %.3f primary %.3f shadow %.3f resolve %.3f  sum %.3f ms" % (name, t["entry_ms"]/20, t["primary_ms"]/20, t["shadow_ms"]/20, t["resolve_ms"]/20,
          (t["entry_ms"]+t["primary_ms"]+t["shadow_ms"]+t["resolve_ms"])/20), flush=True)
for y0 in (0, 540, 945, 1350, 1890):
    report("band rows %d..%d" % (y0, y0 + 270), lambda: ctx.render_rect_dev(cam, (0, y0, W, 270), buf.data_ptr(), None, 0, 0.0))
report("square 1358x764 centre", lambda: ctx.render_rect_dev(cam, (1241, 698, 1358, 764), buf.data_ptr(), None, 0, 0.0))
for tile in (64, 256):
    tiling = capi.Tiling(W, H, tile, 8)
    report("tiles %d, rank 0 of 8" % tile, lambda: ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0))
report("full frame", lambda: ctx.render_rect_dev(cam, (0, 0, W, H), buf.data_ptr(), None, 0, 0.0))
