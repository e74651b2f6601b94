"""Stage times of a centred band of the 4K frame against its height: what does a launch cost whatever its size?"""
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
for rows in (8, 16, 32, 64, 128, 256, 512, 1024, 2160):
    y0 = (H - rows) // 2 // 8 * 8
    f = lambda: ctx.render_rect_dev(cam, (0, y0, W, rows), buf.data_ptr(), None, 0, 0.0)
    for _ in range(3): f()
    ctx.synchronize(); ctx.timing()
    for _ in range(20): f()
    t = ctx.timing()
    print("rows %4d: entry %.3f primary %.3f shadow %.3f resolve %.3f  sum %.3f ms" % (rows, t["entry_ms"]/20, t["primary_ms"]/20, t["shadow_ms"]/20, t["resolve_ms"]/20,
          (t["entry_ms"]+t["primary_ms"]+t["shadow_ms"]+t["resolve_ms"])/20), flush=True)
