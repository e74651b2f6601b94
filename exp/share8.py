import sys, torch
from raytracer_amd import capi, scenes as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
tiling = capi.Tiling(W, H, 64, n)
buf = torch.empty(capi.tiles_per_rank(tiling) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
for _ in range(20):
    ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
ctx.synchronize()
