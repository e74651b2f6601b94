"""A band of the 4K frame of config 3, rendered a few times: rows from argv (profiling target)."""
import sys
import torch
from raytracer_amd import capi, scenes as S
rows = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
buf = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
y0 = (H - rows) // 2 // 8 * 8
for _ in range(reps):
    ctx.render_rect_dev(cam, (0, y0, W, rows), buf.data_ptr(), None, 0, 0.0)
ctx.synchronize()
