"""Resolve stage when every material's highlight is too sharp for the mixed-precision form (roughness 0.1): every slot goes
through the second pass.  PYTHONPATH=. [RT_HIP_LIB=...] python exp/resolve_glossy.py"""
import numpy as np, torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
prims["roughness"] = 0.1
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
rgb = torch.empty(W*H*3, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, 0, 0.0)
ctx.synchronize(); ctx.set_timing(True); ctx.timing()
for _ in range(10):
    ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, 0, 0.0)
t = ctx.timing()
print("resolve %.3f ms per frame; checksum %d" % (t["resolve_ms"] / 10, int(rgb.to(torch.int64).sum().item())))
