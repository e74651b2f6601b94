"""Where a shadow wave's swap goes (library built with -DRT_SWAP_TIMING): results of finished rays, queue access, waiting for
the new rays' surface records, their set-up and first step.  PYTHONPATH=. RT_HIP_LIB=exp/librt_hip_swapt.so python exp/swap_split.py"""
import torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
rgb = torch.empty(W*H*3, dtype=torch.uint8, device="cuda")
for _ in range(2):
    ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, 0, 0.0)
ctx.synchronize()
ctx.reset_counters()
ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, 0, 0.0)
ctx.synchronize()
d = ctx.counters()
res, fetch, load, comp = d["nodes_primary"], d["prims_primary"], d["nodes_shadow"], d["prims_shadow"]
n, total, rays = d["primary_hits"], d["bounce_rays"], d["primary_rays"]
print("swaps %d (%.1f rays each); of the wave lifetime: results %.1f%% queue %.1f%% waiting for records %.1f%% set-up + first step %.1f%%"
      % (n, rays / n, 100 * res / total, 100 * fetch / total, 100 * load / total, 100 * comp / total))
print("cycles per swap: results %.0f queue %.0f records %.0f set-up %.0f" % (res / n, fetch / n, load / n, comp / n))
