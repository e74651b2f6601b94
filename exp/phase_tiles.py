"""Where a shadow wave's cycles go for one rank's share of the tiled 4K frame (library built with -DRT_PHASE_TIMING).
PYTHONPATH=. RT_HIP_LIB=exp/librt_hip_pt.so python exp/phase_tiles.py N_RANKS"""
import sys
import torch
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
for _ in range(3):
    ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
ctx.synchronize()
ctx.reset_counters()
ctx.render_tiles_dev(cam, tiling, 0, buf.data_ptr(), 0, 0.0)
ctx.synchronize()
d = ctx.counters()
swap, inner, leaf, total = d["nodes_primary"], d["prims_primary"], d["nodes_shadow"], d["prims_shadow"]
it_i, lanes_i, it_l, lanes_l = d["primary_hits"], d["bounce_rays"], d["primary_rays"], d["shadow_rays"]
print("ranks %d: cycles swap %.1f%% inner %.1f%% leaf %.1f%% (sum %.1f%% of wave lifetime %.3g)" % (n, 100*swap/total, 100*inner/total, 100*leaf/total, 100*(swap+inner+leaf)/total, total))
print("  inner: %d iterations, %.1f lanes each, %.0f cycles each" % (it_i, lanes_i/it_i, inner/it_i))
print("  leaf : %d rounds, %.1f lanes each, %.0f cycles each" % (it_l, lanes_l/it_l, leaf/it_l))
