"""Where a shadow wave's cycles go on the two-level scene (100 M instanced triangles): library built with -DRT_PHASE_TIMING.
PYTHONPATH=. RT_HIP_LIB=exp/librt_hip_timing.so python exp/phase_inst.py [max_scatters]"""
import sys
import torch
from raytracer_amd import capi, scenes as S
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prims, instances, lights, camd = S.instanced_field(n_side=100)
W, H = 3840, 2160
cam = capi.make_camera(**dict(camd, res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create_instanced(prims, instances, lights))
rgb = torch.empty(W*H*3, dtype=torch.uint8, device="cuda")
for _ in range(2):
    ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, ms, 0.0)
ctx.synchronize()
ctx.reset_counters()
ctx.render_rect_dev(cam, (0, 0, W, H), rgb.data_ptr(), None, ms, 0.0)
ctx.synchronize()
d = ctx.counters()
swap, inner, leaf, total = d["nodes_primary"], d["prims_primary"], d["nodes_shadow"], d["prims_shadow"]
it_i, lanes_i, it_l, lanes_l = d["primary_hits"], d["bounce_rays"], d["primary_rays"], d["shadow_rays"]
print("cycles: swap %.1f%% inner %.1f%% leaf %.1f%% (sum %.1f%% of wave lifetime)" % (100*swap/total, 100*inner/total, 100*leaf/total, 100*(swap+inner+leaf)/total))
print("inner: %d iterations, %.1f lanes each, %.0f cycles each" % (it_i, lanes_i/it_i, inner/it_i))
print("leaf : %d rounds, %.1f lanes each, %.0f cycles each" % (it_l, lanes_l/it_l, leaf/it_l))
