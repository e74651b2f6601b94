import sys, torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
buf = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
for rows in (8, 64, 256, 2160):
    y0 = (H - rows) // 2 // 8 * 8
    f = lambda: ctx.render_rect_dev(cam, (0, y0, W, rows), buf.data_ptr(), None, 0, 0.0)
    for _ in range(2): f()
    ctx.synchronize(); ctx.reset_counters()
    f(); ctx.synchronize()
    d = ctx.counters()
    swap, inner, leaf, total = d["nodes_primary"], d["prims_primary"], d["nodes_shadow"], d["prims_shadow"]
    it_i, lanes_i, it_l, lanes_l = d["primary_hits"], d["bounce_rays"], d["primary_rays"], d["shadow_rays"]
    print("rows %d: wave lifetime total %.3g cycles; swap %.1f%% inner %.1f%% leaf %.1f%%; inner %d it, %.1f lanes, %.0f cyc; leaf %d rounds, %.1f lanes, %.0f cyc" % (
        rows, total, 100*swap/total, 100*inner/total, 100*leaf/total, it_i, lanes_i/max(it_i,1), inner/max(it_i,1), it_l, lanes_l/max(it_l,1), leaf/max(it_l,1)), flush=True)
