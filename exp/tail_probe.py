"""How rectangular is a small launch?  Mean lifetime of a shadow wave (s_memtime ticks, RT_PHASE_TIMING build) against
the launch's duration (HIP events) for bands of the 4K frame of different heights."""
import torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
ctx = capi.Context(0)
ctx.upload(capi.Scene.create(prims, lights))
buf = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
n_waves_max = 4096
for rows in (2160, 1080, 512, 256, 128, 64):
    y0 = (H - rows) // 2 // 8 * 8
    f = lambda: ctx.render_rect_dev(cam, (0, y0, W, rows), buf.data_ptr(), None, 0, 0.0)
    for _ in range(2): f()
    ctx.synchronize(); ctx.reset_counters()
    ctx.set_timing(True); ctx.timing()
    f(); ctx.synchronize()
    tm = ctx.timing(); ctx.set_timing(False)
    d = ctx.counters()
    swap, inner, leaf, total = d["nodes_primary"], d["prims_primary"], d["nodes_shadow"], d["prims_shadow"]
    it_i, lanes_i = d["primary_hits"], d["bounce_rays"]
    n_rays = rows * W * 4
    waves = min(n_waves_max, max(1, n_rays // 512))
    print("rows %4d: shadow launch %.3f ms; %d waves; mean wave lifetime %.0f ticks = %.3f of the launch at 100 MHz; inner %.1f lanes, %.0f ticks/it; swap %.1f%% inner %.1f%% leaf %.1f%%" % (
        rows, tm["shadow_ms"], waves, total / waves, total / waves / (tm["shadow_ms"] * 1e5), lanes_i / max(it_i, 1), inner / max(it_i, 1),
        100 * swap / total, 100 * inner / total, 100 * leaf / total), flush=True)
