"""Localise a pixel whose shadow walks fetch a different number of boxes on the device and in the oracle."""
import numpy as np, torch
from raytracer_amd import capi, scenes as S
from oracle import bind
O = bind.Oracle()
prims = S.mixed_field(n_side=30, seed=77)
lights = S.mixed_field_lights(half_extent=15.0, height=8.0, intensity=60.0)
cam = capi.make_camera(**S.mixed_field_camera(half_extent=15.0, res=(200, 120)))
ocam = bind.Camera.from_buffer_copy(bytes(cam))
sc = capi.Scene.create(prims, lights, max_leaf=1)
ctx = capi.Context(0); ctx.upload(sc)
ctx.set_counting(True); ctx.set_entry_search(False)
nodes, slots, n_slots, n_loose, radius = sc.bvh()
ob = O.bvh(nodes, slots, n_slots, n_loose, radius, sc.wide_bvh())
osc = O.scene(prims, lights)
buf = torch.empty(200 * 120 * 3, dtype=torch.uint8, device="cuda")
def both(rect):
    ctx.reset_counters()
    ctx.render_rect_dev(cam, rect, buf.data_ptr(), None, 0, 0.0); ctx.synchronize()
    c = ctx.counters()
    _, _, counts, hits = O.bvh_render(osc, ob, ocam, rect=rect)
    return (c["nodes_shadow"], c["prims_shadow"]), (int(counts[2]), int(counts[3]))
rect = (0, 0, 200, 120)
print(rect, both(rect))
while rect[2] * rect[3] > 1:
    x0, y0, w, h = rect
    halves = [(x0, y0, w, h // 2), (x0, y0 + h // 2, w, h - h // 2)] if h > 1 else [(x0, y0, w // 2, 1), (x0 + w // 2, y0, w - w // 2, 1)]
    for r in halves:
        d, o = both(r)
        if d != o:
            rect = r; print(r, d, o, flush=True); break
    else:
        print("no half differs"); break
x0, y0, w, h = rect
ids, rgb, counts, hits = O.bvh_render(osc, ob, ocam, rect=rect)
print("pixel", rect, "oracle id", ids, "kind", prims[int(ids[0, 0])] if ids[0, 0] >= 0 else None)
