"""Config 2 of the baseline list: 100k-triangle cube field (8 334 cubes x 12), 1920x1080, primary rays only, one GPU.
Run from the repository root on the GPU box: PYTHONPATH=. python profiles/config2.py"""
import json
import time

import torch

from raytracer_amd import capi, scenes as S

cubes = S.cube_field()
tris = S.cubes_to_triangles(cubes)
cam = capi.make_camera(**S.CUBE_FIELD_CAMERA)
W, H = S.CUBE_FIELD_CAMERA["res"]
ctx = capi.Context(0)
out = {}
for name, prims in (("triangles", tris), ("cubes", cubes)):
    t0 = time.perf_counter()
    sc = capi.Scene.create(prims)
    build_s = time.perf_counter() - t0
    ctx.upload(sc)
    ids = torch.empty(H * W, dtype=torch.int32, device="cuda")
    dist = torch.empty(H * W, dtype=torch.float64, device="cuda")
    for _ in range(5):
        ctx.trace_primary_dev(cam, (0, 0, W, H), ids.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        ctx.trace_primary_dev(cam, (0, 0, W, H), ids.data_ptr(), dist.data_ptr())
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    out[name] = {"n_prims": int(len(prims)), "ms_per_frame": round(ms, 4), "Mrays_per_s": round(W * H / ms / 1e3, 1),
                 "hit_fraction": round(float((ids >= 0).float().mean()), 4), "host_build_s": round(build_s, 3)}
print(json.dumps({"workload": "cube field, %dx%d, primary rays only" % (W, H), **out}))
