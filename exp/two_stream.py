"""Experiment: a rank's 1/8 share rendered as two 1/16 shares on two contexts (two streams) at once."""
import time
import torch
from raytracer_amd import capi, scenes as S
prims = S.mixed_field(n_side=1000, seed=0x100000)
lights = S.mixed_field_lights()
W, H = 3840, 2160
cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
sc = capi.Scene.create(prims, lights)
A = capi.Context(0); A.upload(sc)
B = capi.Context(0); B.upload(sc)
for n in (1, 2, 4, 8):
    t1 = capi.Tiling(W, H, 64, n)
    t2 = capi.Tiling(W, H, 64, 2 * n)
    buf1 = torch.empty(capi.tiles_per_rank(t1) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
    bufa = torch.empty(capi.tiles_per_rank(t2) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
    bufb = torch.empty_like(bufa)
    def one():
        A.render_tiles_dev(cam, t1, 0, buf1.data_ptr(), 0, 0.0)
    def two():
        A.render_tiles_dev(cam, t2, 0, bufa.data_ptr(), 0, 0.0)
        B.render_tiles_dev(cam, t2, n, bufb.data_ptr(), 0, 0.0)
    for f, name in ((one, "one stream"), (two, "two streams")):
        for _ in range(3): f()
        A.synchronize(); B.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): f()
        A.synchronize(); B.synchronize()
        print("ranks %d %s: %.3f ms" % (n, name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
