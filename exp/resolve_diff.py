"""Frame of the headline scene (and of the shipped Test2 scene at its own size) saved to argv[1], or compared with the file argv[2]:
how many pixels two builds colour differently, and by how much.  PYTHONPATH=. RT_HIP_LIB=... python exp/resolve_diff.py out.npy [other.npy]"""
import sys
import numpy as np
from raytracer_amd import capi, scenes as S
frames = []
ctx = capi.Context(0)
for n_side, res in ((1000, (3840, 2160)), (300, (1280, 720))):
    half = n_side / 2.0
    prims = S.mixed_field(n_side=n_side, seed=0x100000)
    lights = S.mixed_field_lights(half_extent=half, height=120.0 * half / 500.0, intensity=4000.0 * (half / 500.0) ** 2)
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=half, res=res))
    ctx.upload(capi.Scene.create(prims, lights))
    rgb, _ = ctx.render(cam, max_scatters=0, want_ids=True)
    frames.append(rgb.reshape(-1, 3))
    rgb2, _ = ctx.render(cam, max_scatters=2, want_ids=True)
    frames.append(rgb2.reshape(-1, 3))
allf = np.concatenate(frames)
np.save(sys.argv[1], allf)
if len(sys.argv) > 2:
    other = np.load(sys.argv[2])
    d = np.abs(allf.astype(np.int16) - other.astype(np.int16))
    o = 0
    for f in frames:
        dd = d[o:o + len(f)]; o += len(f)
        print("pixels %d: differing %.5f%%, max difference %d" % (len(f), 100.0 * (dd > 0).any(-1).mean(), dd.max()))
