"""GPU parity where the scene no longer fits any cache: 4000 x 4000 cells of the mixed field (21.3 M primitives: 3.4 GB
of records, 1.3 GB of wide nodes on the device), 3840x2160, four lights - `bench.py --n-side 4000`.

Fixture: tests/golden/hbm_sample.npz (oracle/make_golden_large.py hbm, made in the development container): 4 096 seeded
pixels, every primary ray and every shadow ray cast from its hit resolved by the oracle's loop over ALL primitives in
ascending id order - no tree of anyone's.  Checked: ids and FP64 distance bits equal, one verdict bit per light equal,
colour within +-1; the frame's counters agree with its id plane; a rank's tiles reproduce the single launch; the frame
through a hierarchy built on the device equals the frame through the host's."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from raytracer_amd import capi
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu

N_SIDE = 4000


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def test_sixteen_million_cells_4k():
    import torch
    W, H = 3840, 2160
    g = np.load(os.path.join(GOLDEN, "hbm_sample.npz"))
    half = N_SIDE / 2.0
    prims = S.mixed_field(n_side=N_SIDE, seed=0x100000)
    lights = S.mixed_field_lights(half_extent=half, height=120.0 * half / 500.0, intensity=4000.0 * (half / 500.0) ** 2)
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=half, res=(W, H)))
    assert tuple(g["res"]) == (W, H) and int(g["n_prims"]) == len(prims) and int(g["n_side"]) == N_SIDE
    sc = capi.Scene.create(prims, lights)
    info = sc.info()
    assert info.depth <= 31 and sc.n_wide() > 0
    # the working set of the walks: far beyond the 256 MiB Infinity Cache
    assert 128 * sc.n_wide() + 160 * int(info.n_slots) > (4 << 30)
    ctx = capi.Context(0)
    try:
        ctx.upload(sc)
        ctx.reset_counters()
        rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
        c = ctx.counters()
        hits = int((ids >= 0).sum())
        assert c["primary_rays"] == W * H and c["primary_hits"] == hits and c["shadow_rays"] == len(lights) * hits
        tids, dist = ctx.trace_primary(cam)
        assert np.array_equal(tids, ids)
        mask = torch.empty(H * W, dtype=torch.uint8, device="cuda")
        ctx.trace_shadows_dev(cam, (0, 0, W, H), mask.data_ptr())
        ctx.synchronize()
        mask = mask.cpu().numpy().reshape(H, W)
        px, py = g["px"].astype(np.int64), g["py"].astype(np.int64)
        assert int((g["id"] >= 0).sum()) > 3000
        assert np.array_equal(ids[py, px], g["id"])
        assert np.array_equal(bits(dist[py, px]), bits(g["dist"]))
        assert np.array_equal(mask[py, px], g["lit_mask"])
        lut = capi.gamma_lut().astype(np.int16)
        lin = g["rgb_linear"].astype(np.int16)
        got = rgb[py, px].astype(np.int16)
        assert ((got >= lut[np.clip(lin - 1, 0, 255)]) & (got <= lut[np.clip(lin + 1, 0, 255)])).all()
        assert (got != lut[lin]).any(-1).mean() < 2e-3
        # the tile partition at this size: rank 3 of 8 renders its tiles; they equal the same pixels of the single launch
        tiling = capi.Tiling(W, H, 64, 8)
        per = capi.tiles_per_rank(tiling)
        buf = torch.empty(per * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
        ctx.render_tiles_dev(cam, tiling, 3, buf.data_ptr(), 0, 0.0)
        ctx.synchronize()
        tiles = buf.cpu().numpy().reshape(per, 64, 64, 3)
        tiles_x = (W + 63) // 64
        n_tiles = tiles_x * ((H + 63) // 64)
        for lt in range(0, per, 37):
            t = lt * 8 + 3
            if t >= n_tiles:
                break
            x0, y0 = (t % tiles_x) * 64, (t // tiles_x) * 64
            h, w = min(64, H - y0), min(64, W - x0)
            assert np.array_equal(tiles[lt, :h, :w], rgb[y0:y0 + h, x0:x0 + w])
        # the hierarchy built on the device for the same primitives (beyond 16.7 M of them its wide form is not made:
        # wide node i = inner node i would pass the 2^25 nodes the kernels can address, so the binary nodes are walked):
        # results do not depend on whose tree it is
        sc = None
        dev = capi.Scene.create(prims, lights, on_device=True)
        ctx.upload(dev)
        assert ctx.build_time_ms() > 0.0
        rgb2, ids2 = ctx.render(cam, max_scatters=0, want_ids=True)
        assert np.array_equal(ids2, ids) and np.array_equal(rgb2, rgb)
    finally:
        ctx.close()
