"""GPU check of the two forms of the resolve stage against each other.

A frame whose lights fit one batch is resolved in two passes (mixed precision for every slot whose colour that decides,
FP64 for the listed rest) when few of the scene's materials have highlights too sharp for the mixed form; small
rectangles (under 16 384 pixels: the list would not fit its plane) and scenes with many sharp highlights take the single
kernel that holds both forms.  All of them must write the colours of the all-FP64 evaluation, so the same frame rendered
whole and rendered as small rectangles has to come out byte for byte the same - with sharp highlights in the scene (below
the sixty-fourth of the materials at which the choice flips) and with none."""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu

N_SIDE = 300
W, H = 1280, 720
PIECE_W, PIECE_H = 128, 96          # 12 288 pixels a piece: always the single kernel


def scene(sharp_fraction):
    half = N_SIDE / 2.0
    prims = S.mixed_field(n_side=N_SIDE, seed=0x5EED)
    n = len(prims)
    rng = S.SplitMix64(0xC0FFEE)
    u = rng.uniform(n)
    sharp = u < sharp_fraction
    # exponents of 798 and 198: both above the 64 up to which the mixed form is used
    prims["roughness"] = np.where(sharp, np.where(u < sharp_fraction / 2, 0.05, 0.1), prims["roughness"])
    lights = S.mixed_field_lights(half_extent=half, height=120.0 * half / 500.0, intensity=4000.0 * (half / 500.0) ** 2)
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=half, res=(W, H)))
    return prims, lights, cam, sharp


@pytest.mark.parametrize("sharp_fraction", [0.0, 0.01, 0.05])
def test_whole_frame_equals_its_pieces(sharp_fraction):
    import torch
    prims, lights, cam, sharp = scene(sharp_fraction)
    ctx = capi.Context(0)
    try:
        ctx.upload(capi.Scene.create(prims, lights))
        rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
        hit = ids >= 0
        assert hit.mean() > 0.5
        if sharp_fraction > 0.0:
            # the sharp materials are in view, so both forms of the light term are exercised
            assert sharp[ids[hit]].sum() > 1000
        assert len(np.unique(rgb.reshape(-1, 3), axis=0)) > 500
        buf = torch.empty(PIECE_W * PIECE_H * 3, dtype=torch.uint8, device="cuda")
        for y0 in range(0, H, PIECE_H):
            for x0 in range(0, W, PIECE_W):
                w, h = min(PIECE_W, W - x0), min(PIECE_H, H - y0)
                ctx.render_rect_dev(cam, (x0, y0, w, h), buf.data_ptr(), None, 0, 0.0)
                ctx.synchronize()
                piece = buf[: w * h * 3].cpu().numpy().reshape(h, w, 3)
                assert np.array_equal(piece, rgb[y0:y0 + h, x0:x0 + w]), (x0, y0)
    finally:
        ctx.close()
