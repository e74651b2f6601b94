"""rt_render_rect_static - the entry point that allocates nothing - against rt_render_rect, through the C ABI:
two calls in a row on the library's fixed workspace, the second with a smaller rectangle, bytes and ids equal."""
import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu


def test_static_workspace_render_matches_the_allocating_one():
    import torch
    prims, lights, camd = S.random_scene(400, 77, kinds=(S.CUBE, S.PLANE, S.CYLINDER, S.SPHERE, S.TRIANGLE), lights=3)
    W, H = 640, 480
    cam = capi.make_camera(**dict(camd, res=(W, H)))
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.upload(capi.Scene.create(prims, lights))

    def both(rect):
        n = rect[2] * rect[3]
        want_rgb = torch.zeros(n * 3, dtype=torch.uint8, device="cuda")
        want_ids = torch.zeros(n, dtype=torch.int32, device="cuda")
        got_rgb = torch.full((n * 3,), 7, dtype=torch.uint8, device="cuda")
        got_ids = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        ctx.render_rect_dev(cam, rect, want_rgb.data_ptr(), want_ids.data_ptr(), 0, 0.0)
        ctx.synchronize()
        free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: what the driver has left on this GPU
        ctx.render_rect_static_dev(cam, rect, got_rgb.data_ptr(), got_ids.data_ptr())
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the GPU's free memory"
        return want_rgb.cpu().numpy(), want_ids.cpu().numpy(), got_rgb.cpu().numpy(), got_ids.cpu().numpy()

    # first call: 245 760 of the 262 144 pixels the workspace takes; second: a smaller, ragged rectangle
    for rect in ((64, 0, 512, 480), (100, 37, 203, 101)):
        want_rgb, want_ids, got_rgb, got_ids = both(rect)
        assert (want_ids >= 0).mean() > 0.2 and want_rgb.any()
        assert np.array_equal(got_ids, want_ids)
        assert np.array_equal(got_rgb, want_rgb)
    # one pixel more than the workspace holds: refused with a clear error, nothing launched
    big = capi.make_camera(**dict(camd, res=(1024, 1024)))
    out = torch.zeros(513 * 512 * 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.RtError) as e:
        ctx.render_rect_static_dev(big, (0, 0, 513, 512), out.data_ptr(), None)
    assert e.value.status == -7 and "262144" in str(e.value)
    # a strip too thin for the workspace's tile planes (it spans 63 blocks of 128 x 128 pixels, 32 fit): refused as well
    strip = capi.make_camera(**dict(camd, res=(8000, 32)))
    with pytest.raises(capi.RtError) as e:
        ctx.render_rect_static_dev(strip, (0, 0, 8000, 32), out.data_ptr(), None)
    assert e.value.status == -7 and "blocks" in str(e.value)
    # ids are optional, and a scene without lights gives the reference's black frame
    rect = (0, 0, 128, 64)
    rgb = torch.full((128 * 64 * 3,), 9, dtype=torch.uint8, device="cuda")
    ctx.render_rect_static_dev(cam, rect, rgb.data_ptr(), None)
    ref = torch.zeros(128 * 64 * 3, dtype=torch.uint8, device="cuda")
    ctx.render_rect_dev(cam, rect, ref.data_ptr(), None, 0, 0.0)
    ctx.synchronize()
    assert torch.equal(rgb, ref)
    ctx.upload(capi.Scene.create(prims, None))
    ids = torch.full((128 * 64,), -7, dtype=torch.int32, device="cuda")
    ids_ref = torch.full((128 * 64,), -8, dtype=torch.int32, device="cuda")
    ctx.render_rect_static_dev(cam, rect, rgb.data_ptr(), ids.data_ptr())
    ctx.render_rect_dev(cam, rect, ref.data_ptr(), ids_ref.data_ptr(), 0, 0.0)
    ctx.synchronize()
    assert not rgb.any() and torch.equal(ids, ids_ref) and (ids >= 0).any()
    ctx.close()


def test_static_workspace_with_eight_lights():
    """Eight lights fill a whole batch.  A 512 x 512 rectangle (4096 tiles) still fits the workspace's per-tile planes and
    runs with the entry search; 600 x 400 (5120 tiles of the pixel map) does not and runs without it.  Same bytes either
    way, against rt_render_rect and against the context with the search switched off."""
    import torch
    prims, lights, camd = S.random_scene(300, 5, kinds=(S.CUBE, S.PLANE, S.SPHERE, S.TRIANGLE), lights=8)
    cam = capi.make_camera(**dict(camd, res=(640, 512)))
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.upload(capi.Scene.create(prims, lights))
    for rect in ((0, 0, 512, 512), (20, 50, 600, 400), (7, 3, 64, 40)):
        n = rect[2] * rect[3]
        want = torch.zeros(n * 3, dtype=torch.uint8, device="cuda")
        plain = torch.zeros(n * 3, dtype=torch.uint8, device="cuda")
        got = torch.full((n * 3,), 7, dtype=torch.uint8, device="cuda")
        ctx.render_rect_dev(cam, rect, want.data_ptr(), None, 0, 0.0)
        ctx.set_entry_search(False)
        ctx.render_rect_dev(cam, rect, plain.data_ptr(), None, 0, 0.0)
        ctx.set_entry_search(True)
        ctx.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        ctx.render_rect_static_dev(cam, rect, got.data_ptr(), None)
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        lit_pixels = int((want.view(-1, 3).max(dim=1).values > 0).sum())
        assert n < 10000 or lit_pixels > n // 50, (rect, lit_pixels)
        assert torch.equal(got, want) and torch.equal(plain, want), rect
    ctx.close()

