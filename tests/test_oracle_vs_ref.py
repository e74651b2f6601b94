"""Development-container only: the C oracle against the real reference engine run live
(oracle/_ref/libref_rt.so, compiled from the reference's own sources).  Skipped where the
reference is absent (GPU box); the committed goldens cover the same ground there."""
import os
import tempfile

import numpy as np
import pytest

from conftest import SCENES, SHIPPED
from oracle import bind
from raytracer_amd import scenes as S


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize("seed,n,nl", [(101, 30, 2), (102, 200, 2), (103, 7, 1)])
def test_random_scene_live(oracle, ref, seed, n, nl):
    prims, lights, camd = S.random_scene(n, seed, lights=nl)
    camd = dict(camd, res=(96, 54))
    with tempfile.TemporaryDirectory() as td:
        jp = os.path.join(td, "s.json")
        S.scene_to_json(prims, lights, {"3": camd}, jp)
        h = ref.load(jp)
        # the independent Python reader sees what the reference loader sees
        p2, l2, c2 = bind.load_scene_json(jp)
        assert p2.tobytes() == prims.tobytes() and l2.tobytes() == lights.tobytes()
        rays = ref.primary_rays(h, "3")
        cam = bind.camera_from_dict(oracle, camd)
        sc = oracle.scene(prims, lights)
        ys, xs = np.mgrid[0:54, 0:96]
        mine = oracle.pixel_rays(cam, np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
        assert np.array_equal(bits(mine), bits(rays))
        a, b = ref.trace_rays(h, rays), oracle.trace_rays(sc, rays)
        assert np.array_equal(a["id"], b["id"])
        for f in ("dist", "p", "n", "r", "u", "v"):
            assert np.array_equal(bits(a[f]), bits(b[f])), f
        assert np.array_equal(ref.shade_rays(h, rays), oracle.shade_rays(sc, rays))
        # the blocked brute-force loop that makes the fixtures at a million primitives, with and without the
        # per-primitive boxes that let it pass over primitives: the reference's hits and colours, bit for bit
        for scb in (oracle.scene(prims, lights), oracle.with_prim_bounds(oracle.scene(prims, lights))):
            hb, lit, lin = oracle.brute_shade_rays(scb, rays)
            assert np.array_equal(a["id"], hb["id"]) and np.array_equal(bits(a["dist"]), bits(hb["dist"]))
            assert np.array_equal(ref.shade_rays(h, rays), lin)
            assert not lit[hb["hit"] == 0].any()
        for ms in (0, 3):
            out = os.path.join(td, "o.ppm")
            assert ref.render_ppm(h, "3", out, 1, ms, 0.0)
            data = open(out, "rb").read()
            img = np.frombuffer(data[len(b"P6\n96 54\n255\n"):], dtype=np.uint8).reshape(54, 96, 3)
            mine_img, _, _ = oracle.render(sc, cam, max_scatters=ms)
            assert np.array_equal(img, mine_img), ms
        # min_scatter_intensity cut-off
        out = os.path.join(td, "o2.ppm")
        assert ref.render_ppm(h, "3", out, 1, 2, 0.3)
        data = open(out, "rb").read()
        img = np.frombuffer(data[len(b"P6\n96 54\n255\n"):], dtype=np.uint8).reshape(54, 96, 3)
        mine_img, _, _ = oracle.render(sc, cam, max_scatters=2, min_scatter_intensity=0.3)
        assert np.array_equal(img, mine_img)
        ref.free(h)


def test_camera_with_up_vector_and_vertical_fit(oracle, ref):
    """Both updateRotationMatrix overloads and both sensor-fit branches, through the reference's scene loader."""
    prims, lights, _ = S.random_scene(5, 5)
    cams = {
        "1": dict(location=[1, -9, 4], direction=[0.1, 0.9, -0.3], up=[0.2, 0.1, 1.0], focal=50.0, sensor=(36.0, 24.0), res=(64, 48)),
        "2": dict(location=[1, -9, 4], direction=[0, 0, -1], up=None, focal=28.0, sensor=(36.0, 24.0), res=(40, 70)),
        "3": dict(location=[1, -9, 4], direction=[0.1, 0.9, -0.3], up=[0.1, 0.9, -0.3], focal=35.0, sensor=(32.0, 32.0), res=(50, 50)),
    }
    with tempfile.TemporaryDirectory() as td:
        jp = os.path.join(td, "s.json")
        S.scene_to_json(prims, lights, cams, jp)
        h = ref.load(jp)
        for cid, c in cams.items():
            rays = ref.primary_rays(h, cid)
            cam = bind.camera_from_dict(oracle, c)
            W, H = c["res"]
            ys, xs = np.mgrid[0:H, 0:W]
            mine = oracle.pixel_rays(cam, np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
            assert np.array_equal(bits(mine), bits(rays)), cid
        ref.free(h)


@pytest.mark.parametrize("name,cam_id", SHIPPED[:1] + SHIPPED[2:3])
def test_read_camera_from_json_path(oracle, ref, name, cam_id):
    """read_camera_from_json (world-up overload only) at fractional pixel coordinates."""
    path = os.path.join(SCENES, name + ".json")
    _, _, cams = bind.load_scene_json(path)
    cam = bind.camera_from_dict(oracle, dict(cams[cam_id], up=None))
    rng = np.random.default_rng(3)
    pxy = rng.uniform(0, 480, size=(500, 2))
    assert np.array_equal(bits(ref.camera_rays(path, cam_id, pxy)), bits(oracle.pixel_rays(cam, pxy)))
