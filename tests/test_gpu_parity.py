"""GPU parity: the HIP path, called through the C ABI, against golden fixtures from the real reference and
against the CPU oracle on seeded inputs.  Bar: primitive ids and FP64 distances bit-exact; 8-bit colours are
allowed +-1 (north-star tolerance; device pow() may differ from libm in the last ulp) and the number of
pixels that differ at all is asserted to be tiny."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, SHIPPED
from raytracer_amd import capi
from raytracer_amd import scenes as S

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def assert_rgb_close(got, want, max_frac=2e-4):
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1, "8-bit colour differs by more than 1 (max %d)" % diff.max()
    frac = (diff > 0).any(-1).mean()
    assert frac <= max_frac, "%.3g of pixels differ by 1" % frac


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_shipped_scenes_ids_and_distances(ctx, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    for max_leaf in (1, 2):
        sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"), max_leaf=max_leaf)
        cam, _ = sc.camera(cam_id)
        ctx.upload(sc)
        ids, dist = ctx.trace_primary(cam)
        assert np.array_equal(ids, g["ids"].astype(np.int32))
        assert np.array_equal(bits(dist[::4]), bits(g["dist_rows"]))


@pytest.mark.parametrize("name,cam_id", [s for s in SHIPPED])
def test_shipped_scenes_renders(ctx, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    cam, _ = sc.camera(cam_id)
    ctx.upload(sc)
    ctx.reset_counters()
    rgb0, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    c = ctx.counters()
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert_rgb_close(rgb0, g["render0"])
    n_lights = sc.info().n_lights
    if n_lights:
        assert c["primary_rays"] == cam.width * cam.height
        assert c["primary_hits"] == int((g["ids"] >= 0).sum())
        assert c["shadow_rays"] == c["primary_hits"] * n_lights
    rgb2 = ctx.render(cam, max_scatters=2)
    assert_rgb_close(rgb2, g["render2"])


@pytest.mark.parametrize("name,cam_id", [s for s in SHIPPED if s[0] != "200_rand_1_cam"])
def test_shipped_scenes_secondary_rays(ctx, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    ctx.upload(sc)
    ids, dist = ctx.trace_rays(g["sec_rays"])
    assert np.array_equal(ids, g["sec_ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["sec_dist"]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_synthetic_scenes(ctx, tag):
    g = np.load(os.path.join(GOLDEN, "synthetic_%s.npz" % tag))
    prims, lights, camd = S.random_scene(int(g["n"]), int(g["seed"]), lights=int(g["n_lights"]))
    sc = capi.Scene.create(prims, lights)
    cam = capi.make_camera(**camd)
    ctx.upload(sc)
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["dist"]))
    for ms in (0, 1, 2):
        assert_rgb_close(ctx.render(cam, max_scatters=ms), g["render%d" % ms], max_frac=1e-3)
    ids, dist = ctx.trace_rays(g["sec_rays"])
    assert np.array_equal(ids, g["sec_ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["sec_dist"]))


def test_triangles_against_reference_planes(ctx):
    g = np.load(os.path.join(GOLDEN, "mixed_field_planes.npz"))
    n_side, seed = int(g["n_side"]), int(g["seed"])
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=10.0, res=(240, 135)))
    quads = S.mixed_field(n_side=n_side, seed=seed, planes_as_triangles=False)
    ctx.upload(capi.Scene.create(quads, S.mixed_field_lights(half_extent=10.0, height=6.0, intensity=40.0)))
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["dist"]))
    assert_rgb_close(ctx.render(cam, max_scatters=0), g["render0"], max_frac=1e-3)
    tris = S.mixed_field(n_side=n_side, seed=seed, planes_as_triangles=True)
    ctx.upload(capi.Scene.create(tris))
    tids, tdist = ctx.trace_primary(cam)
    n_cube = int((quads["kind"] == S.CUBE).sum()); n_quad = int((quads["kind"] == S.PLANE).sum())
    n_sph = int((quads["kind"] == S.SPHERE).sum())
    mapped = tids.copy()
    mapped[tids == n_cube] = n_cube + n_quad - 1
    m = (tids > n_cube) & (tids < n_cube + 1 + n_sph); mapped[m] = tids[m] - 1 + n_quad
    m = tids >= n_cube + 1 + n_sph; mapped[m] = n_cube + (tids[m] - (n_cube + 1 + n_sph)) // 2
    assert np.array_equal(mapped, g["ids"].astype(np.int32))
    hit = g["ids"] >= 0
    assert (np.abs(tdist[hit] - g["dist"][hit]) / g["dist"][hit]).max() < 1e-12
