"""The C oracle against golden fixtures produced by the REAL reference engine (oracle/make_golden.py).
Bit-exact everywhere: ids, float64 distances and hit records, 8-bit colours."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, SHIPPED
from oracle import bind
from oracle.make_golden import row_hash
from raytracer_amd import scenes as S


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def load(name, cam_id, oracle):
    prims, lights, cams = bind.load_scene_json(os.path.join(SCENES, name + ".json"))
    return prims, lights, bind.camera_from_dict(oracle, cams[cam_id])


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_shipped_scene_primary_hits(oracle, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    prims, lights, cam = load(name, cam_id, oracle)
    assert (cam.width, cam.height) == (int(g["width"]), int(g["height"]))
    assert len(prims) == int(g["n_meshes"])
    ids, dist = oracle.trace_primary(oracle.scene(prims, lights), cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist[::4]), bits(g["dist_rows"]))
    assert np.array_equal(row_hash(dist), g["dist_row_hash"])


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_shipped_scene_hit_records_and_aabbs(oracle, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    prims, lights, cam = load(name, cam_id, oracle)
    W, H = cam.width, cam.height
    idx = g["rec_idx"]
    pxy = np.stack([idx % W + 0.5, idx // W + 0.5], axis=1)
    rays = oracle.pixel_rays(cam, pxy)
    hits = oracle.trace_rays(oracle.scene(prims, lights), rays)
    m = hits["hit"] == 1
    assert np.array_equal(hits["id"], g["ids"].reshape(-1)[idx].astype(np.int32))
    for f in ("p", "n", "r", "u", "v"):
        assert np.array_equal(bits(hits[f][m]), bits(g["rec_" + f][m])), f
    boxes = np.array([oracle.aabb(p) for p in prims])
    assert np.array_equal(bits(boxes), bits(g["aabbs"]))


@pytest.mark.parametrize("name,cam_id", [s for s in SHIPPED if s[0] != "200_rand_1_cam"])
def test_shipped_scene_shading_and_renders(oracle, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    prims, lights, cam = load(name, cam_id, oracle)
    sc = oracle.scene(prims, lights)
    W, H = cam.width, cam.height
    ys, xs = np.mgrid[0:H:3, 0:W:3]
    rays = oracle.pixel_rays(cam, np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
    assert np.array_equal(oracle.shade_rays(sc, rays), g["shade_linear"][::3, ::3].reshape(-1, 3))
    for ms in (0, 2):
        img, n_hits, n_shadow = oracle.render(sc, cam, max_scatters=ms)
        assert np.array_equal(img, g["render%d" % ms]), "max_scatters=%d" % ms
        assert n_hits == int((g["ids"] >= 0).sum())
        if ms == 0:
            assert n_shadow == n_hits * len(lights)


def test_no_lights_scene_renders_black(oracle):
    g = np.load(os.path.join(GOLDEN, "shipped_200_rand_1_cam.npz"))
    prims, lights, cam = load("200_rand_1_cam", "1", oracle)
    assert len(lights) == 0
    img, _, n_shadow = oracle.render(oracle.scene(prims, lights), cam, max_scatters=2)
    assert not img.any() and n_shadow == 0
    assert np.array_equal(img, g["render2"])


@pytest.mark.parametrize("name,cam_id", [s for s in SHIPPED if s[0] not in ("200_rand_1_cam",)])
def test_shipped_scene_secondary_rays(oracle, name, cam_id):
    """Shadow-style rays that start on surfaces (origins inside BVH boxes)."""
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    prims, lights, cam = load(name, cam_id, oracle)
    hits = oracle.trace_rays(oracle.scene(prims, lights), g["sec_rays"])
    assert np.array_equal(hits["id"], g["sec_ids"].astype(np.int32))
    assert np.array_equal(bits(hits["dist"]), bits(g["sec_dist"]))


@pytest.mark.parametrize("kind", [bind.CUBE, bind.PLANE, bind.CYLINDER, bind.SPHERE])
def test_single_primitives(oracle, kind):
    """Random and adversarial rays (inside origins, axis-parallel, zero direction, zero scale,
    degenerate quads) against single primitives built through the reference's constructors."""
    g = np.load(os.path.join(GOLDEN, "single_prims.npz"))
    k = S.KIND_NAMES[kind].lower()
    for t in range(len(g[k + "_params"])):
        prim = bind.new_prims(1)
        prim["kind"] = kind
        prim["q"][0] = g[k + "_params"][t]
        hits = oracle.intersect(prim, g[k + "_rays"][t])
        assert np.array_equal(hits["hit"], g[k + "_hit"][t]), (k, t)
        m = hits["hit"] == 1
        for f in ("dist", "p", "n", "r", "u", "v"):
            a, b = hits[f][m], g[k + "_" + f][t][m]
            same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (k, t, f)
        box = oracle.aabb(prim)
        assert np.array_equal(bits(box), bits(g[k + "_aabb"][t])), (k, t)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_synthetic_scenes(oracle, tag):
    """Seeded mixed scenes (all four kinds, materials, 1-3 lights) that went through the reference's own loader."""
    g = np.load(os.path.join(GOLDEN, "synthetic_%s.npz" % tag))
    prims, lights, camd = S.random_scene(int(g["n"]), int(g["seed"]), lights=int(g["n_lights"]))
    cam = bind.camera_from_dict(oracle, camd)
    sc = oracle.scene(prims, lights)
    ids, dist = oracle.trace_primary(sc, cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["dist"]))
    for ms in (0, 1, 2):
        img, _, _ = oracle.render(sc, cam, max_scatters=ms)
        assert np.array_equal(img, g["render%d" % ms]), ms
    hits = oracle.trace_rays(sc, g["sec_rays"])
    assert np.array_equal(hits["id"], g["sec_ids"].astype(np.int32))
    assert np.array_equal(bits(hits["dist"]), bits(g["sec_dist"]))


def test_triangle_pairs_match_reference_planes(oracle):
    """The reference has no triangle.  A parallelogram split into two triangles must give the reference
    Plane's id (tri_id -> owning parallelogram) and the same distance to 1e-12 relative."""
    g = np.load(os.path.join(GOLDEN, "mixed_field_planes.npz"))
    n_side, seed = int(g["n_side"]), int(g["seed"])
    quads = S.mixed_field(n_side=n_side, seed=seed, planes_as_triangles=False)
    tris = S.mixed_field(n_side=n_side, seed=seed, planes_as_triangles=True)
    assert np.array_equal(quads["kind"], g["kinds"])
    cam = bind.camera_from_dict(oracle, S.mixed_field_camera(half_extent=10.0, res=(240, 135)))
    ids, dist = oracle.trace_primary(oracle.scene(tris), cam)
    # map triangle-scene ids back to quad-scene ids
    n_cube = int((quads["kind"] == S.CUBE).sum())
    n_quad = int((quads["kind"] == S.PLANE).sum())            # parallelograms + ground (last)
    n_sph = int((quads["kind"] == S.SPHERE).sum())
    assert int((tris["kind"] == S.PLANE).sum()) == 1          # ground only
    mapped = ids.copy()
    is_ground = ids == n_cube
    is_sph = (ids > n_cube) & (ids < n_cube + 1 + n_sph)
    is_tri = ids >= n_cube + 1 + n_sph
    mapped[is_ground] = n_cube + n_quad - 1
    mapped[is_sph] = ids[is_sph] - 1 + n_quad
    mapped[is_tri] = n_cube + (ids[is_tri] - (n_cube + 1 + n_sph)) // 2
    ref_ids = g["ids"].astype(np.int32)
    assert np.array_equal(mapped, ref_ids)
    hit = ref_ids >= 0
    rel = np.abs(dist[hit] - g["dist"][hit]) / g["dist"][hit]
    assert rel.max() < 1e-12


def test_textured_scene(oracle, tmp_path):
    """Albedo textures: the restatement of Mesh::evaluate_albedo against the reference's loader + Image + shading."""
    g = np.load(os.path.join(GOLDEN, "textured.npz"))
    prims, lights, camd, textures, prim_tex, _ = S.textured_scene(str(tmp_path))
    # what the reference ended up with: files that do not load leave the solid colour
    assert np.array_equal(prim_tex, g["prim_tex"])
    cam = bind.camera_from_dict(oracle, camd)
    sc = oracle.scene(prims, lights, textures, prim_tex)
    ids, dist = oracle.trace_primary(sc, cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist), bits(g["dist"]))
    W, H = camd["res"]
    ys, xs = np.divmod(np.arange(W * H), W)
    rays = oracle.pixel_rays(cam, np.stack([xs + 0.5, ys + 0.5], axis=1))
    hits = oracle.trace_rays(sc, rays)
    hit = hits["hit"] == 1
    assert np.array_equal(bits(hits["u"][hit]), bits(g["u"].reshape(-1)[hit]))
    assert np.array_equal(bits(hits["v"][hit]), bits(g["v"].reshape(-1)[hit]))
    assert np.array_equal(oracle.shade_rays(sc, rays).reshape(H, W, 3), g["shade_linear"])
    for ms in (0, 2):
        img, _, _ = oracle.render(sc, cam, max_scatters=ms)
        assert np.array_equal(img, g["render%d" % ms])
    # the textures matter: without them the frame differs
    plain, _, _ = oracle.render(oracle.scene(prims, lights), cam, max_scatters=0)
    assert not np.array_equal(plain, g["render0"])


@pytest.mark.parametrize("seed,kinds", [(31, (0, 1, 2, 3, 4)), (32, (0, 3)), (33, (1, 4))])
def test_blocked_brute_force_equals_the_plain_loop(oracle, seed, kinds):
    """rto_brute_shade_rays - the loop that makes the fixtures at a million primitives - against rto_trace_rays /
    rto_shade_rays (both pinned on the reference's fixtures above): hits, verdict-derived colours, with and without the
    per-primitive boxes, with degenerate (zero-scale) primitives in the scene."""
    from raytracer_amd import scenes as S
    prims, lights, camd = S.random_scene(700, seed, kinds=kinds, lights=3)
    prims["q"][5, 6:9] = 0.0                      # a cube / plane record with zeroed tail: unbounded or degenerate
    cam = bind.camera_from_dict(oracle, dict(camd, res=(64, 48)))
    ys, xs = np.mgrid[0:48, 0:64]
    rays = oracle.pixel_rays(cam, np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
    sc = oracle.scene(prims, lights)
    want, want_lin = oracle.trace_rays(sc, rays), oracle.shade_rays(sc, rays)
    for scb in (sc, oracle.with_prim_bounds(oracle.scene(prims, lights))):
        hits, lit, lin = oracle.brute_shade_rays(scb, rays, threads=4)
        assert hits.tobytes() == want.tobytes()
        assert np.array_equal(lin, want_lin)
        assert lit.shape == (len(rays), 3) and not lit[hits["hit"] == 0].any()
        assert oracle.trace_rays(scb, rays).tobytes() == want.tobytes()
