"""GPU parity at the benchmark configurations' sizes, against fixtures that owe nothing to the product's hierarchy
(oracle/make_golden_large.py, made in the development container):
  config3_sample / config4_sample - 65 536 / 16 384 seeded pixels of the 4K / 8K frame over the 1M-cell scene, every
      primary and shadow ray resolved by the oracle's loop over ALL 1.33 M primitives: ids, FP64 distances, one verdict
      bit per light, linear colour;
  config3_ref_bvh_ids - the id of every pixel of the 4K frame from the REAL reference's own hierarchy
      (get_BVH_tree + accelerated_one_pass_intersection) over the same cells built through its public constructors;
  config5_sample - 65 536 pixels of the all-mirror instanced scene with two bounces, from the oracle's loop over all
      instances.
Beside them: strided samples against the CPU walk of the product's own tree, and size-independent properties - tile
partition == single launch, ids of the trace kernel == ids of the render kernel, counters consistent with the id plane,
triangle ids / 12 == cube ids."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

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


def oracle_bvh(oracle, sc):
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    return oracle.bvh(nodes, slots, n_slots, n_loose, radius, sc.wide_bvh(), sc.loose_boxes())      # walked the way the kernels walk it


def ocamera(cam):
    from oracle import bind
    return bind.Camera.from_buffer_copy(bytes(cam))


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def check_sample(ctx, cam, g, rgb, ids, dist, mask):
    """A frame's planes against a brute-force fixture at the fixture's pixels: ids and FP64 distances exact, shadow
    verdicts exact, colour within +-1 of the fixture's linear colour (under the display transfer table)."""
    px, py = g["px"].astype(np.int64), g["py"].astype(np.int64)
    assert np.array_equal(ids[py, px], g["id"])
    assert np.array_equal(bits(dist[py, px]), bits(g["dist"]))
    assert np.array_equal(mask[py, px], g["lit_mask"])
    lut = capi.gamma_lut().astype(np.int16)
    lin = g["rgb_linear"].astype(np.int16)
    got = rgb[py, px].astype(np.int16)
    ok = (got >= lut[np.clip(lin - 1, 0, 255)]) & (got <= lut[np.clip(lin + 1, 0, 255)])
    assert ok.all()
    assert (got != lut[lin]).any(-1).mean() < 1e-3
    return int((g["id"] >= 0).sum())


def frame_planes(ctx, cam, W, H):
    import torch
    rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    tids, dist = ctx.trace_primary(cam)
    assert np.array_equal(tids, ids)
    mask = torch.empty(H * W, dtype=torch.uint8, device="cuda")
    ctx.trace_shadows_dev(cam, (0, 0, W, H), mask.data_ptr())
    ctx.synchronize()
    return rgb, ids, dist, mask.cpu().numpy().reshape(H, W)


def test_medium_mixed_field_full_frame(ctx, oracle):
    """120k primitives at 1280x720: every pixel against the CPU walk (ids exact, colours +-1)."""
    n_side = 300
    prims = S.mixed_field(n_side=n_side, seed=0x100000)
    lights = S.mixed_field_lights(half_extent=150.0, height=36.0, intensity=360.0)
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=150.0, res=(1280, 720)))
    sc = capi.Scene.create(prims, lights)
    ctx.upload(sc)
    rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    want_ids, want_rgb, _, hits = oracle.bvh_render(oracle.scene(prims, lights), oracle_bvh(oracle, sc), ocamera(cam))
    assert np.array_equal(ids, want_ids)
    diff = np.abs(rgb.astype(np.int16) - want_rgb.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).any(-1).mean() < 2e-4
    # brute force (the reference's algorithm) on a 4096-ray sample
    rng = np.random.default_rng(4)
    pxy = np.stack([rng.integers(0, 1280, 4096) + 0.5, rng.integers(0, 720, 4096) + 0.5], axis=1)
    rays = capi.pixel_rays(cam, pxy)
    brute = oracle.trace_rays(oracle.scene(prims, lights), rays)
    gid, gdist = ctx.trace_rays(rays)
    assert np.array_equal(gid, brute["id"]) and np.array_equal(bits(gdist), bits(brute["dist"]))


def test_one_million_cells_4k(ctx, oracle):
    """The benchmark workload: 1M-cell mixed scene, 3840x2160, primary + shadow rays to 4 lights."""
    import torch
    W, H = 3840, 2160
    prims = S.mixed_field(n_side=1000, seed=0x100000)
    lights = S.mixed_field_lights()
    cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
    sc = capi.Scene.create(prims, lights)
    assert sc.info().depth <= 31
    ctx.upload(sc)
    ctx.reset_counters()
    rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    c = ctx.counters()
    hits = int((ids >= 0).sum())
    assert c["primary_rays"] == W * H and c["primary_hits"] == hits and c["shadow_rays"] == 4 * hits
    tids, tdist = ctx.trace_primary(cam)
    assert np.array_equal(tids, ids)
    assert np.isinf(tdist[ids < 0]).all() and np.isfinite(tdist[ids >= 0]).all()
    # strided sample against the CPU walk: 240 x 135 pixels
    step = 16
    want_ids, want_rgb, _, _ = oracle.bvh_render(oracle.scene(prims, lights), oracle_bvh(oracle, sc), ocamera(cam), step=step)
    assert np.array_equal(ids[::step, ::step], want_ids)
    diff = np.abs(rgb[::step, ::step].astype(np.int16) - want_rgb.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).any(-1).mean() < 1e-3
    # brute force (the reference's algorithm: every ray against all 1.33 M primitives) on 65 536 seeded pixels: ids,
    # distances, shadow verdicts per light, colour
    g = load_fixture("config3_sample")
    assert tuple(g["res"]) == (W, H) and int(g["n_prims"]) == len(prims)
    mask = torch.empty(H * W, dtype=torch.uint8, device="cuda")
    ctx.trace_shadows_dev(cam, (0, 0, W, H), mask.data_ptr())
    ctx.synchronize()
    assert check_sample(ctx, cam, g, rgb, ids, tdist, mask.cpu().numpy().reshape(H, W)) > 50000
    # the arbitrary-ray kernels on a part of the same sample (they walk the tree in one go, lane by lane)
    sub = capi.pixel_rays(cam, np.stack([g["px"][:4096] + 0.5, g["py"][:4096] + 0.5], axis=1))
    gid, gdist = ctx.trace_rays(sub)
    assert np.array_equal(gid, g["id"][:4096]) and np.array_equal(bits(gdist), bits(g["dist"][:4096]))
    rt = torch.from_numpy(sub).cuda()
    lin = torch.empty((len(sub), 3), dtype=torch.uint8, device="cuda")
    ctx.shade_rays_dev(rt.data_ptr(), len(sub), lin.data_ptr())
    ctx.synchronize()
    assert np.abs(lin.cpu().numpy().astype(np.int16) - g["rgb_linear"][:4096].astype(np.int16)).max() <= 1
    # the 8-GPU partition on this frame: 8 virtual ranks, 64-pixel tiles
    tiling = capi.Tiling(W, H, 64, 8)
    per = capi.tiles_per_rank(tiling) * 64 * 64 * 3
    gathered = torch.empty(8 * per, dtype=torch.uint8, device="cuda")
    for r in range(8):
        ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, 0, 0.0)
    frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), rgb)


def test_one_million_cells_against_the_reference_hierarchy(ctx):
    """Every pixel of the 4K frame against the REAL reference's own BVH path (valid for camera rays: the camera is
    outside the scene's bounds).  The reference has no triangle, so its scene holds each parallelogram as a Plane: the
    product is checked on that scene id for id, and on the benchmark's scene (two triangles per parallelogram) through
    the map from triangle pairs to planes."""
    W, H = 3840, 2160
    want = load_fixture("config3_ref_bvh_ids")["ids"]
    cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
    lights = S.mixed_field_lights()
    planes = S.mixed_field(n_side=1000, seed=0x100000, planes_as_triangles=False)
    assert len(planes) == int(load_fixture("config3_ref_bvh_ids")["n_prims"])
    ctx.upload(capi.Scene.create(planes, lights))
    ids, _ = ctx.trace_primary(cam)
    assert np.array_equal(ids, want)
    tris = S.mixed_field(n_side=1000, seed=0x100000)
    ctx.upload(capi.Scene.create(tris, lights))
    tid, _ = ctx.trace_primary(cam)
    # reference order: cubes, planes, spheres (, triangles).  In the plane scene the planes are the parallelograms in
    # cell order followed by the ground; in the triangle scene the ground is the only plane and the triangles come last,
    # two per parallelogram in the same cell order.
    n_cubes = int((tris["kind"] == S.CUBE).sum())
    n_quads = int((planes["kind"] == S.PLANE).sum()) - 1
    first_tri = int(np.argmax(tris["kind"] == S.TRIANGLE))
    assert first_tri + 2 * n_quads == len(tris)
    to_plane_scene = np.empty(len(tris) + 1, dtype=np.int64)
    to_plane_scene[-1] = -1                                              # a miss stays a miss
    to_plane_scene[:n_cubes] = np.arange(n_cubes)
    to_plane_scene[n_cubes] = n_cubes + n_quads                          # the ground
    n_sph = first_tri - n_cubes - 1
    to_plane_scene[n_cubes + 1:first_tri] = n_cubes + n_quads + 1 + np.arange(n_sph)
    to_plane_scene[first_tri:len(tris)] = n_cubes + np.arange(2 * n_quads) // 2
    assert np.array_equal(to_plane_scene[tid], want)


def test_cube_field_as_triangles(ctx, oracle):
    """The triangle workload: 8334 cubes -> 100 008 triangles, 1920x1080, primary rays only.  The reference has no
    triangle: triangle id // 12 must equal the reference Cube's id and the distance agree to 1e-6 relative."""
    cubes = S.cube_field()
    cam = capi.make_camera(**S.CUBE_FIELD_CAMERA)
    sc_c = capi.Scene.create(cubes)
    ctx.upload(sc_c)
    cid, cdist = ctx.trace_primary(cam)
    step = 8
    want_ids, _, _, _ = oracle.bvh_render(oracle.scene(cubes), oracle_bvh(oracle, sc_c), ocamera(cam), step=step)
    assert np.array_equal(cid[::step, ::step], want_ids)
    rng = np.random.default_rng(2)
    px, py = rng.integers(0, 1920, 2048), rng.integers(0, 1080, 2048)
    brute = oracle.trace_rays(oracle.scene(cubes), capi.pixel_rays(cam, np.stack([px + 0.5, py + 0.5], axis=1)))
    assert np.array_equal(cid[py, px], brute["id"]) and np.array_equal(bits(cdist[py, px]), bits(brute["dist"]))
    tris = S.cubes_to_triangles(cubes)
    assert len(tris) == 100008
    ctx.upload(capi.Scene.create(tris))
    tid, tdist = ctx.trace_primary(cam)
    owner = np.where(tid >= 0, tid // 12, -1)
    mismatch = owner != cid
    # silhouette pixels may differ where the triangle corners (placed with numpy's sin/cos) and the analytic cube
    # disagree by an ulp; they must be isolated and only between hit and miss or neighbouring surfaces at equal depth
    assert mismatch.mean() < 2e-5
    both = (cid >= 0) & ~mismatch
    rel = np.abs(tdist[both] - cdist[both]) / cdist[both]
    assert rel.max() < 1e-6


def test_mirror_bounces_on_a_dense_field(ctx, oracle):
    """The level-by-level bounce pipeline where secondary rays are incoherent: 30k primitives, two bounces, against the
    brute-force restatement of trace_ray_recursive on a 24x16 window in the middle of the field, plus the partition
    property on the whole frame."""
    import torch
    prims = S.mixed_field(n_side=150, seed=0x100000)
    prims["metallic"][::4] = 0.7
    prims["roughness"][::5] = 0.05
    lights = S.mixed_field_lights(half_extent=75.0, height=18.0, intensity=90.0)
    W, H = 640, 360
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=75.0, res=(W, H)))
    sc = capi.Scene.create(prims, lights)
    ctx.upload(sc)
    ctx.reset_counters()
    rgb = ctx.render(cam, max_scatters=2)
    c = ctx.counters()
    assert c["bounce_rays"] > c["primary_hits"]            # dielectrics always reflect a little: two levels of bounces
    rect = (300, 200, 24, 16)
    want, _, _ = oracle.render(oracle.scene(prims, lights), ocamera(cam), rect=rect, max_scatters=2, threads=8)
    got = rgb[rect[1]:rect[1] + rect[3], rect[0]:rect[0] + rect[2]]
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).any(-1).mean() < 0.02
    # min_scatter_intensity cuts weak bounces: between the no-bounce and the full frame
    strong = ctx.render(cam, max_scatters=2, min_scatter_intensity=0.5)
    assert not np.array_equal(strong, rgb) and not np.array_equal(strong, ctx.render(cam, max_scatters=0))
    tiling = capi.Tiling(W, H, 48, 5)
    per = capi.tiles_per_rank(tiling) * 48 * 48 * 3
    gathered = torch.empty(5 * per, dtype=torch.uint8, device="cuda")
    for r in range(5):
        ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, 2, 0.0)
    frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), rgb)


def test_8k_frame(ctx, oracle):
    """The 8K workload's frame size on one GPU: 7680x4320 over the 1M-cell scene, sampled against the CPU walk."""
    W, H = 7680, 4320
    prims = S.mixed_field(n_side=1000, seed=0x100000)
    lights = S.mixed_field_lights()
    cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
    sc = capi.Scene.create(prims, lights)
    ctx.upload(sc)
    ctx.reset_counters()
    rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    c = ctx.counters()
    hits = int((ids >= 0).sum())
    assert c["primary_rays"] == W * H and c["primary_hits"] == hits and c["shadow_rays"] == 4 * hits
    step = 48
    want_ids, want_rgb, _, _ = oracle.bvh_render(oracle.scene(prims, lights), oracle_bvh(oracle, sc), ocamera(cam), step=step)
    assert np.array_equal(ids[::step, ::step], want_ids)
    diff = np.abs(rgb[::step, ::step].astype(np.int16) - want_rgb.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).any(-1).mean() < 1e-3
    # brute force over all primitives on 16 384 seeded pixels of this frame
    import torch
    g = load_fixture("config4_sample")
    assert tuple(g["res"]) == (W, H)
    _, dist = ctx.trace_primary(cam)
    mask = torch.empty(H * W, dtype=torch.uint8, device="cuda")
    ctx.trace_shadows_dev(cam, (0, 0, W, H), mask.data_ptr())
    ctx.synchronize()
    assert check_sample(ctx, cam, g, rgb, ids, dist, mask.cpu().numpy().reshape(H, W)) > 12000
    del mask, dist
    # the partition this frame is rendered with on 8 GPUs, as 8 virtual ranks on this one: the same bytes
    tiling = capi.Tiling(W, H, 64, 8)
    per = capi.tiles_per_rank(tiling) * 64 * 64 * 3
    gathered = torch.empty(8 * per, dtype=torch.uint8, device="cuda")
    for r in range(8):
        ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, 0, 0.0)
    frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), rgb)


def test_device_built_hierarchy_at_a_million_primitives(ctx):
    """The 1M-cell scene with the hierarchy built on the GPU: same frame as with the host's SAH tree, byte for byte
    (every primitive test is exact, so the tree cannot change a pixel); build time and depth as promised."""
    W, H = 1920, 1080
    prims = S.mixed_field(n_side=1000, seed=0x100000)
    lights = S.mixed_field_lights()
    cam = capi.make_camera(**S.mixed_field_camera(res=(W, H)))
    host = capi.Scene.create(prims, lights)
    ctx.upload(host)
    want_rgb, want_ids = ctx.render(cam, max_scatters=0, want_ids=True)
    dev = capi.Scene.create(prims, lights, on_device=True)
    assert dev.info().depth <= 31 and dev.info().n_nodes == 2 * dev.info().n_slots
    ctx.upload(dev)
    assert 0.0 < ctx.build_time_ms() < 100.0
    rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
    assert np.array_equal(ids, want_ids) and np.array_equal(rgb, want_rgb)


def test_hundred_million_instanced_triangles(ctx, oracle):
    """10 000 instances of a 10 008-triangle mesh at 3840x2160 with two mirror bounces.  The oracle loops over all
    instances and tests the mesh of each one the ray can touch, so a seeded sample of pixels and one block of the
    frame are within its reach: ids and distances exact, hit records exact, colours +-1.  Frame-wide: the tile
    partition reproduces the single launch, and the counters agree with the id plane."""
    import torch
    prims, inst, lights, camd = S.instanced_field(mirror=False)
    assert int(inst["n_prims"].sum()) == 100_080_000
    sc = capi.Scene.create_instanced(prims, inst, lights)
    ctx.upload(sc)
    cam = capi.make_camera(**camd)
    osc = oracle.instanced_scene(prims, inst, lights)
    W, H = camd["res"]
    rng = np.random.default_rng(55)
    px = np.stack([rng.integers(0, W, 2000) + 0.5, rng.integers(H // 2, H, 2000) + 0.5], axis=1)
    rays = capi.pixel_rays(cam, px)
    a = oracle.trace_rays(osc, rays)
    rec = ctx.trace_rays_full(rays)
    h = a["hit"] == 1
    assert h.sum() > 800
    assert np.array_equal(rec[:, 1].astype(np.int64), a["id"].astype(np.int64))
    assert np.array_equal(bits(rec[h, 2]), bits(a["dist"][h]))
    for lo, name in ((3, "p"), (6, "n"), (9, "r")):
        assert np.array_equal(bits(rec[h, lo:lo + 3]), bits(a[name][h])), name
    # one block of the frame through the whole pixel pipeline, bounces included
    x0, y0, w, h_ = W // 2 - 24, (5 * H) // 6, 48, 32
    want_rgb, _, _ = oracle.render(osc, ocamera(cam), rect=(x0, y0, w, h_), max_scatters=2)
    want_ids, _ = oracle.trace_primary(osc, ocamera(cam), rect=(x0, y0, w, h_))
    sub = torch.empty(h_ * w * 3, dtype=torch.uint8, device="cuda")
    sid = torch.empty(h_ * w, dtype=torch.int32, device="cuda")
    ctx.render_rect_dev(cam, (x0, y0, w, h_), sub.data_ptr(), sid.data_ptr(), 2, 0.0)
    ctx.synchronize()
    assert (want_ids >= 0).mean() > 0.1 and want_rgb.any()
    assert np.array_equal(sid.cpu().numpy().reshape(h_, w), want_ids)
    diff = np.abs(sub.cpu().numpy().reshape(h_, w, 3).astype(np.int16) - want_rgb.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).any(-1).mean() < 2e-3
    # the whole frame: counters against the id plane, tiles against the single launch
    ctx.reset_counters()
    full, ids = ctx.render(cam, max_scatters=2, want_ids=True)
    c = ctx.counters()
    assert c["primary_rays"] == W * H and c["primary_hits"] == int((ids >= 0).sum())
    assert np.array_equal(full[y0:y0 + h_, x0:x0 + w], sub.cpu().numpy().reshape(h_, w, 3))
    assert ids.max() < 100_080_000 and (ids >= 0).mean() > 0.3
    tiling = capi.Tiling(W, H, 64, 8)
    per = capi.tiles_per_rank(tiling) * 64 * 64 * 3
    gathered = torch.zeros((8 * per,), dtype=torch.uint8, device="cuda")
    for r in range(8):
        ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, 2, 0.0)
    frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), full)


def test_all_mirror_instanced_scene_against_the_fixture(ctx):
    """The 100 M-triangle scene as the benchmark configuration states it - every material a mirror, two bounces - on
    65 536 seeded pixels against the oracle's loop over all instances (no hierarchy involved): primary ids and distances
    exact, colour after two bounces and without bounces within +-1 of the fixture's linear colour."""
    prims, inst, lights, camd = S.instanced_field(mirror=True)
    g = load_fixture("config5_sample")
    W, H = camd["res"]
    assert tuple(g["res"]) == (W, H) and int(g["n_instances"]) == len(inst) and int(g["n_mesh_prims"]) == len(prims)
    ctx.upload(capi.Scene.create_instanced(prims, inst, lights))
    cam = capi.make_camera(**camd)
    px, py = g["px"].astype(np.int64), g["py"].astype(np.int64)
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids[py, px], g["id"]) and (g["id"] >= 0).sum() > 20000
    assert np.array_equal(bits(dist[py, px]), bits(g["dist"]))
    lut = capi.gamma_lut().astype(np.int16)
    for ms, key in ((2, "rgb_linear_2_bounces"), (0, "rgb_linear_0_bounces")):
        rgb = ctx.render(cam, max_scatters=ms)
        lin = g[key].astype(np.int16)
        got = rgb[py, px].astype(np.int16)
        ok = (got >= lut[np.clip(lin - 1, 0, 255)]) & (got <= lut[np.clip(lin + 1, 0, 255)])
        assert ok.all(), ms
        assert (got != lut[lin]).any(-1).mean() < 2e-3, ms
