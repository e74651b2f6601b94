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


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_brute_force_equals_the_hierarchy(ctx, name, cam_id):
    """rt_ctx_set_brute_force: the reference's own loop over every primitive (raytracer.cpp:143-157) on the device gives the
    reference's ids and distances, and the same hit records as the walk through the hierarchy, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    cam, _ = sc.camera(cam_id)
    ctx.upload(sc)
    rays = g["sec_rays"] if "sec_rays" in g else None
    rec_tree = ctx.trace_rays_full(rays) if rays is not None else None
    try:
        ctx.set_brute_force(True)
        ctx.set_counting(True)
        ctx.reset_counters()
        ids, dist = ctx.trace_primary(cam)
        c = ctx.counters()
        assert np.array_equal(ids, g["ids"].astype(np.int32))
        assert np.array_equal(bits(dist[::4]), bits(g["dist_rows"]))
        # every primitive against every ray, nothing skipped
        assert c["primary_rays"] == cam.width * cam.height and c["prims_primary"] == cam.width * cam.height * sc.info().n_prims
        assert c["nodes_primary"] == 0
        if rays is not None:
            ids2, dist2 = ctx.trace_rays(rays)
            assert np.array_equal(ids2, g["sec_ids"].astype(np.int32)) and np.array_equal(bits(dist2), bits(g["sec_dist"]))
            assert np.array_equal(bits(ctx.trace_rays_full(rays)), bits(rec_tree))
    finally:
        ctx.set_counting(False)
        ctx.set_brute_force(False)


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


def oracle_bvh(oracle, sc):
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    return oracle.bvh(nodes, slots, n_slots, n_loose, radius, sc.wide_bvh(), sc.loose_boxes())      # walked the way the kernels walk it


def ocamera(cam):
    from oracle import bind
    return bind.Camera.from_buffer_copy(bytes(cam))


def test_loose_degenerate_and_tied_primitives(ctx, oracle):
    prims, lights, camd = S.random_scene(40, 31, lights=2)
    extra = S.new_prims(8)
    extra["kind"] = [S.CUBE, S.SPHERE, S.PLANE, S.PLANE, S.SPHERE, S.SPHERE, S.CYLINDER, S.PLANE]
    extra["q"][0, :9] = [0.5, 0.2, 0.1, 0.3, 0.2, 0.1, 0.0, 0.7, 0.4]
    extra["q"][1, :6] = [-1.0, 0.5, 0.3, 0.6, 0.0, 0.5]
    extra["q"][2, :12] = [0, 0, 0, 1, 1, 1, 5, 5, 5, 2, 2, 2]
    extra["q"][3, :12] = [0, 0, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0]; extra["reserved"][3] = 1
    extra["q"][4, :6] = [1.0, -1.0, 0.5, 0.8, 0.8, 0.8]
    extra["q"][5, :6] = [1.0, -1.0, 0.5, 0.8, 0.8, 0.8]
    extra["q"][6, :9] = [-2.0, 1.0, 0.0, 0.4, 1.0, 2.0, 0.5, 0.5, 1.5]
    extra["q"][7, :12] = [-3, -3, -1.5, 3, -3, -1.5, 9, 9, 4, -3, 3, -1.5]
    allp = S.sort_reference_order(np.concatenate([prims, extra]))
    sc = capi.Scene.create(allp, lights)
    assert sc.info().n_loose == 2
    ctx.upload(sc)
    cam = capi.make_camera(**camd)
    osc = oracle.scene(allp, lights)
    want_ids, want_dist = oracle.trace_primary(osc, ocamera(cam))
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, want_ids)
    assert np.array_equal(bits(dist), bits(want_dist))
    for ms in (0, 2):
        want_rgb, _, _ = oracle.render(osc, ocamera(cam), max_scatters=ms)
        assert_rgb_close(ctx.render(cam, max_scatters=ms), want_rgb, max_frac=1e-3)
    rng = np.random.default_rng(9)
    o = rng.uniform(-4, 4, size=(20000, 3))
    d = rng.normal(size=(20000, 3)) * rng.uniform(0.01, 30, size=(20000, 1))
    d[:50, 0] = 0.0
    d[50:100, 1:] = 0.0
    d[100] = 0.0
    rays = np.concatenate([o, d], axis=1)
    a = oracle.trace_rays(osc, rays)
    ids, dist = ctx.trace_rays(rays)
    assert np.array_equal(ids, a["id"])
    assert np.array_equal(bits(dist), bits(a["dist"]))


def test_empty_and_single_primitive_scenes(ctx, oracle):
    cam = capi.make_camera([0, -5, 1], [0, 1, -0.1], res=(64, 48))
    lights = np.zeros(1, S.LIGHT_DT); lights["pos"] = [2, -3, 4]; lights["intensity"] = 30.0
    ctx.upload(capi.Scene.create(S.new_prims(0), lights))
    ids, dist = ctx.trace_primary(cam)
    assert (ids == -1).all() and np.isinf(dist).all()
    assert not ctx.render(cam, max_scatters=2).any()
    one = S.new_prims(1); one["kind"] = S.SPHERE; one["q"][0, :6] = [0, 0, 0.5, 1, 1.5, 0.7]
    ctx.upload(capi.Scene.create(one, lights))
    osc = oracle.scene(one, lights)
    want_ids, want_dist = oracle.trace_primary(osc, ocamera(cam))
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, want_ids) and np.array_equal(bits(dist), bits(want_dist))
    want_rgb, _, _ = oracle.render(osc, ocamera(cam), max_scatters=2)
    assert_rgb_close(ctx.render(cam, max_scatters=2), want_rgb, max_frac=1e-3)


def test_counting_kernels_match_the_cpu_walk(ctx, oracle):
    """Node / primitive fetch counts (V_n, V_p of the roofline accounting) from the instrumented kernels equal the
    CPU restatement's walk of the same flattened BVH, ray class by ray class."""
    prims = S.mixed_field(n_side=30, seed=77)
    lights = S.mixed_field_lights(half_extent=15.0, height=8.0, intensity=60.0)
    cam = capi.make_camera(**S.mixed_field_camera(half_extent=15.0, res=(200, 120)))
    for max_leaf in (1, 2):
        sc = capi.Scene.create(prims, lights, max_leaf=max_leaf)
        ctx.upload(sc)
        ctx.set_counting(True)
        ctx.set_entry_search(False)            # the CPU restatement starts every walk at the root
        ctx.reset_counters()
        rgb, ids = ctx.render(cam, max_scatters=0, want_ids=True)
        c = ctx.counters()
        want_ids, want_rgb, counts, hits = oracle.bvh_render(oracle.scene(prims, lights), oracle_bvh(oracle, sc), ocamera(cam))
        assert np.array_equal(ids, want_ids)
        assert_rgb_close(rgb, want_rgb, max_frac=1e-3)
        assert c["primary_hits"] == hits and c["shadow_rays"] == hits * 4
        assert [c["nodes_primary"], c["prims_primary"], c["nodes_shadow"], c["prims_shadow"]] == list(counts)
        # walks that start where the tile's beam allows: the same frame from fewer boxes (the search's own included)
        ctx.set_entry_search(True)
        ctx.reset_counters()
        rgb_e, ids_e = ctx.render(cam, max_scatters=0, want_ids=True)
        ce = ctx.counters()
        ctx.set_counting(False)
        assert np.array_equal(ids_e, ids) and np.array_equal(rgb_e, rgb)
        assert ce["nodes_primary"] < c["nodes_primary"] and ce["nodes_shadow"] < c["nodes_shadow"]
        # the entry node puts subtrees of different levels side by side, so a ray may meet them in another order than
        # the tree's own nesting gives: a few primitive tests more or fewer, never a different result
        assert ce["prims_primary"] <= 1.01 * c["prims_primary"] and ce["prims_shadow"] <= 1.01 * c["prims_shadow"]
        assert ce["primary_hits"] == hits and ce["shadow_rays"] == hits * 4
        # the plain kernels give the same picture and leave the fetch counters alone
        ctx.reset_counters()
        rgb2 = ctx.render(cam, max_scatters=0)
        c2 = ctx.counters()
        assert np.array_equal(rgb, rgb2) and c2["nodes_primary"] == 0 and c2["primary_hits"] == hits


def test_tiles_and_subrectangles(ctx):
    """Multi-GPU partition exercised on one GPU with virtual ranks: rendering every rank's tiles and assembling
    them reproduces the single-launch frame byte for byte; so does a sub-rectangle."""
    import torch
    prims, lights, camd = S.random_scene(80, 5, lights=2)
    camd = dict(camd, res=(200, 136))          # not a multiple of the tile size
    cam = capi.make_camera(**camd)
    ctx.upload(capi.Scene.create(prims, lights))
    W, H = camd["res"]
    for ms in (0, 1):
        full = ctx.render(cam, max_scatters=ms)
        for tile, n_ranks in ((16, 1), (32, 2), (64, 3), (48, 8)):
            tiling = capi.Tiling(W, H, tile, n_ranks)
            per = capi.tiles_per_rank(tiling) * tile * tile * 3
            assert capi.tile_count(tiling) == ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
            gathered = torch.full((n_ranks * per,), 7, dtype=torch.uint8, device="cuda")
            for r in range(n_ranks):
                ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, ms, 0.0)
            frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
            ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
            ctx.synchronize()
            assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), full), (tile, n_ranks)
            # the host restatement of the layout (what the gloo tests exchange) is what the kernels produce
            from raytracer_amd import tiles as T
            g_host = gathered.cpu().numpy()
            for r in range(n_ranks):
                assert np.array_equal(g_host[r * per:(r + 1) * per], T.pack_rank_tiles(full, tile, n_ranks, r)), (tile, n_ranks, r)
            assert np.array_equal(T.assemble(g_host, W, H, tile, n_ranks), full)
        x0, y0, w, h = 37, 21, 101, 77
        sub = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda")
        sid = torch.empty(h * w, dtype=torch.int32, device="cuda")
        ctx.render_rect_dev(cam, (x0, y0, w, h), sub.data_ptr(), sid.data_ptr(), ms, 0.0)
        ctx.synchronize()
        assert np.array_equal(sub.cpu().numpy().reshape(h, w, 3), full[y0:y0 + h, x0:x0 + w])
    ids, _ = ctx.trace_primary(cam)
    assert np.array_equal(sid.cpu().numpy().reshape(h, w), ids[y0:y0 + h, x0:x0 + w])


def test_gather_tiles_through_rccl_with_one_rank(ctx):
    """rt_gather_tiles is what a multi-GPU host calls per frame.  One GPU can only be a group of one: that still binds
    RCCL, creates a communicator, copies rank 0's own tiles and scatters them into the frame - the frame must be the
    single launch's.  Arguments that do not fit the communicator are refused."""
    import torch
    prims, lights, camd = S.random_scene(60, 15, lights=2)
    W, H = 328, 200
    cam = capi.make_camera(**dict(camd, res=(W, H)))
    ctx.upload(capi.Scene.create(prims, lights))
    full = ctx.render(cam, max_scatters=0)
    comm = capi.Comm(ctx, 1, 0, capi.comm_unique_id())
    tiling = capi.Tiling(W, H, 64, 1)
    local = torch.empty(capi.tiles_per_rank(tiling) * 64 * 64 * 3, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        ctx.render_tiles_dev(cam, tiling, 0, local.data_ptr(), 0, 0.0)
        comm.gather_tiles(tiling, local.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), full)
    with pytest.raises(capi.RtError) as e:
        comm.gather_tiles(capi.Tiling(W, H, 64, 2), local.data_ptr(), frame.data_ptr())     # two ranks asked of a group of one
    assert e.value.status == -1
    with pytest.raises(capi.RtError):
        comm.gather_tiles(tiling, local.data_ptr(), None)                                    # rank 0 without a frame
    with pytest.raises(capi.RtError):
        capi.Comm(ctx, 2, 2, capi.comm_unique_id())                                          # rank outside the group
    comm.close()
    # the form for one process that drives several GPUs, with the one GPU there is
    import ctypes as C
    ctxs = (C.c_void_p * 1)(ctx.h)
    comms = (C.c_void_p * 1)()
    assert capi.lib().rt_comm_create_all(ctxs, 1, comms) == 0
    bufs = (C.c_void_p * 1)(local.data_ptr())
    frame.zero_()
    assert capi.lib().rt_gather_tiles_all(comms, 1, tiling, bufs, frame.data_ptr()) == 0
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), full)
    # ... and the whole frame for a host-memory caller (what the C++ mirror calls under RT_GPUS)
    host = np.zeros((H, W, 3), dtype=np.uint8)
    assert capi.lib().rt_render_frame_multi_host(comms, 1, C.byref(cam), 64, 0, 0.0, host.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(host, full)
    twice = (C.c_void_p * 2)(ctx.h, ctx.h)
    two = (C.c_void_p * 2)()
    assert capi.lib().rt_comm_create_all(twice, 2, two) == -1                                # one GPU cannot be two ranks
    assert capi.lib().rt_comm_destroy(comms[0]) == 0


def test_callers_stream_and_device_pointers(ctx):
    import torch
    prims, lights, camd = S.random_scene(30, 8, lights=1)
    cam = capi.make_camera(**camd)
    ctx.upload(capi.Scene.create(prims, lights))
    W, H = camd["res"]
    want_ids, want_dist = ctx.trace_primary(cam)
    s = torch.cuda.Stream()
    try:
        ctx.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            ids = torch.empty(H * W, dtype=torch.int32, device="cuda")
            dist = torch.empty(H * W, dtype=torch.float64, device="cuda")
            ctx.trace_primary_dev(cam, (0, 0, W, H), ids.data_ptr(), dist.data_ptr())
            rays = torch.from_numpy(capi.pixel_rays(cam, [[10.5, 20.5], [80.5, 45.5]])).cuda()
            rid = torch.empty(2, dtype=torch.int32, device="cuda")
            rgb = torch.empty(6, dtype=torch.uint8, device="cuda")
            ctx.trace_rays_dev(rays.data_ptr(), 2, rid.data_ptr(), None)
            ctx.shade_rays_dev(rays.data_ptr(), 2, rgb.data_ptr())
        s.synchronize()
        assert np.array_equal(ids.cpu().numpy().reshape(H, W), want_ids)
        assert np.array_equal(bits(dist.cpu().numpy().reshape(H, W)), bits(want_dist))
        assert rid.cpu().tolist() == [int(want_ids[20, 10]), int(want_ids[45, 80])]
    finally:
        ctx.use_own_stream()


def test_shade_rays_against_oracle(ctx, oracle):
    g = np.load(os.path.join(GOLDEN, "shipped_Test2.npz"))
    import torch
    sc = capi.Scene.load_json(os.path.join(SCENES, "Test2.json"))
    cam, _ = sc.camera("142")
    ctx.upload(sc)
    ys, xs = np.mgrid[0:270:3, 0:480:3]
    rays = capi.pixel_rays(cam, np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
    d_rays = torch.from_numpy(rays).cuda()
    rgb = torch.empty(len(rays) * 3, dtype=torch.uint8, device="cuda")
    ctx.set_stream(0)
    try:
        ctx.shade_rays_dev(d_rays.data_ptr(), len(rays), rgb.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    assert_rgb_close(rgb.cpu().numpy().reshape(-1, 3)[None], g["shade_linear"][::3, ::3].reshape(-1, 3)[None], max_frac=1e-3)


def test_error_statuses_on_device():
    c = capi.Context(0)
    cam = capi.make_camera([0, -5, 1], [0, 1, 0], res=(32, 32))
    with pytest.raises(capi.RtError) as e:
        c.trace_primary(cam)
    assert e.value.status == -6
    c.upload(capi.Scene.create(S.random_scene(5, 1)[0], S.random_scene(5, 1)[1]))
    import torch
    buf = torch.empty(32 * 32 * 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.RtError) as e:
        c.render_rect_dev(cam, (0, 0, 33, 32), buf.data_ptr())
    assert e.value.status == -1
    with pytest.raises(capi.RtError) as e:
        c.render_rect_dev(cam, (0, 0, 32, 32), buf.data_ptr(), None, 9)
    assert e.value.status == -7
    with pytest.raises(capi.RtError) as e:
        c.render_tiles_dev(cam, capi.Tiling(32, 32, 24, 2), 0, buf.data_ptr())
    assert e.value.status == -1
    c.close()


# ---- whole hit records, shading of supplied hits, multi-sample frames ------------------------------------

@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_full_hit_records_against_reference(ctx, name, cam_id):
    """The 16-double records against the real reference's Hit fields on every 31st primary ray."""
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    cam, _ = sc.camera(cam_id)
    ctx.upload(sc)
    idx = g["rec_idx"]
    pxy = np.stack([(idx % cam.width) + 0.5, (idx // cam.width) + 0.5], axis=1)
    rays = capi.pixel_rays(cam, pxy)
    rec = ctx.trace_rays_full(rays)
    want_ids = g["ids"].reshape(-1)[idx].astype(np.int32)
    hit = want_ids >= 0
    assert np.array_equal(rec[:, 0] != 0, hit)
    assert np.array_equal(rec[:, 1].astype(np.int32), want_ids)
    assert np.all(rec[~hit, 3:] == 0)
    assert np.array_equal(bits(rec[hit, 3:6]), bits(g["rec_p"][hit]))
    assert np.array_equal(bits(rec[hit, 6:9]), bits(g["rec_n"][hit]))
    assert np.array_equal(bits(rec[hit, 9:12]), bits(g["rec_r"][hit]))
    # u, v go through atan2 / asin, whose device versions may differ from libm in the last bits
    assert np.abs(rec[hit, 12] - g["rec_u"][hit]).max() <= 1e-12
    assert np.abs(rec[hit, 13] - g["rec_v"][hit]).max() <= 1e-12
    assert np.all(rec[:, 14:] == 0)


@pytest.mark.parametrize("name,cam_id", [s for s in SHIPPED if s[0] != "200_rand_1_cam"])
def test_shade_hits_matches_reference_shade(ctx, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    cam, _ = sc.camera(cam_id)
    ctx.upload(sc)
    idx = np.arange(0, cam.width * cam.height, 7)
    pxy = np.stack([(idx % cam.width) + 0.5, (idx // cam.width) + 0.5], axis=1)
    rays = capi.pixel_rays(cam, pxy)
    rec = ctx.trace_rays_full(rays)
    hits10 = np.zeros((len(idx), 10))
    hits10[:, 0:6] = rec[:, 3:9]
    hits10[:, 6] = rec[:, 1]
    hits10[:, 7:9] = rec[:, 12:14]
    hits10[:, 9] = 1.0
    rgb = ctx.shade_hits(hits10, rays[:, 0:3])
    want = g["shade_linear"].reshape(-1, 3)[idx]
    assert_rgb_close(rgb, want, max_frac=1e-3)


def jitter_offsets(seed, s, xs, ys):
    """numpy restatement of the per-pixel sub-pixel offset of sample s (s >= 1)."""
    M = np.uint64(0xFFFFFFFF)

    def mix(h):
        h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x7FEB352D)) & M
        h = h ^ (h >> np.uint64(15)); h = (h * np.uint64(0x846CA68B)) & M
        return h ^ (h >> np.uint64(16))
    js = ((((seed ^ 0xA511E9B3) * 0x9E3779B1) & 0xFFFFFFFF) + s * 0x85EBCA6B) & 0xFFFFFFFF
    js = js or 1
    x = xs.astype(np.uint64); y = ys.astype(np.uint64)
    h = ((np.uint64(js) * np.uint64(0x9E3779B1)) & M) ^ ((x * np.uint64(0x85EBCA77)) & M) ^ ((y * np.uint64(0xC2B2AE3D)) & M)
    h = mix(h)
    gg = mix((h * np.uint64(0x9E3779B1) + np.uint64(0x7F4A7C15)) & M)
    return h.astype(np.float64) / 4294967296.0 - 0.5, gg.astype(np.float64) / 4294967296.0 - 0.5


def test_multi_sample_frame(ctx, oracle):
    g = np.load(os.path.join(GOLDEN, "synthetic_b.npz"))
    prims, lights, camd = S.random_scene(int(g["n"]), int(g["seed"]), lights=int(g["n_lights"]))
    sc = capi.Scene.create(prims, lights)
    cam = capi.make_camera(**camd)
    ctx.upload(sc)
    assert np.array_equal(ctx.render_samples(cam, 1, seed=5, max_scatters=0), ctx.render(cam, max_scatters=0))
    spp, seed = 3, 77
    got = ctx.render_samples(cam, spp, seed=seed, max_scatters=0)
    again = ctx.render_samples(cam, spp, seed=seed, max_scatters=0)
    assert np.array_equal(got, again)                       # a frame is a pure function of the seed
    assert not np.array_equal(got, ctx.render_samples(cam, spp, seed=seed + 1, max_scatters=0))
    W, H = cam.width, cam.height
    ys, xs = np.divmod(np.arange(W * H), W)
    osc = oracle.scene(prims, lights)
    ocam = oracle.camera(camd["location"], camd["direction"], None, camd["focal"], camd["sensor"], camd["res"])
    acc = np.zeros((W * H, 3))
    for s in range(spp):
        jx, jy = (0.0, 0.0) if s == 0 else jitter_offsets(seed, s, xs, ys)
        rays = oracle.pixel_rays(ocam, np.stack([xs + 0.5 + jx, ys + 0.5 + jy], axis=1))
        acc += oracle.shade_rays(osc, rays)
    lut = oracle.gamma_lut().astype(np.int16)
    want_lin = np.clip(np.round(acc / spp), 0, 255).astype(np.int64).reshape(H, W, 3)
    # Compared as LINEAR means: every sample's linear colour may be off by one level, so the mean may be off by one
    # level before the display transfer; the transfer is monotone, so the frame must lie between the encodings of
    # mean - 1 and mean + 1 (on dark pixels, where the curve is steep, that is several display levels; elsewhere none).
    lo, hi = lut[np.clip(want_lin - 1, 0, 255)], lut[np.clip(want_lin + 1, 0, 255)]
    g16 = got.astype(np.int16)
    assert ((g16 >= lo) & (g16 <= hi)).all()
    assert (g16 != lut[want_lin]).any(-1).mean() < 2e-3


@pytest.mark.parametrize("kind", [S.CUBE, S.PLANE, S.CYLINDER, S.SPHERE])
def test_single_primitive_intersect_against_reference(ctx, kind):
    """Mesh::intersect of lone primitives built through the reference's public constructors: random rays plus
    origins inside, axis-parallel and zero directions, zero scales and degenerate corners."""
    g = np.load(os.path.join(GOLDEN, "single_prims.npz"))
    k = S.KIND_NAMES[kind].lower()
    for t in range(len(g[k + "_params"])):
        prim = np.zeros(1, dtype=S.PRIM_DT)
        prim["kind"] = kind
        prim["q"][0] = g[k + "_params"][t]
        prim["alpha"] = 1.0; prim["roughness"] = 0.5; prim["ior"] = 1.5; prim["albedo"] = 1.0
        rec = ctx.intersect_prim(prim, g[k + "_rays"][t])
        assert np.array_equal(rec[:, 0] != 0, g[k + "_hit"][t] == 1), (k, t)
        m = rec[:, 0] != 0
        for f, cols in (("dist", slice(2, 3)), ("p", slice(3, 6)), ("n", slice(6, 9)), ("r", slice(9, 12))):
            a, b = rec[m, cols].reshape(-1), np.asarray(g[k + "_" + f][t][m], dtype=np.float64).reshape(-1)
            same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (k, t, f)
        for f, col in (("u", 12), ("v", 13)):
            a, b = rec[m, col], g[k + "_" + f][t][m]
            assert (np.isnan(a) == np.isnan(b)).all() and np.nanmax(np.abs(a - b), initial=0.0) <= 1e-12, (k, t, f)
        assert np.all(rec[~m, 1] == -1) and np.all(np.isinf(rec[~m, 2]))


def test_textured_scene(ctx, tmp_path):
    """Albedo textures end to end: scene file with texture paths -> library loader -> texel pool on the GPU -> the
    albedo under each hit's (u, v), against the reference's own renders of the same files."""
    g = np.load(os.path.join(GOLDEN, "textured.npz"))
    prims, lights, camd, textures, prim_tex, paths = S.textured_scene(str(tmp_path))
    jp = str(tmp_path / "scene.json")
    S.scene_to_json(prims, lights, {"3": camd}, jp, texture_paths=paths)
    from_file = capi.Scene.load_json(jp)
    in_memory = capi.Scene.create(prims, lights)
    in_memory.set_textures(textures, prim_tex)
    cam = capi.make_camera(**camd)
    W, H = camd["res"]
    for sc in (from_file, in_memory):
        ctx.upload(sc)
        ids, dist = ctx.trace_primary(cam)
        assert np.array_equal(ids, g["ids"].astype(np.int32))
        assert np.array_equal(bits(dist), bits(g["dist"]))
        for ms in (0, 2):
            assert_rgb_close(ctx.render(cam, max_scatters=ms), g["render%d" % ms], max_frac=1e-3)
    ys, xs = np.divmod(np.arange(W * H), W)
    rays = capi.pixel_rays(cam, np.stack([xs + 0.5, ys + 0.5], axis=1))
    rec = ctx.trace_rays_full(rays)
    hit = rec[:, 0] != 0
    assert np.abs(rec[hit, 12] - g["u"].reshape(-1)[hit]).max() <= 1e-12
    assert np.abs(rec[hit, 13] - g["v"].reshape(-1)[hit]).max() <= 1e-12
    # RayTracer::shade through both entry points: whole rays, and hits handed back in
    import torch
    d_rays = torch.from_numpy(rays).cuda()
    d_rgb = torch.empty((len(rays), 3), dtype=torch.uint8, device="cuda")
    ctx.shade_rays_dev(d_rays.data_ptr(), len(rays), d_rgb.data_ptr())
    ctx.synchronize()
    assert_rgb_close(d_rgb.cpu().numpy().reshape(H, W, 3), g["shade_linear"], max_frac=1e-3)
    hits10 = np.zeros((len(rays), 10))
    hits10[:, 0:6] = rec[:, 3:9]; hits10[:, 6] = rec[:, 1]; hits10[:, 7:9] = rec[:, 12:14]; hits10[:, 9] = 1.0
    assert_rgb_close(ctx.shade_hits(hits10, rays[:, 0:3]).reshape(H, W, 3), g["shade_linear"], max_frac=1e-3)
    # a Hit without surface coordinates falls back to the solid colour (mesh.cpp:23): differs from the textured frame
    hits10[:, 9] = 0.0
    assert not np.array_equal(ctx.shade_hits(hits10, rays[:, 0:3]).reshape(H, W, 3), g["shade_linear"])
    # tiles of a textured frame (the multi-GPU partition) give the same pixels
    full = ctx.render(cam, max_scatters=0)
    tiling = capi.Tiling(W, H, 32, 3)
    per = capi.tiles_per_rank(tiling) * 32 * 32 * 3
    gathered = torch.zeros(3 * per, dtype=torch.uint8, device="cuda")
    for r in range(3):
        ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, 0, 0.0)
    frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
    ctx.synchronize()
    assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), full)


def test_frame_replays_from_a_hip_graph(ctx):
    """After one warm-up frame (which sizes the context's working set) a frame is memsets and launches only, so it can
    be captured on the caller's stream and replayed; the replayed frames equal the directly rendered ones."""
    import torch
    prims, lights, camd = S.random_scene(60, 21, lights=3)
    cam = capi.make_camera(**dict(camd, res=(256, 144)))
    ctx.upload(capi.Scene.create(prims, lights))
    want0 = ctx.render(cam, max_scatters=0)
    want2 = ctx.render(cam, max_scatters=2)
    W, H = 256, 144
    out0 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    out2 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            ctx.render_rect_dev(cam, (0, 0, W, H), out0.data_ptr(), None, 0, 0.0)       # warm-up on this stream
            ctx.render_rect_dev(cam, (0, 0, W, H), out2.data_ptr(), None, 2, 0.0)
            side.synchronize()
            out0.zero_(); out2.zero_()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                ctx.render_rect_dev(cam, (0, 0, W, H), out0.data_ptr(), None, 0, 0.0)
                ctx.render_rect_dev(cam, (0, 0, W, H), out2.data_ptr(), None, 2, 0.0)
            assert not out0.any().item()                     # capture launched nothing
            for _ in range(2):
                graph.replay()
                torch.cuda.synchronize()
                assert np.array_equal(out0.cpu().numpy().reshape(H, W, 3), want0)
                assert np.array_equal(out2.cpu().numpy().reshape(H, W, 3), want2)
                out0.zero_(); out2.zero_()
                torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()


def test_more_lights_than_one_shadow_launch_holds(ctx, oracle):
    """Eleven lights: two shadow launches (8 + 3) with the running light sum parked in between, plain and with bounces,
    in rectangle and in tile mode.  The oracle sums the lights in the same (file) order."""
    import torch
    prims, lights, camd = S.random_scene(50, 91, lights=11)
    lights["intensity"] *= 0.25
    cam = capi.make_camera(**dict(camd, res=(192, 108)))
    ctx.upload(capi.Scene.create(prims, lights))
    osc = oracle.scene(prims, lights)
    ctx.reset_counters()
    for ms in (0, 2):
        want, _, _ = oracle.render(osc, ocamera(cam), max_scatters=ms, threads=8)
        got = ctx.render(cam, max_scatters=ms)
        assert_rgb_close(got, want, max_frac=2e-3)
        tiling = capi.Tiling(192, 108, 32, 2)
        per = capi.tiles_per_rank(tiling) * 32 * 32 * 3
        gathered = torch.empty(2 * per, dtype=torch.uint8, device="cuda")
        for r in range(2):
            ctx.render_tiles_dev(cam, tiling, r, gathered.data_ptr() + r * per, ms, 0.0)
        frame = torch.empty(108 * 192 * 3, dtype=torch.uint8, device="cuda")
        ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
        ctx.synchronize()
        assert np.array_equal(frame.cpu().numpy().reshape(108, 192, 3), got)
    c = ctx.counters()
    assert c["shadow_rays"] % 11 == 0


def check_tree(nodes, slots, n_slots):
    """Structural invariants of a flattened hierarchy: every slot in exactly one leaf, sibling pairs at even indices,
    parents contain children, node 1 unused.  Returns the depth."""
    seen = np.zeros(n_slots, dtype=np.int32)
    depth, stack, visited = 0, [(0, 0)], 0
    while stack:
        k, d = stack.pop()
        visited += 1
        depth = max(depth, d)
        first, count = int(nodes["first"][k]), int(nodes["count"][k])
        if count:
            seen[first:first + count] += 1
        else:
            assert first % 2 == 0 and first >= 2 and first + 1 < len(nodes)
            for c in (first, first + 1):
                assert (nodes["lo"][c] >= nodes["lo"][k]).all() and (nodes["hi"][c] <= nodes["hi"][k]).all()
                stack.append((c, d + 1))
    assert (seen == 1).all() and visited + 1 == len(nodes)
    return depth


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_hierarchy_built_on_the_device(ctx, oracle, name, cam_id):
    """RT_BUILD_ON_DEVICE: Morton sort and top-down split on the GPU.  A different tree, the same answers: ids and FP64
    distances against the reference's, frames against its renders; the tree itself is downloaded and checked, and the
    counting kernels agree with the CPU walk of that downloaded tree."""
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"), on_device=True)
    info = sc.info()
    cam, _ = sc.camera(cam_id)
    ctx.upload(sc)
    assert ctx.build_time_ms() > 0.0
    nodes, slots = ctx.download_bvh()
    assert len(nodes) == info.n_nodes == 2 * info.n_slots and len(slots) == info.n_slots + info.n_loose
    assert sorted(slots.tolist()) == sorted(set(slots.tolist()))
    assert check_tree(nodes, slots, info.n_slots) <= info.depth <= 31       # the host knows a bound, the data decide
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert np.array_equal(bits(dist[::4]), bits(g["dist_rows"]))
    for ms in (0, 2):
        assert_rgb_close(ctx.render(cam, max_scatters=ms), g["render%d" % ms])
    if "sec_rays" in g:
        sid, sdist = ctx.trace_rays(g["sec_rays"])
        assert np.array_equal(sid, g["sec_ids"].astype(np.int32)) and np.array_equal(bits(sdist), bits(g["sec_dist"]))
    # node / primitive fetch counts of the device-built tree: GPU counting kernels == CPU walk of the downloaded tree
    prims, lights = sc.prims(), sc.lights()
    radius = float(np.nextafter(np.float32(info.radius), np.float32(np.inf)))
    wide = ctx.download_wide_bvh()
    if info.n_slots >= 2:
        assert len(wide) == 2 * info.n_slots                # the builder wrote wide nodes beside the binary ones
    b = oracle.bvh(nodes, slots, info.n_slots, info.n_loose, radius, wide, sc.loose_boxes())
    _, _, counts, hits = oracle.bvh_render(oracle.scene(prims, lights), b, ocamera(cam))
    ctx.set_counting(True); ctx.set_entry_search(False); ctx.reset_counters()     # the CPU walk starts at the root
    ctx.render(cam, max_scatters=0)
    c = ctx.counters()
    ctx.set_counting(False); ctx.set_entry_search(True)
    if len(lights):
        assert [c["nodes_primary"], c["prims_primary"], c["nodes_shadow"], c["prims_shadow"]] == counts.tolist()
        assert c["primary_hits"] == hits


def test_device_build_edge_sizes(ctx, oracle):
    """0, 1, 2, 3, 5 and 64 primitives, unbounded ones mixed in, and coincident primitives (equal curve keys)."""
    cam = capi.make_camera([0.3, -7, 2.5], [0, 1, -0.3], res=(96, 64))
    lights = np.zeros(1, S.LIGHT_DT); lights["pos"] = [2, -3, 6]; lights["intensity"] = 40.0
    for n in (0, 1, 2, 3, 5, 64):
        prims, _, _ = S.random_scene(max(n, 1), 300 + n, lights=1)
        prims = prims[:n]
        if n >= 3:
            extra = S.new_prims(1); extra["kind"] = S.SPHERE; extra["q"][0, :6] = [0.5, 0.5, 1.0, 0.7, 0.0, 0.6]   # zero scale: unbounded
            prims = S.sort_reference_order(np.concatenate([prims, extra]))
        sc = capi.Scene.create(prims, lights, on_device=True)
        ctx.upload(sc)
        info = sc.info()
        nodes, slots = ctx.download_bvh()
        assert len(nodes) == 2 * info.n_slots
        if info.n_slots:
            assert check_tree(nodes, slots, info.n_slots) <= info.depth
        want_ids, want_dist = oracle.trace_primary(oracle.scene(prims, lights), ocamera(cam))
        ids, dist = ctx.trace_primary(cam)
        assert np.array_equal(ids, want_ids) and np.array_equal(bits(dist), bits(want_dist)), n
        want_rgb, _, _ = oracle.render(oracle.scene(prims, lights), ocamera(cam), max_scatters=1)
        assert_rgb_close(ctx.render(cam, max_scatters=1), want_rgb, max_frac=2e-3)
    # 37 shells around one centre (one curve key for all of them: the range can only be halved) among 30 others
    shells = S.new_prims(37); shells["kind"] = S.SPHERE
    shells["q"][:, :3] = [0.4, 0.3, 1.1]; shells["q"][:, 3:6] = (0.2 + 0.05 * np.arange(37))[:, None]
    others, _, _ = S.random_scene(30, 77, lights=1)
    prims = S.sort_reference_order(np.concatenate([shells, others]))
    sc = capi.Scene.create(prims, lights, on_device=True)
    ctx.upload(sc)
    nodes, slots = ctx.download_bvh()
    assert check_tree(nodes, slots, sc.info().n_slots) <= sc.info().depth
    want_ids, want_dist = oracle.trace_primary(oracle.scene(prims, lights), ocamera(cam))
    ids, dist = ctx.trace_primary(cam)
    assert np.array_equal(ids, want_ids) and np.array_equal(bits(dist), bits(want_dist))
    with pytest.raises(capi.RtError):
        capi.Scene.create(prims, lights, on_device=True).bvh()           # nothing on the host to hand out


@pytest.mark.parametrize("seed", list(range(400, 412)))
def test_random_scenes_against_brute_force(ctx, oracle, seed):
    """A dozen more seeded scenes of all five kinds, some with duplicated, degenerate and scene-spanning primitives, both
    host- and device-built hierarchies: ids and FP64 distances equal brute force bit for bit, colours within one level
    for 0 and 2 bounces, arbitrary rays (unnormalised, axis-parallel, starting inside primitives) included."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 90))
    prims, lights, camd = S.random_scene(n, seed, kinds=(S.CUBE, S.PLANE, S.CYLINDER, S.SPHERE, S.TRIANGLE), lights=int(rng.integers(1, 4)))
    extra = []
    if seed % 2 == 0:                                   # a copy of some primitive: equal distances, lowest id wins
        extra.append(prims[int(rng.integers(0, n))].copy())
    if seed % 3 == 0:                                   # a floor under everything
        g = S.new_prims(1); g["kind"] = S.PLANE; g["q"][0, :12] = [-9, -9, -3.5, 9, -9, -3.5, 9, 9, -3.5, -9, 9, -3.5]
        g["metallic"] = 0.5
        extra.append(g[0])
    if seed % 4 == 0:                                   # a zero scale: unbounded, tested against every ray
        z = S.new_prims(1); z["kind"] = S.CYLINDER; z["q"][0, :9] = [0.3, -0.4, 0.2, 0.5, 0.1, -0.7, 0.6, 0.0, 0.9]
        extra.append(z[0])
    if extra:
        prims = S.sort_reference_order(np.concatenate([prims, np.array(extra, dtype=prims.dtype)]))
    camd = dict(camd, res=(112, 63))
    cam = capi.make_camera(**camd)
    osc = oracle.scene(prims, lights)
    want_ids, want_dist = oracle.trace_primary(osc, ocamera(cam))
    want = {ms: oracle.render(osc, ocamera(cam), max_scatters=ms)[0] for ms in (0, 2)}
    o = rng.uniform(-4, 4, size=(3000, 3))
    d = rng.normal(size=(3000, 3)) * rng.uniform(0.01, 30, size=(3000, 1))
    d[:30, 0] = 0.0
    d[30:60, 1:] = 0.0
    o[60:200] = prims["q"][rng.integers(0, len(prims), 140), :3] + rng.normal(size=(140, 3)) * 0.05      # from inside / on things
    rays = np.concatenate([o, d], axis=1)
    a = oracle.trace_rays(osc, rays)
    for on_device in (False, True):
        ctx.upload(capi.Scene.create(prims, lights, on_device=on_device, max_leaf=1 + seed % 3))
        ids, dist = ctx.trace_primary(cam)
        assert np.array_equal(ids, want_ids) and np.array_equal(bits(dist), bits(want_dist)), on_device
        for ms in (0, 2):
            assert_rgb_close(ctx.render(cam, max_scatters=ms), want[ms], max_frac=2e-3)
        rid, rdist = ctx.trace_rays(rays)
        assert np.array_equal(rid, a["id"]) and np.array_equal(bits(rdist), bits(a["dist"])), on_device


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 17, 101, 2024])
def test_shadow_verdicts_where_a_ray_leaves_its_own_primitive(ctx, oracle, seed):
    """The shadow pass does not test a ray against the primitive it starts on when that test is known to miss.  Verdict
    bits must still equal the every-primitive loop's (which tests everything, the ray's own primitive included) where that
    rule is under strain: scales from 1e-3 to 1e3 and of either sign, lights inside solids, lights in a primitive's own
    plane and grazing its surface, the camera inside a solid, coincident primitives."""
    import torch
    rng = np.random.default_rng(seed)
    prims, lights, camd = S.random_scene(60, seed, kinds=(S.CUBE, S.PLANE, S.CYLINDER, S.SPHERE, S.TRIANGLE), lights=6)
    prims = prims.copy()
    for i in range(len(prims)):
        k = int(prims["kind"][i])
        if k in (S.CUBE, S.CYLINDER, S.SPHERE):
            sl = slice(6, 9) if k != S.SPHERE else slice(3, 6)
            r = rng.uniform()
            if r < 0.15:
                prims["q"][i, sl] *= rng.choice([1e-3, 1e3]) if seed % 2 else 1.0           # tiny / huge
            elif r < 0.3:
                prims["q"][i, sl] *= rng.choice([-1.0, 1.0], size=3)                         # mirrored axes
            elif r < 0.35:
                prims["q"][i, sl.start] = 0.0                                               # flattened: kept beside the tree
    solids = [i for i in range(len(prims)) if int(prims["kind"][i]) in (S.CUBE, S.CYLINDER, S.SPHERE)]
    flats = [i for i in range(len(prims)) if int(prims["kind"][i]) in (S.PLANE, S.TRIANGLE)]
    lights = lights.copy()
    lights["pos"][0] = prims["q"][solids[0], 0:3]                                           # inside a solid
    lights["pos"][1] = prims["q"][solids[1], 0:3] + 1e-3
    q = prims["q"][flats[0]]
    lights["pos"][2] = q[0:3] + 0.3 * (q[3:6] - q[0:3]) + 0.6 * (q[6:9] - q[0:3])           # in a flat primitive's plane
    lights["pos"][3] = q[0:3] + 5.0 * (q[3:6] - q[0:3])                                     # in that plane, far outside
    twin = S.new_prims(2); twin[0] = prims[solids[2]]; twin[1] = prims[flats[1]]            # coincident copies
    inside_at = prims["q"][solids[3], 0:3] + 1e-2                                            # a camera inside a solid
    prims = S.sort_reference_order(np.concatenate([prims, twin]))
    for where in ("outside", "inside"):
        cd = dict(camd, res=(128, 72))
        if where == "inside":
            cd = dict(cd, location=list(inside_at))
        cam = capi.make_camera(**cd)
        ctx.upload(capi.Scene.create(prims, lights))
        W, H = cam.width, cam.height
        mask = torch.empty(H * W, dtype=torch.uint8, device="cuda")
        ctx.trace_shadows_dev(cam, (0, 0, W, H), mask.data_ptr()); ctx.synchronize()
        got = mask.cpu().numpy().reshape(H, W)
        ys, xs = np.mgrid[0:H, 0:W]
        rays = oracle.pixel_rays(ocamera(cam), np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1))
        hits, lit, rgb = oracle.brute_shade_rays(oracle.scene(prims, lights), rays)
        want = (lit.astype(np.uint32) << np.arange(lit.shape[1], dtype=np.uint32)).sum(1).astype(np.uint8).reshape(H, W)
        want[hits["id"].reshape(H, W) < 0] = 0
        assert (hits["id"] >= 0).mean() > 0.05
        assert np.array_equal(got, want), "%d of %d verdict bytes differ (%s)" % ((got != want).sum(), got.size, where)
