"""Host-side half of the product (no GPU needed): the C ABI loads and exports everything include/rt_hip.h
declares, the scene loader agrees with an independent reader, the flattened BVH is well formed, and walking
it with the product's traversal rules (restated in the oracle) gives exactly the brute-force answers."""
import ctypes as C
import os
import re
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SCENES, SHIPPED
from oracle import bind
from raytracer_amd import capi
from raytracer_amd import scenes as S


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    raw = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert declared == {s[0] for s in capi.SYMBOLS}
    assert capi.lib().rt_abi_version() == 1
    assert capi.lib().rt_strerror(-6) == b"no scene uploaded"


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU the device side refuses to start instead of computing elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.RtError) as e:
        capi.Context(0)
    assert e.value.status == -4


def test_error_statuses():
    with pytest.raises(capi.RtError) as e:
        capi.Scene.load_json("/nonexistent/scene.json")
    assert e.value.status == -2
    sc = capi.Scene.load_json(os.path.join(SCENES, "Test1.json"))
    with pytest.raises(capi.RtError) as e:
        sc.camera("999")
    assert e.value.status == -3
    with pytest.raises(capi.RtError) as e:
        capi.camera_from_json(os.path.join(SCENES, "Test1.json"), "999")
    assert e.value.status == -3
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        f.write('{"MESH": {"Sphere": {"1": {"location": {"x": 1e, "y": 0, "z": 0}}}}}')
    try:
        sc2 = capi.Scene.load_json(f.name)      # "1e" is accepted by strtod as 1 - like std::stod
        assert sc2.info().n_prims == 1
        open(f.name, "w").write('{"MESH": {"Sphere": {"1": {"location": {"x": --, "y": 0, "z": 0}}}}}')
        with pytest.raises(capi.RtError) as e:
            capi.Scene.load_json(f.name)
        assert e.value.status == -8
        open(f.name, "w").write("this is not a scene")
        assert capi.Scene.load_json(f.name).info().n_prims == 0   # load_from_json returns true even if nothing parses
    finally:
        os.unlink(f.name)


ALL_SCENE_FILES = [os.path.join(SCENES, n + ".json") for n, _ in SHIPPED] + \
                  sorted(os.path.join(SCENES, "export", f) for f in os.listdir(os.path.join(SCENES, "export")))


@pytest.mark.parametrize("path", ALL_SCENE_FILES, ids=[os.path.basename(p) for p in ALL_SCENE_FILES])
def test_scene_loader_matches_independent_reader(oracle, path):
    prims, lights, cams = bind.load_scene_json(path)
    sc = capi.Scene.load_json(path)
    assert sc.prims().tobytes() == prims.astype(S.PRIM_DT).tobytes()
    assert sc.lights().tobytes() == lights.astype(S.LIGHT_DT).tobytes()
    info = sc.info()
    assert info.n_cameras == len(cams)
    for i, (cid, c) in enumerate(cams.items()):
        cam, ident = sc.camera(index=i)
        assert ident == cid
        ocam = bind.camera_from_dict(oracle, c)
        assert bytes(cam) == bytes(ocam)
        cam2, _ = sc.camera(cid)
        assert bytes(cam2) == bytes(cam)


def test_generated_scene_roundtrip_and_camera_rays(oracle):
    prims, lights, camd = S.random_scene(60, 77, lights=3)
    camd = dict(camd, up=[0.1, 0.2, 1.0])
    with tempfile.TemporaryDirectory() as td:
        jp = os.path.join(td, "s.json")
        S.scene_to_json(prims, lights, {"12": camd}, jp)
        sc = capi.Scene.load_json(jp)
        assert sc.prims().tobytes() == prims.tobytes()
        cam, _ = sc.camera("12")
        ocam = bind.camera_from_dict(oracle, camd)
        assert bytes(cam) == bytes(ocam)
        # read_camera_from_json ignores "up"
        cam_wu = capi.camera_from_json(jp, "12")
        assert bytes(cam_wu) == bytes(bind.camera_from_dict(oracle, dict(camd, up=None)))
    rng = np.random.default_rng(1)
    pxy = rng.uniform(0, 160, size=(200, 2))
    assert np.array_equal(bits(capi.pixel_rays(cam, pxy)), bits(oracle.pixel_rays(ocam, pxy)))
    assert np.array_equal(capi.gamma_lut(), oracle.gamma_lut())


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_reference_aabbs(name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"))
    boxes = np.array([capi.prim_aabb(p) for p in sc.prims()])
    assert np.array_equal(bits(boxes), bits(g["aabbs"]))


def check_bvh(sc, oracle):
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    info = sc.info()
    prims = sc.prims()
    assert n_slots + n_loose <= len(prims)
    assert len(np.unique(slots)) == len(slots)
    if n_slots == 0:
        assert len(nodes) == 0
        return 0
    assert len(nodes) % 2 == 0
    seen = np.zeros(n_slots, dtype=int)
    max_depth = 0
    stack = [(0, 0)]
    while stack:
        i, d = stack.pop()
        max_depth = max(max_depth, d)
        nd = nodes[i]
        assert (nd["lo"] <= nd["hi"]).all()
        if nd["count"] > 0:
            for s in range(nd["first"], nd["first"] + nd["count"]):
                seen[s] += 1
                box = oracle.aabb(prims[slots[s]])
                p = prims[slots[s]]
                if p["kind"] == S.PLANE:      # reachable points: the parallelogram, not the four given corners
                    q = p["q"].reshape(4, 3)
                    pts = np.array([q[0], q[1], q[3], q[1] + q[3] - q[0]])
                    box = np.concatenate([pts.min(0), pts.max(0)])
                assert (nd["lo"].astype(np.float64) <= box[:3]).all() and (nd["hi"].astype(np.float64) >= box[3:]).all()
        else:
            c = int(nd["first"])
            assert c % 2 == 0 and c >= 2
            for k in (c, c + 1):
                ch = nodes[k]
                assert (ch["lo"] >= nd["lo"]).all() and (ch["hi"] <= nd["hi"]).all()
                stack.append((k, d + 1))
    assert (seen == 1).all()
    assert max_depth == info.depth and max_depth <= 31
    return max_depth


def check_wide(sc, may_be_dropped=False):
    """The wide nodes are the binary tree with up to two inner children opened per node: every binary leaf appears
    exactly once with its slot range and its box, every child box is one of the binary tree's, node 0 holds the root's
    children, and a binary walk and a wide walk reach the same leaves."""
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    wide = sc.wide_bvh()
    if len(nodes) == 0 or nodes[0]["count"] > 0:
        assert len(wide) == 0
        return
    if may_be_dropped and len(wide) == 0:      # a tree so lopsided that the wide walk could outgrow its stack: binary walk
        return
    assert len(wide) > 0
    leaf_box = {}
    stack = [0]
    while stack:
        nd = nodes[stack.pop()]
        if nd["count"]:
            leaf_box[(int(nd["first"]), int(nd["count"]))] = (nd["lo"].copy(), nd["hi"].copy())
        else:
            stack += [int(nd["first"]), int(nd["first"]) + 1]
    seen, visited, stack = {}, 0, [0]
    while stack:
        w = wide[stack.pop()]
        visited += 1
        kids = [k for k in range(4) if w["ref"][k] != 0xFFFFFFFF]
        assert len(kids) >= 2 and kids == list(range(len(kids)))          # filled from the front, never a lone child
        for k in kids:
            ref = int(w["ref"][k])
            if ref >> 28:
                key = (ref & 0x0FFFFFFF, ref >> 28)
                assert key not in seen
                seen[key] = True
                lo, hi = leaf_box[key]
                assert np.array_equal(w["lo"][:, k], lo) and np.array_equal(w["hi"][:, k], hi)
            else:
                assert 0 < ref < len(wide)
                stack.append(ref)
    assert visited == len(wide) and set(seen) == set(leaf_box)


@pytest.mark.parametrize("max_leaf", [1, 2, 4])
def test_bvh_structure(oracle, max_leaf):
    for name, _ in SHIPPED:
        sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"), max_leaf=max_leaf)
        check_bvh(sc, oracle)
        check_wide(sc)
    prims = S.mixed_field(n_side=40, seed=5)
    sc = capi.Scene.create(prims, max_leaf=max_leaf)
    check_wide(sc)
    d = check_bvh(sc, oracle)
    assert d < 31
    assert sc.info().n_prims == len(prims)


def test_bvh_depth_is_bounded_on_adversarial_input(oracle):
    """Exponentially spaced, exponentially sized spheres make SAH peel one primitive per level; the builder must
    fall back to balanced splits before the depth exceeds the traversal stack."""
    n = 3000
    p = S.new_prims(n)
    p["kind"] = S.SPHERE
    x = 1.0001 ** np.arange(n) * np.cumsum(1.5 ** (np.arange(n) % 40))
    p["q"][:, 0] = np.cumsum(2.0 ** (np.arange(n) / 40.0))
    p["q"][:, 3:6] = (2.0 ** (np.arange(n) / 40.0))[:, None] * 0.4
    sc = capi.Scene.create(p, max_leaf=1)
    assert check_bvh(sc, oracle) <= 31
    check_wide(sc, may_be_dropped=True)
    n = 70
    p = S.new_prims(n)
    p["kind"] = S.SPHERE
    p["q"][:, 0] = 3.0 ** np.arange(n)
    p["q"][:, 3:6] = (3.0 ** np.arange(n))[:, None] * 0.3
    sc = capi.Scene.create(p, max_leaf=1)
    assert check_bvh(sc, oracle) <= 31
    check_wide(sc, may_be_dropped=True)


def oracle_bvh(oracle, sc):
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    return oracle.bvh(nodes, slots, n_slots, n_loose, radius, sc.wide_bvh(), sc.loose_boxes())      # walked the way the kernels walk it


@pytest.mark.parametrize("name,cam_id", SHIPPED)
def test_bvh_walk_equals_brute_force_on_shipped_scenes(oracle, name, cam_id):
    g = np.load(os.path.join(GOLDEN, "shipped_%s.npz" % name))
    sc = capi.Scene.load_json(os.path.join(SCENES, name + ".json"), max_leaf=2)
    cam, _ = sc.camera(cam_id)
    ocam = bind.Camera.from_buffer_copy(bytes(cam))
    osc = oracle.scene(sc.prims(), sc.lights())
    ids, rgb, counts, hits = oracle.bvh_render(osc, oracle_bvh(oracle, sc), ocam)
    assert np.array_equal(ids, g["ids"].astype(np.int32))
    assert hits == int((g["ids"] >= 0).sum())
    assert np.array_equal(rgb, g["render0"])          # shadow verdicts through the BVH, then gamma
    assert counts[0] > 0 and counts[1] > 0
    if "sec_rays" in g.files:
        h, _ = oracle.bvh_trace_rays(osc, oracle_bvh(oracle, sc), g["sec_rays"])
        assert np.array_equal(h["id"], g["sec_ids"].astype(np.int32))
        assert np.array_equal(bits(h["dist"]), bits(g["sec_dist"]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_bvh_walk_equals_brute_force_on_synthetic_scenes(oracle, tag):
    g = np.load(os.path.join(GOLDEN, "synthetic_%s.npz" % tag))
    prims, lights, camd = S.random_scene(int(g["n"]), int(g["seed"]), lights=int(g["n_lights"]))
    for max_leaf in (1, 3):
        sc = capi.Scene.create(prims, lights, max_leaf=max_leaf)
        ocam = bind.camera_from_dict(oracle, camd)
        osc = oracle.scene(prims, lights)
        b = oracle_bvh(oracle, sc)
        ids, rgb, counts, hits = oracle.bvh_render(osc, b, ocam)
        assert np.array_equal(ids, g["ids"].astype(np.int32))
        assert np.array_equal(rgb, g["render0"])
        h, _ = oracle.bvh_trace_rays(osc, b, g["sec_rays"])
        assert np.array_equal(h["id"], g["sec_ids"].astype(np.int32))
        assert np.array_equal(bits(h["dist"]), bits(g["sec_dist"]))


def test_bvh_walk_with_loose_degenerate_and_tied_primitives(oracle):
    """Zero-scale primitives (unbounded), planes that can never be hit, coincident primitives (exact ties ->
    lowest id), non-parallelogram quads, rays starting inside boxes, unnormalised directions."""
    prims, lights, camd = S.random_scene(40, 31, lights=2)
    extra = S.new_prims(8)
    extra["kind"] = [S.CUBE, S.SPHERE, S.PLANE, S.PLANE, S.SPHERE, S.SPHERE, S.CYLINDER, S.PLANE]
    extra["q"][0, :9] = [0.5, 0.2, 0.1, 0.3, 0.2, 0.1, 0.0, 0.7, 0.4]           # zero scale: loose
    extra["q"][1, :6] = [-1.0, 0.5, 0.3, 0.6, 0.0, 0.5]                           # zero scale: loose
    extra["q"][2, :12] = [0, 0, 0, 1, 1, 1, 5, 5, 5, 2, 2, 2]                     # collinear edges: never hit
    extra["q"][3, :12] = [0, 0, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0]; extra["reserved"][3] = 1
    extra["q"][4, :6] = [1.0, -1.0, 0.5, 0.8, 0.8, 0.8]
    extra["q"][5, :6] = [1.0, -1.0, 0.5, 0.8, 0.8, 0.8]                           # exact duplicate of the previous
    extra["q"][6, :9] = [-2.0, 1.0, 0.0, 0.4, 1.0, 2.0, 0.5, 0.5, 1.5]
    extra["q"][7, :12] = [-3, -3, -1.5, 3, -3, -1.5, 9, 9, 4, -3, 3, -1.5]        # corner 2 far away, ignored
    allp = S.sort_reference_order(np.concatenate([prims, extra]))
    sc = capi.Scene.create(allp, lights, max_leaf=2)
    info = sc.info()
    assert info.n_loose == 2 and info.n_slots == len(allp) - 4
    osc = oracle.scene(allp, lights)
    b = oracle_bvh(oracle, sc)
    ocam = bind.camera_from_dict(oracle, camd)
    ids, rgb, _, _ = oracle.bvh_render(osc, b, ocam)
    ref_ids, _ = oracle.trace_primary(osc, ocam)
    ref_rgb, _, _ = oracle.render(osc, ocam, max_scatters=0)
    assert np.array_equal(ids, ref_ids)
    assert np.array_equal(rgb, ref_rgb)
    rng = np.random.default_rng(9)
    o = rng.uniform(-4, 4, size=(4000, 3))
    d = rng.normal(size=(4000, 3)) * rng.uniform(0.01, 30, size=(4000, 1))
    d[:50, 0] = 0.0
    d[50:100, 1:] = 0.0
    rays = np.concatenate([o, d], axis=1)
    a = oracle.trace_rays(osc, rays)
    h, _ = oracle.bvh_trace_rays(osc, b, rays)
    assert np.array_equal(a["id"], h["id"])
    same = (bits(a["dist"]) == bits(h["dist"]))
    assert same.all()
    # shadow verdicts against the reference rule evaluated by brute force
    L = rng.uniform(0.5, 6.0, size=4000)
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays_n = np.concatenate([o, dn], axis=1)
    a = oracle.trace_rays(osc, rays_n)
    want = (a["hit"] == 1) & (a["dist"] > 1e-4) & (a["dist"] < L - 1e-4)
    got = oracle.bvh_occluded(osc, b, rays_n, L)
    assert np.array_equal(got.astype(bool), want)


def test_scene_spanning_primitives_stay_beside_the_tree(oracle):
    """A primitive that spans a quarter of the scene or more (a ground plane) is kept out of the hierarchy and tested
    against every ray; at most 8 of them, and only in scenes of more than 16 primitives.  The walk still equals brute
    force."""
    prims = S.mixed_field(n_side=24, seed=9)                      # 24 x 24 cells and a ground plane
    lights = S.mixed_field_lights(half_extent=12.0, height=6.0, intensity=40.0)
    sc = capi.Scene.create(prims, lights)
    info = sc.info()
    nodes, slots, n_slots, n_loose, radius = sc.bvh()
    ground = int(np.flatnonzero(prims["kind"] == S.PLANE)[-1])
    assert info.n_loose == 1 and slots[n_slots:].tolist() == [ground]
    check_bvh(sc, oracle)
    cam = oracle.camera(**{k: v for k, v in zip(("pos", "direction"), ([0.0, -17.0, 8.5], [0.0, 0.8, -0.45]))}, res=(160, 90))
    osc = oracle.scene(prims, lights)
    ids, rgb, _, _ = oracle.bvh_render(osc, oracle_bvh(oracle, sc), cam)
    want_ids, _ = oracle.trace_primary(osc, cam)
    assert (want_ids == ground).mean() > 0.05
    assert np.array_equal(ids, want_ids) and np.array_equal(rgb, oracle.render(osc, cam, max_scatters=0)[0])
    # nine spanning planes are more than the rule takes out: they all stay in the tree
    many = np.concatenate([prims] + [prims[ground:ground + 1]] * 8)
    assert capi.Scene.create(S.sort_reference_order(many), lights).info().n_loose == 0
    # a small scene keeps everything in the tree
    assert capi.Scene.create(np.concatenate([prims[:10], prims[ground:ground + 1]]), lights).info().n_loose == 0


@pytest.mark.parametrize("kind", [S.CUBE, S.PLANE, S.CYLINDER, S.SPHERE])
def test_library_aabbs_of_constructed_primitives(kind):
    """compute_aabb of primitives built through the reference's constructors, bit for bit: this pins the library's
    own rotation matrices (the reference's build, g++ -O2, evaluates each angle with one sincos() call)."""
    g = np.load(os.path.join(GOLDEN, "single_prims.npz"))
    k = S.KIND_NAMES[kind].lower()
    for t in range(len(g[k + "_params"])):
        prim = S.new_prims(1)
        prim["kind"] = kind
        prim["q"][0] = g[k + "_params"][t]
        got = np.asarray(capi.prim_aabb(prim[0])).reshape(-1)
        want = np.asarray(g[k + "_aabb"][t], dtype=np.float64).reshape(-1)
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), (k, t)


def test_textures_from_the_scene_file(tmp_path):
    """albedo_texture_ppm: P6 and P3 files (comments, maxval 100) are read as the reference's Image reads them; files
    that are missing or cut short leave the primitive with its solid colour (mesh.cpp:105-110)."""
    prims, lights, camd, textures, prim_tex, paths = S.textured_scene(str(tmp_path))
    jp = str(tmp_path / "scene.json")
    S.scene_to_json(prims, lights, {"3": camd}, jp, texture_paths=paths)
    sc = capi.Scene.load_json(jp)
    assert sc.info().n_prims == len(prims)
    for i in range(len(prims)):
        t = sc.texture(i)
        if prim_tex[i] < 0:
            assert t is None, i
        else:
            assert np.array_equal(t, textures[prim_tex[i]]), i
    px, maxval = capi.read_ppm(paths[1])            # index 1 -> tex1: ASCII, comments, maxval 100
    assert maxval == 100 and np.array_equal(px, textures[1]) and px.max() <= 100
    with pytest.raises(capi.RtError) as e:
        capi.read_ppm(str(tmp_path / "does_not_exist.ppm"))
    assert e.value.status == -2
    with pytest.raises(capi.RtError) as e:
        capi.read_ppm(str(tmp_path / "broken.ppm"))
    assert e.value.status == -8 and "Unexpected EOF" in str(e.value)
    bad = tmp_path / "bad.ppm"
    for content, msg in ((b"P5\n1 1\n255\n\x00", "Unsupported PPM format"), (b"P6\n0 1\n255\n", "Invalid PPM header"),
                         (b"P6\n1 1\n70000\n", "max value"), (b"P3\n1 1\n9\n1 2 10\n", "out of range")):
        bad.write_bytes(content)
        with pytest.raises(capi.RtError) as e:
            capi.read_ppm(str(bad))
        assert msg in str(e.value), (content, str(e.value))
    # the same textures attached to a scene built in memory
    sc2 = capi.Scene.create(prims, lights)
    assert sc2.texture(0) is None
    sc2.set_textures(textures, prim_tex)
    for i in range(len(prims)):
        t = sc2.texture(i)
        assert (t is None) if prim_tex[i] < 0 else np.array_equal(t, textures[prim_tex[i]])
    with pytest.raises(capi.RtError):
        sc2.set_textures(textures, np.full(len(prims), len(textures), dtype=np.int32))
