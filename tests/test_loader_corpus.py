"""The scene loader against the REAL reference loader on a corpus of documents (whitespace, member order, number
formats, missing members and their defaults, odd ids, planes with the wrong number of corners, degenerate geometry,
several cameras, strings holding braces, duplicate keys, non-JSON input).  tests/golden/loader_fuzz.npz holds the
documents and what the reference made of each (oracle/make_loader_fuzz.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import bind
from raytracer_amd import capi

G = np.load(os.path.join(GOLDEN, "loader_fuzz.npz"))
NAMES = [str(n) for n in G["names"]]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def load(k, tmp_path):
    path = str(tmp_path / ("doc%d.json" % k))
    with open(path, "w") as f:
        f.write(str(G["texts"][k]))
    return path, capi.Scene.load_json(path)


@pytest.mark.parametrize("k", range(len(NAMES)), ids=NAMES)
def test_document_loads_like_the_reference(oracle, tmp_path, k):
    path, sc = load(k, tmp_path)
    from_text = capi.Scene.load_json_text(str(G["texts"][k]))
    prims = sc.prims()
    assert len(prims) == int(G["n_%d" % k]) == len(from_text.prims())
    aabbs = np.array([capi.prim_aabb(p) for p in prims]).reshape(-1, 6)
    assert np.array_equal(bits(aabbs), bits(G["aabbs_%d" % k]))
    # lights and materials show in what a fixed fan of rays sees: ids, distances, linear colour
    rays = G["rays"]
    osc = oracle.scene(prims, sc.lights())
    hits = oracle.trace_rays(osc, rays)
    assert np.array_equal(hits["id"], G["ids_%d" % k])
    hit = hits["hit"] == 1
    assert np.array_equal(bits(hits["dist"][hit]), bits(G["dist_%d" % k][hit]))
    got = oracle.shade_rays(osc, rays).astype(np.int16)
    want = G["rgb_%d" % k].astype(np.int16)
    # with three or more lights the reference sums them in unordered_map order, the library in file order: +-1
    assert np.abs(got - want).max() <= (1 if len(sc.lights()) > 2 else 0)
    # cameras: RayTracer's reader (defaults, optional up vector) and the stand-alone reader (all members mandatory)
    for cid in G["cams_%d" % k]:
        cid = str(cid)
        count = int(G["count_%d_%s" % (k, cid)])
        try:
            cam, _ = sc.camera(cid)
        except capi.RtError:
            cam = None
        assert (cam is not None) == (count > 0), cid
        if cam is not None:
            assert cam.width * cam.height == count
            key = "primary_%d_%s" % (k, cid)
            if key in G:
                assert np.array_equal(bits(capi.pixel_rays(cam, G["primary_px_%d_%s" % (k, cid)])), bits(G[key]))
        try:
            alone = capi.camera_from_json(path, cid)
        except capi.RtError:
            alone = None
        assert (alone is not None) == bool(G["alone_ok_%d_%s" % (k, cid)]), cid
        if alone is not None:
            assert np.array_equal(bits(capi.pixel_rays(alone, G["pixels"])), bits(G["alone_%d_%s" % (k, cid)]))


@pytest.mark.gpu
def test_corpus_on_the_gpu(tmp_path):
    ctx = capi.Context(0)
    rays = G["rays"]
    for k in range(len(NAMES)):
        _, sc = load(k, tmp_path)
        ctx.upload(sc)
        ids, dist = ctx.trace_rays(rays)
        assert np.array_equal(ids, G["ids_%d" % k]), NAMES[k]
        hit = ids >= 0
        assert np.array_equal(bits(dist[hit]), bits(G["dist_%d" % k][hit])), NAMES[k]
        rec = ctx.trace_rays_full(rays)
        hits10 = np.zeros((len(rays), 10))
        hits10[:, 0:6] = rec[:, 3:9]; hits10[:, 6] = rec[:, 1]; hits10[:, 7:9] = rec[:, 12:14]; hits10[:, 9] = 1.0
        got = ctx.shade_hits(hits10, rays[:, 0:3]).astype(np.int16)
        assert np.abs(got - G["rgb_%d" % k].astype(np.int16)).max() <= 1, NAMES[k]
    ctx.close()
