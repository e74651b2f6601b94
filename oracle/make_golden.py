"""Generate tests/golden/ from the REAL reference engine (oracle/_ref/libref_rt.so).

Run in the development container only (needs /root/reference):   python -m oracle.make_golden
Everything written is data: scene files the reference ships, seeded inputs, and the reference's
outputs for them.  No reference source text is stored.
"""
import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bind                      # noqa: E402
from raytracer_amd import scenes as S        # noqa: E402

REF_ROOT = os.environ.get("RT_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
SHIPPED = [("toy_scene_for_render", "5"), ("toy_scene_for_render_with_materials", "6"), ("Test1", "21"),
           ("Test2", "142"), ("200_rand_1_cam", "1")]


def read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6"
    w, h = map(int, parts[1].split())
    img = np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)
    return img, hashlib.md5(data).hexdigest()


def row_hash(a):
    """FNV-1a style 64-bit hash of each row's float64 bit patterns (vectorised over rows)."""
    bits = np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    h = np.full(bits.shape[0], 1469598103934665603, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in range(bits.shape[1]):
            h = (h ^ bits[:, c]) * np.uint64(1099511628211)
    return h


def render(R, h, cam_id, max_scatters):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ppm")
        assert R.render_ppm(h, cam_id, out, 1, max_scatters, 0.0)
        return read_ppm(out)


def secondary_rays(hits, lights, rng, count):
    """Shadow-style rays: from hit points (offset along the normal like the reference's shade) to lights."""
    idx = np.flatnonzero(hits["hit"] == 1)
    if len(idx) == 0 or len(lights) == 0:
        return np.zeros((0, 6))
    pick = idx[(rng.uniform(count) * len(idx)).astype(int)]
    li = (rng.uniform(count) * len(lights)).astype(int)
    p, n = hits["p"][pick], hits["n"][pick]
    l = lights["pos"][li] - p
    ld = l / np.linalg.norm(l, axis=1, keepdims=True)
    s = np.where((n * ld).sum(1) >= 0, 1.0, -1.0)[:, None]
    return np.concatenate([p + s * n * 1e-4, ld], axis=1)


def main():
    R = bind.Ref()
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)

    # 1. scene files the reference ships (data fixtures)
    for name, _ in SHIPPED:
        shutil.copyfile(os.path.join(REF_ROOT, "s2899221", "ASCII", name + ".json"), os.path.join(GOLD, "scenes", name + ".json"))
    os.makedirs(os.path.join(GOLD, "scenes", "export"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF_ROOT, "tests", "export"))):
        if f.endswith(".json"):
            shutil.copyfile(os.path.join(REF_ROOT, "tests", "export", f), os.path.join(GOLD, "scenes", "export", f))
    for f in os.listdir(os.path.join(GOLD, "scenes")) + [os.path.join("export", x) for x in os.listdir(os.path.join(GOLD, "scenes", "export"))]:
        p = os.path.join(GOLD, "scenes", f)
        if os.path.isfile(p):
            os.chmod(p, 0o644)

    # 2. shipped scenes: primary-ray ids / distances / hit records, linear shade, renders
    for name, cam_id in SHIPPED:
        path = os.path.join(GOLD, "scenes", name + ".json")
        h = R.load(path)
        rays = R.primary_rays(h, cam_id)
        _, lights, cams = bind.load_scene_json(path)
        W, H = cams[cam_id]["res"]
        hits = R.trace_rays(h, rays)
        bvh = R.trace_rays(h, rays, bvh=True)
        out = dict(camera_id=cam_id, width=W, height=H,
                   ids=hits["id"].reshape(H, W).astype(np.int16),
                   dist_rows=hits["dist"].reshape(H, W)[::4],          # every 4th row in full ...
                   dist_row_hash=row_hash(hits["dist"].reshape(H, W)),  # ... and a bit-pattern hash of every row
                   ref_bvh_ids=bvh["id"].reshape(H, W).astype(np.int16),
                   aabbs=R.aabbs(h), n_meshes=R.num_meshes(h))
        # full hit records on a strided sample (every 31st ray)
        out["rec_idx"] = np.arange(0, len(rays), 31)
        for f in ("p", "n", "r", "u", "v"):
            out["rec_" + f] = hits[f][::31]
        if len(lights):
            out["shade_linear"] = R.shade_rays(h, rays).reshape(H, W, 3)
        for ms in (0, 2):
            img, md5 = render(R, h, cam_id, ms)
            out["render%d" % ms] = img
            out["render%d_md5" % ms] = md5
        rng = S.SplitMix64(sum(map(ord, name)))
        sec = secondary_rays(hits, lights, rng, 4096)
        if len(sec):
            sh = R.trace_rays(h, sec)
            shb = R.trace_rays(h, sec, bvh=True)
            out.update(sec_rays=sec, sec_ids=sh["id"].astype(np.int16), sec_dist=sh["dist"],
                       sec_ref_bvh_ids=shb["id"].astype(np.int16))
        np.savez_compressed(os.path.join(GOLD, "shipped_%s.npz" % name), **out)
        R.free(h)
        print(name, "hits", int((hits["id"] >= 0).sum()), out["render2_md5"])

    # 3. single primitives through the public constructors: random + adversarial rays
    rng = S.SplitMix64(20251226)
    prim_out = {}
    for kind, nparam in ((bind.CUBE, 9), (bind.PLANE, 12), (bind.CYLINDER, 9), (bind.SPHERE, 6)):
        plist, rlist, reclist, aabbs = [], [], [], []
        for trial in range(24):
            u = rng.uniform(nparam)
            if kind in (bind.CUBE, bind.CYLINDER):
                q = np.concatenate([(u[0:3] - 0.5) * 4, (u[3:6] - 0.5) * 2 * np.pi, 0.1 + 1.4 * u[6:9]])
                if trial == 3: q[3:6] = 0.0                     # axis aligned: parallel-ray branches
                if trial == 4: q[6] = 0.0                       # zero scale on one axis
                centre = q[0:3]
            elif kind == bind.SPHERE:
                q = np.concatenate([(u[0:3] - 0.5) * 4, 0.1 + 1.4 * u[3:6]])
                if trial == 4: q[4] = 0.0
                centre = q[0:3]
            else:
                p0 = (u[0:3] - 0.5) * 4
                e1, e2 = (u[3:6] - 0.5) * 3, (u[6:9] - 0.5) * 3
                q = np.concatenate([p0, p0 + e1, p0 + e1 + e2 + (u[9:12] - 0.5) * (0.3 if trial % 2 else 0.0), p0 + e2])
                if trial == 3: q = np.array([-1, -1, 0.5, 1, -1, 0.5, 1, 1, 0.5, -1, 1, 0.5], dtype=float)
                if trial == 4: q[3:6] = q[0:3]                  # degenerate edge
                centre = q.reshape(4, 3).mean(0)
            n = 96
            ur = rng.uniform(n * 6).reshape(n, 6)
            o = centre + (ur[:, 0:3] - 0.5) * 8
            tgt = centre + (ur[:, 3:6] - 0.5) * 2.2
            d = tgt - o
            d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)   # half normalised, half not
            o[80:88] = centre + (ur[80:88, 0:3] - 0.5) * 0.3                     # origins inside
            d[88:92] = np.eye(3)[[0, 1, 2, 0]] * np.array([1, -1, 1, -1])[:, None]  # axis-parallel
            d[92] = 0.0                                                           # zero direction
            rays = np.concatenate([o, d], axis=1)
            rec = R.intersect_prim(kind, q, rays)
            plist.append(np.pad(q, (0, 12 - len(q)))); rlist.append(rays); reclist.append(rec)
            aabbs.append(R.prim_aabb(kind, q))
        k = S.KIND_NAMES[kind].lower()
        prim_out[k + "_params"] = np.array(plist)
        prim_out[k + "_rays"] = np.array(rlist)
        prim_out[k + "_aabb"] = np.array(aabbs)
        rec = np.array(reclist)
        for f in ("hit", "dist", "p", "n", "r", "u", "v"):
            prim_out[k + "_" + f] = rec[f]
        print(k, "hit fraction", rec["hit"].mean())
    np.savez_compressed(os.path.join(GOLD, "single_prims.npz"), **prim_out)

    # 4. seeded mixed scenes with materials and lights, through the reference's own JSON loader
    for tag, n, seed, nl in (("a", 48, 11, 2), ("b", 120, 12, 3), ("c", 16, 13, 1)):
        prims, lights, cam = S.random_scene(n, seed, lights=nl)
        with tempfile.TemporaryDirectory() as td:
            jp = os.path.join(td, "s.json")
            S.scene_to_json(prims, lights, {"7": cam}, jp)
            h = R.load(jp)
            assert R.num_meshes(h) == n
            rays = R.primary_rays(h, "7")
            W, H = cam["res"]
            hits = R.trace_rays(h, rays)
            out = dict(n=n, seed=seed, n_lights=nl, ids=hits["id"].reshape(H, W).astype(np.int16), dist=hits["dist"].reshape(H, W),
                       shade_linear=R.shade_rays(h, rays).reshape(H, W, 3))
            for ms in (0, 1, 2):
                out["render%d" % ms], out["render%d_md5" % ms] = render(R, h, "7", ms)
            rngs = S.SplitMix64(seed * 977)
            sec = secondary_rays(hits, lights, rngs, 2048)
            sh = R.trace_rays(h, sec)
            out.update(sec_rays=sec, sec_ids=sh["id"].astype(np.int16), sec_dist=sh["dist"])
            R.free(h)
        np.savez_compressed(os.path.join(GOLD, "synthetic_%s.npz" % tag), **out)
        print("synthetic", tag, "hits", int((hits["id"] >= 0).sum()))

    # 5. parallelograms as reference Planes: pins the triangle extension indirectly
    prims = S.mixed_field(n_side=20, seed=0x100000, planes_as_triangles=False)
    lights = S.mixed_field_lights(half_extent=10.0, height=6.0, intensity=40.0)
    cam = S.mixed_field_camera(half_extent=10.0, res=(240, 135))
    with tempfile.TemporaryDirectory() as td:
        jp = os.path.join(td, "s.json")
        S.scene_to_json(prims, lights, {"1": cam}, jp)
        h = R.load(jp)
        rays = R.primary_rays(h, "1")
        hits = R.trace_rays(h, rays)
        W, H = cam["res"]
        img0, md5 = render(R, h, "1", 0)
        np.savez_compressed(os.path.join(GOLD, "mixed_field_planes.npz"), n_side=20, seed=0x100000,
                            ids=hits["id"].reshape(H, W).astype(np.int16), dist=hits["dist"].reshape(H, W),
                            kinds=prims["kind"], render0=img0, render0_md5=md5)
        R.free(h)
        print("mixed_field_planes hits", int((hits["id"] >= 0).sum()))

    golden_textured(R)


def golden_textured(R):
    """6. albedo textures (P6 and P3 files, a maxval below 255, unreadable files) through the reference's own loader,
    Image reader and Mesh::evaluate_albedo."""
    with tempfile.TemporaryDirectory() as td:
        prims, lights, cam, textures, prim_tex, paths = S.textured_scene(td)
        jp = os.path.join(td, "s.json")
        S.scene_to_json(prims, lights, {"3": cam}, jp, texture_paths=paths)
        h = R.load(jp)
        assert R.num_meshes(h) == len(prims)
        rays = R.primary_rays(h, "3")
        W, H = cam["res"]
        hits = R.trace_rays(h, rays)
        out = dict(ids=hits["id"].reshape(H, W).astype(np.int16), dist=hits["dist"].reshape(H, W),
                   u=hits["u"].reshape(H, W), v=hits["v"].reshape(H, W), prim_tex=prim_tex,
                   shade_linear=R.shade_rays(h, rays).reshape(H, W, 3))
        for ms in (0, 2):
            out["render%d" % ms], out["render%d_md5" % ms] = render(R, h, "3", ms)
        R.free(h)
    np.savez_compressed(os.path.join(GOLD, "textured.npz"), **out)
    print("textured hits", int((hits["id"] >= 0).sum()), out["render2_md5"])


if __name__ == "__main__":
    if "--textured-only" in sys.argv:
        golden_textured(bind.Ref())
    else:
        main()
