"""ctypes bindings for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

`Oracle`  - oracle/librt_oracle.so, the plain-C FP64 restatement (oracle/rt_oracle.c).
`Ref`     - oracle/_ref/libref_rt.so, the real reference engine compiled from its own sources
            (development container only; `Ref.available()` is False on the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (raytracer_amd) never does.

Also here: `load_scene_json`, an independent (Python `json`) reader of the Blender-export scene
format with the defaults of the reference loader (s2899221/Code/raytracer.cpp:29-121,
s2899221/Code/mesh.cpp:52-114, 316-363, 544-590, 707-782), used to cross-check the product's own loader.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CUBE, PLANE, CYLINDER, SPHERE, TRIANGLE = 0, 1, 2, 3, 4

PRIM_DT = np.dtype([("kind", "<i4"), ("reserved", "<i4"), ("q", "<f8", (12,)),
                    ("alpha", "<f8"), ("metallic", "<f8"), ("roughness", "<f8"), ("ior", "<f8"),
                    ("albedo", "<f8", (3,))])
LIGHT_DT = np.dtype([("pos", "<f8", (3,)), ("intensity", "<f8")])
HIT_DT = np.dtype([("hit", "<i4"), ("id", "<i4"), ("dist", "<f8"), ("p", "<f8", (3,)), ("n", "<f8", (3,)),
                   ("r", "<f8", (3,)), ("u", "<f8"), ("v", "<f8"), ("prim", "<i4"), ("inst", "<i4")])
INSTANCE_DT = np.dtype([("rot", "<f8", (9,)), ("trans", "<f8", (3,)), ("first_prim", "<u4"), ("n_prims", "<u4")])
NODE_DT = np.dtype([("lo", "<f4", (3,)), ("first", "<u4"), ("hi", "<f4", (3,)), ("count", "<u4")])
NODE4_DT = np.dtype([("lo", "<f4", (3, 4)), ("hi", "<f4", (3, 4)), ("ref", "<u4", (4,)), ("pad", "<u4", (4,))])
assert NODE4_DT.itemsize == 128
assert PRIM_DT.itemsize == 160 and LIGHT_DT.itemsize == 32 and HIT_DT.itemsize == 112 and INSTANCE_DT.itemsize == 104 and NODE_DT.itemsize == 32


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("rot", C.c_double * 9), ("focal", C.c_double),
                ("sensor_w", C.c_double), ("sensor_h", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class Bvh(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_int64), ("slot_prim", C.c_void_p), ("n_slots", C.c_int64),
                ("n_loose", C.c_int64), ("radius", C.c_float), ("nodes4", C.c_void_p), ("n_nodes4", C.c_int64),
                ("loose_boxes", C.c_void_p)]


class _Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class _Scene(C.Structure):
    _fields_ = [("prims", C.c_void_p), ("n_prims", C.c_int64), ("lights", C.c_void_p), ("n_lights", C.c_int64),
                ("textures", C.c_void_p), ("n_textures", C.c_int64), ("prim_tex", C.c_void_p),
                ("instanced", C.c_int64), ("inst", C.c_void_p), ("n_inst", C.c_int64), ("inst_bounds", C.c_void_p),
                ("prim_bounds", C.c_void_p)]


def new_prims(n):
    p = np.zeros(n, dtype=PRIM_DT)
    p["alpha"] = 1.0
    p["roughness"] = 0.5
    p["ior"] = 1.5
    p["albedo"] = 1.0
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def build(ref=True):
    """Compile the checkers (idempotent; make decides what is stale)."""
    subprocess.run(["make", "-s", "-C", HERE, "all" if ref else os.path.join(HERE, "librt_oracle.so")], check=True)


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "librt_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        dp, vp, i64p = C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int64)
        L.rto_camera_look.argtypes = [dp, dp, dp, C.POINTER(Camera)]
        L.rto_pixel_to_ray.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, dp, dp]
        L.rto_intersect.argtypes = [vp, dp, dp, vp]
        L.rto_intersect.restype = C.c_int
        L.rto_aabb.argtypes = [vp, dp, dp]
        L.rto_trace_rays.argtypes = [C.POINTER(_Scene), vp, C.c_int64, vp, C.c_int]
        L.rto_shade_rays.argtypes = [C.POINTER(_Scene), vp, C.c_int64, vp, C.c_int]
        L.rto_trace_primary.argtypes = [C.POINTER(_Scene), C.POINTER(Camera)] + [C.c_int] * 4 + [vp, vp, C.c_int]
        L.rto_render.argtypes = [C.POINTER(_Scene), C.POINTER(Camera)] + [C.c_int] * 4 + \
                                [C.c_int, C.c_double, vp, i64p, i64p, C.c_int]
        L.rto_gamma_lut.argtypes = [vp]
        L.rto_trace_rays_rgb.argtypes = [C.POINTER(_Scene), vp, C.c_int64, C.c_int, C.c_double, vp, C.c_int]
        L.rto_hit_bounds.argtypes = [vp, C.c_int64, vp]
        L.rto_brute_shade_rays.argtypes = [C.POINTER(_Scene), vp, C.c_int64, vp, vp, vp, C.c_int]
        L.rto_bvh_closest_hit.argtypes = [C.POINTER(_Scene), C.POINTER(Bvh), dp, dp, vp, i64p]
        L.rto_bvh_closest_hit.restype = C.c_int
        L.rto_bvh_occluded.argtypes = [C.POINTER(_Scene), C.POINTER(Bvh), dp, dp, C.c_double, i64p]
        L.rto_bvh_occluded.restype = C.c_int
        L.rto_bvh_trace_rays.argtypes = [C.POINTER(_Scene), C.POINTER(Bvh), vp, C.c_int64, vp, i64p, C.c_int]
        L.rto_bvh_render.argtypes = [C.POINTER(_Scene), C.POINTER(Bvh), C.POINTER(Camera)] + [C.c_int] * 5 + \
                                    [vp, vp, i64p, i64p, C.c_int]

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def scene(prims, lights=None, textures=None, prim_tex=None):
        """textures: list of (h, w, 3) uint8 arrays; prim_tex: texture index per primitive (-1 = none)."""
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, LIGHT_DT), dtype=LIGHT_DT)
        s = _Scene(_ptr(prims), len(prims), _ptr(lights), len(lights), None, 0, None, 0, None, 0, None, None)
        s._keep = (prims, lights)
        if textures:
            arrs = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
            table = (_Texture * len(arrs))(*[_Texture(a.shape[1], a.shape[0], a.ctypes.data) for a in arrs])
            pt = np.ascontiguousarray(prim_tex, dtype=np.int32)
            assert len(pt) == len(prims)
            s.textures, s.n_textures, s.prim_tex = C.cast(table, C.c_void_p), len(arrs), _ptr(pt)
            s._keep += (arrs, table, pt)
        return s

    def with_prim_bounds(self, scene):
        """Hand the brute-force loop a box per primitive so that it can pass over primitives a ray cannot touch;
        results do not depend on it."""
        prims = scene._keep[0]
        bounds = np.empty((len(prims), 6), dtype=np.float64)
        self.lib.rto_hit_bounds(_ptr(prims), len(prims), _ptr(bounds))
        scene.prim_bounds = _ptr(bounds)
        scene._keep += (bounds,)
        return scene

    def brute_shade_rays(self, scene, rays6, threads=0):
        """Closest hit, shadow verdict per light and linear colour per ray, every ray against every primitive."""
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        hits = np.zeros(len(rays6), dtype=HIT_DT)
        lit = np.zeros((len(rays6), max(int(scene.n_lights), 1)), dtype=np.uint8)
        rgb = np.zeros((len(rays6), 3), dtype=np.uint8)
        self.lib.rto_brute_shade_rays(C.byref(scene), _ptr(rays6), len(rays6), _ptr(hits), _ptr(lit), _ptr(rgb), threads)
        return hits, lit[:, :int(scene.n_lights)], rgb

    def _hit_bounds(self, p):
        """Box that holds every point the intersector can return for p (the reference's own AABB of a plane spans the
        four corners although its intersector walks the parallelogram of corners 0, 1, 3)."""
        if int(p["kind"]) == PLANE:
            q = p["q"]
            p0, p1, p3 = q[0:3], q[3:6], q[9:12]
            c = np.stack([p0, p1, p3, p1 + p3 - p0])
            return np.concatenate([c.min(0), c.max(0)])
        return self.aabb(p)

    def instanced_scene(self, prims, instances, lights=None, skip_far_instances=True):
        """Two-level scene (rto_instance).  skip_far_instances hands the oracle each mesh's bounds so that it can pass
        over instances a ray cannot touch; results do not depend on it."""
        s = self.scene(prims, lights)
        inst = np.ascontiguousarray(instances, dtype=INSTANCE_DT)
        s.instanced, s.inst, s.n_inst = 1, _ptr(inst), len(inst)
        s._keep += (inst,)
        if skip_far_instances and len(inst):
            prims = s._keep[0]
            boxes = np.array([self._hit_bounds(p) for p in prims]) if len(prims) else np.zeros((0, 6))
            bounds = np.empty((len(inst), 6))
            cache = {}
            for k, it in enumerate(inst):
                key = (int(it["first_prim"]), int(it["n_prims"]))
                if key not in cache:
                    b = boxes[key[0]:key[0] + key[1]]
                    b = b[np.isfinite(b).all(axis=1)] if len(b) else b
                    cache[key] = np.concatenate([b[:, :3].min(0), b[:, 3:].max(0)]) if len(b) else np.array([1.0, 1, 1, -1, -1, -1])
                bounds[k] = cache[key]
            bounds = np.ascontiguousarray(bounds)
            s.inst_bounds = _ptr(bounds)
            s._keep += (bounds,)
        return s

    def camera(self, pos, direction, up=None, focal=35.0, sensor=(36.0, 24.0), res=(1920, 1080)):
        cam = Camera()
        pos = (C.c_double * 3)(*pos)
        d = (C.c_double * 3)(*direction)
        u = (C.c_double * 3)(*up) if up is not None else None
        self.lib.rto_camera_look(pos, d, u, C.byref(cam))
        cam.focal, cam.sensor_w, cam.sensor_h = focal, sensor[0], sensor[1]
        cam.width, cam.height = res
        return cam

    def pixel_rays(self, cam, pxy):
        pxy = np.asarray(pxy, dtype=np.float64).reshape(-1, 2)
        out = np.empty((len(pxy), 6), dtype=np.float64)
        o = (C.c_double * 3)()
        d = (C.c_double * 3)()
        for i, (x, y) in enumerate(pxy):
            self.lib.rto_pixel_to_ray(C.byref(cam), float(x), float(y), o, d)
            out[i, :3] = o[:]
            out[i, 3:] = d[:]
        return out

    def intersect(self, prim, rays6):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        prim = np.ascontiguousarray(prim, dtype=PRIM_DT).reshape(1)
        out = np.zeros(len(rays6), dtype=HIT_DT)
        dp = C.POINTER(C.c_double)
        for i in range(len(rays6)):
            self.lib.rto_intersect(_ptr(prim), rays6[i, :3].ctypes.data_as(dp), rays6[i, 3:].ctypes.data_as(dp),
                                   C.c_void_p(out.ctypes.data + i * HIT_DT.itemsize))
        return out

    def aabb(self, prim):
        prim = np.ascontiguousarray(prim, dtype=PRIM_DT).reshape(1)
        mn = (C.c_double * 3)()
        mx = (C.c_double * 3)()
        self.lib.rto_aabb(_ptr(prim), mn, mx)
        return np.array(mn[:] + mx[:])

    def trace_rays(self, scene, rays6, threads=0):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(len(rays6), dtype=HIT_DT)
        self.lib.rto_trace_rays(C.byref(scene), _ptr(rays6), len(rays6), _ptr(out), threads)
        return out

    def shade_rays(self, scene, rays6, threads=0):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((len(rays6), 3), dtype=np.uint8)
        self.lib.rto_shade_rays(C.byref(scene), _ptr(rays6), len(rays6), _ptr(out), threads)
        return out

    def trace_rays_rgb(self, scene, rays6, max_scatters=2, min_scatter_intensity=0.0, threads=0):
        """trace_ray_recursive per ray: linear 8-bit colour after up to max_scatters bounces (no gamma)."""
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((len(rays6), 3), dtype=np.uint8)
        self.lib.rto_trace_rays_rgb(C.byref(scene), _ptr(rays6), len(rays6), max_scatters, min_scatter_intensity, _ptr(out), threads)
        return out

    def trace_primary(self, scene, cam, rect=None, threads=0):
        x0, y0, w, h = rect or (0, 0, cam.width, cam.height)
        ids = np.empty((h, w), dtype=np.int32)
        dist = np.empty((h, w), dtype=np.float64)
        self.lib.rto_trace_primary(C.byref(scene), C.byref(cam), x0, y0, w, h, _ptr(ids), _ptr(dist), threads)
        return ids, dist

    def render(self, scene, cam, rect=None, max_scatters=2, min_scatter_intensity=0.0, threads=0):
        x0, y0, w, h = rect or (0, 0, cam.width, cam.height)
        rgb = np.empty((h, w, 3), dtype=np.uint8)
        nh, ns = C.c_int64(0), C.c_int64(0)
        self.lib.rto_render(C.byref(scene), C.byref(cam), x0, y0, w, h, max_scatters, min_scatter_intensity,
                            _ptr(rgb), C.byref(nh), C.byref(ns), threads)
        return rgb, nh.value, ns.value

    def gamma_lut(self):
        lut = np.empty(256, dtype=np.uint8)
        self.lib.rto_gamma_lut(_ptr(lut))
        return lut

    @staticmethod
    def bvh(nodes, slot_prim, n_slots, n_loose, radius, nodes4=None, loose_boxes=None):
        """Descriptor of the product's flattened BVH (numpy views from raytracer_amd.capi.Scene.bvh()); with nodes4
        (Scene.wide_bvh()) the walk goes through the wide nodes, as the product's does when it has them; with loose_boxes
        (Scene.loose_boxes()) the primitives beside the tree are culled by their boxes, as the product does."""
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DT)
        slot_prim = np.ascontiguousarray(slot_prim, dtype=np.uint32)
        b = Bvh(_ptr(nodes), len(nodes), _ptr(slot_prim), int(n_slots), int(n_loose), float(radius), None, 0, None)
        b._keep = (nodes, slot_prim)
        if loose_boxes is not None and len(loose_boxes):
            loose_boxes = np.ascontiguousarray(loose_boxes, dtype=np.float32).reshape(-1, 8)
            assert len(loose_boxes) == int(n_loose)
            b.loose_boxes = _ptr(loose_boxes)
            b._keep += (loose_boxes,)
        if nodes4 is not None and len(nodes4):
            nodes4 = np.ascontiguousarray(nodes4, dtype=NODE4_DT)
            b.nodes4, b.n_nodes4 = _ptr(nodes4), len(nodes4)
            b._keep += (nodes4,)
        return b

    def bvh_render(self, scene, bvh, cam, rect=None, step=1, threads=0):
        """Primary + shadow pass through the flattened BVH; returns ids, rgb, counts[4], hits."""
        x0, y0, w, h = rect or (0, 0, cam.width, cam.height)
        ow, oh = (w + step - 1) // step, (h + step - 1) // step
        ids = np.empty((oh, ow), dtype=np.int32)
        rgb = np.empty((oh, ow, 3), dtype=np.uint8)
        counts = (C.c_int64 * 4)()
        nh = C.c_int64(0)
        self.lib.rto_bvh_render(C.byref(scene), C.byref(bvh), C.byref(cam), x0, y0, w, h, step, _ptr(ids), _ptr(rgb),
                                counts, C.byref(nh), threads)
        return ids, rgb, np.array(counts[:], dtype=np.int64), nh.value

    def bvh_trace_rays(self, scene, bvh, rays6, threads=0):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(len(rays6), dtype=HIT_DT)
        counts = (C.c_int64 * 2)()
        self.lib.rto_bvh_trace_rays(C.byref(scene), C.byref(bvh), _ptr(rays6), len(rays6), _ptr(out), counts, threads)
        return out, np.array(counts[:], dtype=np.int64)

    def bvh_occluded(self, scene, bvh, rays6, light_dist):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        light_dist = np.broadcast_to(np.asarray(light_dist, dtype=np.float64), (len(rays6),))
        out = np.zeros(len(rays6), dtype=np.int32)
        dp = C.POINTER(C.c_double)
        for i in range(len(rays6)):
            out[i] = self.lib.rto_bvh_occluded(C.byref(scene), C.byref(bvh), rays6[i, :3].ctypes.data_as(dp),
                                               rays6[i, 3:].ctypes.data_as(dp), float(light_dist[i]), None)
        return out


REC_DOUBLES = 16   # oracle/ref_harness.cpp HitRecord


def _records_to_hits(rec):
    out = np.zeros(len(rec), dtype=HIT_DT)
    out["hit"] = rec[:, 0].astype(np.int32)
    out["id"] = rec[:, 1].astype(np.int32)
    out["dist"] = rec[:, 2]
    out["p"] = rec[:, 3:6]
    out["n"] = rec[:, 6:9]
    out["r"] = rec[:, 9:12]
    out["u"] = rec[:, 12]
    out["v"] = rec[:, 13]
    return out


class Ref:
    """The real reference engine (oracle/_ref/libref_rt.so)."""
    PATH = os.path.join(HERE, "_ref", "libref_rt.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = L = C.CDLL(self.PATH)
        vp = C.c_void_p
        L.ref_scene_load.argtypes = [C.c_char_p]
        L.ref_scene_load.restype = vp
        L.ref_scene_free.argtypes = [vp]
        L.ref_scene_num_meshes.argtypes = [vp]
        L.ref_scene_aabbs.argtypes = [vp, vp]
        L.ref_primary_rays.argtypes = [vp, C.c_char_p, vp, C.c_long]
        L.ref_primary_rays.restype = C.c_long
        L.ref_trace_rays.argtypes = [vp, vp, C.c_long, vp]
        L.ref_trace_rays_bvh.argtypes = [vp, vp, C.c_long, vp]
        L.ref_shade_rays.argtypes = [vp, vp, C.c_long, vp]
        L.ref_render_ppm.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double]
        L.ref_intersect_prim.argtypes = [C.c_int, vp, vp, C.c_long, vp]
        L.ref_prim_aabb.argtypes = [C.c_int, vp, vp]
        L.ref_camera_rays.argtypes = [C.c_char_p, C.c_char_p, vp, C.c_long, vp]
        L.ref_prims_create.argtypes = [vp, vp, C.c_long]
        L.ref_prims_create.restype = vp
        L.ref_prims_free.argtypes = [vp]
        L.ref_prims_trace_bvh.argtypes = [vp, vp, C.c_long, vp, vp, C.c_int]
        L.ref_prims_trace_brute.argtypes = [vp, vp, C.c_long, vp, vp, C.c_int]
        L.ref_time_one_pass.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_time_one_pass.restype = C.c_double
        L.ref_time_accelerated.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_time_accelerated.restype = C.c_double

    def load(self, path):
        h = self.lib.ref_scene_load(path.encode())
        if not h:
            raise FileNotFoundError(path)
        return h

    def free(self, h):
        self.lib.ref_scene_free(h)

    def num_meshes(self, h):
        return self.lib.ref_scene_num_meshes(h)

    def aabbs(self, h):
        out = np.empty((self.num_meshes(h), 6), dtype=np.float64)
        self.lib.ref_scene_aabbs(h, _ptr(out))
        return out

    def primary_rays(self, h, cam_id):
        n = self.lib.ref_primary_rays(h, cam_id.encode(), None, 0)
        rays = np.empty((n, 6), dtype=np.float64)
        self.lib.ref_primary_rays(h, cam_id.encode(), _ptr(rays), n)
        return rays

    def trace_rays(self, h, rays6, bvh=False):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        rec = np.empty((len(rays6), REC_DOUBLES), dtype=np.float64)
        (self.lib.ref_trace_rays_bvh if bvh else self.lib.ref_trace_rays)(h, _ptr(rays6), len(rays6), _ptr(rec))
        return _records_to_hits(rec)

    def shade_rays(self, h, rays6):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        out = np.empty((len(rays6), 3), dtype=np.uint8)
        self.lib.ref_shade_rays(h, _ptr(rays6), len(rays6), _ptr(out))
        return out

    def render_ppm(self, h, cam_id, out_path, spp=1, max_scatters=2, min_scatter_intensity=0.0):
        return bool(self.lib.ref_render_ppm(h, cam_id.encode(), out_path.encode(), spp, max_scatters,
                                            min_scatter_intensity))

    def intersect_prim(self, kind, params, rays6):
        """kind uses this module's constants; params as the reference constructors take them."""
        refkind = {CUBE: 0, PLANE: 1, CYLINDER: 2, SPHERE: 3}[kind]
        params = np.ascontiguousarray(params, dtype=np.float64)
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        rec = np.empty((len(rays6), REC_DOUBLES), dtype=np.float64)
        self.lib.ref_intersect_prim(refkind, _ptr(params), _ptr(rays6), len(rays6), _ptr(rec))
        return _records_to_hits(rec)

    def prim_aabb(self, kind, params):
        refkind = {CUBE: 0, PLANE: 1, CYLINDER: 2, SPHERE: 3}[kind]
        params = np.ascontiguousarray(params, dtype=np.float64)
        out = np.empty(6, dtype=np.float64)
        self.lib.ref_prim_aabb(refkind, _ptr(params), _ptr(out))
        return out

    def prims_create(self, prims):
        """A reference scene from PRIM_DT records (cubes, planes, cylinders, spheres only, already in that order),
        built through the reference's public constructors."""
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        kinds = np.ascontiguousarray(prims["kind"], dtype=np.int32)
        assert (np.diff(kinds) >= 0).all() and kinds.max(initial=0) <= SPHERE
        q = np.ascontiguousarray(prims["q"], dtype=np.float64)
        h = self.lib.ref_prims_create(_ptr(kinds), _ptr(q), len(prims))
        assert h
        return h

    def prims_free(self, h):
        self.lib.ref_prims_free(h)

    def _prims_trace(self, fn, h, rays6, threads):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        ids = np.empty(len(rays6), dtype=np.int32)
        dist = np.empty(len(rays6), dtype=np.float64)
        fn(h, _ptr(rays6), len(rays6), _ptr(ids), _ptr(dist), threads)
        return ids, dist

    def prims_trace_bvh(self, h, rays6, threads=1):
        """(ids, distances) through the reference's own get_BVH_tree + accelerated_one_pass_intersection."""
        return self._prims_trace(self.lib.ref_prims_trace_bvh, h, rays6, threads)

    def prims_trace_brute(self, h, rays6, threads=1):
        """(ids, distances) through the loop of one_pass_intersection over the same objects."""
        return self._prims_trace(self.lib.ref_prims_trace_brute, h, rays6, threads)

    def camera_rays(self, path, cam_id, pxy):
        pxy = np.ascontiguousarray(pxy, dtype=np.float64).reshape(-1, 2)
        rays = np.empty((len(pxy), 6), dtype=np.float64)
        ok = self.lib.ref_camera_rays(path.encode(), cam_id.encode(), _ptr(pxy), len(pxy), _ptr(rays))
        return rays if ok else None


# ------------------------------------------------------------------------------------------------
# Independent scene reader (Python json), same defaults as the reference loader.

def _vec(d, key, default):
    v = d.get(key)
    if not isinstance(v, dict) or not all(k in v for k in "xyz"):
        return list(default)
    return [float(v["x"]), float(v["y"]), float(v["z"])]


def _material(p, obj):
    m = obj.get("material")
    if isinstance(m, dict):
        p["alpha"] = float(m.get("alpha", 1.0))
        p["metallic"] = float(m.get("metallic", 0.0))
        p["roughness"] = float(m.get("roughness", 0.5))
        p["ior"] = float(m.get("ior", 1.5))
        p["albedo"] = _vec(m, "albedo", (1.0, 1.0, 1.0))


def load_scene_json(path):
    """Returns (prims, lights, cameras) with prims in the reference's id order
    (cubes, planes, cylinders, spheres; file order inside a class), lights and cameras in file order.
    cameras: dict id -> dict(location, direction, up|None, focal, sensor, res)."""
    with open(path) as f:
        doc = json.load(f)
    mesh = doc.get("MESH", {}) or {}
    rows = []
    for kind, cls in ((CUBE, "Cube"), (PLANE, "Plane"), (CYLINDER, "Cylinder"), (SPHERE, "Sphere")):
        for _id, obj in (mesh.get(cls, {}) or {}).items():
            p = new_prims(1)[0]
            p["kind"] = kind
            if kind in (CUBE, CYLINDER):
                rot = obj.get("rotation", {}) or {}
                p["q"][:9] = _vec(obj, "translation", (0, 0, 0)) + \
                    [float(rot.get("roll", 0.0)), float(rot.get("pitch", 0.0)), float(rot.get("yaw", 0.0))] + \
                    _vec(obj, "scale", (1, 1, 1))
            elif kind == SPHERE:
                p["q"][:6] = _vec(obj, "location", (0, 0, 0)) + _vec(obj, "scale", (1, 1, 1))
            else:
                corners = obj.get("corners")
                if not corners:
                    continue
                flat = []
                for c in corners[:4]:
                    flat += [float(c.get("x", 0.0)), float(c.get("y", 0.0)), float(c.get("z", 0.0))]
                if len(flat) < 12:
                    # fewer than four corners: the reference keeps the primitive but it never intersects
                    flat += [0.0] * (12 - len(flat))
                    p["reserved"] = len(corners) if len(corners) else -1
                p["q"][:12] = flat
            _material(p, obj)
            rows.append(p)
    prims = np.array(rows, dtype=PRIM_DT) if rows else np.zeros(0, PRIM_DT)
    lights = []
    for _id, obj in ((doc.get("LIGHT", {}) or {}).get("POINT", {}) or {}).items():
        if "location" not in obj:
            continue
        lights.append((tuple(_vec(obj, "location", (0, 0, 0))), float(obj.get("radiant_intensity", 1.0))))
    lights = np.array(lights, dtype=LIGHT_DT) if lights else np.zeros(0, LIGHT_DT)
    cams = {}
    for cid, obj in ((doc.get("CAMERA", {}) or {}).get("PERSP", {}) or {}).items():
        if "location" not in obj or "direction" not in obj:
            continue
        res = obj.get("film_resolution", {"x": 1920, "y": 1080})
        cams[cid] = dict(location=_vec(obj, "location", (0, 0, 0)), direction=_vec(obj, "direction", (0, 0, 1)),
                         up=_vec(obj, "up", (0, 0, 1)) if "up" in obj else None,
                         focal=float(obj.get("focal_length", 35.0)),
                         sensor=(float(obj.get("sensor_width", 36.0)), float(obj.get("sensor_height", 24.0))),
                         res=(int(res["x"]), int(res["y"])))
    return prims, lights, cams


def camera_from_dict(oracle, c):
    return oracle.camera(c["location"], c["direction"], c["up"], c["focal"], c["sensor"], c["res"])
