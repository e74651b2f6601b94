"""ctypes binding of librt_hip.so (include/rt_hip.h).

The HIP library is the product: nothing here computes on the CPU.  Loading fails loudly when the
library has not been built (`python -m raytracer_amd.build`); there is no fallback path.
"""
import ctypes as C
import os

import numpy as np

from .scenes import INSTANCE_DT, LIGHT_DT, PRIM_DT

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB: another build of the library (experiments with compile-time variants, profiles/variant.py)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(HERE, "lib", "librt_hip.so")

NODE_DT = np.dtype([("lo", "<f4", (3,)), ("first", "<u4"), ("hi", "<f4", (3,)), ("count", "<u4")])
NODE4_DT = np.dtype([("lo", "<f4", (3, 4)), ("hi", "<f4", (3, 4)), ("ref", "<u4", (4,)), ("pad", "<u4", (4,))])   # rt_node4

RT_OK = 0
STATUS = {0: "RT_OK", -1: "RT_ERR_INVALID", -2: "RT_ERR_IO", -3: "RT_ERR_NOT_FOUND", -4: "RT_ERR_NO_DEVICE",
          -5: "RT_ERR_HIP", -6: "RT_ERR_NO_SCENE", -7: "RT_ERR_LIMIT", -8: "RT_ERR_PARSE", -9: "RT_ERR_COMM"}


class RtError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("%s (%d): %s" % (STATUS.get(status, "?"), status, detail))
        self.status = status


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("rot", C.c_double * 9), ("focal", C.c_double),
                ("sensor_w", C.c_double), ("sensor_h", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class Tiling(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tile", C.c_int32), ("n_ranks", C.c_int32)]


class BuildOpts(C.Structure):
    _fields_ = [("max_leaf", C.c_uint32), ("sah_bins", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32 * 5)]


BUILD_ON_DEVICE = 1


class SceneInfo(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_slots", C.c_uint64), ("n_loose", C.c_uint64),
                ("n_nodes", C.c_uint32), ("depth", C.c_uint32), ("n_lights", C.c_uint32), ("n_cameras", C.c_uint32),
                ("n_by_kind", C.c_uint32 * 8), ("bounds", C.c_double * 6), ("radius", C.c_double),
                ("n_instances", C.c_uint64), ("n_instanced_prims", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary_rays", "primary_hits", "shadow_rays", "bounce_rays",
                                          "nodes_primary", "prims_primary", "nodes_shadow", "prims_shadow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("primary_ms", C.c_double), ("shadow_ms", C.c_double), ("resolve_ms", C.c_double), ("bounce_ms", C.c_double),
                ("primary_launches", C.c_uint32), ("shadow_launches", C.c_uint32), ("resolve_launches", C.c_uint32),
                ("bounce_launches", C.c_uint32), ("entry_ms", C.c_double), ("entry_launches", C.c_uint32), ("reserved", C.c_uint32)]


# every symbol include/rt_hip.h declares: (name, restype, argtypes)
_VP, _I32, _U32, _U64, _D = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
_DP = C.POINTER(C.c_double)
SYMBOLS = [
    ("rt_scene_create", C.c_int, [_VP, _U64, _VP, _U32, C.POINTER(BuildOpts), C.POINTER(_VP)]),
    ("rt_scene_create_instanced", C.c_int, [_VP, _U64, _VP, _U64, _VP, _U32, C.POINTER(BuildOpts), C.POINTER(_VP)]),
    ("rt_scene_load_json", C.c_int, [C.c_char_p, C.POINTER(BuildOpts), C.POINTER(_VP)]),
    ("rt_scene_load_json_text", C.c_int, [C.c_char_p, _U64, C.POINTER(BuildOpts), C.POINTER(_VP)]),
    ("rt_scene_set_textures", C.c_int, [_VP, _VP, _U32, _VP]),
    ("rt_scene_get_texture", C.c_int, [_VP, _U64, _VP]),
    ("rt_image_read_ppm", C.c_int, [C.c_char_p, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), _VP, _U64]),
    ("rt_scene_destroy", C.c_int, [_VP]),
    ("rt_scene_get_info", C.c_int, [_VP, C.POINTER(SceneInfo)]),
    ("rt_scene_get_prims", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_U64)]),
    ("rt_scene_get_lights", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_U32)]),
    ("rt_scene_get_bvh", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_U32), C.POINTER(_VP), C.POINTER(_U64)]),
    ("rt_scene_get_loose_boxes", C.c_int, [_VP, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("rt_scene_get_wide_bvh", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_U32)]),
    ("rt_scene_camera", C.c_int, [_VP, C.c_char_p, _I32, C.POINTER(Camera), C.c_char_p]),
    ("rt_prim_aabb", C.c_int, [_VP, _DP]),
    ("rt_camera_look", C.c_int, [_DP, _DP, _DP, C.POINTER(Camera)]),
    ("rt_camera_from_json", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(Camera)]),
    ("rt_camera_pixel_to_ray", C.c_int, [C.POINTER(Camera), _D, _D, _DP, _DP]),
    ("rt_ctx_create", C.c_int, [C.c_int, C.POINTER(_VP)]),
    ("rt_ctx_destroy", C.c_int, [_VP]),
    ("rt_ctx_set_stream", C.c_int, [_VP, _VP]),
    ("rt_ctx_use_own_stream", C.c_int, [_VP]),
    ("rt_ctx_synchronize", C.c_int, [_VP]),
    ("rt_scene_upload", C.c_int, [_VP, _VP]),
    ("rt_ctx_build_time", C.c_int, [_VP, C.POINTER(_D)]),
    ("rt_ctx_download_bvh", C.c_int, [_VP, _VP, _U32, C.POINTER(_U32), _VP, _U64, C.POINTER(_U64)]),
    ("rt_ctx_download_wide_bvh", C.c_int, [_VP, _VP, _U32, C.POINTER(_U32)]),
    ("rt_trace_primary", C.c_int, [_VP, C.POINTER(Camera), Rect, _VP, _VP]),
    ("rt_trace_rays", C.c_int, [_VP, _VP, _U64, _VP, _VP]),
    ("rt_shade_rays", C.c_int, [_VP, _VP, _U64, _VP]),
    ("rt_trace_rays_full", C.c_int, [_VP, _VP, _U64, _VP]),
    ("rt_shade_hits", C.c_int, [_VP, _VP, _VP, _U64, _VP]),
    ("rt_render_samples", C.c_int, [_VP, C.POINTER(Camera), Rect, C.c_int, _U32, C.c_int, _D, _VP]),
    ("rt_render_rect", C.c_int, [_VP, C.POINTER(Camera), Rect, C.c_int, _D, _VP, _VP]),
    ("rt_trace_shadows", C.c_int, [_VP, C.POINTER(Camera), Rect, _VP]),
    ("rt_render_rect_static", C.c_int, [_VP, C.POINTER(Camera), Rect, _VP, _VP]),
    ("rt_render_tiles", C.c_int, [_VP, C.POINTER(Camera), Tiling, C.c_int, C.c_int, _D, _VP]),
    ("rt_assemble_tiles", C.c_int, [_VP, Tiling, _VP, _VP]),
    ("rt_comm_unique_id", C.c_int, [_VP]),
    ("rt_comm_create", C.c_int, [_VP, C.c_int, C.c_int, _VP, C.POINTER(_VP)]),
    ("rt_comm_create_all", C.c_int, [C.POINTER(_VP), C.c_int, C.POINTER(_VP)]),
    ("rt_comm_destroy", C.c_int, [_VP]),
    ("rt_gather_tiles", C.c_int, [_VP, Tiling, _VP, _VP]),
    ("rt_gather_tiles_all", C.c_int, [C.POINTER(_VP), C.c_int, Tiling, C.POINTER(_VP), _VP]),
    ("rt_render_frame_multi_host", C.c_int, [C.POINTER(_VP), C.c_int, C.POINTER(Camera), C.c_int, C.c_int, _D, _VP]),
    ("rt_tiles_per_rank", _I32, [Tiling]),
    ("rt_tile_count", _I32, [Tiling]),
    ("rt_counters_reset", C.c_int, [_VP]),
    ("rt_counters_read", C.c_int, [_VP, C.POINTER(Counters)]),
    ("rt_ctx_set_counting", C.c_int, [_VP, C.c_int]),
    ("rt_ctx_set_timing", C.c_int, [_VP, C.c_int]),
    ("rt_ctx_set_entry_search", C.c_int, [_VP, C.c_int]),
    ("rt_ctx_set_brute_force", C.c_int, [_VP, C.c_int]),
    ("rt_probe_fetch_rate", C.c_int, [_VP, C.c_int, _U64, _U32, C.POINTER(C.c_double)]),
    ("rt_timing_read", C.c_int, [_VP, C.POINTER(Timing)]),
    ("rt_render_frame_host", C.c_int, [_VP, C.POINTER(Camera), C.c_int, _D, _VP, _VP]),
    ("rt_trace_primary_host", C.c_int, [_VP, C.POINTER(Camera), _VP, _VP]),
    ("rt_trace_rays_host", C.c_int, [_VP, _VP, _U64, _VP, _VP]),
    ("rt_trace_rays_full_host", C.c_int, [_VP, _VP, _U64, _VP]),
    ("rt_shade_hits_host", C.c_int, [_VP, _VP, _VP, _U64, _VP]),
    ("rt_intersect_prim_host", C.c_int, [_VP, _VP, _VP, _U64, _VP]),
    ("rt_render_samples_host", C.c_int, [_VP, C.POINTER(Camera), C.c_int, _U32, C.c_int, _D, _VP]),
    ("rt_gamma_lut", C.c_int, [_VP]),
    ("rt_strerror", C.c_char_p, [C.c_int]),
    ("rt_last_error", C.c_char_p, []),
    ("rt_abi_version", C.c_int, []),
]

_lib = None


def lib():
    """Load librt_hip.so; raises when it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librt_hip.so is not built (%s): run `python -m raytracer_amd.build`" % LIB_PATH)
        # A process must hold ONE HIP runtime.  torch wheels bundle their own libamdhip64.so.7 and look it up by
        # file name, so if the system copy is loaded first torch loads a second runtime and finds no GPU.  Loading
        # torch first makes librt_hip.so bind to torch's copy (same SONAME), which also makes torch's streams and
        # events valid handles for this library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_camera(location, direction, up=None, focal=35.0, sensor=(36.0, 24.0), res=(1920, 1080)):
    cam = Camera()
    pos = (C.c_double * 3)(*location)
    d = (C.c_double * 3)(*direction)
    u = (C.c_double * 3)(*up) if up is not None else None
    _check(lib().rt_camera_look(pos, d, u, C.byref(cam)))
    cam.focal, cam.sensor_w, cam.sensor_h = float(focal), float(sensor[0]), float(sensor[1])
    cam.width, cam.height = int(res[0]), int(res[1])
    return cam


def camera_from_json(path, camera_id):
    cam = Camera()
    _check(lib().rt_camera_from_json(path.encode(), str(camera_id).encode(), C.byref(cam)))
    return cam


def pixel_rays(cam, pxy):
    pxy = np.asarray(pxy, dtype=np.float64).reshape(-1, 2)
    out = np.empty((len(pxy), 6), dtype=np.float64)
    o, d = (C.c_double * 3)(), (C.c_double * 3)()
    for i, (x, y) in enumerate(pxy):
        _check(lib().rt_camera_pixel_to_ray(C.byref(cam), float(x), float(y), o, d))
        out[i, :3], out[i, 3:] = o[:], d[:]
    return out


def read_ppm(path):
    """(pixels (h, w, 3) uint8, maxval) through the library's PPM reader; raises RtError as the reference's Image throws."""
    w, h, mv = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib().rt_image_read_ppm(str(path).encode(), C.byref(w), C.byref(h), C.byref(mv), None, 0))
    px = np.empty((h.value, w.value, 3), dtype=np.uint8)
    _check(lib().rt_image_read_ppm(str(path).encode(), C.byref(w), C.byref(h), C.byref(mv), _ptr(px), px.size))
    return px, mv.value


def prim_aabb(prim):
    prim = np.ascontiguousarray(prim, dtype=PRIM_DT).reshape(1)
    out = (C.c_double * 6)()
    _check(lib().rt_prim_aabb(_ptr(prim), out))
    return np.array(out[:])


def gamma_lut():
    lut = np.empty(256, dtype=np.uint8)
    _check(lib().rt_gamma_lut(_ptr(lut)))
    return lut


class Scene:
    """Host-side scene: descriptions, flattened records, BVH (rt_scene)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def create(cls, prims, lights=None, max_leaf=0, sah_bins=0, on_device=False):
        """on_device: build no hierarchy on the host; Context.upload sorts and builds on the GPU."""
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, LIGHT_DT), dtype=LIGHT_DT)
        opts = BuildOpts(max_leaf, sah_bins, BUILD_ON_DEVICE if on_device else 0)
        h = C.c_void_p()
        _check(lib().rt_scene_create(_ptr(prims), len(prims), _ptr(lights), len(lights), C.byref(opts), C.byref(h)))
        return cls(h)

    @classmethod
    def create_instanced(cls, prims, instances, lights=None, max_leaf=0, sah_bins=0):
        """Two-level scene: prims is a pool of meshes, instances (INSTANCE_DT) place runs of it with rigid transforms."""
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        instances = np.ascontiguousarray(instances, dtype=INSTANCE_DT)
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, LIGHT_DT), dtype=LIGHT_DT)
        opts = BuildOpts(max_leaf, sah_bins, 0)
        h = C.c_void_p()
        _check(lib().rt_scene_create_instanced(_ptr(prims), len(prims), _ptr(instances), len(instances), _ptr(lights), len(lights),
                                               C.byref(opts), C.byref(h)))
        return cls(h)

    @classmethod
    def load_json(cls, path, max_leaf=0, sah_bins=0, on_device=False):
        opts = BuildOpts(max_leaf, sah_bins, BUILD_ON_DEVICE if on_device else 0)
        h = C.c_void_p()
        _check(lib().rt_scene_load_json(str(path).encode(), C.byref(opts), C.byref(h)))
        return cls(h)

    def set_textures(self, textures, prim_texture):
        """textures: list of (h, w, 3) uint8 arrays; prim_texture: index per primitive, -1 = none."""
        arrs = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        table = (Texture * max(len(arrs), 1))()
        for k, a in enumerate(arrs):
            assert a.ndim == 3 and a.shape[2] == 3
            table[k] = Texture(a.shape[1], a.shape[0], a.ctypes.data)
        pt = np.ascontiguousarray(prim_texture, dtype=np.int32)
        if len(arrs) and len(pt) != int(self.info().n_prims):      # the library reads one entry per primitive
            raise ValueError("prim_texture needs one entry per primitive (%d), got %d" % (int(self.info().n_prims), len(pt)))
        _check(lib().rt_scene_set_textures(self.h, table, len(arrs), _ptr(pt)))

    def texture(self, prim_id):
        t = Texture()
        _check(lib().rt_scene_get_texture(self.h, prim_id, C.byref(t)))
        if t.width == 0:
            return None
        buf = (C.c_uint8 * (t.width * t.height * 3)).from_address(t.rgb)
        return np.frombuffer(buf, dtype=np.uint8).reshape(t.height, t.width, 3).copy()

    @classmethod
    def load_json_text(cls, text, max_leaf=0, sah_bins=0):
        data = text.encode() if isinstance(text, str) else bytes(text)
        opts = BuildOpts(max_leaf, sah_bins)
        h = C.c_void_p()
        _check(lib().rt_scene_load_json_text(data, len(data), C.byref(opts), C.byref(h)))
        return cls(h)

    def close(self):
        if self.h:
            lib().rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = SceneInfo()
        _check(lib().rt_scene_get_info(self.h, C.byref(i)))
        return i

    def prims(self):
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().rt_scene_get_prims(self.h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PRIM_DT)
        buf = (C.c_char * (n.value * PRIM_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=PRIM_DT).copy()

    def lights(self):
        p, n = C.c_void_p(), C.c_uint32()
        _check(lib().rt_scene_get_lights(self.h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, LIGHT_DT)
        buf = (C.c_char * (n.value * LIGHT_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=LIGHT_DT).copy()

    def bvh(self):
        """(nodes, slot_prim incl. loose primitives, n_slots, n_loose, radius) as numpy copies."""
        pn, nn, ps, ns = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint64()
        _check(lib().rt_scene_get_bvh(self.h, C.byref(pn), C.byref(nn), C.byref(ps), C.byref(ns)))
        info = self.info()
        nodes = np.zeros(0, NODE_DT)
        if nn.value:
            nodes = np.frombuffer((C.c_char * (nn.value * 32)).from_address(pn.value), dtype=NODE_DT).copy()
        total = ns.value + info.n_loose
        slots = np.zeros(0, np.uint32)
        if total:
            slots = np.frombuffer((C.c_char * (total * 4)).from_address(ps.value), dtype=np.uint32).copy()
        return nodes, slots, int(ns.value), int(info.n_loose), float(np.nextafter(np.float32(info.radius), np.float32(np.inf)))

    def loose_boxes(self):
        """(n_loose, 8) float32: boxes of the primitives kept beside the tree (min xyz, -, max xyz, -)."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().rt_scene_get_loose_boxes(self.h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 8), np.float32)
        return np.frombuffer((C.c_char * (n.value * 32)).from_address(p.value), dtype=np.float32).reshape(-1, 8).copy()

    def n_wide(self):
        """Number of wide nodes (0: the kernels walk the binary tree)."""
        n = C.c_uint32()
        _check(lib().rt_scene_get_wide_bvh(self.h, None, C.byref(n)))
        return int(n.value)

    def wide_bvh(self):
        """The wide nodes the kernels walk (NODE4_DT), empty when the scene is walked through the binary tree."""
        p, n = C.c_void_p(), C.c_uint32()
        _check(lib().rt_scene_get_wide_bvh(self.h, C.byref(p), C.byref(n)))
        if not n.value:
            return np.zeros(0, NODE4_DT)
        buf = (C.c_char * (n.value * NODE4_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=NODE4_DT).copy()

    def camera(self, camera_id=None, index=-1):
        cam = Camera()
        ident = C.create_string_buffer(32)
        cid = str(camera_id).encode() if camera_id is not None else None
        _check(lib().rt_scene_camera(self.h, cid, index, C.byref(cam), ident))
        return cam, ident.value.decode()


class Context:
    """One GPU (rt_ctx)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().rt_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            lib().rt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """hip_stream: integer handle (0 = HIP's default stream, which is torch's default stream)."""
        _check(lib().rt_ctx_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def use_own_stream(self):
        _check(lib().rt_ctx_use_own_stream(self.h))

    def synchronize(self):
        _check(lib().rt_ctx_synchronize(self.h))

    def upload(self, scene):
        _check(lib().rt_scene_upload(self.h, scene.h))

    def build_time_ms(self):
        ms = C.c_double()
        _check(lib().rt_ctx_build_time(self.h, C.byref(ms)))
        return ms.value

    def download_bvh(self):
        """(nodes, slot_prim) of the uploaded hierarchy as the kernels walk it (host- or device-built)."""
        nn, ns = C.c_uint32(), C.c_uint64()
        _check(lib().rt_ctx_download_bvh(self.h, None, 0, C.byref(nn), None, 0, C.byref(ns)))
        nodes = np.zeros(nn.value, dtype=NODE_DT)
        slots = np.zeros(ns.value, dtype=np.uint32)
        _check(lib().rt_ctx_download_bvh(self.h, _ptr(nodes), len(nodes), C.byref(nn), _ptr(slots), len(slots), C.byref(ns)))
        return nodes, slots

    def download_wide_bvh(self):
        """The wide nodes the kernels walk on this GPU (NODE4_DT), empty when they walk the binary tree."""
        n = C.c_uint32()
        _check(lib().rt_ctx_download_wide_bvh(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=NODE4_DT)
        if n.value:
            _check(lib().rt_ctx_download_wide_bvh(self.h, _ptr(out), n.value, C.byref(n)))
        return out

    def set_counting(self, on):
        _check(lib().rt_ctx_set_counting(self.h, 1 if on else 0))

    def set_entry_search(self, on):
        """Start each tile's walks at the deepest node its beam of rays must pass through (default on)."""
        _check(lib().rt_ctx_set_entry_search(self.h, 1 if on else 0))

    def set_brute_force(self, on):
        """Closest-hit entry points test every primitive against every ray (the reference's one_pass_intersection)."""
        _check(lib().rt_ctx_set_brute_force(self.h, 1 if on else 0))

    def probe_fetch_rate(self, pattern=0, table_bytes=16384, steps=4096):
        """Nodes (pattern 2: 16-byte loads) per second the GPU delivers to lanes with the walk's load sequence and nothing
        else: pattern 0 every lane its own node of the table, 1 one node per wave, 2 streaming."""
        out = C.c_double(0.0)
        _check(lib().rt_probe_fetch_rate(self.h, int(pattern), int(table_bytes), int(steps), C.byref(out)))
        return out.value

    def set_timing(self, on):
        _check(lib().rt_ctx_set_timing(self.h, 1 if on else 0))

    def timing(self):
        """Device milliseconds and launch counts per pipeline stage since the last call (synchronises)."""
        t = Timing()
        _check(lib().rt_timing_read(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timing._fields_}

    def reset_counters(self):
        _check(lib().rt_counters_reset(self.h))

    def counters(self):
        c = Counters()
        _check(lib().rt_counters_read(self.h, C.byref(c)))
        return c.as_dict()

    # -- device-pointer entry points (ints from torch's data_ptr() or any hipMalloc) ------------------
    def trace_primary_dev(self, cam, rect, ids_ptr, dist_ptr):
        _check(lib().rt_trace_primary(self.h, C.byref(cam), Rect(*rect), ids_ptr, dist_ptr))

    def trace_rays_dev(self, rays_ptr, n, ids_ptr, dist_ptr):
        _check(lib().rt_trace_rays(self.h, rays_ptr, n, ids_ptr, dist_ptr))

    def shade_rays_dev(self, rays_ptr, n, rgb_ptr):
        _check(lib().rt_shade_rays(self.h, rays_ptr, n, rgb_ptr))

    def trace_rays_full_dev(self, rays_ptr, n, records_ptr):
        _check(lib().rt_trace_rays_full(self.h, rays_ptr, n, records_ptr))

    def shade_hits_dev(self, hits_ptr, view_ptr, n, rgb_ptr):
        _check(lib().rt_shade_hits(self.h, hits_ptr, view_ptr, n, rgb_ptr))

    def render_samples_dev(self, cam, rect, spp, seed, rgb_ptr, max_scatters=0, min_scatter_intensity=0.0):
        _check(lib().rt_render_samples(self.h, C.byref(cam), Rect(*rect), spp, seed, max_scatters, min_scatter_intensity, rgb_ptr))

    def render_rect_dev(self, cam, rect, rgb_ptr, ids_ptr=None, max_scatters=0, min_scatter_intensity=0.0):
        _check(lib().rt_render_rect(self.h, C.byref(cam), Rect(*rect), max_scatters, min_scatter_intensity, rgb_ptr, ids_ptr))

    def trace_shadows_dev(self, cam, rect, mask_ptr):
        """One byte per pixel: bit l set = light l reaches the pixel's primary hit."""
        _check(lib().rt_trace_shadows(self.h, C.byref(cam), Rect(*rect), mask_ptr))

    def render_rect_static_dev(self, cam, rect, rgb_ptr, ids_ptr=None):
        """rt_render_rect (no bounces) on the library's fixed workspace: no device allocation; at most 262144 pixels."""
        _check(lib().rt_render_rect_static(self.h, C.byref(cam), Rect(*rect), rgb_ptr, ids_ptr))

    def render_tiles_dev(self, cam, tiling, rank, rgb_ptr, max_scatters=0, min_scatter_intensity=0.0):
        _check(lib().rt_render_tiles(self.h, C.byref(cam), tiling, rank, max_scatters, min_scatter_intensity, rgb_ptr))

    def assemble_tiles_dev(self, tiling, gathered_ptr, frame_ptr):
        _check(lib().rt_assemble_tiles(self.h, tiling, gathered_ptr, frame_ptr))

    # -- host-memory conveniences -----------------------------------------------------------------------
    def render(self, cam, max_scatters=2, min_scatter_intensity=0.0, want_ids=False):
        rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
        ids = np.empty((cam.height, cam.width), dtype=np.int32) if want_ids else None
        _check(lib().rt_render_frame_host(self.h, C.byref(cam), max_scatters, min_scatter_intensity, _ptr(rgb), _ptr(ids)))
        return (rgb, ids) if want_ids else rgb

    def trace_primary(self, cam):
        ids = np.empty((cam.height, cam.width), dtype=np.int32)
        dist = np.empty((cam.height, cam.width), dtype=np.float64)
        _check(lib().rt_trace_primary_host(self.h, C.byref(cam), _ptr(ids), _ptr(dist)))
        return ids, dist

    def trace_rays(self, rays6):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        ids = np.empty(len(rays6), dtype=np.int32)
        dist = np.empty(len(rays6), dtype=np.float64)
        _check(lib().rt_trace_rays_host(self.h, _ptr(rays6), len(rays6), _ptr(ids), _ptr(dist)))
        return ids, dist

    def trace_rays_full(self, rays6):
        """(n, 16) float64 records: hit flag, id, distance, point, normal, reflected direction, u, v, 0, 0."""
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        rec = np.empty((len(rays6), 16), dtype=np.float64)
        _check(lib().rt_trace_rays_full_host(self.h, _ptr(rays6), len(rays6), _ptr(rec)))
        return rec

    def intersect_prim(self, prim, rays6):
        """Mesh::intersect of one primitive (a PRIM_DT record) for n rays; (n, 16) records."""
        prim = np.ascontiguousarray(prim).reshape(1)
        rays6 = np.ascontiguousarray(rays6, dtype=np.float64).reshape(-1, 6)
        rec = np.empty((len(rays6), 16), dtype=np.float64)
        _check(lib().rt_intersect_prim_host(self.h, _ptr(prim), _ptr(rays6), len(rays6), _ptr(rec)))
        return rec

    def shade_hits(self, hits10, view3):
        """hits10: point, normal, primitive id, u, v, has-uv flag per hit; view3: origin of the ray that found it."""
        hits10 = np.ascontiguousarray(hits10, dtype=np.float64).reshape(-1, 10)
        view3 = np.ascontiguousarray(view3, dtype=np.float64).reshape(-1, 3)
        assert len(hits10) == len(view3)
        rgb = np.empty((len(hits10), 3), dtype=np.uint8)
        _check(lib().rt_shade_hits_host(self.h, _ptr(hits10), _ptr(view3), len(hits10), _ptr(rgb)))
        return rgb

    def render_samples(self, cam, spp, seed=1, max_scatters=2, min_scatter_intensity=0.0):
        rgb = np.empty((cam.height, cam.width, 3), dtype=np.uint8)
        _check(lib().rt_render_samples_host(self.h, C.byref(cam), spp, seed, max_scatters, min_scatter_intensity, _ptr(rgb)))
        return rgb


def comm_unique_id():
    """128 bytes that name a new group of ranks (rank 0 makes them, every rank passes them to Comm)."""
    buf = (C.c_uint8 * 128)()
    _check(lib().rt_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """One rank of a group of GPUs (rt_comm): RCCL point-to-point gather of tile buffers to rank 0."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        assert len(unique_id) == 128
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().rt_comm_create(ctx.h, n_ranks, rank, buf, C.byref(h)))
        self.h, self.ctx, self.n_ranks, self.rank = h, ctx, n_ranks, rank

    def gather_tiles(self, tiling, tiles_ptr, frame_ptr=None):
        _check(lib().rt_gather_tiles(self.h, tiling, tiles_ptr, frame_ptr))

    def close(self):
        if self.h:
            lib().rt_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tiles_per_rank(tiling):
    return int(lib().rt_tiles_per_rank(tiling))


def tile_count(tiling):
    return int(lib().rt_tile_count(tiling))
