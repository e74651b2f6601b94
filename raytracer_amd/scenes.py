"""Deterministic synthetic scenes and the scene-description record shared with the C ABI.

Primitive kinds and their id order follow the reference's flat primitive list
(cubes, planes, cylinders, spheres: s2899221/Code/raytracer.cpp:20-27); triangles are an extension
and always come last.  All generators are pure functions of their seed (splitmix64), so the same
scene is rebuilt bit-for-bit on any machine.
"""
import json

import numpy as np

CUBE, PLANE, CYLINDER, SPHERE, TRIANGLE = 0, 1, 2, 3, 4
KIND_NAMES = {CUBE: "Cube", PLANE: "Plane", CYLINDER: "Cylinder", SPHERE: "Sphere", TRIANGLE: "Triangle"}

# rt_prim_desc / rt_light of include/rt_hip.h
PRIM_DT = np.dtype([("kind", "<i4"), ("reserved", "<i4"), ("q", "<f8", (12,)),
                    ("alpha", "<f8"), ("metallic", "<f8"), ("roughness", "<f8"), ("ior", "<f8"),
                    ("albedo", "<f8", (3,))])
LIGHT_DT = np.dtype([("pos", "<f8", (3,)), ("intensity", "<f8")])
# rt_instance: x_world = rot * x_mesh + trans for the run [first_prim, first_prim + n_prims) of the primitive pool
INSTANCE_DT = np.dtype([("rot", "<f8", (9,)), ("trans", "<f8", (3,)), ("first_prim", "<u4"), ("n_prims", "<u4")])


def new_prims(n):
    """n primitives with the reference's default material (mesh.h:17, mesh.cpp:79-80)."""
    p = np.zeros(n, dtype=PRIM_DT)
    p["alpha"] = 1.0
    p["roughness"] = 0.5
    p["ior"] = 1.5
    p["albedo"] = 1.0
    return p


def sort_reference_order(prims):
    """Stable sort into the reference's id order (kind ascending, original order inside a kind)."""
    return prims[np.argsort(prims["kind"], kind="stable")]


class SplitMix64:
    """Counter-based splitmix64 stream; `uniform(n)` returns the next n doubles in [0, 1)."""

    def __init__(self, seed):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def next_u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = z[-1] if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self.next_u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u


def rotation_rpy(roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll), vectorised; used only to place triangle corners."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.empty(np.shape(roll) + (3, 3))
    R[..., 0, 0] = cy * cp
    R[..., 0, 1] = cy * sp * sr - sy * cr
    R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp
    R[..., 1, 1] = sy * sp * sr + cy * cr
    R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp
    R[..., 2, 1] = cp * sr
    R[..., 2, 2] = cp * cr
    return R


_CUBE_TRIS = np.array([  # corner index = 4*(x>0) + 2*(y>0) + (z>0); two triangles per face
    (0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1),
    (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)])


def cubes_to_triangles(cubes):
    """12 triangles per cube, triangle k of cube i at index 12*i + k."""
    q = cubes["q"]
    R = rotation_rpy(q[:, 3], q[:, 4], q[:, 5])
    signs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    local = signs[None, :, :] * q[:, None, 6:9]
    corners = np.einsum("nij,nkj->nki", R, local) + q[:, None, 0:3]
    tri = corners[:, _CUBE_TRIS, :]                       # n, 12, 3 corners, xyz
    out = new_prims(len(cubes) * 12)
    out["kind"] = TRIANGLE
    out["q"][:, :9] = tri.reshape(-1, 9)
    for f in ("alpha", "metallic", "roughness", "ior", "albedo"):
        out[f] = np.repeat(cubes[f], 12, axis=0)
    return out


def cube_field(n_cubes=8334, seed=0xC0FFEE, grid=(96, 87), half_extent=50.0):
    """Jittered grid of randomly rotated, randomly scaled cubes on x,y in [-half_extent, half_extent]."""
    rng = SplitMix64(seed)
    gx, gy = grid
    assert n_cubes <= gx * gy
    u = rng.uniform(n_cubes * 9).reshape(n_cubes, 9)
    cell = np.arange(n_cubes)
    cx, cy = cell % gx, cell // gx
    p = new_prims(n_cubes)
    p["kind"] = CUBE
    p["q"][:, 0] = -half_extent + (cx + u[:, 0]) * (2 * half_extent / gx)
    p["q"][:, 1] = -half_extent + (cy + u[:, 1]) * (2 * half_extent / gy)
    p["q"][:, 2] = 2.0 * u[:, 2]
    p["q"][:, 3:6] = -np.pi + 2 * np.pi * u[:, 3:6]
    p["q"][:, 6:9] = 0.10 + 0.35 * u[:, 6:9]
    return p


CUBE_FIELD_CAMERA = dict(location=[0.0, -90.0, 45.0], direction=[0.0, 0.8, -0.45], up=None, focal=35.0,
                         sensor=(36.0, 24.0), res=(1920, 1080))


def mixed_field(n_side=1000, seed=0x100000, half_extent=None, planes_as_triangles=True, ground=True):
    """n_side x n_side jittered grid with one primitive per cell chosen by a uniform draw:
    u < 1/3 parallelogram (two triangles, or one reference Plane when planes_as_triangles is False),
    u < 2/3 ellipsoid, else oriented cube; plus a ground plane.  Cell pitch is 1 world unit.
    Returned in reference id order."""
    n = n_side * n_side
    half = float(half_extent) if half_extent is not None else n_side / 2.0
    pitch = 2.0 * half / n_side
    rng = SplitMix64(seed)
    u = rng.uniform(n * 12).reshape(n, 12)
    cell = np.arange(n)
    cx, cy = cell % n_side, cell // n_side
    centre = np.stack([-half + (cx + u[:, 1]) * pitch, -half + (cy + u[:, 2]) * pitch, 2.0 * u[:, 3]], axis=1)
    kind_draw = u[:, 0]
    is_par = kind_draw < 1.0 / 3.0
    is_sph = (~is_par) & (kind_draw < 2.0 / 3.0)
    is_cub = ~(is_par | is_sph)
    scale = 0.10 + 0.35 * u[:, 4:7]
    parts = []

    cubes = new_prims(int(is_cub.sum()))
    cubes["kind"] = CUBE
    cubes["q"][:, 0:3] = centre[is_cub]
    cubes["q"][:, 3:6] = -np.pi + 2 * np.pi * u[is_cub, 7:10]
    cubes["q"][:, 6:9] = scale[is_cub]
    parts.append(cubes)

    # parallelogram: two edge vectors from random directions, lengths in [0.2, 0.9]
    m = int(is_par.sum())
    up = u[is_par]
    def edge(a, b, ln):
        z = 2.0 * a - 1.0
        r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        ph = 2.0 * np.pi * b
        return np.stack([r * np.cos(ph), r * np.sin(ph), z], axis=1) * (0.2 + 0.7 * ln)[:, None]
    e1 = edge(up[:, 7], up[:, 8], up[:, 4])
    e2 = edge(up[:, 9], up[:, 10], up[:, 5])
    p0 = centre[is_par] - 0.5 * (e1 + e2)
    p1, p3, p2 = p0 + e1, p0 + e2, p0 + e1 + e2
    if planes_as_triangles:
        tris = new_prims(2 * m)
        tris["kind"] = TRIANGLE
        tris["q"][0::2, :9] = np.concatenate([p0, p1, p3], axis=1)
        tris["q"][1::2, :9] = np.concatenate([p2, p3, p1], axis=1)
        quads = new_prims(0)
    else:
        quads = new_prims(m)
        quads["kind"] = PLANE
        quads["q"][:, :12] = np.concatenate([p0, p1, p2, p3], axis=1)
        tris = new_prims(0)
    if ground:
        g = new_prims(1)
        g["kind"] = PLANE
        e = half * 1.05
        g["q"][0, :12] = [-e, -e, -0.5, e, -e, -0.5, e, e, -0.5, -e, e, -0.5]
        quads = np.concatenate([quads, g])
    parts.append(quads)

    sph = new_prims(int(is_sph.sum()))
    sph["kind"] = SPHERE
    sph["q"][:, 0:3] = centre[is_sph]
    sph["q"][:, 3:6] = scale[is_sph]
    parts.append(sph)
    parts.append(tris)
    prims = np.concatenate(parts)
    # a little material variety, still deterministic
    mrng = SplitMix64(seed ^ 0xABCDEF)
    mu = mrng.uniform(len(prims) * 5).reshape(len(prims), 5)
    prims["albedo"] = 0.25 + 0.75 * mu[:, 0:3]
    prims["metallic"] = np.where(mu[:, 3] < 0.2, 1.0, 0.0)
    prims["roughness"] = 0.2 + 0.8 * mu[:, 4]
    return sort_reference_order(prims)


def mixed_field_lights(half_extent=500.0, height=120.0, intensity=4000.0):
    L = np.zeros(4, dtype=LIGHT_DT)
    h = half_extent / 2.0
    L["pos"] = [[h, h, height], [-h, h, height], [h, -h, height], [-h, -h, height]]
    L["intensity"] = intensity
    return L


def mixed_field_camera(half_extent=500.0, res=(3840, 2160)):
    s = half_extent / 500.0
    return dict(location=[0.0, -700.0 * s, 350.0 * s], direction=[0.0, 0.8, -0.45], up=None, focal=35.0,
                sensor=(36.0, 24.0), res=tuple(res))


def make_instances(rot, trans, first_prim, n_prims):
    """INSTANCE_DT records from (n, 3, 3) rotations, (n, 3) translations and the mesh run(s) they place."""
    inst = np.zeros(len(trans), dtype=INSTANCE_DT)
    inst["rot"] = np.asarray(rot, dtype=np.float64).reshape(-1, 9)
    inst["trans"] = trans
    inst["first_prim"] = first_prim
    inst["n_prims"] = n_prims
    return inst


def instanced_field(n_side=100, mesh_cubes=834, seed=0x5, pitch=32.0, mirror=True):
    """One mesh - a cluster of mesh_cubes rotated cubes as 12 triangles each, about 30 x 30 units - placed n_side x n_side
    times on a jittered grid, every instance with its own yaw, a small tilt and a lift.  With the defaults: 10 008
    triangles x 10 000 instances.  mirror: every material metallic 1, roughness 0.001.
    Returns (prims, instances, lights, camera dict)."""
    side = int(np.ceil(np.sqrt(mesh_cubes)))
    cubes = cube_field(n_cubes=mesh_cubes, seed=seed, grid=(side, side), half_extent=side * 1.04 / 2.0)
    mrng = SplitMix64(seed ^ 0x51ED)
    cubes["albedo"] = 0.35 + 0.65 * mrng.uniform(mesh_cubes * 3).reshape(mesh_cubes, 3)
    if mirror:
        cubes["metallic"] = 1.0
        cubes["roughness"] = 0.001
    else:
        cubes["metallic"] = np.where(mrng.uniform(mesh_cubes) < 0.3, 1.0, 0.0)
    prims = cubes_to_triangles(cubes)
    n = n_side * n_side
    rng = SplitMix64(seed * 7919 + 13)
    u = rng.uniform(n * 6).reshape(n, 6)
    cell = np.arange(n)
    half = n_side * pitch / 2.0
    trans = np.stack([-half + (cell % n_side + 0.25 + 0.5 * u[:, 0]) * pitch,
                      -half + (cell // n_side + 0.25 + 0.5 * u[:, 1]) * pitch, 3.0 * u[:, 2]], axis=1)
    rot = rotation_rpy(0.25 * (u[:, 3] - 0.5), 0.25 * (u[:, 4] - 0.5), -np.pi + 2 * np.pi * u[:, 5])
    inst = make_instances(rot, trans, 0, len(prims))
    s = half / 500.0
    lights = mixed_field_lights(half_extent=half, height=120.0 * s, intensity=4000.0 * s * s)
    return prims, inst, lights, mixed_field_camera(half_extent=half)


def random_scene(n, seed, kinds=(CUBE, PLANE, CYLINDER, SPHERE), extent=6.0, lights=2):
    """Small cluttered test scene of mixed reference primitives around the origin, with varied materials."""
    rng = SplitMix64(seed)
    u = rng.uniform(n * 24).reshape(n, 24)
    p = new_prims(n)
    k = np.asarray(kinds)[np.minimum((u[:, 0] * len(kinds)).astype(int), len(kinds) - 1)]
    p["kind"] = k
    c = (u[:, 1:4] - 0.5) * extent
    for i in range(n):
        if k[i] in (CUBE, CYLINDER):
            p["q"][i, :9] = list(c[i]) + list(-np.pi + 2 * np.pi * u[i, 4:7]) + list(0.15 + 0.6 * u[i, 7:10])
        elif k[i] == SPHERE:
            p["q"][i, :6] = list(c[i]) + list(0.15 + 0.6 * u[i, 7:10])
        else:
            e1 = (u[i, 4:7] - 0.5) * 3.0
            e2 = (u[i, 7:10] - 0.5) * 3.0
            p0 = c[i] - 0.5 * (e1 + e2)
            if k[i] == PLANE:
                p["q"][i, :12] = list(p0) + list(p0 + e1) + list(p0 + e1 + e2) + list(p0 + e2)
            else:
                p["q"][i, :9] = list(p0) + list(p0 + e1) + list(p0 + e2)
    p["albedo"] = 0.2 + 0.8 * u[:, 10:13]
    p["metallic"] = np.where(u[:, 13] < 0.3, u[:, 14], 0.0)
    p["roughness"] = np.where(u[:, 15] < 0.2, 0.0, u[:, 16])
    p["ior"] = 1.0 + u[:, 17]
    L = np.zeros(lights, dtype=LIGHT_DT)
    lu = rng.uniform(lights * 4).reshape(lights, 4)
    L["pos"] = (lu[:, :3] - 0.5) * 2.5 * extent
    L["pos"][:, 2] = np.abs(L["pos"][:, 2]) + extent * 0.6
    L["intensity"] = 20.0 + 60.0 * lu[:, 3]
    cam = dict(location=[0.3, -2.2 * extent, 0.9 * extent], direction=[-0.02, 0.9, -0.38], up=None, focal=35.0,
               sensor=(36.0, 24.0), res=(160, 90))
    return sort_reference_order(p), L, cam


# -------------------------------------------------------------------------------------------------
# Blender-export JSON writer (format of s2899221/ASCII/*.json; producer s2899221/Blend/Export.py)

def _v(x, y, z):
    return {"x": float(x), "y": float(y), "z": float(z)}


def make_texture(seed, width, height, maxval=255):
    """Deterministic (height, width, 3) uint8 texture with samples in [0, maxval]."""
    rng = SplitMix64(seed)
    u = rng.uniform(width * height * 3).reshape(height, width, 3)
    y, x = np.mgrid[0:height, 0:width]
    checker = (((x // 2) + (y // 2)) % 2)[..., None]
    return np.clip(np.floor((0.25 + 0.5 * checker + 0.25 * u) * (maxval + 1)), 0, maxval).astype(np.uint8)


def write_ppm(path, pixels, maxval=255, ascii_format=False, comments=False):
    """P6 (or P3) file as the reference's Image reads it: samples are stored as they are, never rescaled."""
    h, w, _ = pixels.shape
    head = "%s\n%s%d %d\n%s%d\n" % ("P3" if ascii_format else "P6", "# made by scenes.write_ppm\n" if comments else "",
                                  w, h, "# maxval next\n" if comments else "", maxval)
    with open(path, "wb") as f:
        f.write(head.encode())
        if ascii_format:
            rows = [" ".join(str(int(v)) for v in row.reshape(-1)) for row in pixels]
            f.write(("\n".join(rows) + "\n").encode())
        else:
            f.write(np.ascontiguousarray(pixels, dtype=np.uint8).tobytes())


def textured_scene(directory):
    """Seeded scene of all four reference kinds where most primitives carry an albedo texture.  Writes the texture
    files into `directory` and returns (prims, lights, camera dict, texture arrays, texture index per primitive,
    texture path per primitive).  Two primitives name files that cannot be loaded and keep their solid colour."""
    import os
    prims, lights, cam = random_scene(36, 77, lights=2)
    cam = dict(cam, res=(320, 180))
    prims["metallic"][::3] = 0.6                        # textured metals: the tint and F0 paths
    specs = [(8, 8, 255, False, False), (5, 3, 100, True, True), (1, 1, 255, False, False), (64, 32, 255, False, True), (2, 2, 255, True, False)]
    textures, files = [], []
    for k, (w, h, mv, asc, com) in enumerate(specs):
        t = make_texture(1000 + k, w, h, mv)
        p = os.path.join(directory, "tex%d.ppm" % k)
        write_ppm(p, t, mv, asc, com)
        textures.append(t); files.append(p)
    broken = os.path.join(directory, "broken.ppm")
    with open(broken, "wb") as f:
        f.write(b"P6\n4 4\n255\nshort")
    prim_tex = np.full(len(prims), -1, dtype=np.int32)
    paths = [""] * len(prims)
    for i in range(len(prims)):
        if i % 6 == 5:
            continue                                    # solid colour
        if i == 7:
            paths[i] = os.path.join(directory, "does_not_exist.ppm")
        elif i == 13:
            paths[i] = broken
        else:
            prim_tex[i] = i % len(specs)
            paths[i] = files[prim_tex[i]]
    return prims, lights, cam, textures, prim_tex, paths


def scene_to_json(prims, lights, cameras, path=None, texture_paths=None):
    """Serialise reference-kind primitives (no triangles) to the scene format the reference loads.
    repr()-style floats round-trip exactly, so the loaded scene equals the in-memory one bit for bit."""
    mesh = {"Cube": {}, "Sphere": {}, "Plane": {}, "Cylinder": {}}
    next_id = 1
    for index, p in enumerate(prims):
        k = int(p["kind"])
        if k == TRIANGLE:
            raise ValueError("the reference scene format has no triangle")
        q = p["q"]
        mat = {"alpha": float(p["alpha"]), "metallic": float(p["metallic"]), "roughness": float(p["roughness"]),
               "ior": float(p["ior"]), "albedo": _v(*p["albedo"])}
        if texture_paths is not None and texture_paths[index]:
            mat["albedo_texture_ppm"] = texture_paths[index]
        if k in (CUBE, CYLINDER):
            obj = {"translation": _v(*q[0:3]), "rotation": {"roll": float(q[3]), "pitch": float(q[4]), "yaw": float(q[5])},
                   "scale": _v(*q[6:9]), "material": mat}
        elif k == SPHERE:
            obj = {"location": _v(*q[0:3]), "scale": _v(*q[3:6]), "material": mat}
        else:
            obj = {"corners": [_v(*q[3 * c:3 * c + 3]) for c in range(4)], "material": mat}
        mesh[KIND_NAMES[k]][str(next_id)] = obj
        next_id += 1
    doc = {"MESH": {k: v for k, v in mesh.items() if v}}
    if len(lights):
        doc["LIGHT"] = {"POINT": {}}
        for L in lights:
            doc["LIGHT"]["POINT"][str(next_id)] = {"location": _v(*L["pos"]), "radiant_intensity": float(L["intensity"])}
            next_id += 1
    doc["CAMERA"] = {"PERSP": {}}
    for cid, c in cameras.items():
        e = {"location": _v(*c["location"]), "direction": _v(*c["direction"]), "focal_length": float(c["focal"]),
             "sensor_width": float(c["sensor"][0]), "sensor_height": float(c["sensor"][1]),
             "film_resolution": {"x": int(c["res"][0]), "y": int(c["res"][1])}}
        if c.get("up") is not None:
            e["up"] = _v(*c["up"])
        doc["CAMERA"]["PERSP"][str(cid)] = e
    text = json.dumps(doc, indent=1)
    if path:
        with open(path, "w") as f:
            f.write(text)
    return text
