// TEST INFRASTRUCTURE ONLY - never linked into the product library.
//
// Thin extern "C" shim around the *real* reference engine.  It is compiled in
// the development container together with the reference's own translation
// units, taken in place from $(REF)/s2899221/Code (see oracle/Makefile), and
// lands in oracle/_ref/libref_rt.so.  Nothing of the reference is copied into
// this repository; this file only calls its public API:
//
//   RayTracer::load_from_json            s2899221/Code/raytracer.h:25
//   RayTracer::get_one_ray_per_pixel     s2899221/Code/raytracer.h:28
//   RayTracer::one_pass_intersection     s2899221/Code/raytracer.h:31
//   RayTracer::shade                     s2899221/Code/raytracer.h:34
//   RayTracer::render_unaccelerated_ppm  s2899221/Code/raytracer.h:41
//   get_BVH_tree / accelerated_one_pass_intersection  raytracer.h:78,81
//   Cube/Sphere/Plane/Cylinder ctors + intersect + compute_aabb  mesh.h
//   Camera::pixelToRay                   s2899221/Code/persp_camera.h:60
//
// Uses: (1) generate the golden fixtures under tests/golden/, (2) cross-check
// the plain-C restatement in oracle/rt_oracle.c, (3) optional "reference"
// CPU baseline.  The library is absent on the GPU box.

#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "raytracer.h"

namespace {

struct RefScene {
    RayTracer rt;
    std::vector<std::unique_ptr<BoundingBox>> node_store;
    const Mesh* bvh_root = nullptr;
    bool bvh_built = false;
};

struct HitRecord {        // 16 doubles, must match REC_DOUBLES in oracle/refbind.py
    double hit;           // 1.0 / 0.0
    double id;            // index in get_scene_mesh_ptrs() order, -1 on miss
    double dist;
    double p[3];
    double n[3];
    double r[3];
    double u, v;
    double pad[2];
};
static_assert(sizeof(HitRecord) == 16 * sizeof(double), "layout");

int mesh_index(const std::vector<const Mesh*>& ptrs, const Mesh* m) {
    for (size_t i = 0; i < ptrs.size(); ++i) if (ptrs[i] == m) return static_cast<int>(i);
    return -1;
}

void fill_record(const std::vector<const Mesh*>& ptrs, bool ok, const Hit& h, HitRecord& out) {
    std::memset(&out, 0, sizeof(out));
    out.id = -1.0;
    out.dist = std::numeric_limits<double>::infinity();
    if (!ok) return;
    out.hit = 1.0;
    if (h.hasMesh()) out.id = mesh_index(ptrs, h.getMesh());
    out.dist = h.getDistanceAlongRay();
    const HitVec3& p = h.getIntersectionPoint();
    const HitVec3& n = h.getSurfaceNormal();
    const HitVec3& r = h.getReflectedDirection();
    out.p[0] = p.x; out.p[1] = p.y; out.p[2] = p.z;
    out.n[0] = n.x; out.n[1] = n.y; out.n[2] = n.z;
    out.r[0] = r.x; out.r[1] = r.y; out.r[2] = r.z;
    if (h.hasUV()) { out.u = h.getU(); out.v = h.getV(); }
}

Ray make_ray(const double* r6) {
    return Ray(RayVec3{ r6[0], r6[1], r6[2] }, RayVec3{ r6[3], r6[4], r6[5] });
}

} // namespace

extern "C" {

void* ref_scene_load(const char* json_path) {
    auto* s = new RefScene();
    if (!s->rt.load_from_json(json_path)) { delete s; return nullptr; }
    return s;
}

void ref_scene_free(void* scene) { delete static_cast<RefScene*>(scene); }

int ref_scene_num_meshes(void* scene) {
    return static_cast<int>(static_cast<RefScene*>(scene)->rt.get_scene_mesh_ptrs().size());
}

// World AABB of every primitive, in get_scene_mesh_ptrs() order: out[i*6 + {0..2}] = min, {3..5} = max.
void ref_scene_aabbs(void* scene, double* out) {
    const auto& ptrs = static_cast<RefScene*>(scene)->rt.get_scene_mesh_ptrs();
    for (size_t i = 0; i < ptrs.size(); ++i) {
        Float3 mn, mx; ptrs[i]->compute_aabb(mn, mx);
        out[i*6+0] = mn.x; out[i*6+1] = mn.y; out[i*6+2] = mn.z;
        out[i*6+3] = mx.x; out[i*6+4] = mx.y; out[i*6+5] = mx.z;
    }
}

// Primary rays exactly as get_one_ray_per_pixel builds them.  Returns the ray count
// (0 when the camera id is unknown); rays6 may be null to query the count only.
long ref_primary_rays(void* scene, const char* camera_id, double* rays6, long capacity) {
    std::vector<Ray> rays = static_cast<RefScene*>(scene)->rt.get_one_ray_per_pixel(camera_id);
    if (rays6) {
        for (size_t i = 0; i < rays.size() && static_cast<long>(i) < capacity; ++i) {
            const RayVec3& o = rays[i].getPosition(); const RayVec3& d = rays[i].getDirection();
            double* r = rays6 + i*6;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
        }
    }
    return static_cast<long>(rays.size());
}

// Brute-force closest hit (RayTracer::one_pass_intersection) for n arbitrary rays.
void ref_trace_rays(void* scene, const double* rays6, long n, double* records16) {
    RefScene* s = static_cast<RefScene*>(scene);
    const auto& ptrs = s->rt.get_scene_mesh_ptrs();
    for (long i = 0; i < n; ++i) {
        Hit h;
        const bool ok = s->rt.one_pass_intersection(make_ray(rays6 + i*6), h);
        fill_record(ptrs, ok, h, reinterpret_cast<HitRecord*>(records16)[i]);
    }
}

// Same rays through the reference's own BVH (get_BVH_tree + accelerated_one_pass_intersection).
void ref_trace_rays_bvh(void* scene, const double* rays6, long n, double* records16) {
    RefScene* s = static_cast<RefScene*>(scene);
    const auto& ptrs = s->rt.get_scene_mesh_ptrs();
    if (!s->bvh_built) {
        s->node_store.clear();
        s->bvh_root = get_BVH_tree(ptrs, s->node_store);
        s->bvh_built = true;
    }
    for (long i = 0; i < n; ++i) {
        Hit h;
        const bool ok = accelerated_one_pass_intersection(s->bvh_root, make_ray(rays6 + i*6), h);
        fill_record(ptrs, ok, h, reinterpret_cast<HitRecord*>(records16)[i]);
    }
}

// one_pass_intersection followed by RayTracer::shade for n arbitrary rays; rgb = n*3 bytes (linear, pre-gamma).
void ref_shade_rays(void* scene, const double* rays6, long n, unsigned char* rgb) {
    RefScene* s = static_cast<RefScene*>(scene);
    for (long i = 0; i < n; ++i) {
        const Ray ray = make_ray(rays6 + i*6);
        Hit h; (void)s->rt.one_pass_intersection(ray, h);
        const Pixel px = s->rt.shade(h, ray);
        rgb[i*3+0] = px.r; rgb[i*3+1] = px.g; rgb[i*3+2] = px.b;
    }
}

int ref_render_ppm(void* scene, const char* camera_id, const char* out_path, int spp, int max_scatters, double min_scatter_intensity) {
    return static_cast<RefScene*>(scene)->rt.render_unaccelerated_ppm(camera_id, out_path, spp, max_scatters, min_scatter_intensity) ? 1 : 0;
}

// Single-primitive intersection through the public constructors (no JSON).
// kind: 0 Cube(T,rpy,S) 1 Plane(4 corners) 2 Cylinder(T,rpy,S) 3 Sphere(C,S); params: 9 / 12 / 9 / 6 doubles.
void ref_intersect_prim(int kind, const double* q, const double* rays6, long n, double* records16) {
    std::unique_ptr<Mesh> m;
    switch (kind) {
        case 0: m.reset(new Cube(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]})); break;
        case 1: m.reset(new Plane(std::vector<Float3>{ {q[0],q[1],q[2]}, {q[3],q[4],q[5]}, {q[6],q[7],q[8]}, {q[9],q[10],q[11]} })); break;
        case 2: m.reset(new Cylinder(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]})); break;
        default: m.reset(new Sphere(Float3{q[0],q[1],q[2]}, Float3{q[3],q[4],q[5]})); break;
    }
    std::vector<const Mesh*> one{ m.get() };
    for (long i = 0; i < n; ++i) {
        Hit h;
        const bool ok = m->intersect(make_ray(rays6 + i*6), h) && h.hasDistanceAlongRay();
        if (ok) h.setMesh(m.get());
        fill_record(one, ok, h, reinterpret_cast<HitRecord*>(records16)[i]);
    }
}

void ref_prim_aabb(int kind, const double* q, double* out6) {
    std::unique_ptr<Mesh> m;
    switch (kind) {
        case 0: m.reset(new Cube(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]})); break;
        case 1: m.reset(new Plane(std::vector<Float3>{ {q[0],q[1],q[2]}, {q[3],q[4],q[5]}, {q[6],q[7],q[8]}, {q[9],q[10],q[11]} })); break;
        case 2: m.reset(new Cylinder(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]})); break;
        default: m.reset(new Sphere(Float3{q[0],q[1],q[2]}, Float3{q[3],q[4],q[5]})); break;
    }
    Float3 mn, mx; m->compute_aabb(mn, mx);
    out6[0] = mn.x; out6[1] = mn.y; out6[2] = mn.z; out6[3] = mx.x; out6[4] = mx.y; out6[5] = mx.z;
}

// Camera::pixelToRay for a camera read through read_camera_from_json (world-up variant) at arbitrary pixel coordinates.
int ref_camera_rays(const char* json_path, const char* camera_id, const double* pxy, long n, double* rays6) {
    Camera cam;
    if (!read_camera_from_json(json_path, camera_id, cam)) return 0;
    for (long i = 0; i < n; ++i) {
        Vec3 o, d; cam.pixelToRay(pxy[i*2], pxy[i*2+1], o, d);
        double* r = rays6 + i*6;
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    }
    return 1;
}

// A scene built through the reference's public constructors (no JSON: its loader is quadratic in the number of
// objects), in the reference's id order - cubes, planes, cylinders, spheres (raytracer.cpp:20-27); the caller passes
// the primitives already in that order.  kind / params as ref_intersect_prim (12 doubles per primitive).
struct RefPrims {
    std::vector<std::unique_ptr<Mesh>> owned;
    std::vector<const Mesh*> ptrs;
    std::unordered_map<const Mesh*, int> index;
    std::vector<std::unique_ptr<BoundingBox>> node_store;
    const Mesh* root = nullptr;
    bool built = false;
};

void* ref_prims_create(const int* kinds, const double* params12, long n) {
    auto* s = new RefPrims();
    s->owned.reserve(n); s->ptrs.reserve(n);
    for (long i = 0; i < n; ++i) {
        const double* q = params12 + i * 12;
        Mesh* m = nullptr;
        switch (kinds[i]) {
            case 0: m = new Cube(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]}); break;
            case 1: m = new Plane(std::vector<Float3>{ {q[0],q[1],q[2]}, {q[3],q[4],q[5]}, {q[6],q[7],q[8]}, {q[9],q[10],q[11]} }); break;
            case 2: m = new Cylinder(Float3{q[0],q[1],q[2]}, EulerAngles{q[3],q[4],q[5]}, Float3{q[6],q[7],q[8]}); break;
            case 3: m = new Sphere(Float3{q[0],q[1],q[2]}, Float3{q[3],q[4],q[5]}); break;
            default: delete s; return nullptr;
        }
        s->owned.emplace_back(m);
        s->ptrs.push_back(m);
        s->index[m] = static_cast<int>(i);
    }
    return s;
}

void ref_prims_free(void* h) { delete static_cast<RefPrims*>(h); }

// get_BVH_tree + accelerated_one_pass_intersection (raytracer.cpp:454-551) for n rays: ids and distances.  The
// traversal only reads the tree, so rays are spread over OpenMP threads.
void ref_prims_trace_bvh(void* h, const double* rays6, long n, int* ids, double* dist, int threads) {
    RefPrims* s = static_cast<RefPrims*>(h);
    if (!s->built) { s->root = get_BVH_tree(s->ptrs, s->node_store); s->built = true; }
    #pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
    for (long i = 0; i < n; ++i) {
        Hit hit;
        const bool ok = s->root && accelerated_one_pass_intersection(s->root, make_ray(rays6 + i*6), hit);
        ids[i] = -1; dist[i] = std::numeric_limits<double>::infinity();
        if (ok && hit.hasMesh()) { ids[i] = s->index.at(hit.getMesh()); dist[i] = hit.getDistanceAlongRay(); }
    }
}

// The loop of RayTracer::one_pass_intersection (raytracer.cpp:143-157) over the same primitives: every primitive's
// intersect, strict '<' on the distance, first in order wins.
void ref_prims_trace_brute(void* h, const double* rays6, long n, int* ids, double* dist, int threads) {
    RefPrims* s = static_cast<RefPrims*>(h);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (long i = 0; i < n; ++i) {
        const Ray ray = make_ray(rays6 + i*6);
        double best = std::numeric_limits<double>::infinity();
        int best_id = -1;
        for (size_t k = 0; k < s->ptrs.size(); ++k) {
            Hit tmp;
            if (s->ptrs[k]->intersect(ray, tmp) && tmp.hasDistanceAlongRay()) {
                const double d = tmp.getDistanceAlongRay();
                if (d < best) { best = d; best_id = static_cast<int>(k); }
            }
        }
        ids[i] = best_id; dist[i] = best;
    }
}

// The reference's own timing functions (s2899221/Code/raytracer.cpp:553,567), for the "reference" CPU baseline.
double ref_time_one_pass(const char* json_path, const char* camera_id) { return time_trace_all_one_pass(json_path, camera_id); }
double ref_time_accelerated(const char* json_path, const char* camera_id) { return time_trace_all_accelerated_one_pass(json_path, camera_id); }

} // extern "C"
