/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, FP64 restatement of the reference ray tracer's hot path (primary-ray generation,
 * brute-force closest hit over analytic primitives, Blinn-Phong shading with hard shadows,
 * Fresnel-weighted mirror bounces, gamma).  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (hit ids, distances,
 * hit records, 8-bit colours) against the real reference engine compiled from its own sources
 * (oracle/_ref, built by oracle/Makefile) by tests/test_oracle_vs_ref.py in the development
 * container, and against golden fixtures generated from that engine (tests/golden/) everywhere.
 * Two exceptions, both extensions the reference does not have: the triangle primitive is pinned only
 * indirectly (two triangles of a parallelogram against the reference Plane), and two-level scenes
 * (rto_instance) are parity unpinned - tied to the pinned plain path by construction tests only (an
 * identity instance reproduces the plain scene bit for bit, tests/test_instancing.py).
 *
 * Arithmetic follows the reference operation by operation so that results agree to the last bit
 * when compiled without FMA contraction (-ffp-contract=off, no -ffast-math).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Primitive kinds; the numeric order is the reference's primitive-id order
 * (cubes, planes, cylinders, spheres: s2899221/Code/raytracer.cpp:20-27); triangles come last. */
enum { RTO_CUBE = 0, RTO_PLANE = 1, RTO_CYLINDER = 2, RTO_SPHERE = 3, RTO_TRIANGLE = 4 };

/* One primitive as the reference's constructors take it (s2899221/Code/mesh.h:81,107,132,153):
 *   cube / cylinder : q[0..2] translation, q[3..5] roll,pitch,yaw, q[6..8] scale
 *   sphere          : q[0..2] location, q[3..5] scale
 *   plane           : q[0..11] four corners
 *   triangle        : q[0..8] three corners
 * followed by the material block of mesh.h:52-59. */
typedef struct rto_prim {
    int32_t kind;
    int32_t reserved;
    double  q[12];
    double  alpha, metallic, roughness, ior;
    double  albedo[3];
} rto_prim;

typedef struct rto_light { double pos[3]; double intensity; } rto_light;

/* Camera state of s2899221/Code/persp_camera.h:29-36; rot is row-major, columns right/up/forward. */
typedef struct rto_camera {
    double  pos[3];
    double  rot[9];
    double  focal, sensor_w, sensor_h;
    int32_t width, height;
} rto_camera;

typedef struct rto_hit {
    int32_t hit;         /* 1 when something was hit */
    int32_t id;          /* primitive index, -1 on miss */
    double  dist;        /* distance along the ray, +inf on miss */
    double  p[3], n[3], r[3];
    double  u, v;
    int32_t prim;        /* primitive whose material applies: id, or in an instanced scene the mesh primitive */
    int32_t inst;        /* instanced scenes: which instance was hit; -1 otherwise */
} rto_hit;

/* Two-level scenes (an extension of the product; the reference has no instancing): an instance places the run
 * [first_prim, first_prim + n_prims) of the primitive list with x_world = rot * x_mesh + trans (rot row-major,
 * orthonormal).  When a scene has instanced != 0 only the instances are visible.  A ray enters an instance as
 * o' = rot^T (o - trans), d' = rot^T d and is tested there with the ordinary intersectors; the winning hit returns as
 * p = rot p' + trans, n = rot n', r = rot r' with the distance found in the instance's frame.  Ids: instance k owns
 * [base_k, base_k + n_prims_k), base_k = sum of n_prims of the instances before it; strict '<' over ascending ids. */
typedef struct rto_instance { double rot[9]; double trans[3]; uint32_t first_prim, n_prims; } rto_instance;

/* Albedo texture as Image keeps it (utils.h:129-151): 8-bit RGB, rows top to bottom. */
typedef struct rto_texture { int32_t width, height; const uint8_t* rgb; } rto_texture;

typedef struct rto_scene {
    const rto_prim*  prims;  int64_t n_prims;
    const rto_light* lights; int64_t n_lights;
    const rto_texture* textures; int64_t n_textures;   /* may be NULL / 0 */
    const int32_t* prim_tex;                           /* per primitive: index into textures or -1; may be NULL */
    int64_t instanced;                                 /* non-zero: two-level scene, see rto_instance */
    const rto_instance* inst; int64_t n_inst;
    /* optional, 6 doubles per instance: bounds of its mesh in the mesh's frame (min xyz, max xyz).  Only used to skip
     * instances a ray cannot touch (a padded FP64 slab test); results are the same with or without it. */
    const double* inst_bounds;
    /* optional, 6 doubles per primitive (rto_hit_bounds): a box that holds every point the primitive's intersector can
     * return.  Only used by the brute-force loop to pass over primitives a ray cannot touch (the same padded FP64 slab
     * test); every primitive the ray can touch is still tested, in ascending id order, so results are the same with
     * or without it (tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py). */
    const double* prim_bounds;
} rto_scene;

/* 32-byte BVH node shared with the product's flattened layout: internal nodes have count == 0 and
 * first = index of the left child (the right child is first + 1); leaves hold count > 0 primitive
 * slots starting at first in the slot->primitive-id table. */
typedef struct rto_node32 { float lo[3]; uint32_t first; float hi[3]; uint32_t count; } rto_node32;

void rto_camera_look(const double pos[3], const double dir[3], const double* up_or_null, rto_camera* cam);
void rto_pixel_to_ray(const rto_camera* cam, double px, double py, double o[3], double d[3]);

int  rto_intersect(const rto_prim* prim, const double o[3], const double d[3], rto_hit* out);
void rto_aabb(const rto_prim* prim, double mn[3], double mx[3]);
int  rto_closest_hit(const rto_scene* sc, const double o[3], const double d[3], rto_hit* out);
void rto_shade(const rto_scene* sc, const rto_hit* hit, const double ray_o[3], uint8_t rgb[3]);
void rto_trace(const rto_scene* sc, const double o[3], const double d[3], int depth, int max_scatters,
               double min_scatter_intensity, uint8_t rgb[3]);

/* Batched forms (OpenMP over rays / rows; threads <= 0 means "all cores"). */
void rto_trace_rays(const rto_scene* sc, const double* rays6, int64_t n, rto_hit* out, int threads);
void rto_shade_rays(const rto_scene* sc, const double* rays6, int64_t n, uint8_t* rgb, int threads);
/* Primary rays for the pixel rectangle [x0,x0+w) x [y0,y0+h): ids and distances, row-major in the rectangle. */
void rto_trace_primary(const rto_scene* sc, const rto_camera* cam, int x0, int y0, int w, int h,
                       int32_t* ids, double* dist, int threads);
/* Full pixel pipeline (spp = 1): trace, shade, bounces, gamma.  rgb is w*h*3 bytes.  With no lights: all black.
 * shadow_rays (may be NULL) receives the number of shadow rays cast, n_hits the primary hits. */
void rto_render(const rto_scene* sc, const rto_camera* cam, int x0, int y0, int w, int h,
                int max_scatters, double min_scatter_intensity, uint8_t* rgb,
                int64_t* n_hits, int64_t* shadow_rays, int threads);
void rto_gamma_lut(uint8_t lut[256]);
/* rto_trace (trace_ray_recursive, raytracer.cpp:281-371) per ray: linear 8-bit colour, 3 bytes per ray. */
void rto_trace_rays_rgb(const rto_scene* sc, const double* rays6, int64_t n, int max_scatters, double min_scatter_intensity,
                        uint8_t* rgb, int threads);

/* Box holding every point each primitive's intersector can return (min xyz, max xyz per primitive): the reference's
 * compute_aabb for cube / cylinder / ellipsoid, the parallelogram of corners 0, 1, 3 for a plane (its intersector
 * ignores corner 2, mesh.cpp:799-804), the corners of a triangle; a primitive with a zero or non-finite scale is not
 * bounded by its corner box and gets an infinite one. */
void rto_hit_bounds(const rto_prim* prims, int64_t n, double* out6);
/* Brute force at sizes where one ray costs a million primitive tests: for each ray the closest hit of
 * rto_closest_hit (hits), then RayTracer::shade's shadow ray towards every light from every hit, each resolved by
 * the same loop over ALL primitives (lit[i * n_lights + l] = 1 when light l reaches hit i, 0 when it is blocked or ray
 * i missed), and the linear 8-bit colour those verdicts give (rgb, 3 bytes per ray; no gamma).  Rays and primitives are
 * visited in cache-sized blocks; per ray the primitives are still met in ascending id order with strict '<'. */
void rto_brute_shade_rays(const rto_scene* sc, const double* rays6, int64_t n, rto_hit* hits, uint8_t* lit,
                          uint8_t* rgb, int threads);

/* The product's flattened BVH, as rt_scene_get_bvh / rt_scene_get_info hand it out: slot_prim maps
 * primitive slots to ids, the first n_slots in BVH leaf order followed by n_loose unbounded primitives
 * that every ray tests; radius scales the FP32 box padding (2^-18 * max(|origin|, radius)). */
/* The product's wide node (rt_node4): up to four subtrees of the binary tree side by side, boxes by component
 * (lo[axis][child]); ref[child] = count << 28 | first slot for a leaf, the index of another wide node for an inner
 * child, all ones for an empty place.  Wide node 0 holds the children of the binary root. */
typedef struct rto_node4 { float lo[3][4]; float hi[3][4]; uint32_t ref[4]; uint32_t pad[4]; } rto_node4;

typedef struct rto_bvh {
    const rto_node32* nodes; int64_t n_nodes;
    const uint32_t* slot_prim; int64_t n_slots; int64_t n_loose;
    float radius;
    /* when set, the walk goes through the wide nodes exactly as the product's does (root box from nodes[0], then four
     * boxes per step, entered children ordered by a five-exchange network, nearest first) */
    const rto_node4* nodes4; int64_t n_nodes4;
    /* optional (rt_scene_get_loose_boxes): 8 floats per loose primitive - min xyz, -, max xyz, -.  When set, a ray that
     * misses a loose primitive's padded box is spared that primitive's test, as in the product. */
    const float* loose_boxes;
} rto_bvh;

/* Closest hit through the BVH with FP32 padded boxes and FP64 leaf tests; same result as rto_closest_hit by
 * construction (equal distances -> lowest id).  counts[0] += nodes fetched, counts[1] += primitive records fetched. */
int  rto_bvh_closest_hit(const rto_scene* sc, const rto_bvh* bvh, const double o[3], const double d[3],
                         rto_hit* out, int64_t counts[2]);
/* Shadow verdict with the reference's rule (blocked iff the closest hit lies in (1e-4, L - 1e-4)),
 * evaluated through the BVH with the early exits the product uses. */
int  rto_bvh_occluded(const rto_scene* sc, const rto_bvh* bvh, const double o[3], const double d[3],
                      double light_dist, int64_t counts[2]);
void rto_bvh_trace_rays(const rto_scene* sc, const rto_bvh* bvh, const double* rays6, int64_t n, rto_hit* out,
                        int64_t counts[2], int threads);
/* Primary + shadow pass over a rectangle through the BVH (max_scatters = 0 pipeline), every step-th pixel,
 * counting node / primitive fetches per ray class: counts[0..1] primary, counts[2..3] shadow. */
void rto_bvh_render(const rto_scene* sc, const rto_bvh* bvh, const rto_camera* cam, int x0, int y0, int w, int h,
                    int step, int32_t* ids, uint8_t* rgb, int64_t counts[4], int64_t* n_hits, int threads);

#ifdef __cplusplus
}
#endif
#endif /* RT_ORACLE_H */
