/* rt_hip.h - C ABI of the MI355X ray-tracing hot path (librt_hip.so).
 *
 * The reference engine (kailen-hargenrader/RayTracer, s2899221/Code) has no plugin or FFI layer: its
 * boundary is a C++ class API.  This C ABI sits directly beneath a host-side mirror of that API
 * (raytracer_amd/csrc/host/{raytracer,mesh,persp_camera,utils}.h) and is what any binding for the path
 * calls: each entry point below names the reference interface whose work it takes over.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; nothing throws across the
 * boundary; all pointers are caller-owned; "_dev" pointers are device memory on the context's GPU
 * (hipMalloc, or a torch CUDA tensor's data_ptr); calls on one rt_ctx are ordered on its stream and are
 * not re-entrant; different contexts (GPUs) may be driven from different host threads.
 * Primitive ids are indices into the reference's flat primitive list, i.e. cubes, planes, cylinders,
 * spheres in file order (s2899221/Code/raytracer.cpp:20-27), then triangles (extension); -1 = miss.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* null pointer, bad size, bad rectangle ... */
    RT_ERR_IO = -2,           /* scene file missing / unreadable */
    RT_ERR_NOT_FOUND = -3,    /* unknown camera id */
    RT_ERR_NO_DEVICE = -4,    /* no usable GPU / HIP runtime error at context creation */
    RT_ERR_HIP = -5,          /* a HIP call failed; rt_last_error() has the text */
    RT_ERR_NO_SCENE = -6,     /* trace/render called before rt_scene_upload */
    RT_ERR_LIMIT = -7,        /* size beyond a documented maximum */
    RT_ERR_PARSE = -8,        /* malformed number in a scene file */
    RT_ERR_COMM = -9          /* RCCL missing or an RCCL call failed; rt_last_error() has the text */
} rt_status;

enum { RT_CUBE = 0, RT_PLANE = 1, RT_CYLINDER = 2, RT_SPHERE = 3, RT_TRIANGLE = 4 };

/* One primitive as the reference constructors take it (s2899221/Code/mesh.h:81,107,132,153):
 *   cube / cylinder: q[0..2] translation, q[3..5] roll,pitch,yaw (R = Rz Ry Rx), q[6..8] scale
 *   sphere: q[0..2] location, q[3..5] scale      plane: q[0..11] four corners (corner 2 is ignored by
 *   the intersector, as in mesh.cpp:799-804)     triangle: q[0..8] three corners
 * and the material block of mesh.h:52-59.  reserved on a plane: 0 = four corners; 1..3 = only that many were given
 * (the plane is kept, spans those corners and is never hit, mesh.cpp:796, 922-930); any other value = no corners. */
typedef struct rt_prim_desc {
    int32_t kind;
    int32_t reserved;
    double  q[12];
    double  alpha, metallic, roughness, ior;
    double  albedo[3];
} rt_prim_desc;

/* PointLight, s2899221/Code/raytracer.h:14-17. */
typedef struct rt_light { double pos[3]; double intensity; } rt_light;

/* Camera state, s2899221/Code/persp_camera.h:29-36; rot row-major with columns right, up, forward. */
typedef struct rt_camera {
    double  pos[3];
    double  rot[9];
    double  focal, sensor_w, sensor_h;
    int32_t width, height;
} rt_camera;

typedef struct rt_rect { int32_t x0, y0, w, h; } rt_rect;

/* Flattened BVH node, 32 bytes.  count == 0: internal, children at first and first + 1.
 * count > 0: leaf over primitive slots [first, first + count). */
typedef struct rt_node32 { float lo[3]; uint32_t first; float hi[3]; uint32_t count; } rt_node32;

/* Wide node, 128 bytes: up to four subtrees of the binary tree side by side (boxes by component: lo[axis][child],
 * hi[axis][child]).  ref[child]: bits 31..28 = primitive count of a leaf over slots [ref & 0x0FFFFFFF, +count), or 0 for
 * an inner node whose wide node has that index; 0xFFFFFFFF = no child.  Wide node 0 holds the children of the root. */
typedef struct rt_node4 { float lo[3][4]; float hi[3][4]; uint32_t ref[4]; uint32_t pad[4]; } rt_node4;

typedef struct rt_build_opts {
    uint32_t max_leaf;        /* primitives per leaf, 1..8 (0 = default 1) */
    uint32_t sah_bins;        /* 4..32 (0 = default 32) */
    uint32_t flags;           /* RT_BUILD_* */
    uint32_t reserved[5];
} rt_build_opts;
/* Leave the hierarchy to the GPU: no tree is built on the host, rt_scene_upload sorts the primitives along a Morton
 * curve and cuts the sorted sequence top-down on the device (about a millisecond per million primitives, for scenes that
 * change every frame; the tree is of the same quality as the host's for evenly spread scenes, results are the same in
 * any case).  rt_scene_get_bvh has nothing to return for such a scene and rt_scene_info.depth is an upper bound;
 * rt_ctx_download_bvh fetches what the device built. */
#define RT_BUILD_ON_DEVICE 1u

typedef struct rt_scene_info {
    uint64_t n_prims;         /* primitives in id order */
    uint64_t n_slots;         /* primitives reachable through the BVH */
    uint64_t n_loose;         /* primitives tested against every ray: unbounded ones (a zero / non-finite scale) and, in
                                 scenes of more than 16, up to 8 that span a quarter of the scene or more */
    uint32_t n_nodes, depth;
    uint32_t n_lights, n_cameras;
    uint32_t n_by_kind[8];
    double   bounds[6];       /* min xyz, max xyz of the bounded primitives */
    double   radius;          /* max |coordinate| of bounds and lights: scales the FP32 box padding */
    uint64_t n_instances;     /* instanced scenes: instances, and the primitives they stand for (the id range) */
    uint64_t n_instanced_prims;
} rt_scene_info;

/* Frame partition for multi-GPU runs: square tiles dealt round-robin (tile t -> rank t % n_ranks);
 * rank r stores its tiles back to back, each as tile*tile pixels (edge tiles padded). */
typedef struct rt_tiling { int32_t width, height, tile, n_ranks; } rt_tiling;

typedef struct rt_counters {
    uint64_t primary_rays, primary_hits, shadow_rays, bounce_rays;
    uint64_t nodes_primary, prims_primary;   /* V_n, V_p: only filled by the counting variants */
    uint64_t nodes_shadow, prims_shadow;
} rt_counters;

/* Device time of the frame pipeline's launches, measured with HIP events on the context's stream. */
typedef struct rt_timing {
    double   primary_ms, shadow_ms, resolve_ms, bounce_ms;        /* summed over the launches below */
    uint32_t primary_launches, shadow_launches, resolve_launches, bounce_launches;
    double   entry_ms;                                            /* the entry searches of both ray classes, on their own */
    uint32_t entry_launches, reserved;
} rt_timing;

typedef struct rt_scene rt_scene;   /* host side: descriptions + flattened records + BVH */
typedef struct rt_ctx rt_ctx;       /* one GPU: stream, device copy of a scene, scratch outputs */

/* ---- host side: scenes ------------------------------------------------------------------- */

/* Build a scene from in-memory descriptions (ids = array order; the caller is responsible for the
 * reference's kind order if it wants reference ids).  Takes over what RayTracer keeps after load
 * (raytracer.h:48-58) plus get_BVH_tree (raytracer.h:78, raytracer.cpp:454-499), re-designed: binned SAH,
 * FP32 boxes rounded outward, depth <= 31. */
int rt_scene_create(const rt_prim_desc* prims, uint64_t n_prims, const rt_light* lights, uint32_t n_lights,
                    const rt_build_opts* opts_or_null, rt_scene** out);
/* Two-level scenes (extension; the reference has no instancing).  The primitive list is a pool of meshes - a mesh is a
 * contiguous run [first_prim, first_prim + n_prims) - and what rays see is the instances: each places one mesh with a
 * rigid transform x_world = rot * x_mesh + trans (rot row-major, orthonormal; not checked).  The hierarchy has two
 * levels, one tree per distinct mesh and one over the instances, so a scene of 10^4 instances of a 10^4-primitive mesh
 * costs the memory of 2 * 10^4 items.  A ray is carried into an instance's frame as o' = rot^T (o - trans),
 * d' = rot^T d, tested there with the ordinary intersectors, and the hit is carried back as p = rot p' + trans,
 * n = rot n', r = rot r'; the distance is the one found in the instance's frame.  Hit ids: instance k owns the ids
 * [base_k, base_k + n_prims_k), base_k = sum of n_prims over the instances before it; equal distances resolve to the
 * lowest id.  Materials are those of the mesh primitive.  Every mesh primitive must be bounded (RT_ERR_INVALID
 * otherwise); textures and RT_BUILD_ON_DEVICE are not available for such scenes; at most 2^31 - 1 ids. */
typedef struct rt_instance { double rot[9]; double trans[3]; uint32_t first_prim, n_prims; } rt_instance;
int rt_scene_create_instanced(const rt_prim_desc* prims, uint64_t n_prims, const rt_instance* instances, uint64_t n_instances,
                              const rt_light* lights, uint32_t n_lights, const rt_build_opts* opts_or_null, rt_scene** out);
/* RayTracer::load_from_json (raytracer.h:25, raytracer.cpp:107-121) for the Blender-export format, in
 * linear time, same defaults; the file's cameras are kept and served by rt_scene_camera. */
int rt_scene_load_json(const char* path, const rt_build_opts* opts_or_null, rt_scene** out);
/* The same from text already in memory (the per-class readers Cube/Plane/Cylinder/Sphere::read_from_json,
 * mesh.cpp:52,316,544,707, take text blocks rather than paths). */
int rt_scene_load_json_text(const char* text, uint64_t len, const rt_build_opts* opts_or_null, rt_scene** out);
/* Albedo textures (Mesh::setAlbedoTexture, mesh.h:47; sampled by Mesh::evaluate_albedo, mesh.cpp:21-45: wrap, nearest,
 * x = floor(u * (w - 1))).  rt_scene_load_json reads the files named by "albedo_texture_ppm" itself and, like the
 * reference (mesh.cpp:105-110), keeps the solid colour when a file does not load.  For scenes built in memory:
 * prim_texture[i] = index into textures, or -1; call before rt_scene_upload.  Pixels are copied. */
typedef struct rt_texture { int32_t width, height; const uint8_t* rgb; } rt_texture;   /* rows top to bottom, 3 bytes per pixel */
int rt_scene_set_textures(rt_scene* scene, const rt_texture* textures, uint32_t n_textures, const int32_t* prim_texture);
/* Borrowed view of the texture of one primitive (width 0 when it has none). */
int rt_scene_get_texture(const rt_scene* scene, uint64_t prim_id, rt_texture* out);
/* Image::loadFromFile (utils.cpp:212-250): P3 / P6, maxval <= 255, samples as stored.  With rgb_or_null == NULL only the
 * header is returned.  RT_ERR_IO: cannot open; RT_ERR_PARSE: not a PPM this reader accepts. */
int rt_image_read_ppm(const char* path, int32_t* width, int32_t* height, int32_t* maxval, uint8_t* rgb_or_null, uint64_t capacity);
int rt_scene_destroy(rt_scene* scene);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);
/* Borrowed views, valid until rt_scene_destroy. */
int rt_scene_get_prims(const rt_scene* scene, const rt_prim_desc** prims, uint64_t* n);
int rt_scene_get_lights(const rt_scene* scene, const rt_light** lights, uint32_t* n);
int rt_scene_get_bvh(const rt_scene* scene, const rt_node32** nodes, uint32_t* n_nodes,
                     const uint32_t** slot_prim, uint64_t* n_slots);
/* Boxes of the primitives kept beside the tree (rt_scene_info.n_loose of them, in slot order): 8 floats each - min xyz,
 * unused, max xyz, unused - rounded outward; infinite for an unbounded primitive.  The kernels spare a ray the FP64
 * test of such a primitive when the ray misses the padded box (the test of a box in the tree, no more and no less). */
int rt_scene_get_loose_boxes(const rt_scene* scene, const float** boxes8, uint64_t* n);
/* The hierarchy as the kernels walk it for host-built scenes: the binary tree collapsed into wide nodes (one 128-byte
 * fetch and four box tests per step instead of two steps; the same boxes, the same leaves).  For a two-level scene one
 * array holds the tree over the instances first and then each mesh's tree, inner references counted from the start of
 * the array.  n = 0 when the scene is walked through the binary nodes (a single leaf, or a tree so lopsided that the
 * wide walk could need more than the traversal stack holds) and for device-built scenes (see rt_ctx_download_wide_bvh).
 * The reference has no counterpart (its tree is binary, raytracer.cpp:454-499). */
int rt_scene_get_wide_bvh(const rt_scene* scene, const rt_node4** nodes4, uint32_t* n_nodes4);
/* Camera by id as read_all_cameras builds it (raytracer.cpp:29-68); RT_ERR_NOT_FOUND if absent.
 * index >= 0 with id == NULL enumerates cameras in file order; id_out (>= 32 bytes) may be NULL. */
int rt_scene_camera(const rt_scene* scene, const char* id, int32_t index, rt_camera* out, char* id_out);
/* World AABB of primitive `id` exactly as the reference's compute_aabb (mesh.cpp:288,517,683,922). */
int rt_prim_aabb(const rt_prim_desc* prim, double out6[6]);

/* Camera::updateRotationMatrix, both overloads (persp_camera.cpp:35-66); up may be NULL. */
int rt_camera_look(const double pos[3], const double dir[3], const double* up_or_null, rt_camera* cam);
/* read_camera_from_json (persp_camera.h:66, persp_camera.cpp:120-168): world-up orientation only. */
int rt_camera_from_json(const char* path, const char* camera_id, rt_camera* out);
/* Camera::pixelToRay (persp_camera.cpp:94-118) on the host, for callers that want rays. */
int rt_camera_pixel_to_ray(const rt_camera* cam, double px, double py, double origin[3], double dir[3]);

/* ---- device side -------------------------------------------------------------------------- */

int rt_ctx_create(int device, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);
/* Launch on the caller's stream: a hipStream_t such as torch.cuda.current_stream().cuda_stream, where NULL is
 * HIP's default stream (torch's default).  A fresh context launches on a private non-blocking stream;
 * rt_ctx_use_own_stream returns to it. */
int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream);
int rt_ctx_use_own_stream(rt_ctx* ctx);
int rt_ctx_synchronize(rt_ctx* ctx);
/* Copy the scene to the context's GPU (synchronous; replaces a previously uploaded scene). */
int rt_scene_upload(rt_ctx* ctx, const rt_scene* scene);

/* Milliseconds the last rt_scene_upload spent building the hierarchy on the device (0 for a host-built scene). */
int rt_ctx_build_time(rt_ctx* ctx, double* ms_out);
/* The uploaded hierarchy as the kernels walk it: up to node_capacity nodes and slot_capacity slot -> primitive id
 * entries (bounded primitives in leaf order, then the unbounded ones); the counts are returned either way. */
int rt_ctx_download_bvh(rt_ctx* ctx, rt_node32* nodes_out, uint32_t node_capacity, uint32_t* n_nodes_out,
                        uint32_t* slot_prim_out, uint64_t slot_capacity, uint64_t* n_slots_out);

/* The wide nodes the kernels walk on this GPU, if they walk any (n = 0 otherwise): for a host-built scene what
 * rt_scene_get_wide_bvh returns; for a device-built one 2 * n_slots places of which place i holds the wide form of inner
 * node i of the binary tree (place 0: the root; places of leaves are unused). */
int rt_ctx_download_wide_bvh(rt_ctx* ctx, rt_node4* nodes4_out, uint32_t capacity, uint32_t* n_nodes4_out);

/* RayTracer::get_one_ray_per_pixel + one_pass_intersection over a pixel rectangle (raytracer.cpp:123-157):
 * primary rays generated in registers, closest hit.  Outputs row-major in the rectangle; either may be NULL. */
int rt_trace_primary(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int32_t* hit_id_dev, double* dist_dev);
/* RayTracer::one_pass_intersection (raytracer.h:31) for n arbitrary rays (origin xyz, direction xyz, FP64). */
int rt_trace_rays(rt_ctx* ctx, const double* rays6_dev, uint64_t n, int32_t* hit_id_dev, double* dist_dev);
/* one_pass_intersection + RayTracer::shade (raytracer.h:34) for n arbitrary rays: linear 8-bit colour, no gamma. */
int rt_shade_rays(rt_ctx* ctx, const double* rays6_dev, uint64_t n, uint8_t* rgb_dev);
/* one_pass_intersection returning the reference's whole Hit (utils.h:68-112) per ray, 16 doubles each:
 * [0] 1 = hit / 0 = miss, [1] primitive id (-1 on a miss), [2] distance along the ray, [3..5] intersection point,
 * [6..8] surface normal, [9..11] reflected direction, [12..13] surface u, v (mesh.cpp:262-278, 497-512, 669-679, 918;
 * triangles: barycentric edge coordinates), [14..15] zero. */
int rt_trace_rays_full(rt_ctx* ctx, const double* rays6_dev, uint64_t n, double* records16_dev);
/* RayTracer::shade (raytracer.h:34, raytracer.cpp:159-279) for caller-supplied hits: 10 doubles each ([0..2] point,
 * [3..5] normal, [6] primitive id, [7..8] surface u, v, [9] non-zero when u, v are set; id -2 = a hit that names no
 * primitive, shaded white with the default material as the reference does; any other id outside the scene = black;
 * in an instanced scene the id names the mesh primitive, i.e. an index into the primitive pool)
 * and the origin of the ray that produced each hit (3 doubles).  Linear 8-bit colour. */
int rt_shade_hits(rt_ctx* ctx, const double* hits10_dev, const double* view3_dev, uint64_t n, uint8_t* rgb_dev);
/* RayTracer::render_unaccelerated_ppm's pixel loop with samples_per_pixel = 1 (raytracer.cpp:401-438):
 * trace, shade with hard shadows, up to max_scatters mirror bounces (<= 8), gamma.  rgb_dev is
 * rect.w * rect.h * 3 bytes; hit_id_dev (may be NULL) receives the primary hit ids. */
int rt_render_rect(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int max_scatters, double min_scatter_intensity,
                   uint8_t* rgb_dev, int32_t* hit_id_dev);
/* The shadow test of RayTracer::shade alone (raytracer.cpp:210-240) for the primary hit of every pixel of the rectangle:
 * one byte per pixel, bit l set = point light l (file order, at most 8 lights: RT_ERR_LIMIT beyond) is NOT blocked, i.e.
 * no closest hit of the shadow ray lies in (1e-4, distance to the light - 1e-4); 0 where the camera ray hits nothing. */
int rt_trace_shadows(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, uint8_t* lit_mask_dev);
/* rt_render_rect with max_scatters = 0 for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The frame working
 * set (72 bytes per pixel: hit points, normals, light sums, record slots; 64 bytes of shadow verdicts per 8 x 8 tile and
 * light of a batch), the work-queue heads and up to 5 MiB of per-tile planes of the entry search live in a fixed
 * workspace compiled into the library - one __device__ array of 18 MiB + 4 MiB + 5 MiB (27.0 MiB, RT_STATIC_SCRATCH_BYTES
 * in rt_device.h), of which every GPU the library runs on holds its own copy until the process ends - so the call makes
 * no hipMalloc / hipFree and calls no allocating routine.  Limits: rect.w * rect.h <= RT_STATIC_MAX_PIXELS (262144, e.g.
 * 512 x 512), and the rectangle, cut into blocks of 128 x 128 pixels from its top left corner, spans at most
 * RT_STATIC_MAX_TILES / 256 = 32 blocks (only a strip thinner than 32 pixels can have the pixels and not the blocks);
 * beyond either the call returns RT_ERR_LIMIT - render in pieces.  A call whose tiles and lights need more than the
 * 5 MiB of per-tile planes (more than 4096 tiles with 8 lights, say) runs without the entry search: same bytes out,
 * more boxes tested.  Launch timing (rt_ctx_set_timing) must be off (RT_ERR_INVALID otherwise: it creates events).  All
 * calls on one GPU share that GPU's workspace: calls that overlap in time on one GPU (two contexts, two streams) race;
 * calls ordered on one stream do not, and each call clears what it needs cleared.  The stream must be set before a
 * capture begins (rt_ctx_set_stream orders the new stream behind the old one with an event, which a capturing stream
 * cannot wait for; a capturing stream is taken as it is).  Outputs are byte-identical to rt_render_rect's. */
#define RT_STATIC_MAX_PIXELS 262144u
#define RT_STATIC_MAX_TILES 8192u
int rt_render_rect_static(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, uint8_t* rgb_dev, int32_t* hit_id_dev);
/* The same loop with samples_per_pixel > 1 (raytracer.cpp:403-436): sample 0 through the pixel centre, later samples
 * offset by up to half a pixel; per-sample linear 8-bit colours are averaged, rounded, then gamma encoded.  The
 * reference seeds its offsets from std::random_device, so frames differ run to run; here they are a pure function
 * of (seed, sample, pixel). */
int rt_render_samples(rt_ctx* ctx, const rt_camera* cam, rt_rect rect, int samples_per_pixel, uint32_t seed, int max_scatters,
                      double min_scatter_intensity, uint8_t* rgb_dev);
/* The same pixel pipeline over this rank's share of a tiled frame; rgb_dev holds
 * rt_tiles_per_rank(tiling) * tile * tile * 3 bytes. */
int rt_render_tiles(rt_ctx* ctx, const rt_camera* cam, rt_tiling tiling, int rank, int max_scatters,
                    double min_scatter_intensity, uint8_t* rgb_dev);
/* Scatter the gathered per-rank tile buffers (rank-major, as rt_render_tiles wrote them) into a row-major frame. */
int rt_assemble_tiles(rt_ctx* ctx, rt_tiling tiling, const uint8_t* gathered_dev, uint8_t* frame_rgb_dev);
/* The exchange of a tiled frame: every rank's tile buffer (as rt_render_tiles wrote it) travels to rank 0 by RCCL
 * point-to-point over xGMI - one send per peer, the matching receives on rank 0 in one group, no ring and no reduction -
 * and rank 0 scatters them into the row-major frame (the assemble kernel), all ordered on each context's stream.  The
 * reference has no counterpart: it is one thread on one CPU (render loop raytracer.cpp:401-438).
 *   One process (or host thread) per GPU: rank 0 makes an id (rt_comm_unique_id), hands its 128 bytes to the others by
 *   any channel, every rank calls rt_comm_create, then rt_gather_tiles once per frame (frame_rgb_dev is read on rank 0
 *   only and may be NULL elsewhere).
 *   One process driving n GPUs: rt_comm_create_all over n contexts (one GPU each, in rank order), then
 *   rt_gather_tiles_all with every rank's tile buffer; the frame lands on the GPU of contexts[0].
 * n_ranks = 1 works (rank 0 copies and scatters its own tiles).  RCCL is looked up at first use: without it these
 * calls return RT_ERR_COMM and everything else works.  A failing RCCL call returns RT_ERR_COMM with its text in
 * rt_last_error(); a communicator that failed is not usable afterwards. */
#define RT_COMM_ID_BYTES 128
typedef struct rt_comm rt_comm;
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_create(rt_ctx* ctx, int n_ranks, int rank, const uint8_t id[RT_COMM_ID_BYTES], rt_comm** out);
int rt_comm_create_all(rt_ctx** ctxs, int n, rt_comm** comms_out);
int rt_comm_destroy(rt_comm* comm);
int rt_gather_tiles(rt_comm* comm, rt_tiling tiling, const uint8_t* tiles_dev, uint8_t* frame_rgb_dev);
int rt_gather_tiles_all(rt_comm** comms, int n, rt_tiling tiling, const uint8_t* const* tiles_dev, uint8_t* frame_rgb_dev);
/* render_unaccelerated_ppm's frame (samples_per_pixel = 1, raytracer.cpp:373-442) on the n GPUs of a group made with
 * rt_comm_create_all, for host-memory callers: every GPU renders its tiles of `tile` x `tile` pixels, rt_gather_tiles_all
 * brings them to the first GPU, the frame is copied to rgb_host.  Each context must hold the scene.  Synchronises. */
int rt_render_frame_multi_host(rt_comm** comms, int n, const rt_camera* cam, int tile, int max_scatters,
                               double min_scatter_intensity, uint8_t* rgb_host);
int32_t rt_tiles_per_rank(rt_tiling tiling);
int32_t rt_tile_count(rt_tiling tiling);

/* Ray and (for the counting variants) node / primitive fetch counters accumulated since the last reset. */
int rt_counters_reset(rt_ctx* ctx);
int rt_counters_read(rt_ctx* ctx, rt_counters* out);          /* synchronises the stream */
/* When on, every pipeline launch is bracketed by HIP events on the context's stream; rt_timing_read synchronises,
 * returns the sums since the last read and clears them.  Off by default (events cost a little host time). */
int rt_ctx_set_timing(rt_ctx* ctx, int on);
int rt_timing_read(rt_ctx* ctx, rt_timing* out);
/* When on, trace / render launches use the instrumented kernels that also count BVH node and primitive
 * record fetches per ray class (slower; for roofline accounting and tests). */
int rt_ctx_set_counting(rt_ctx* ctx, int on);

/* Entry search (on by default; scenes walked through wide nodes, two-level ones through their top-level tree): before a
 * closest-hit or shadow launch of the frame pipeline one lane per 8x8-pixel tile refines, from the root down, a set of up
 * to four subtrees that the tile's beam of rays can touch, and the tile's rays start their walks from that set (its
 * "entry node").  A subtree is passed over only when no ray of the beam can touch its box, so results are identical with
 * it on or off; only the node counters differ (the counting kernels then include the boxes the search itself fetched,
 * and rays may meet subtrees in another order).  The reference has no counterpart (its traversal starts every ray at the
 * root, raytracer.cpp:501-512). */
int rt_ctx_set_entry_search(rt_ctx* ctx, int on);
/* When on, the closest-hit entry points (rt_trace_primary, rt_trace_rays, rt_trace_rays_full and their _host forms) run
 * RayTracer::one_pass_intersection as the reference does (raytracer.cpp:143-157): every primitive against every ray, no
 * hierarchy - the loop time_trace_all_one_pass times (raytracer.cpp:553-565).  Results equal the hierarchy's (ids,
 * distances, records bit for bit); cost is O(rays x primitives), meant for scenes of a few thousand primitives.  Off by
 * default.  Instanced scenes: RT_ERR_INVALID.  The render entry points are not affected. */
int rt_ctx_set_brute_force(rt_ctx* ctx, int on);
/* Measurement helper with no counterpart in the reference: the rate at which this GPU delivers 128-byte wide nodes to
 * lanes when the walk's load sequence (seven 16-byte loads per lane and node) runs with nothing behind it, in the ray
 * kernels' launch shape.  pattern 0: every lane its own node of a table of table_bytes (a power of two; 16 KiB sits in a
 * CU's L1, a few MiB in L2); pattern 1: the lanes of a wave share the node; pattern 2: plain streaming of 16-byte loads
 * over table_bytes.  per_second_out = nodes (pattern 2: 16-byte loads) per second over all lanes.  bench.py prices the
 * dominant kernel's node fetches against pattern 0 on an L1-resident table.  Uses the context's host-convenience
 * scratch; synchronises. */
int rt_probe_fetch_rate(rt_ctx* ctx, int pattern, uint64_t table_bytes, uint32_t steps, double* per_second_out);

/* Convenience for host-memory callers (the C++ host mirror): whole-frame render / trace into host buffers.
 * These allocate device scratch inside the context and synchronise. */
int rt_render_frame_host(rt_ctx* ctx, const rt_camera* cam, int max_scatters, double min_scatter_intensity,
                         uint8_t* rgb_host, int32_t* hit_id_host_or_null);
int rt_trace_primary_host(rt_ctx* ctx, const rt_camera* cam, int32_t* hit_id_host, double* dist_host);
int rt_trace_rays_host(rt_ctx* ctx, const double* rays6_host, uint64_t n, int32_t* hit_id_host, double* dist_host);
int rt_trace_rays_full_host(rt_ctx* ctx, const double* rays6_host, uint64_t n, double* records16_host);
int rt_shade_hits_host(rt_ctx* ctx, const double* hits10_host, const double* view3_host, uint64_t n, uint8_t* rgb_host);
/* Cube/Plane/Cylinder/Sphere::intersect (mesh.cpp:159,371,597,795) of ONE primitive, outside any scene, for n rays;
 * records as rt_trace_rays_full with id 0 on a hit.  Needs no uploaded scene. */
int rt_intersect_prim_host(rt_ctx* ctx, const rt_prim_desc* prim, const double* rays6_host, uint64_t n, double* records16_host);
int rt_render_samples_host(rt_ctx* ctx, const rt_camera* cam, int samples_per_pixel, uint32_t seed, int max_scatters,
                           double min_scatter_intensity, uint8_t* rgb_host);

/* 256-entry table of the reference's gamma_encode lambda (raytracer.cpp:428-432). */
int rt_gamma_lut(uint8_t lut[256]);

const char* rt_strerror(int status);
const char* rt_last_error(void);      /* thread-local detail for the last failing call */
int rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
