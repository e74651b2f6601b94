// Kernel argument blocks and launcher prototypes shared by rt_kernels.hip and rt_capi.cpp.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include "rt_internal.h"

#ifndef RT_STACK_LDS
#define RT_STACK_LDS 12            // traversal stack entries per lane kept in LDS (8 bytes each); deeper ones go private
#endif
#define RT_CTR_SHARDS 64           // counter replicas, one cache line each, summed on read
#define RT_CTR_STRIDE 16           // 64-bit words between replicas (128 bytes)
#define RT_LIGHT_BATCH 8           // lights per shadow launch (bounds the verdict planes)
#ifndef RT_RAYS_MIN_WAVES
#define RT_RAYS_MIN_WAVES 4        // waves per SIMD the persistent kernel is compiled for (caps VGPRs at 128)
#endif
#define RT_QUEUE_PARTS 8u          // work-queue heads per launch: one per XCD
#ifndef RT_RESOLVE_PARTS
#define RT_RESOLVE_PARTS 256u      // parts (each with its own count) of the list between the two passes of the resolve
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 256u              // consecutive rays a persistent wave takes per queue access (4 tiles)
#endif
#ifndef RT_CHUNK_PRIMARY
#define RT_CHUNK_PRIMARY 512u      // the same for the camera rays of a plain frame
#endif
#define RT_REFILL_MIN 32           // finished lanes that must pile up before a wave swaps rays (default)
#define RT_REFILL_MIN_INST 48      // the same for two-level scenes
// per 8x8 tile: one 128-byte entry node for its camera rays, one per light of a batch for its shadow rays, and the box of
// its hits (24 bytes, padded to 32)
#define RT_ENTRY_NODE_BYTES 128u
#define RT_TILE_BYTES(n_batch) (RT_ENTRY_NODE_BYTES * (1u + (n_batch)) + 32u)
// workspace of rt_render_rect_static: the frame planes (72 bytes per pixel), the queue heads, the shadow verdicts of one
// batch of lights (64 bytes per 8x8 tile and light) and 5 MiB for the entry search's per-tile planes (a call whose tiles
// need more goes without the entry search; results are the same either way)
#define RT_FRAME_BYTES_PER_SLOT 72u
#define RT_STATIC_TILE_BYTES (5u << 20)
#define RT_STATIC_SCRATCH_BYTES (RT_STATIC_MAX_PIXELS * RT_FRAME_BYTES_PER_SLOT + 4096u + RT_STATIC_MAX_TILES * 64u * RT_LIGHT_BATCH + 512u + RT_STATIC_TILE_BYTES)

struct DevScene {
    const rt_node32*   nodes;      // root at 0, sibling pairs at even indices
    const rt_prim_rec* recs;       // BVH slot order, loose primitives after n_slots
    const rt_mat_rec*  mats;       // by primitive id
    const rt_light*    lights;
    const uint8_t*     gamma;      // 256-entry table
    const uint8_t*     texels;     // albedo textures, 8-bit RGB
    const rt_tex_rec*  tex_table;
    const float4*      nodes4;     // wide nodes (8 x 16 bytes each), or null: walk the binary nodes
    const float4*      loose_boxes;// per loose primitive: lo xyz -, hi xyz - (infinite: unbounded); null when there are none
    float root_lo[3], root_hi[3];  // wide walk: the root's box ...
    uint32_t root_ref;             // ... and where the walk goes once inside it: a wide node, or the leaf the root is
    const rt_inst_rec* inst;       // two-level scenes: instances in the leaf order of the top-level tree
    uint32_t n_inst;
    uint32_t n_nodes, n_slots, n_loose, n_lights, n_prims, n_textures;
    float radius;                  // max |coordinate| of the scene: scales the box padding
};

struct DevCamera {                 // persp_camera.h:29-36 with the sensor-fit branch already taken
    double pos[3];
    double rot[9];
    double eff_w, eff_h, focal;
    double width, height;
    uint32_t jitter_seed;          // 0: pixel centres; otherwise seeds a per-pixel sub-pixel offset
};

// How blocks map to pixels and output slots.
//   mode 0: rectangle (x0, y0, w, h) of the frame, output row-major in the rectangle;
//           super_x = 128x128-pixel super tiles per row (256 waves each).
//   mode 1: this rank's tiles of a tiled frame (w, h = frame size, x0 = y0 = 0); output tile after tile.
// A divisor fixed for a launch: n / d is umulhi(n, m) with m = ceil(2^32 / d) (m = 0 stands for d = 1), which exceeds the
// quotient by one at most for any n < 2^32; one multiply-subtract tells (the kernels' div_by).  The compiler's own
// division by a run-time value is twenty instructions, and the pixel maps are evaluated for every ray that is set up.
struct DivU32 { uint32_t d, m; };
inline DivU32 make_div_u32(uint32_t d) { DivU32 k; k.d = d; k.m = d > 1u ? static_cast<uint32_t>((1ull << 32) / d) + (((1ull << 32) % d) ? 1u : 0u) : 0u; return k; }

struct PixelMap {
    int mode;
    int x0, y0, w, h;
    uint32_t super_x;
    int tile, tiles_x, n_tiles, rank, n_ranks;
    // the divisors the maps use (pixel_map_divisors): super_x, w | waves per tile, tiles_x, waves per tile row, tile^2, tile
    DivU32 d_super_x, d_w, d_wpt, d_tiles_x, d_wpt_x, d_t2, d_tile;
};
inline void pixel_map_divisors(PixelMap& pm) {
    pm.d_super_x = make_div_u32(pm.super_x ? pm.super_x : 1u);
    pm.d_w = make_div_u32(pm.w > 0 ? static_cast<uint32_t>(pm.w) : 1u);
    const uint32_t tile = pm.tile > 0 ? static_cast<uint32_t>(pm.tile) : 8u, wpt_x = tile / 8u ? tile / 8u : 1u;
    pm.d_wpt = make_div_u32(wpt_x * wpt_x);
    pm.d_tiles_x = make_div_u32(pm.tiles_x > 0 ? static_cast<uint32_t>(pm.tiles_x) : 1u);
    pm.d_wpt_x = make_div_u32(wpt_x);
    pm.d_t2 = make_div_u32(tile * tile);
    pm.d_tile = make_div_u32(tile);
}

// Per-frame working set of the frame pipeline (closest hit, surface records, shadow verdicts, resolve) (all planes have n entries, indexed by output slot).
struct FrameBufs {
    double *px, *py, *pz;          // hit point
    double *nx, *ny, *nz;          // unit normal, re-normalised as RayTracer::shade does
    double *lighting;              // running light sum between light batches
    uint8_t *lit;                  // [tiles][lights of the batch][64] shadow verdicts in ray order: 1 = the light reaches the point
    int32_t *id;                   // primitive whose material applies to the hit, -1 = miss / padding
    uint32_t *slot, *inst;         // what the closest-hit launch found: record slot (all ones = nothing), instance slot
    size_t n;
    // Mirror-bounce frames only (null otherwise): the chain of trace_ray_recursive is walked level by level.
    double *ray;                   // 6 planes: origin xyz, direction xyz of the ray traced at this level
    double *rawn;                  // 3 planes: the intersector's normal as it left the primitive test
    uint8_t *alive;                // levels >= 1: this slot still carries a ray
    int level;
    uint32_t *albedo;              // scenes with textures only: albedo under the hit, r | g << 8 | b << 16
    // Entry search (scenes walked through wide nodes; null otherwise): where the walks of a tile's 64 rays start instead of at the
    // root - a wide node made for the tile's beam, whose up to four children are the subtrees the beam can touch at all
    // (layout of a wide node: six planes of box coordinates, four references, all ones = empty place).
    float4   *entry_p;             // [tiles][8] camera rays of the tile
    float4   *entry_s;             // [tiles][lights of the batch][8] shadow rays of the tile towards one light
    float    *tile_box;            // [tiles][8] min xyz, max xyz of the tile's hit points (min > max: no hit), 2 unused
};

// Per-level results of a mirror-bounce frame, folded into the final colour at the end.
struct LevelBufs {
    uint32_t *local;               // [level][n] local colour, r | g << 8 | b << 16 (linear 8 bit)
    double   *fr;                  // [level][n] Fresnel weight of the bounce leaving this level
    uint32_t *tint;                // [level][n] albedo | metallic flag << 24
    uint8_t  *last;                // [n] deepest level reached
};

extern "C" {
// n_tiles: 8x8 pixel tiles of the pixel map (64 rays each); n_waves: persistent waves to launch; queue: one zeroed
// 32-bit counter per launch.
hipError_t rt_launch_primary(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm,
                             uint32_t n_tiles, uint32_t n_waves, int counting, int32_t* ids, double* dist,
                             const FrameBufs* fb, rt_counters* ctr, unsigned int* queue, int refill_min);
// Entry references for the beams of n_tiles tiles: camera rays (n_batch = 0) or shadow rays towards lights
// light0 .. light0 + n_batch - 1 (after the surface records of the tiles are known).
hipError_t rt_launch_entry_search(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, uint32_t n_tiles,
                                  uint32_t light0, uint32_t n_batch, int counting, const FrameBufs* fb, rt_counters* ctr);
hipError_t rt_launch_shadow(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm,
                            uint32_t n_tiles, uint32_t n_waves, uint32_t light0, uint32_t n_batch, int counting,
                            const FrameBufs* fb, rt_counters* ctr, unsigned int* queue, int refill_min);
hipError_t rt_launch_resolve(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const FrameBufs* fb, uint32_t light0, uint32_t n_batch,
                             int first, int last, uint8_t* rgb, unsigned int* list_heads);
hipError_t rt_launch_verdict_mask(hipStream_t st, const PixelMap* pm, const FrameBufs* fb, uint32_t n_batch, uint8_t* mask);
hipError_t rt_launch_trace_rays(hipStream_t st, const DevScene* sc, const double* rays, uint64_t n, int counting, int shade_it,
                                int32_t* ids, double* dist, uint8_t* rgb, double* records, rt_counters* ctr);
// Every primitive against every ray (no hierarchy); rays = null: the camera rays of the rectangle in pm, row-major.
hipError_t rt_launch_brute(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const double* rays, uint64_t n,
                           int counting, int32_t* ids, double* dist, double* records, rt_counters* ctr);
hipError_t rt_launch_accumulate(hipStream_t st, const uint8_t* rgb, uint32_t* acc, size_t n, int first);
hipError_t rt_launch_finish_samples(hipStream_t st, const uint32_t* acc, size_t n, int spp, const uint8_t* gamma, uint8_t* rgb);
hipError_t rt_launch_intersect_one(hipStream_t st, const rt_prim_rec* rec, const double* rays, uint64_t n, double* records);
hipError_t rt_launch_shade_hits(hipStream_t st, const DevScene* sc, const double* hits, const double* view, uint64_t n,
                                uint8_t* rgb, rt_counters* ctr);
hipError_t rt_launch_bounce_resolve(hipStream_t st, const DevScene* sc, const DevCamera* cam, const PixelMap* pm, const FrameBufs* fb, const LevelBufs* lv,
                                    uint32_t light0, uint32_t n_batch, int first, int last, int max_scatters, double min_si);
hipError_t rt_launch_bounce_fold(hipStream_t st, const DevScene* sc, const LevelBufs* lv, size_t n, uint8_t* rgb);
hipError_t rt_device_build_workspace(uint32_t n, size_t* bytes);
hipError_t rt_device_build(hipStream_t st, uint32_t n, const float* boxes, const float bounds[6], const rt_prim_rec* recs_in,
                           rt_node32* nodes_out, rt_prim_rec* recs_out, uint32_t* order_out, void* work, size_t work_bytes,
                           rt_node4* wide_out, uint8_t* wide_need);
hipError_t rt_launch_slot_ids(hipStream_t st, const rt_prim_rec* recs, uint32_t n, uint32_t* slot_prim);
// Address and size, on the current device, of the library's fixed workspace (a file-scope __device__ array in
// rt_kernels.hip; every GPU holds its own copy at its own address).  `device` must be the current device.
hipError_t rt_static_scratch(int device, void** ptr, size_t* bytes);
hipError_t rt_launch_assemble(hipStream_t st, int width, int height, int tile, int tiles_x, int n_ranks, int tiles_per_rank,
                              const uint8_t* gathered, uint8_t* frame);
}

#endif
