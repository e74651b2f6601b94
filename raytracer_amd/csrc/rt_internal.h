// Internal layouts shared by the host-side scene builder and the HIP kernels.
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H

#include <stdint.h>
#include "rt_hip.h"

// Flattened primitive record, 160 bytes, read by the kernels with 16-byte loads.
//   cube / cylinder : d[0..2] T, d[3..11] R row-major, d[12..14] S, d[15..17] 1/S (0 where S == 0)
//   sphere          : d[0..2] C, d[3..5] S, d[6..8] 1/S
//   plane / triangle: d[0..8] inverse of [e1 e2 n] row-major, d[9..11] P0, d[12..14] e1, d[15..17] e2
// All derived values are computed on the host with the reference's operation order
// (s2899221/Code/mesh.cpp:122-145, 176-178, 803-847) so the device never evaluates sin/cos.
struct rt_prim_rec {
    double   d[18];
    uint32_t kind;
    uint32_t id;        // index in the caller's primitive list (reference id order)
    uint32_t pad[2];
};

// Per-primitive shading inputs, indexed by id: Mesh material getters + solid evaluate_albedo
// (s2899221/Code/mesh.h:36-59, mesh.cpp:15-45).
struct rt_mat_rec {
    double  metallic, roughness, ior;
    uint8_t albedo[3];             // solid colour, already 8 bit (mesh.cpp:40-43)
    uint8_t pad;
    int32_t tex;                   // index into the scene's texture table, -1 = solid colour
};

// One albedo texture inside the scene's texel pool (8-bit RGB, rows top to bottom).
struct rt_tex_rec { uint64_t offset; int32_t width, height; };

// One instance of a two-level scene, 128 bytes, in the leaf order of the tree over the instances.
struct rt_inst_rec {
    double   rot[9];               // mesh frame -> world, row-major
    double   trans[3];
    uint32_t root_ref;             // node reference the walk starts from inside the mesh's tree
    int32_t  id_off;               // hit id = record id + id_off (first id of the instance - first primitive of its mesh)
    float    radius;               // max |coordinate| of the mesh in its own frame: scales the box padding there
    uint32_t index;                // position in the caller's instance list
    uint32_t pad[4];
};

#define RT_STACK_DEPTH_INST 64     // two-level scenes: both trees and the marker between them
// A walk that starts from a tile's entry node (up to four subtrees of the hierarchy picked for the tile) can have three
// of them waiting on the stack before it descends into the fourth, where a walk from the root reaches that subtree with
// fewer entries pushed: peak use is at most (what the collapsed tree needs) + RT_ENTRY_STACK_EXTRA.  Every gate that
// decides whether a tree may be walked through its wide nodes keeps that margin.
#define RT_ENTRY_STACK_EXTRA 3
#define RT_WIDE_NODE_LIMIT (1u << 25) // wide nodes a scene may have: the kernels address them as a 32-bit byte offset (index * 128)
#define RT_WIDE_STACK_LIMIT (RT_STACK_DEPTH_INST - 1 - RT_ENTRY_STACK_EXTRA)   // largest `need` the wide walk accepts
#define RT_STACK_DEPTH 32          // traversal stack entries per lane; the builder keeps depth <= 31
#define RT_MAX_SCATTERS 8
#define RT_PAD_SHIFT 18            // FP32 box padding = 2^-18 * max(|ray origin|, scene radius)

#ifdef __cplusplus
static_assert(sizeof(rt_prim_rec) == 160, "record layout");
static_assert(sizeof(rt_mat_rec) == 32, "material layout");
static_assert(sizeof(rt_inst_rec) == 128, "instance layout");
static_assert(sizeof(rt_node32) == 32, "node layout");
static_assert(sizeof(rt_node4) == 128, "wide node layout");
static_assert(sizeof(rt_prim_desc) == 160, "description layout");
#endif

#endif
