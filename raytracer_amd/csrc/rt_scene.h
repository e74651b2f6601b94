// Host-side scene: primitive descriptions, flattened records, BVH.  Internal to librt_hip.so.
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <string>
#include <utility>
#include <vector>

#include "rt_internal.h"

struct rt_box64 { double lo[3], hi[3]; };

enum rt_prim_class { RT_CLASS_BOUNDED = 0, RT_CLASS_LOOSE = 1, RT_CLASS_NEVER = 2 };

struct rt_scene {
    std::vector<rt_prim_desc> prims;                 // id order
    std::vector<rt_light>     lights;
    std::vector<std::pair<std::string, rt_camera>> cameras;   // file order
    std::vector<rt_mat_rec>   mats;                  // by id
    std::vector<rt_prim_rec>  recs;                  // BVH slot order, then the loose primitives
    std::vector<rt_node32>    nodes;                 // node 0 = root, node 1 unused, sibling pairs at even indices
    std::vector<rt_node4>     nodes4;                // the same tree collapsed into wide nodes (empty: walk the binary one)
    uint32_t wide_root_ref = 0;                      // where the wide walk starts: wide node 0, or the leaf the root is
    std::vector<uint32_t>     slot_prim;             // slot -> id (bounded, BVH order; loose ones appended)
    uint64_t n_slots = 0, n_loose = 0;
    std::vector<float>        loose_boxes;           // 8 floats per loose primitive: lo xyz, -, hi xyz, - (infinite: unbounded)
    rt_scene_info info{};
    // albedo textures (Mesh::setAlbedoTexture, mesh.h:47): pool of texels, one table entry per texture, and the
    // texture of each primitive (-1 = none)
    // RT_BUILD_ON_DEVICE: no nodes here; recs / slot_prim are in id order and boxes_f32 (6 floats per bounded primitive,
    // rounded outward) is what the device builds from
    bool device_build = false;
    std::vector<float>      boxes_f32;
    std::vector<uint8_t>    texels;
    std::vector<rt_tex_rec> tex_table;
    std::vector<int32_t>    prim_tex;
    // two-level scenes: the caller's instances and their flattened records in the leaf order of the top-level tree
    // (nodes then holds the top-level tree followed by one tree per distinct mesh)
    bool instanced = false;
    std::vector<rt_instance> instances;
    std::vector<rt_inst_rec> inst_recs;
};

// rt_prep.cpp
void rt_rotation_rpy(double roll, double pitch, double yaw, double R[9]);
rt_prim_class rt_make_record(const rt_prim_desc& p, uint32_t id, rt_prim_rec& rec, rt_box64& box);
void rt_make_material(const rt_prim_desc& p, rt_mat_rec& m);
void rt_reference_aabb(const rt_prim_desc& p, double out6[6]);
void rt_camera_basis(const double pos[3], const double dir[3], const double* up, rt_camera* cam);
void rt_pixel_to_ray(const rt_camera* cam, double px, double py, double o[3], double d[3]);
void rt_gamma_table(uint8_t lut[256]);

// rt_bvh.cpp
struct rt_bvh_result { std::vector<rt_node32> nodes; std::vector<uint32_t> order; uint32_t depth = 0; };
void rt_build_bvh(const std::vector<rt_box64>& boxes, uint32_t max_leaf, uint32_t bins, rt_bvh_result& out);
// Collapse a binary tree into wide nodes; returns the stack entries a walk of the result can need (0 and an empty
// result when the root is a leaf or there is no tree).
uint32_t rt_collapse_wide(const std::vector<rt_node32>& nodes, std::vector<rt_node4>& out);

// rt_json.cpp; tex_paths receives one entry per primitive ("" = no albedo_texture_ppm given)
int rt_parse_scene_file(const char* path, std::vector<rt_prim_desc>& prims, std::vector<rt_light>& lights,
                        std::vector<std::pair<std::string, rt_camera>>& cameras, std::vector<std::string>& tex_paths, std::string& err);
int rt_parse_scene_text(const char* text, size_t len, std::vector<rt_prim_desc>& prims, std::vector<rt_light>& lights,
                        std::vector<std::pair<std::string, rt_camera>>& cameras, std::vector<std::string>& tex_paths, std::string& err);
// rt_image.cpp: Image::loadFromFile (utils.cpp:212-250); false on any error
bool rt_load_ppm(const char* path, int32_t& width, int32_t& height, int32_t& maxval, std::vector<uint8_t>& rgb, std::string& err);
void rt_scene_attach_textures(rt_scene* s, const std::vector<std::string>& tex_paths);
int rt_parse_camera_file(const char* path, const char* camera_id, rt_camera* out, std::string& err);

// rt_scene.cpp
int rt_scene_finalize(rt_scene* s, const rt_build_opts* opts);

#endif
