// Scene finalisation: classify primitives, flatten records in BVH order, collect statistics.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "rt_scene.h"

// Size of the tree the device builds over n primitives (rt_gpu_build.hip): one primitive per leaf, so 2 n node slots
// (root, the unused slot beside it, n - 1 sibling pairs); its depth depends on the data and is bounded like the host
// builder's.
static uint32_t rt_device_build_node_count(uint32_t n) { return n ? 2u * n : 0u; }
static uint32_t rt_device_build_depth(uint32_t n) {
    if (n <= 1) return 0;
    return n - 1 < RT_STACK_DEPTH - 1 ? n - 1 : RT_STACK_DEPTH - 1;
}

// Two-level scene: one tree per distinct mesh (a run of the primitive pool), one tree over the instances' world boxes.
// nodes = [top-level tree | mesh trees...], all references absolute; a top-level leaf holds one instance slot.
static int finalize_instanced(rt_scene* s, uint32_t max_leaf, uint32_t bins) {
    const uint64_t n = s->prims.size();
    rt_scene_info& info = s->info;
    std::memset(&info, 0, sizeof(info));
    for (int k = 0; k < 3; ++k) { info.bounds[k] = INFINITY; info.bounds[3 + k] = -INFINITY; }
    if (!s->prim_tex.empty() || !s->tex_table.empty()) return RT_ERR_INVALID;

    s->mats.resize(n);
    std::vector<rt_prim_rec> rec(n); std::vector<rt_box64> box(n); std::vector<uint8_t> cls(n);
    for (uint64_t i = 0; i < n; ++i) {
        const rt_prim_desc& p = s->prims[i];
        rt_make_material(p, s->mats[i]);
        if (p.kind >= 0 && p.kind < 8) ++info.n_by_kind[p.kind];
        cls[i] = static_cast<uint8_t>(rt_make_record(p, static_cast<uint32_t>(i), rec[i], box[i]));
    }

    struct MeshTree { rt_box64 bounds; bool empty = true; uint32_t root_ref = 0, wide_root_ref = 0, depth = 0; float radius = 0.0f; bool built = false; };
    std::map<std::pair<uint32_t, uint32_t>, MeshTree> meshes;
    uint64_t id_total = 0;
    for (const rt_instance& in : s->instances) {
        if (static_cast<uint64_t>(in.first_prim) + in.n_prims > n) return RT_ERR_INVALID;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {           // rot^T rot = 1
            double dot = 0.0;
            for (int k = 0; k < 3; ++k) dot += in.rot[k*3 + a] * in.rot[k*3 + b];
            if (!(std::fabs(dot - (a == b ? 1.0 : 0.0)) < 1e-9)) return RT_ERR_INVALID;
        }
        for (int k = 0; k < 3; ++k) if (!std::isfinite(in.trans[k])) return RT_ERR_INVALID;
        id_total += in.n_prims;
        MeshTree& m = meshes[std::make_pair(in.first_prim, in.n_prims)];
        if (m.built) continue;
        m.built = true;
        for (int k = 0; k < 3; ++k) { m.bounds.lo[k] = INFINITY; m.bounds.hi[k] = -INFINITY; }
        for (uint32_t i = in.first_prim; i < in.first_prim + in.n_prims; ++i) {
            if (cls[i] == RT_CLASS_LOOSE) return RT_ERR_INVALID;
            if (cls[i] != RT_CLASS_BOUNDED) continue;
            m.empty = false;
            for (int k = 0; k < 3; ++k) { m.bounds.lo[k] = std::min(m.bounds.lo[k], box[i].lo[k]); m.bounds.hi[k] = std::max(m.bounds.hi[k], box[i].hi[k]); }
        }
        double r = 0.0;
        if (!m.empty) for (int k = 0; k < 3; ++k) r = std::max(r, std::max(std::fabs(m.bounds.lo[k]), std::fabs(m.bounds.hi[k])));
        m.radius = std::nextafterf(static_cast<float>(r), INFINITY);
    }
    if (id_total >= (1ull << 31)) return RT_ERR_LIMIT;

    // world boxes of the instances that can be hit, and the tree over them
    std::vector<rt_box64> inst_boxes; std::vector<uint32_t> inst_of_box;
    for (uint32_t k = 0; k < s->instances.size(); ++k) {
        const rt_instance& in = s->instances[k];
        const MeshTree& m = meshes[std::make_pair(in.first_prim, in.n_prims)];
        if (m.empty) continue;
        rt_box64 wb; for (int a = 0; a < 3; ++a) { wb.lo[a] = INFINITY; wb.hi[a] = -INFINITY; }
        for (int corner = 0; corner < 8; ++corner) {
            const double c[3] = { (corner & 1) ? m.bounds.hi[0] : m.bounds.lo[0], (corner & 2) ? m.bounds.hi[1] : m.bounds.lo[1],
                                  (corner & 4) ? m.bounds.hi[2] : m.bounds.lo[2] };
            for (int a = 0; a < 3; ++a) {
                const double w = in.rot[a*3]*c[0] + in.rot[a*3 + 1]*c[1] + in.rot[a*3 + 2]*c[2] + in.trans[a];
                wb.lo[a] = std::min(wb.lo[a], w); wb.hi[a] = std::max(wb.hi[a], w);
            }
        }
        for (int a = 0; a < 3; ++a) {                      // the transform's own rounding, generously
            const double pad = 1e-12 * std::max(std::fabs(wb.lo[a]), std::fabs(wb.hi[a]));
            wb.lo[a] -= pad; wb.hi[a] += pad;
            info.bounds[a] = std::min(info.bounds[a], wb.lo[a]); info.bounds[3 + a] = std::max(info.bounds[3 + a], wb.hi[a]);
        }
        inst_boxes.push_back(wb); inst_of_box.push_back(k);
    }
    std::vector<uint64_t> id_base(s->instances.size() + 1, 0);
    for (size_t k = 0; k < s->instances.size(); ++k) id_base[k + 1] = id_base[k] + s->instances[k].n_prims;

    rt_bvh_result top;
    rt_build_bvh(inst_boxes, 1, bins, top);
    s->nodes = top.nodes;
    // the same trees as wide nodes, one array: the top-level tree's first, each mesh's behind it; references are made
    // absolute as the pieces are appended.  wide_need = stack entries the wide walk can need (both levels and the marker)
    s->nodes4.clear();
    uint32_t wide_need = rt_collapse_wide(top.nodes, s->nodes4), wide_mesh_need = 0;
    s->wide_root_ref = (!top.nodes.empty() && top.nodes[0].count) ? (top.nodes[0].count << 28) | top.nodes[0].first : 0u;
    s->recs.clear(); s->slot_prim.clear();
    uint32_t mesh_depth = 0;
    for (auto& kv : meshes) {
        MeshTree& m = kv.second;
        m.built = false;
        if (m.empty) continue;
        std::vector<rt_box64> boxes; std::vector<uint32_t> prim_of;
        for (uint32_t i = kv.first.first; i < kv.first.first + kv.first.second; ++i)
            if (cls[i] == RT_CLASS_BOUNDED) { boxes.push_back(box[i]); prim_of.push_back(i); }
        rt_bvh_result t;
        rt_build_bvh(boxes, max_leaf, bins, t);
        const uint32_t node_base = static_cast<uint32_t>(s->nodes.size()), slot_base = static_cast<uint32_t>(s->recs.size());
        if (static_cast<uint64_t>(node_base) + t.nodes.size() >= (1ull << 29) || static_cast<uint64_t>(slot_base) + boxes.size() >= (1ull << 28))
            return RT_ERR_LIMIT;
        for (rt_node32 nd : t.nodes) {
            if (nd.count == 0) nd.first += node_base; else nd.first += slot_base;
            s->nodes.push_back(nd);
        }
        s->nodes[node_base + 1] = t.nodes[1];                // the unused slot beside the root stays an inverted box
        const rt_node32& root = s->nodes[node_base];
        m.root_ref = root.count ? (root.count << 28) | root.first : (root.first >> 1);
        {
            std::vector<rt_node4> w4;
            wide_mesh_need = std::max(wide_mesh_need, rt_collapse_wide(t.nodes, w4));
            const uint32_t base4 = static_cast<uint32_t>(s->nodes4.size());
            m.wide_root_ref = root.count ? m.root_ref : base4;
            for (rt_node4 w : w4) {
                for (int k = 0; k < 4; ++k) {
                    if (w.ref[k] == 0xFFFFFFFFu) continue;
                    if (w.ref[k] >> 28) w.ref[k] = (w.ref[k] & 0xF0000000u) | ((w.ref[k] & 0x0FFFFFFFu) + slot_base);
                    else w.ref[k] += base4;
                }
                s->nodes4.push_back(w);
            }
        }
        m.depth = t.depth;
        mesh_depth = std::max(mesh_depth, t.depth);
        for (uint32_t slot = 0; slot < t.order.size(); ++slot) {
            s->recs.push_back(rec[prim_of[t.order[slot]]]);
            s->slot_prim.push_back(prim_of[t.order[slot]]);
        }
    }
    s->inst_recs.clear(); s->inst_recs.reserve(top.order.size());
    for (uint32_t slot = 0; slot < top.order.size(); ++slot) {
        const uint32_t k = inst_of_box[top.order[slot]];
        const rt_instance& in = s->instances[k];
        const MeshTree& m = meshes[std::make_pair(in.first_prim, in.n_prims)];
        rt_inst_rec r; std::memset(&r, 0, sizeof(r));
        for (int a = 0; a < 9; ++a) r.rot[a] = in.rot[a];
        for (int a = 0; a < 3; ++a) r.trans[a] = in.trans[a];
        r.root_ref = m.root_ref; r.pad[0] = m.wide_root_ref;      // the wide one replaces it below when the scene goes wide
        r.id_off = static_cast<int32_t>(static_cast<int64_t>(id_base[k]) - static_cast<int64_t>(in.first_prim));
        r.radius = m.radius;
        r.index = k;
        s->inst_recs.push_back(r);
    }
    if (s->nodes4.size() >= RT_WIDE_NODE_LIMIT || wide_need + 1 + wide_mesh_need >= RT_WIDE_STACK_LIMIT) s->nodes4.clear();
    if (!s->nodes4.empty()) for (rt_inst_rec& r : s->inst_recs) r.root_ref = r.pad[0];
    for (rt_inst_rec& r : s->inst_recs) r.pad[0] = 0;
    s->n_slots = s->recs.size(); s->n_loose = 0; s->loose_boxes.clear();
    s->device_build = false; s->boxes_f32.clear();

    info.n_prims = n; info.n_slots = s->n_slots; info.n_loose = 0;
    info.n_nodes = static_cast<uint32_t>(s->nodes.size());
    info.depth = s->inst_recs.empty() ? 0 : top.depth + 1 + mesh_depth;     // stack entries a walk can need
    info.n_lights = static_cast<uint32_t>(s->lights.size());
    info.n_cameras = static_cast<uint32_t>(s->cameras.size());
    info.n_instances = s->instances.size(); info.n_instanced_prims = id_total;
    double radius = 0.0;
    if (!s->inst_recs.empty()) for (int k = 0; k < 6; ++k) radius = std::max(radius, std::fabs(info.bounds[k]));
    else for (int k = 0; k < 6; ++k) info.bounds[k] = 0.0;
    for (const rt_light& L : s->lights) for (int k = 0; k < 3; ++k) if (std::isfinite(L.pos[k])) radius = std::max(radius, std::fabs(L.pos[k]));
    info.radius = radius;
    return RT_OK;
}

int rt_scene_finalize(rt_scene* s, const rt_build_opts* opts) {
    const uint32_t max_leaf = (opts && opts->max_leaf) ? opts->max_leaf : 1;
    const uint32_t bins = (opts && opts->sah_bins) ? opts->sah_bins : 32;
    if (s->instanced) {
        if (opts && (opts->flags & RT_BUILD_ON_DEVICE)) return RT_ERR_INVALID;
        return finalize_instanced(s, max_leaf, bins);
    }
    const uint64_t n = s->prims.size();
    if (n >= (1ull << 28)) return RT_ERR_LIMIT;      // node references keep 28 bits for a slot / child-pair index

    s->mats.resize(n);
    std::vector<rt_prim_rec> bounded_recs; bounded_recs.reserve(n);
    std::vector<rt_box64> boxes; boxes.reserve(n);
    std::vector<rt_prim_rec> loose;
    std::vector<std::pair<uint32_t, rt_box64>> loose_bounds;       // (id, box) of the bounded ones among them
    rt_scene_info& info = s->info;
    std::memset(&info, 0, sizeof(info));
    for (int k = 0; k < 3; ++k) { info.bounds[k] = INFINITY; info.bounds[3 + k] = -INFINITY; }

    for (uint64_t i = 0; i < n; ++i) {
        const rt_prim_desc& p = s->prims[i];
        rt_make_material(p, s->mats[i]);
        if (i < s->prim_tex.size()) s->mats[i].tex = s->prim_tex[i];
        if (p.kind >= 0 && p.kind < 8) ++info.n_by_kind[p.kind];
        rt_prim_rec rec; rt_box64 box;
        switch (rt_make_record(p, static_cast<uint32_t>(i), rec, box)) {
            case RT_CLASS_BOUNDED:
                bounded_recs.push_back(rec); boxes.push_back(box);
                for (int k = 0; k < 3; ++k) {
                    info.bounds[k] = std::min(info.bounds[k], box.lo[k]);
                    info.bounds[3 + k] = std::max(info.bounds[3 + k], box.hi[k]);
                }
                break;
            case RT_CLASS_LOOSE: loose.push_back(rec); break;
            default: break;
        }
    }

    rt_bvh_result bvh;
    s->device_build = opts && (opts->flags & RT_BUILD_ON_DEVICE);
    s->boxes_f32.clear();
    if (boxes.size() > 16) {
        // A balanced tree over a space-filling curve has no good place for a primitive that spans a large part of the
        // scene (a ground plane): every ancestor of its leaf would span the scene too, and every ray would walk that
        // chain.  Up to 8 such primitives are set beside the tree and tested against every ray instead.
        double ext = 0.0;
        for (int k = 0; k < 3; ++k) ext = std::max(ext, info.bounds[3 + k] - info.bounds[k]);
        std::vector<std::pair<double, uint32_t>> big;
        for (uint32_t i = 0; i < boxes.size(); ++i) {
            double e = 0.0;
            for (int k = 0; k < 3; ++k) e = std::max(e, boxes[i].hi[k] - boxes[i].lo[k]);
            if (e >= 0.25 * ext) big.emplace_back(-e, i);
        }
        if (!big.empty() && big.size() <= 8) {
            std::vector<rt_prim_rec> keep_recs; std::vector<rt_box64> keep_boxes;
            std::vector<uint8_t> out(boxes.size(), 0);
            for (const auto& b : big) out[b.second] = 1;
            for (uint32_t i = 0; i < boxes.size(); ++i) {
                if (out[i]) { loose.push_back(bounded_recs[i]); loose_bounds.emplace_back(bounded_recs[i].id, boxes[i]); }
                else { keep_recs.push_back(bounded_recs[i]); keep_boxes.push_back(boxes[i]); }
            }
            std::sort(loose.begin(), loose.end(), [](const rt_prim_rec& a, const rt_prim_rec& b) { return a.id < b.id; });
            bounded_recs.swap(keep_recs); boxes.swap(keep_boxes);
        }
    }
    if (s->device_build) {
        // identity order; the device sorts.  Boxes go over in FP32, rounded outward like the host builder's.
        bvh.order.resize(boxes.size());
        for (uint32_t i = 0; i < boxes.size(); ++i) bvh.order[i] = i;
        bvh.depth = rt_device_build_depth(static_cast<uint32_t>(boxes.size()));
        s->boxes_f32.reserve(boxes.size() * 6);
        for (const rt_box64& b : boxes) {
            for (int k = 0; k < 3; ++k) { float f = static_cast<float>(b.lo[k]); if (static_cast<double>(f) > b.lo[k]) f = std::nextafterf(f, -INFINITY); s->boxes_f32.push_back(f); }
            for (int k = 0; k < 3; ++k) { float f = static_cast<float>(b.hi[k]); if (static_cast<double>(f) < b.hi[k]) f = std::nextafterf(f, INFINITY); s->boxes_f32.push_back(f); }
        }
    } else {
        rt_build_bvh(boxes, max_leaf, bins, bvh);
    }
    s->nodes.swap(bvh.nodes);
    s->nodes4.clear(); s->wide_root_ref = 0;
    if (!s->device_build && (rt_collapse_wide(s->nodes, s->nodes4) >= RT_WIDE_STACK_LIMIT || s->nodes4.size() >= RT_WIDE_NODE_LIMIT))
        s->nodes4.clear();   // too lopsided for the wide walk's stack, or too many nodes for its 32-bit offsets
    s->n_slots = bounded_recs.size();
    s->n_loose = loose.size();
    s->recs.clear(); s->recs.reserve(s->n_slots + s->n_loose);
    s->slot_prim.clear(); s->slot_prim.reserve(s->n_slots + s->n_loose);
    for (uint32_t slot = 0; slot < bvh.order.size(); ++slot) {
        s->recs.push_back(bounded_recs[bvh.order[slot]]);
        s->slot_prim.push_back(bounded_recs[bvh.order[slot]].id);
    }
    // boxes of the primitives beside the tree (FP32, rounded outward; everything for an unbounded one): a ray that misses
    // the padded box is spared the primitive's FP64 test
    s->loose_boxes.clear();
    for (const rt_prim_rec& r : loose) {
        s->recs.push_back(r); s->slot_prim.push_back(r.id);
        float b[8] = { -INFINITY, -INFINITY, -INFINITY, 0.0f, INFINITY, INFINITY, INFINITY, 0.0f };
        for (const auto& lb : loose_bounds)
            if (lb.first == r.id)
                for (int k = 0; k < 3; ++k) {
                    float lo = static_cast<float>(lb.second.lo[k]); if (static_cast<double>(lo) > lb.second.lo[k]) lo = std::nextafterf(lo, -INFINITY);
                    float hi = static_cast<float>(lb.second.hi[k]); if (static_cast<double>(hi) < lb.second.hi[k]) hi = std::nextafterf(hi, INFINITY);
                    b[k] = lo; b[4 + k] = hi;
                }
        s->loose_boxes.insert(s->loose_boxes.end(), b, b + 8);
    }

    info.n_prims = n; info.n_slots = s->n_slots; info.n_loose = s->n_loose;
    info.n_nodes = s->device_build ? rt_device_build_node_count(static_cast<uint32_t>(s->n_slots)) : static_cast<uint32_t>(s->nodes.size());
    info.depth = bvh.depth;
    info.n_lights = static_cast<uint32_t>(s->lights.size());
    info.n_cameras = static_cast<uint32_t>(s->cameras.size());
    double radius = 0.0;
    if (s->n_slots) for (int k = 0; k < 6; ++k) radius = std::max(radius, std::fabs(info.bounds[k]));
    else for (int k = 0; k < 6; ++k) info.bounds[k] = 0.0;
    for (const rt_light& L : s->lights) for (int k = 0; k < 3; ++k) if (std::isfinite(L.pos[k])) radius = std::max(radius, std::fabs(L.pos[k]));
    info.radius = radius;
    return RT_OK;
}
