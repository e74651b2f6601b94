// Scene finalisation: classify primitives, flatten records in BVH order, collect statistics.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rt_scene.h"

// Shape of the tree the device builds over n primitives (rt_gpu_build.hip): it halves its range at every node, so it has
// 2 n node slots (root, the unused slot beside it, 2 n - 2 others) and depth ceil(log2 n).
static uint32_t rt_device_build_node_count(uint32_t n) { return n ? 2u * n : 0u; }
static uint32_t rt_device_build_depth(uint32_t n) {
    uint32_t d = 0;
    while ((1ull << d) < n) ++d;
    return d;
}

int rt_scene_finalize(rt_scene* s, const rt_build_opts* opts) {
    const uint32_t max_leaf = (opts && opts->max_leaf) ? opts->max_leaf : 1;
    const uint32_t bins = (opts && opts->sah_bins) ? opts->sah_bins : 32;
    const uint64_t n = s->prims.size();
    if (n >= (1ull << 28)) return RT_ERR_LIMIT;      // node references keep 28 bits for a slot / child-pair index

    s->mats.resize(n);
    std::vector<rt_prim_rec> bounded_recs; bounded_recs.reserve(n);
    std::vector<rt_box64> boxes; boxes.reserve(n);
    std::vector<rt_prim_rec> loose;
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
    if (s->device_build && boxes.size() > 16) {
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
                if (out[i]) loose.push_back(bounded_recs[i]);
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
    s->n_slots = bounded_recs.size();
    s->n_loose = loose.size();
    s->recs.clear(); s->recs.reserve(s->n_slots + s->n_loose);
    s->slot_prim.clear(); s->slot_prim.reserve(s->n_slots + s->n_loose);
    for (uint32_t slot = 0; slot < bvh.order.size(); ++slot) {
        s->recs.push_back(bounded_recs[bvh.order[slot]]);
        s->slot_prim.push_back(bounded_recs[bvh.order[slot]].id);
    }
    for (const rt_prim_rec& r : loose) { s->recs.push_back(r); s->slot_prim.push_back(r.id); }

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
