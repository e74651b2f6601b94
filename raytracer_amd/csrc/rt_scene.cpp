// Scene finalisation: classify primitives, flatten records in BVH order, collect statistics.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rt_scene.h"

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
    rt_build_bvh(boxes, max_leaf, bins, bvh);
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
    info.n_nodes = static_cast<uint32_t>(s->nodes.size()); info.depth = bvh.depth;
    info.n_lights = static_cast<uint32_t>(s->lights.size());
    info.n_cameras = static_cast<uint32_t>(s->cameras.size());
    double radius = 0.0;
    if (s->n_slots) for (int k = 0; k < 6; ++k) radius = std::max(radius, std::fabs(info.bounds[k]));
    else for (int k = 0; k < 6; ++k) info.bounds[k] = 0.0;
    for (const rt_light& L : s->lights) for (int k = 0; k < 3; ++k) if (std::isfinite(L.pos[k])) radius = std::max(radius, std::fabs(L.pos[k]));
    info.radius = radius;
    return RT_OK;
}
