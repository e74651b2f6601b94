// Binned-SAH BVH builder producing the flat 32-byte-node layout the kernels walk.
//
// Takes the place of get_BVH_tree (s2899221/Code/raytracer.cpp:454-499: recursive median split on the
// largest-extent axis, one primitive per leaf, heap nodes) with a different design:
//   * surface-area heuristic over `bins` centroid bins on all three axes;
//   * up to max_leaf primitives per leaf;
//   * node 0 is the root, node 1 is unused, the two children of a node sit side by side at an even
//     index, so one 64-byte aligned fetch brings both child boxes;
//   * depth never exceeds RT_STACK_DEPTH - 1 (the per-lane traversal stack holds one entry per level):
//     a subtree that could not otherwise fit is finished with balanced median splits;
//   * boxes are stored in FP32, rounded outward from the FP64 bounds.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "rt_scene.h"

namespace {

struct Builder {
    const std::vector<rt_box64>& boxes;
    std::vector<uint32_t>& order;
    std::vector<rt_node32>& nodes;
    std::vector<rt_box64> bb;          // per-bin scratch of sah_split
    std::vector<uint32_t> bc;
    std::vector<double> right_area;
    uint32_t max_leaf, bins, depth_seen = 0;
    static constexpr uint32_t kMaxDepth = RT_STACK_DEPTH - 1;

    Builder(const std::vector<rt_box64>& b, std::vector<uint32_t>& o, std::vector<rt_node32>& n, uint32_t ml, uint32_t bn)
        : boxes(b), order(o), nodes(n), bb(bn), bc(bn), right_area(bn), max_leaf(ml), bins(bn) {}

    static float down(double x) { float f = static_cast<float>(x); if (static_cast<double>(f) > x) f = std::nextafterf(f, -INFINITY); return f; }
    static float up(double x)   { float f = static_cast<float>(x); if (static_cast<double>(f) < x) f = std::nextafterf(f, INFINITY);  return f; }

    static double area(const rt_box64& b) {
        const double ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
        if (!(ex >= 0.0) || !(ey >= 0.0) || !(ez >= 0.0)) return 0.0;
        return ex*ey + ey*ez + ez*ex;
    }
    static void grow(rt_box64& a, const rt_box64& b) {
        for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
    }
    static rt_box64 empty() {
        rt_box64 b; for (int k = 0; k < 3; ++k) { b.lo[k] = std::numeric_limits<double>::infinity(); b.hi[k] = -b.lo[k]; } return b;
    }
    double centroid(uint32_t prim, int axis) const { return 0.5 * (boxes[prim].lo[axis] + boxes[prim].hi[axis]); }

    // levels a balanced split needs below a node holding n primitives
    uint32_t levels_needed(uint64_t n) const {
        uint32_t k = 0;
        while (n > max_leaf) { n = (n + 1) / 2; ++k; }
        return k;
    }

    void set_box(rt_node32& nd, const rt_box64& b) {
        for (int k = 0; k < 3; ++k) { nd.lo[k] = down(b.lo[k]); nd.hi[k] = up(b.hi[k]); }
    }

    // Fills nodes[self] for the primitives order[begin, end); bounds = their union box.
    void build(uint32_t self, uint32_t begin, uint32_t end, const rt_box64& bounds, uint32_t depth) {
        depth_seen = std::max(depth_seen, depth);
        set_box(nodes[self], bounds);
        const uint32_t n = end - begin;
        if (n <= max_leaf) { nodes[self].first = begin; nodes[self].count = n; return; }

        uint32_t mid = 0;
        bool have_split = false;
        const bool must_balance = depth + levels_needed(n) >= kMaxDepth;
        if (!must_balance) have_split = sah_split(begin, end, bounds, mid);
        if (have_split) {
            // keep the depth guarantee for both children
            const uint32_t big = std::max(mid - begin, end - mid);
            if (depth + 1 + levels_needed(big) > kMaxDepth) have_split = false;
        }
        if (!have_split) mid = median_split(begin, end);

        rt_box64 lb = empty(), rb = empty();
        for (uint32_t i = begin; i < mid; ++i) grow(lb, boxes[order[i]]);
        for (uint32_t i = mid; i < end; ++i) grow(rb, boxes[order[i]]);

        const uint32_t child = static_cast<uint32_t>(nodes.size());
        nodes.push_back(rt_node32{}); nodes.push_back(rt_node32{});
        nodes[self].first = child; nodes[self].count = 0;
        build(child, begin, mid, lb, depth + 1);
        build(child + 1, mid, end, rb, depth + 1);
    }

    uint32_t median_split(uint32_t begin, uint32_t end) {
        rt_box64 cb = empty();
        for (uint32_t i = begin; i < end; ++i) for (int k = 0; k < 3; ++k) {
            const double c = centroid(order[i], k);
            cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c);
        }
        int axis = 0;
        if (cb.hi[1] - cb.lo[1] > cb.hi[axis] - cb.lo[axis]) axis = 1;
        if (cb.hi[2] - cb.lo[2] > cb.hi[axis] - cb.lo[axis]) axis = 2;
        const uint32_t mid = begin + (end - begin) / 2;
        std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end, [&](uint32_t a, uint32_t b) {
            const double ca = centroid(a, axis), cb2 = centroid(b, axis);
            return ca < cb2 || (ca == cb2 && a < b);
        });
        return mid;
    }

    bool sah_split(uint32_t begin, uint32_t end, const rt_box64& bounds, uint32_t& mid_out) {
        const uint32_t n = end - begin;
        rt_box64 cb = empty();
        for (uint32_t i = begin; i < end; ++i) for (int k = 0; k < 3; ++k) {
            const double c = centroid(order[i], k);
            cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c);
        }
        const uint32_t B = bins;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1; uint32_t best_bin = 0;
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0) || !std::isfinite(ext)) continue;
            const double scale = static_cast<double>(B) / ext;
            for (uint32_t b = 0; b < B; ++b) { bb[b] = empty(); bc[b] = 0; }
            for (uint32_t i = begin; i < end; ++i) {
                const uint32_t p = order[i];
                uint32_t b = static_cast<uint32_t>(std::min<double>(B - 1, std::max(0.0, (centroid(p, axis) - lo) * scale)));
                grow(bb[b], boxes[p]); ++bc[b];
            }
            rt_box64 acc = empty();
            for (uint32_t b = B; b-- > 1;) { grow(acc, bb[b]); right_area[b] = area(acc); }
            acc = empty();
            uint32_t left_n = 0;
            for (uint32_t b = 0; b + 1 < B; ++b) {
                grow(acc, bb[b]); left_n += bc[b];
                const uint32_t right_n = n - left_n;
                if (left_n == 0 || right_n == 0) continue;
                const double cost = area(acc) * left_n + right_area[b + 1] * right_n;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        if (best_axis < 0) return false;
        (void)bounds;
        const double lo = cb.lo[best_axis], scale = static_cast<double>(B) / (cb.hi[best_axis] - cb.lo[best_axis]);
        auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t p) {
            const uint32_t b = static_cast<uint32_t>(std::min<double>(B - 1, std::max(0.0, (centroid(p, best_axis) - lo) * scale)));
            return b <= best_bin;
        });
        mid_out = static_cast<uint32_t>(it - order.begin());
        return mid_out > begin && mid_out < end;
    }
};

} // namespace

// A wide node takes the two children of a binary inner node and, twice, replaces the inner child with the largest box by
// that child's own two children: up to four subtrees whose boxes a ray tests in one step.  Children of wide node w that
// are inner nodes become wide nodes of their own; leaves keep their slot range.  The walk pushes up to three postponed
// subtrees per step, so the stack it can need is computed here (children are visited nearest first, the others wait).
uint32_t rt_collapse_wide(const std::vector<rt_node32>& bn, std::vector<rt_node4>& out) {
    out.clear();
    if (bn.empty() || bn[0].count != 0) return 0;
    struct Todo { uint32_t bin, self; };
    std::vector<Todo> todo;
    out.push_back(rt_node4{}); todo.push_back({0u, 0u});
    while (!todo.empty()) {
        const Todo td = todo.back(); todo.pop_back();
        uint32_t ch[4]; int nch = 2;
        ch[0] = bn[td.bin].first; ch[1] = ch[0] + 1;
        while (nch < 4) {
            int best = -1; double best_area = -1.0;
            for (int k = 0; k < nch; ++k) {
                const rt_node32& nd = bn[ch[k]];
                if (nd.count) continue;
                const double ex = double(nd.hi[0]) - nd.lo[0], ey = double(nd.hi[1]) - nd.lo[1], ez = double(nd.hi[2]) - nd.lo[2];
                const double a = ex*ey + ey*ez + ez*ex;
                if (a > best_area) { best_area = a; best = k; }
            }
            if (best < 0) break;
            const uint32_t f = bn[ch[best]].first;
            ch[best] = f; ch[nch++] = f + 1;
        }
        rt_node4 w; std::memset(&w, 0, sizeof(w));
        for (int k = 0; k < 4; ++k) {
            if (k < nch) {
                const rt_node32& nd = bn[ch[k]];
                for (int a = 0; a < 3; ++a) { w.lo[a][k] = nd.lo[a]; w.hi[a][k] = nd.hi[a]; }
                if (nd.count) w.ref[k] = (nd.count << 28) | nd.first;
                else { w.ref[k] = static_cast<uint32_t>(out.size()); out.push_back(rt_node4{}); todo.push_back({ch[k], w.ref[k]}); }
            } else {
                for (int a = 0; a < 3; ++a) { w.lo[a][k] = INFINITY; w.hi[a][k] = -INFINITY; }
                w.ref[k] = 0xFFFFFFFFu;
            }
        }
        out[td.self] = w;
    }
    // stack need: entering child c of a node leaves the node's other children on the stack
    std::vector<uint32_t> need(out.size(), 0);
    for (size_t i = out.size(); i-- > 0;) {                 // children have larger indices than their parent
        uint32_t n_children = 0, deepest = 0;
        for (int k = 0; k < 4; ++k) {
            if (out[i].ref[k] == 0xFFFFFFFFu) continue;
            ++n_children;
            if ((out[i].ref[k] >> 28) == 0) deepest = std::max(deepest, need[out[i].ref[k]]);
        }
        need[i] = (n_children ? n_children - 1 : 0) + deepest;
    }
    return need[0];
}

void rt_build_bvh(const std::vector<rt_box64>& boxes, uint32_t max_leaf, uint32_t bins, rt_bvh_result& out) {
    out.nodes.clear(); out.order.clear(); out.depth = 0;
    const uint32_t n = static_cast<uint32_t>(boxes.size());
    if (n == 0) return;
    if (max_leaf < 1) max_leaf = 1;
    if (max_leaf > 8) max_leaf = 8;
    if (bins < 4) bins = 4;
    if (bins > 32) bins = 32;
    out.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.order[i] = i;
    out.nodes.reserve(2 * static_cast<size_t>(n) + 2);
    out.nodes.push_back(rt_node32{});
    out.nodes.push_back(rt_node32{});          // unused pad so that sibling pairs start at even indices
    // an inverted box for the unused node: never entered even if something were to look at it
    for (int k = 0; k < 3; ++k) { out.nodes[1].lo[k] = INFINITY; out.nodes[1].hi[k] = -INFINITY; }
    Builder b(boxes, out.order, out.nodes, max_leaf, bins);
    rt_box64 all = Builder::empty();
    for (uint32_t i = 0; i < n; ++i) Builder::grow(all, boxes[i]);
    b.build(0, 0, n, all, 0);
    out.depth = b.depth_seen;
}
