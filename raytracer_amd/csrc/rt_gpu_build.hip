// Hierarchy construction on the GPU: Morton order, then a top-down split along the curve.
//
// Takes the place of get_BVH_tree (s2899221/Code/raytracer.cpp:454-499: median split on the largest axis) for scenes
// that change every frame.  Primitive boxes are sorted along a 63-bit Morton curve of their centroids (rocPRIM radix
// sort).  The tree is then cut level by level: a node over the sorted slots [b, e) splits where the highest bit in
// which its first and last key differ flips - the spatial median of the octree cell that holds them - unless that
// would leave too little stack for the larger side, in which case the range is halved (the rule of the host builder:
// no leaf deeper than RT_STACK_DEPTH - 1).  Nothing is allocated per node: the two children of the node that splits
// after slot g live at nodes 2 (g + 1) and 2 (g + 1) + 1, and since every inner node has its own g, the 2 n node
// slots (root, the unused slot beside it, n - 1 pairs) are filled exactly.  Layout is the one the kernels walk: root
// at 0, sibling pairs at even indices, one primitive per leaf.  Boxes are fitted bottom-up, one launch per level.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "rt_device.h"

namespace {

__device__ inline unsigned long long spread21(unsigned long long v) {     // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8)  & 0x100F00F00F00F00Full;
    v = (v | v << 4)  & 0x10C30C30C30C30C3ull;
    v = (v | v << 2)  & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(256) k_morton(const float* __restrict__ boxes, uint32_t n, float lo_x, float lo_y, float lo_z,
                                                float inv_x, float inv_y, float inv_z, unsigned long long* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* b = boxes + static_cast<size_t>(i) * 6;
    const float cx = (0.5f * (b[0] + b[3]) - lo_x) * inv_x, cy = (0.5f * (b[1] + b[4]) - lo_y) * inv_y, cz = (0.5f * (b[2] + b[5]) - lo_z) * inv_z;
    auto q = [](float c) -> unsigned long long {
        const float s = fminf(fmaxf(c, 0.0f), 1.0f) * 2097151.0f;
        return static_cast<unsigned long long>(s == s ? s : 0.0f);
    };
    keys[i] = spread21(q(cx)) | spread21(q(cy)) << 1 | spread21(q(cz)) << 2;
    vals[i] = i;
}

// Primitive records into Morton order.
__global__ void __launch_bounds__(256) k_permute_records(const rt_prim_rec* __restrict__ in, const uint32_t* __restrict__ order, uint32_t n,
                                                         rt_prim_rec* __restrict__ out) {
    // 160-byte records as ten 16-byte pieces: one piece per thread
    const size_t t = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    const size_t slot = t / 10u, piece = t % 10u;
    if (slot >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(in + order[slot]);
    reinterpret_cast<uint4*>(out + slot)[piece] = src[piece];
}

__device__ inline void store_leaf(rt_node32* nodes, uint32_t index, const float* boxes, const uint32_t* order, uint32_t slot) {
    const float* b = boxes + static_cast<size_t>(order[slot]) * 6;
    rt_node32 nd;
    nd.lo[0] = b[0]; nd.lo[1] = b[1]; nd.lo[2] = b[2]; nd.hi[0] = b[3]; nd.hi[1] = b[4]; nd.hi[2] = b[5];
    nd.first = slot; nd.count = 1;
    nodes[index] = nd;
}

constexpr uint32_t kLevels = RT_STACK_DEPTH - 1;       // inner nodes live at depths 0 .. kLevels - 1
constexpr uint32_t kSplitBlocks = 512;

__device__ inline uint32_t ceil_log2(uint32_t m) { return m <= 1u ? 0u : 32u - static_cast<uint32_t>(__clz(m - 1u)); }

// One level of the top-down cut.  items[off[level] .. +cnt[level]) are this level's inner nodes as (node, b, e, -);
// children that are inner nodes themselves are appended to the next level's run, single slots become leaves at once.
__global__ void __launch_bounds__(256) k_split_level(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order,
                                                     const float* __restrict__ boxes, uint32_t level, uint4* __restrict__ items,
                                                     uint32_t* __restrict__ off, uint32_t* __restrict__ cnt, rt_node32* __restrict__ nodes) {
    const uint32_t begin = off[level], count = cnt[level], next = begin + count;
    if (blockIdx.x == 0 && threadIdx.x == 0) off[level + 1] = next;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint4 it = items[begin + i];
        const uint32_t b = it.y, e = it.z, m = e - b;
        bool halve = level + ceil_log2(m) >= kLevels;
        uint32_t g = 0;                                     // last slot of the left child
        if (!halve) {
            const unsigned long long first = keys[b], last = keys[e - 1];
            if (first == last) halve = true;
            else {
                const unsigned long long bit = 1ull << (63 - __clzll(static_cast<long long>(first ^ last)));
                uint32_t lo = b, hi = e - 1;                // keys[lo] has the bit clear, keys[hi] has it set
                while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (keys[mid] & bit) hi = mid; else lo = mid; }
                g = lo;
                const uint32_t big = (g + 1u - b > e - 1u - g) ? g + 1u - b : e - 1u - g;
                if (level + 1u + ceil_log2(big) > kLevels) halve = true;
            }
        }
        if (halve) g = b + (m + 1u) / 2u - 1u;
        const uint32_t pair = 2u * (g + 1u);
        nodes[it.x].first = pair; nodes[it.x].count = 0;
#pragma unroll
        for (uint32_t side = 0; side < 2u; ++side) {
            const uint32_t cb = side ? g + 1u : b, ce = side ? e : g + 1u;
            if (ce - cb == 1u) store_leaf(nodes, pair + side, boxes, order, cb);
            else items[next + atomicAdd(&cnt[level + 1], 1u)] = make_uint4(pair + side, cb, ce, 0u);
        }
    }
}

__device__ inline float box_area(const rt_node32& nd) {
    const float ex = nd.hi[0] - nd.lo[0], ey = nd.hi[1] - nd.lo[1], ez = nd.hi[2] - nd.lo[2];
    return ex*ey + ey*ez + ez*ex;
}

// Boxes of one level's inner nodes: the union of the two children, which are already in place.  With `wide` set, the
// node's wide form is written as well (wide node = its index in the binary tree): its two children, the inner one with
// the larger box replaced by its own children, twice - everything below this level is final by now.  need[i] = stack
// entries a wide walk below node i can need.
__global__ void __launch_bounds__(256) k_fit_level(uint32_t level, const uint4* __restrict__ items, const uint32_t* __restrict__ off,
                                                   const uint32_t* __restrict__ cnt, rt_node32* __restrict__ nodes,
                                                   rt_node4* __restrict__ wide, uint8_t* __restrict__ need) {
    const uint32_t begin = off[level], count = cnt[level];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t node = items[begin + i].x;
        const uint32_t pair = nodes[node].first;
        const rt_node32 l = nodes[pair], r = nodes[pair + 1];
        for (int k = 0; k < 3; ++k) { nodes[node].lo[k] = fminf(l.lo[k], r.lo[k]); nodes[node].hi[k] = fmaxf(l.hi[k], r.hi[k]); }
        if (!wide) continue;
        uint32_t ch[4] = { pair, pair + 1u, 0u, 0u };
        rt_node32 cn[4]; cn[0] = l; cn[1] = r;
        int nch = 2;
        while (nch < 4) {
            int best = -1; float best_area = -1.0f;
            for (int k = 0; k < nch; ++k) if (cn[k].count == 0u) { const float a = box_area(cn[k]); if (a > best_area) { best_area = a; best = k; } }
            if (best < 0) break;
            const uint32_t f = cn[best].first;
            ch[best] = f; cn[best] = nodes[f];
            ch[nch] = f + 1u; cn[nch] = nodes[f + 1u];
            ++nch;
        }
        rt_node4 w;
        uint32_t deepest = 0;
        for (int k = 0; k < 4; ++k) {
            if (k < nch) {
                for (int a = 0; a < 3; ++a) { w.lo[a][k] = cn[k].lo[a]; w.hi[a][k] = cn[k].hi[a]; }
                if (cn[k].count) w.ref[k] = (cn[k].count << 28) | cn[k].first;
                else { w.ref[k] = ch[k]; deepest = max(deepest, static_cast<uint32_t>(need[ch[k]])); }
            } else {
                for (int a = 0; a < 3; ++a) { w.lo[a][k] = __builtin_inff(); w.hi[a][k] = -__builtin_inff(); }
                w.ref[k] = 0xFFFFFFFFu;
            }
            w.pad[k] = 0u;
        }
        wide[node] = w;
        const uint32_t nd = static_cast<uint32_t>(nch - 1) + deepest;
        need[node] = static_cast<uint8_t>(nd > 255u ? 255u : nd);
    }
}

__global__ void k_first_item(uint32_t n, uint4* items, uint32_t* off, uint32_t* cnt) {
    for (uint32_t k = 0; k <= kLevels + 1u; ++k) { off[k] = 0; cnt[k] = 0; }
    items[0] = make_uint4(0u, 0u, n, 0u);
    cnt[0] = 1;
}

__global__ void __launch_bounds__(256) k_slot_ids(const rt_prim_rec* __restrict__ recs, uint32_t n, uint32_t* __restrict__ slot_prim) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) slot_prim[i] = recs[i].id;
}

__global__ void k_root_leaf(const float* boxes, const uint32_t* order, rt_node32* nodes) { store_leaf(nodes, 0, boxes, order, 0); }

__global__ void k_pad_node(rt_node32* nodes) {      // the unused slot beside the root: an inverted box
    rt_node32 nd;
    for (int k = 0; k < 3; ++k) { nd.lo[k] = __builtin_inff(); nd.hi[k] = -__builtin_inff(); }
    nd.first = 0; nd.count = 0;
    nodes[1] = nd;
}

} // namespace

extern "C" hipError_t rt_launch_slot_ids(hipStream_t st, const rt_prim_rec* recs, uint32_t n, uint32_t* slot_prim) {
    if (n) hipLaunchKernelGGL(k_slot_ids, dim3((n + 255u) / 256u), dim3(256), 0, st, recs, n, slot_prim);
    return hipGetLastError();
}

namespace {
size_t round256(size_t b) { return (b + 255u) & ~static_cast<size_t>(255u); }
hipError_t sort_temp_bytes(uint32_t n, size_t& bytes) {
    unsigned long long* k = nullptr; uint32_t* v = nullptr;
    bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0, 63, static_cast<hipStream_t>(nullptr));
}
} // namespace

// Device memory the build needs beside its inputs and outputs: two key arrays, one value array, the sort's own scratch.
extern "C" hipError_t rt_device_build_workspace(uint32_t n, size_t* bytes) {
    size_t temp = 0;
    const hipError_t e = sort_temp_bytes(n, temp);
    *bytes = 2 * round256(sizeof(unsigned long long) * n) + round256(sizeof(uint32_t) * n) + round256(temp ? temp : 16)
           + round256(sizeof(uint4) * n) + round256(sizeof(uint32_t) * 2 * (kLevels + 2));
    return e;
}

// boxes: 6 floats per primitive (lo xyz, hi xyz), already rounded outward; bounds: union of all boxes.
// nodes_out holds 2 n nodes, recs_out and order_out n entries, work rt_device_build_workspace(n) bytes; all device memory.
// wide_out (2 n wide nodes) and wide_need (2 n bytes) may be null: then only the binary tree is made.  Otherwise
// wide_out[i] is the wide form of inner node i (node 0: the root) and wide_need[0] the stack a wide walk can need.
// Launches only: nothing is allocated and the stream is not synchronised.
extern "C" hipError_t rt_device_build(hipStream_t st, uint32_t n, const float* boxes, const float bounds[6], const rt_prim_rec* recs_in,
                                      rt_node32* nodes_out, rt_prim_rec* recs_out, uint32_t* order_out, void* work, size_t work_bytes,
                                      rt_node4* wide_out, uint8_t* wide_need) {
    if (n == 0) return hipSuccess;
    size_t temp_bytes = 0;
    hipError_t e = sort_temp_bytes(n, temp_bytes);
    if (e != hipSuccess) return e;
    const size_t key_bytes = round256(sizeof(unsigned long long) * n), val_bytes = round256(sizeof(uint32_t) * n);
    const size_t sort_bytes = round256(temp_bytes ? temp_bytes : 16), item_bytes = round256(sizeof(uint4) * n);
    if (!work || work_bytes < 2 * key_bytes + val_bytes + sort_bytes + item_bytes + round256(sizeof(uint32_t) * 2 * (kLevels + 2)))
        return hipErrorInvalidValue;
    char* base = static_cast<char*>(work);
    unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(base + key_bytes);
    uint32_t* vals_in = reinterpret_cast<uint32_t*>(base + 2 * key_bytes);
    void* temp = base + 2 * key_bytes + val_bytes;
    uint4* items = reinterpret_cast<uint4*>(base + 2 * key_bytes + val_bytes + sort_bytes);
    uint32_t* off = reinterpret_cast<uint32_t*>(base + 2 * key_bytes + val_bytes + sort_bytes + item_bytes);
    uint32_t* cnt = off + (kLevels + 2);
    {
        // one scale for all three axes (the cube around the scene): a flat scene must not spend curve bits on its thin
        // axis, or the upper levels would cut it into slabs that overlap everywhere
        float ext = 0.0f;
        for (int k = 0; k < 3; ++k) ext = fmaxf(ext, bounds[3 + k] - bounds[k]);
        const float inv1 = (ext > 0.0f && ext < 3.0e38f) ? 1.0f / ext : 0.0f;
        const float inv[3] = { inv1, inv1, inv1 };
        const uint32_t blocks = (n + 255u) / 256u;
        hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, st, boxes, n, bounds[0], bounds[1], bounds[2], inv[0], inv[1], inv[2], keys_in, vals_in);
        e = hipGetLastError();
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, order_out, n, 0, 63, st);
        if (e == hipSuccess) {
            const size_t pieces = static_cast<size_t>(n) * 10u;
            hipLaunchKernelGGL(k_permute_records, dim3(static_cast<uint32_t>((pieces + 255u) / 256u)), dim3(256), 0, st, recs_in, order_out, n, recs_out);
            if (n == 1) {
                // a single primitive: the root is its leaf
                hipLaunchKernelGGL(k_root_leaf, dim3(1), dim3(1), 0, st, boxes, order_out, nodes_out);
            } else {
                const uint32_t blocks_l = (n / 2u + 255u) / 256u < kSplitBlocks ? (n / 2u + 255u) / 256u : kSplitBlocks;
                hipLaunchKernelGGL(k_first_item, dim3(1), dim3(1), 0, st, n, items, off, cnt);
                for (uint32_t level = 0; level < kLevels; ++level)
                    hipLaunchKernelGGL(k_split_level, dim3(blocks_l), dim3(256), 0, st, keys_out, order_out, boxes, level, items, off, cnt, nodes_out);
                for (uint32_t level = kLevels; level-- > 0;)
                    hipLaunchKernelGGL(k_fit_level, dim3(blocks_l), dim3(256), 0, st, level, items, off, cnt, nodes_out, wide_out, wide_need);
            }
            hipLaunchKernelGGL(k_pad_node, dim3(1), dim3(1), 0, st, nodes_out);
            e = hipGetLastError();
        }
    }
    return e;
}
