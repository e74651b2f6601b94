// Hierarchy construction on the GPU: Morton order + a balanced tree of closed-form shape.
//
// Takes the place of get_BVH_tree (s2899221/Code/raytracer.cpp:454-499: median split on the largest axis) for scenes
// that change every frame.  Primitive boxes are sorted along a 63-bit Morton curve of their centroids (rocPRIM radix
// sort), and the tree over the sorted sequence always halves its range: the node at depth d, position j covers the slots
// [floor(j n / 2^d), floor((j + 1) n / 2^d)).  Its shape therefore depends on n alone - every index below is a formula,
// no node is allocated at run time - and its depth is floor(log2 n) + 1, within the traversal stack for any scene the
// library accepts.  Layout is the one the kernels walk: root at 0, node 1 unused, sibling pairs at even indices, one
// primitive per leaf.  Boxes are fitted bottom-up, one launch per level.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "rt_device.h"

namespace {

struct Shape {
    uint32_t n;          // leaves
    uint32_t full;       // d*: deepest level whose 2^d nodes all exist (every node there covers 1 or 2 slots)
    uint32_t last_base;  // index of the first node of level d* + 1
};

__host__ __device__ inline Shape shape_of(uint32_t n) {
    Shape s; s.n = n; s.full = 0;
    while ((2u << s.full) <= n) ++s.full;           // floor(log2 n)
    s.last_base = 2u << s.full;                     // levels 1..d* occupy [2, 2^(d*+1))
    return s;
}

// first slot of the node at depth d, position j
__host__ __device__ inline uint32_t slot_begin(uint32_t n, uint32_t d, uint32_t j) {
    return static_cast<uint32_t>((static_cast<unsigned long long>(j) * n) >> d);
}

// index of the node at depth d <= d*, position j (root: 0; level d >= 1 starts at 2^d)
__host__ __device__ inline uint32_t node_index(uint32_t d, uint32_t j) { return d == 0 ? 0u : (1u << d) + j; }

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

// Level d* + 1 (children of the two-slot nodes of level d*) and the one-slot nodes of level d*: all leaves.
__global__ void __launch_bounds__(256) k_leaves(Shape s, const float* __restrict__ boxes, const uint32_t* __restrict__ order,
                                                rt_node32* __restrict__ nodes) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (1u << s.full)) return;
    const uint32_t b = slot_begin(s.n, s.full, j), e = slot_begin(s.n, s.full, j + 1);
    if (e - b == 1) { store_leaf(nodes, node_index(s.full, j), boxes, order, b); return; }
    // two slots: this node is internal; its pair sits at last_base + 2 * (two-slot nodes before j) = last_base + 2 (b - j)
    const uint32_t pair = s.last_base + 2u * (b - j);
    store_leaf(nodes, pair, boxes, order, b);
    store_leaf(nodes, pair + 1, boxes, order, b + 1);
}

// Internal nodes of one level: box = union of the two children, which are already in place.
__global__ void __launch_bounds__(256) k_fit_level(Shape s, uint32_t d, rt_node32* __restrict__ nodes) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (1u << d)) return;
    uint32_t pair;
    if (d == s.full) {
        const uint32_t b = slot_begin(s.n, d, j), e = slot_begin(s.n, d, j + 1);
        if (e - b < 2) return;                      // a leaf of this level, written by k_leaves
        pair = s.last_base + 2u * (b - j);
    } else {
        pair = node_index(d + 1, 2u * j);
    }
    const rt_node32 l = nodes[pair], r = nodes[pair + 1];
    rt_node32 nd;
    for (int k = 0; k < 3; ++k) { nd.lo[k] = fminf(l.lo[k], r.lo[k]); nd.hi[k] = fmaxf(l.hi[k], r.hi[k]); }
    nd.first = pair; nd.count = 0;
    nodes[node_index(d, j)] = nd;
}

__global__ void __launch_bounds__(256) k_slot_ids(const rt_prim_rec* __restrict__ recs, uint32_t n, uint32_t* __restrict__ slot_prim) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) slot_prim[i] = recs[i].id;
}

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
    *bytes = 2 * round256(sizeof(unsigned long long) * n) + round256(sizeof(uint32_t) * n) + round256(temp ? temp : 16);
    return e;
}

// boxes: 6 floats per primitive (lo xyz, hi xyz), already rounded outward; bounds: union of all boxes.
// nodes_out holds 2 n nodes, recs_out and order_out n entries, work rt_device_build_workspace(n) bytes; all device memory.
// Launches only: nothing is allocated and the stream is not synchronised.
extern "C" hipError_t rt_device_build(hipStream_t st, uint32_t n, const float* boxes, const float bounds[6], const rt_prim_rec* recs_in,
                                      rt_node32* nodes_out, rt_prim_rec* recs_out, uint32_t* order_out, void* work, size_t work_bytes) {
    if (n == 0) return hipSuccess;
    size_t temp_bytes = 0;
    hipError_t e = sort_temp_bytes(n, temp_bytes);
    if (e != hipSuccess) return e;
    const size_t key_bytes = round256(sizeof(unsigned long long) * n), val_bytes = round256(sizeof(uint32_t) * n);
    if (!work || work_bytes < 2 * key_bytes + val_bytes + round256(temp_bytes ? temp_bytes : 16)) return hipErrorInvalidValue;
    char* base = static_cast<char*>(work);
    unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(base + key_bytes);
    uint32_t* vals_in = reinterpret_cast<uint32_t*>(base + 2 * key_bytes);
    void* temp = base + 2 * key_bytes + val_bytes;
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
            const Shape s = shape_of(n);
            if (n == 1) {
                // a single primitive: the root is its leaf
                hipLaunchKernelGGL(k_leaves, dim3(1), dim3(256), 0, st, s, boxes, order_out, nodes_out);
            } else {
                hipLaunchKernelGGL(k_leaves, dim3(((1u << s.full) + 255u) / 256u), dim3(256), 0, st, s, boxes, order_out, nodes_out);
                for (uint32_t d = s.full + 1; d-- > 0;)
                    hipLaunchKernelGGL(k_fit_level, dim3(((1u << d) + 255u) / 256u), dim3(256), 0, st, s, d, nodes_out);
            }
            hipLaunchKernelGGL(k_pad_node, dim3(1), dim3(1), 0, st, nodes_out);
            e = hipGetLastError();
        }
    }
    return e;
}
