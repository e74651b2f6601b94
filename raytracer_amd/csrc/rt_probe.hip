// Fetch-rate probe: how fast can this GPU deliver wide BVH nodes to lanes when nothing else is in the way?
//
// The ray kernels fetch one 128-byte node per lane and step (seven 16-byte loads of which most hit the CU's L1), so the
// ceiling that matters for them is the rate of the vector-memory request path - texture addresser, L1 tag lookup, data
// return - not HBM.  That rate is not in any data sheet for gfx950, so it is measured: the probe runs the walk's own
// load sequence with no arithmetic behind it, in the walk's launch shape (one wave per workgroup, 16 waves per CU), on a
// table of nodes sized to sit in L1 (or in L2, or nowhere) and with the lanes of a wave either spread over the table
// (every lane its own node: the incoherent limit) or sharing one node per wave (the coherent limit).
// rt_probe_fetch_rate (include/rt_hip.h) returns nodes per second; bench.py prices the shadow launch against it.
#include <hip/hip_runtime.h>

#include <string>

#include "rt_device.h"
#include "rt_scene.h"

int rt_fail(int code, const std::string& msg);          // rt_capi.cpp
hipStream_t rt_ctx_stream(rt_ctx* ctx);
int rt_ctx_device(rt_ctx* ctx);
int rt_ctx_scratch(rt_ctx* ctx, int slot, size_t bytes, void** out);
uint32_t rt_ctx_max_waves(rt_ctx* ctx);

namespace {

// PATTERN 0: every lane picks its own node of the table (pseudo-random walk, independent of the data).
// PATTERN 1: the lanes of a wave share the node (one pick per wave).
// PATTERN 2: plain streaming - lane i of wave w reads 16 bytes at (step * waves + w) * 1 KiB + 16 i (table = whole buffer).
template <int PATTERN>
__global__ void __launch_bounds__(64, 4) k_probe_fetch(const float4* __restrict__ table, uint32_t mask, uint32_t steps, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (PATTERN == 1 ? blockIdx.x : blockIdx.x * 64u + lane) * 0x9E3779B1u + 0x7F4A7C15u;
    float acc = 0.0f;
    if (PATTERN == 2) {
        for (uint32_t it = 0; it < steps; ++it) {
            const size_t at = ((static_cast<size_t>(it) * gridDim.x + blockIdx.x) * 64u + lane) & mask;
            acc += table[at].x;
        }
    } else {
        for (uint32_t it = 0; it < steps; ++it) {
            s = s * 1664525u + 1013904223u;
            const float4* n4 = table + 8u * static_cast<size_t>((s >> 7) & mask);
            // the seven loads of a step of the wide walk: six planes of box coordinates and the references
            const float4 a = n4[0], b = n4[1], c = n4[2], d = n4[3], e = n4[4], f = n4[5], g = n4[6];
            acc += (a.x + b.y) + (c.z + d.w) + (e.x + f.y) + g.z;
        }
    }
    if (acc == 1.2345678e-30f) sink[0] = acc;             // never true for the table's contents: keeps the loads alive
}

} // namespace

extern "C" int rt_probe_fetch_rate(rt_ctx* ctx, int pattern, uint64_t table_bytes, uint32_t steps, double* per_second_out) {
    if (!ctx || !per_second_out) return rt_fail(RT_ERR_INVALID, "null argument");
    if (pattern < 0 || pattern > 2 || steps == 0 || table_bytes < 4096 || table_bytes > (1ull << 32) || (table_bytes & (table_bytes - 1)))
        return rt_fail(RT_ERR_INVALID, "probe: pattern 0..2, steps > 0, table a power of two in [4 KiB, 4 GiB]");
    hipError_t e = hipSetDevice(rt_ctx_device(ctx));
    void* base = nullptr;
    if (e == hipSuccess) { if (int rc = rt_ctx_scratch(ctx, 0, table_bytes + 256, &base)) return rc; }
    hipStream_t st = rt_ctx_stream(ctx);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // finite, harmless contents (a float pattern that never sums to the sentinel)
    if (e == hipSuccess) e = hipMemsetAsync(base, 0x3C, table_bytes + 256, st);
    if (e == hipSuccess) e = hipEventCreate(&t0);
    if (e == hipSuccess) e = hipEventCreate(&t1);
    const uint32_t waves = rt_ctx_max_waves(ctx);
    const float4* table = static_cast<const float4*>(base);
    float* sink = reinterpret_cast<float*>(static_cast<unsigned char*>(base) + table_bytes);
    const uint32_t mask = static_cast<uint32_t>((pattern == 2 ? table_bytes / 16u : table_bytes / 128u) - 1u);
    float ms = 0.0f;
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {          // the first pass warms the caches and the clocks
        (void)hipEventRecord(t0, st);
        switch (pattern) {
            case 0: hipLaunchKernelGGL((k_probe_fetch<0>), dim3(waves), dim3(64), 0, st, table, mask, steps, sink); break;
            case 1: hipLaunchKernelGGL((k_probe_fetch<1>), dim3(waves), dim3(64), 0, st, table, mask, steps, sink); break;
            default: hipLaunchKernelGGL((k_probe_fetch<2>), dim3(waves), dim3(64), 0, st, table, mask, steps, sink); break;
        }
        e = hipGetLastError();
        (void)hipEventRecord(t1, st);
        if (e == hipSuccess) e = hipEventSynchronize(t1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    if (e != hipSuccess) return rt_fail(RT_ERR_HIP, std::string("fetch-rate probe: ") + hipGetErrorString(e));
    // nodes (patterns 0, 1) or 16-byte loads (pattern 2) per second, summed over all lanes
    *per_second_out = static_cast<double>(waves) * 64.0 * steps / (static_cast<double>(ms) * 1e-3);
    return RT_OK;
}
