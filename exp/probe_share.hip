// How does the L1 path merge lanes that ask for the same node?  Every lane runs the wide walk's seven 16-byte loads of
// one 128-byte node per step; lanes are grouped so that `distinct` nodes are asked for per wave, with the lanes of a
// group either next to each other (adjacent) or spread with a stride (interleaved).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <bool INTERLEAVED>
__global__ void __launch_bounds__(64, 4) k(const float4* __restrict__ table, uint32_t mask, uint32_t steps, uint32_t distinct, float* sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t group = INTERLEAVED ? lane % distinct : lane / (64u / distinct);
    uint32_t s = (blockIdx.x * 64u + group) * 0x9E3779B1u + 0x7F4A7C15u;
    float acc = 0.0f;
    for (uint32_t it = 0; it < steps; ++it) {
        s = s * 1664525u + 1013904223u;
        const float4* n4 = table + 8u * static_cast<size_t>((s >> 7) & mask);
        const float4 a = n4[0], b = n4[1], c = n4[2], d = n4[3], e = n4[4], f = n4[5], g = n4[6];
        acc += (a.x + b.y) + (c.z + d.w) + (e.x + f.y) + g.z;
    }
    if (acc == 1.2345678e-30f) sink[0] = acc;
}
int main() {
    const size_t bytes = 16384;           // L1-resident table
    void* base; hipMalloc(&base, bytes + 256); hipMemset(base, 0x3C, bytes + 256);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const uint32_t waves = p.multiProcessorCount * 16, steps = 4000, mask = bytes / 128 - 1;
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int inter = 0; inter < 2; ++inter)
        for (uint32_t distinct = 1; distinct <= 64; distinct *= 2) {
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(t0);
                if (inter) hipLaunchKernelGGL(k<true>, dim3(waves), dim3(64), 0, 0, (const float4*)base, mask, steps, distinct, (float*)((char*)base + bytes));
                else hipLaunchKernelGGL(k<false>, dim3(waves), dim3(64), 0, 0, (const float4*)base, mask, steps, distinct, (float*)((char*)base + bytes));
                hipEventRecord(t1); hipEventSynchronize(t1); hipEventElapsedTime(&ms, t0, t1);
            }
            const double wave_steps = double(waves) * steps / (ms * 1e-3);
            printf("%s distinct %2u per wave: %.3f ms, %.1f G wave-steps/s, %.2f clocks per wave-step and CU at 2.4 GHz\n", inter ? "interleaved" : "adjacent   ",
                   distinct, ms, wave_steps * 1e-9, 2.4e9 * p.multiProcessorCount / wave_steps);
        }
    return 0;
}
