// Microbenchmark: divergent 16-B gathers (the BVH node access pattern) -- lane-loads per second by table size.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ tab, uint32_t mask, int iters, int per_node, float4* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx = ((s >> 4) & mask) * 2u;      // 32-B records
        float4 a = tab[idx];
        acc.x += a.x; acc.y += a.y;
        if (per_node == 2) { float4 b = tab[idx + 1]; acc.z += b.z; acc.w += b.w; }
        s ^= __float_as_uint(a.x) & 1u;             // dependent chain like a traversal
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}
int main() {
    const size_t max_rec = 1u << 25;                // 32M records x 32 B = 1 GiB
    float4* tab; hipMalloc(&tab, max_rec * 32); hipMemset(tab, 0, max_rec * 32);
    const int blocks = 256 * 20; float4* out; hipMalloc(&out, blocks * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per_node = 1; per_node <= 2; per_node++)
        for (int lg = 10; lg <= 25; lg += 3) {
            const uint32_t mask = (1u << lg) - 1; const int iters = 2000;
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, mask, 100, per_node, out);
            hipEventRecord(e0); hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, per_node, out); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double visits = (double)blocks * 256 * iters;
            printf("loads/node %d  table %8.1f MB : %.3f ms  %.1f Gvisits/s  %.1f GB/s useful (%.2f lane-loads/clk/CU @2.4GHz)\n", per_node, (double)(mask + 1) * 32 / 1e6, ms,
                   visits / ms / 1e6, visits * 16 * per_node / ms / 1e6, visits * per_node / (ms * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
