// Microbenchmark 2: 32-B records fetched (a) by one lane with two dwordx4 loads, (b) by a lane PAIR with one dwordx4 each
// (two instructions cover 64 records), (c) 64-B records by lane quads.  Reports records/s.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_pair(const float4* __restrict__ tab, uint32_t mask, int iters, float4* out) {
    const uint32_t lane = threadIdx.x & 63u, gw = (blockIdx.x * 256u + threadIdx.x) >> 6;
    // per-RAY state lives in every lane (64 rays per wave): s is the ray's own chain
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = ((s >> 4) & mask) * 2u;                 // this lane's (ray's) record
        // instruction 1 serves rays 0..31: lane l fetches half (l&1) of ray (l>>1)'s record; instruction 2 serves rays 32..63
        const uint32_t i0 = __shfl(idx, lane >> 1, 64), i1 = __shfl(idx, 32 + (lane >> 1), 64);
        const float4 a = tab[i0 + (lane & 1u)], b = tab[i1 + (lane & 1u)];
        // bring both halves back to the owning lane: ray r<32 reads from lanes 2r, 2r+1 of `a`; r>=32 from `b`
        const uint32_t src = (lane & 31u) * 2u; const bool hi = lane >= 32u;
        const float ax = __shfl(a.x, src, 64), bx = __shfl(b.x, src, 64), ay = __shfl(a.y, src + 1, 64), by = __shfl(b.y, src + 1, 64);
        const float x = hi ? bx : ax, y = hi ? by : ay;
        acc.x += x; acc.y += y;
        s ^= __float_as_uint(x) & 1u;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_solo(const float4* __restrict__ tab, uint32_t mask, int iters, float4* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = ((s >> 4) & mask) * 2u;
        const float4 a = tab[idx], b = tab[idx + 1];
        acc.x += a.x; acc.y += b.y;
        s ^= __float_as_uint(a.x) & 1u;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}
// sequential halves: second load issued only after the first returned (so it should hit in L1)
__global__ void __launch_bounds__(256) k_seq(const float4* __restrict__ tab, uint32_t mask, int iters, float4* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = ((s >> 4) & mask) * 2u;
        const float4 a = tab[idx];
        const uint32_t dep = __float_as_uint(a.x) & 1u;               // 0 (table is zero) but the compiler cannot know
        const float4 b = tab[idx + 1 + dep * 2];
        acc.x += a.x; acc.y += b.y;
        s ^= dep;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}
int main() {
    const size_t max_rec = 1u << 25;
    float4* tab; hipMalloc(&tab, max_rec * 32 + 64); hipMemset(tab, 0, max_rec * 32 + 64);
    const int blocks = 256 * 8; float4* out; hipMalloc(&out, blocks * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"solo 2xdwordx4", "lane pair", "sequential halves"};
    for (int v = 0; v < 3; v++)
        for (int lg = 13; lg <= 25; lg += 4) {
            const uint32_t mask = (1u << lg) - 1; const int iters = 2000;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_solo, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
                if (v == 1) hipLaunchKernelGGL(k_pair, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
                if (v == 2) hipLaunchKernelGGL(k_seq, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double recs = (double)blocks * 256 * iters;
            printf("%-18s table %8.1f MB : %8.3f ms  %6.1f Grecords/s\n", names[v], (double)(mask + 1) * 32 / 1e6, ms, recs / ms / 1e6);
        }
    return 0;
}
