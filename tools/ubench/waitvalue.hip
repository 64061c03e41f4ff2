// Does hipStreamWaitValue32 gate a second stream on a flag a running kernel stores?  (experiment for DESIGN.md item 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void a_kernel(uint32_t* sig, uint32_t seq, unsigned long long* t_flag, unsigned long long spin) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }                       // ~spin * 10 ns
    if (threadIdx.x == 0 && blockIdx.x == 0) { __hip_atomic_store(sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); *t_flag = wall_clock64(); }
    while (wall_clock64() - t0 < 2 * spin) { }
}
__global__ void b_kernel(unsigned long long* t_b) { if (threadIdx.x == 0 && blockIdx.x == 0) *t_b = wall_clock64(); }
int main() {
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    uint32_t* sig = nullptr; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    unsigned long long* t = nullptr; CK(hipMalloc((void**)&t, 4 * sizeof(unsigned long long))); CK(hipMemset(t, 0, 32));
    CK(hipMemset(sig, 0, 8));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (uint32_t seq = 1; seq <= 3; seq++) {
        CK(hipStreamWaitValue32(s2, sig, seq, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(b_kernel, dim3(1), dim3(64), 0, s2, t + 1);
        hipLaunchKernelGGL(a_kernel, dim3(64), dim3(256), 0, s1, sig, seq, t, 50000ull);     // 0.5 ms to the flag, 1 ms in all
        CK(hipStreamWriteValue32(s1, sig, seq, 0));                                            // safety net: released at the latest when A is done
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        unsigned long long h[2]; CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
        printf("seq %u: B ran %.1f us after the flag store (A runs 500 us beyond it)\n", seq, ((long long)h[1] - (long long)h[0]) * 0.01);
    }
    printf("ok\n");
    return 0;
}
