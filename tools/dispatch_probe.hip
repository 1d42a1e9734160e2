// dev probe: what a launch of one-per-CU workgroups costs before its first instruction and after its last -- an EMPTY kernel
// with the Gram's launch shape (N workgroups x T threads, L bytes of dynamic LDS) timed back to back, against the same
// number of waves in smaller workgroups.   usage: dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int* p) { extern __shared__ char s[]; if (p && threadIdx.x == 0 && blockIdx.x == 1 << 30) p[0] = s[0]; }
__global__ __launch_bounds__(1024) void spin_kernel(long long cycles, int* p) {
    extern __shared__ char s[];
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 0 && blockIdx.x == 1 << 30) p[0] = s[0];
}
int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int wgs, threads, lds; } cfgs[] = {{245, 1024, 120 * 1024}, {245, 1024, 0}, {245, 512, 120 * 1024}, {490, 512, 60 * 1024}, {245, 256, 120 * 1024},
                                                      {256, 1024, 64 * 1024}, {512, 256, 78 * 1024}, {245, 64, 0}};
    for (auto c : cfgs) {
        hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
        hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
        float best = 1e9f, bests = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            for (int i = 0; i < 23; ++i) { if (i == 3) hipEventRecord(e0); hipLaunchKernelGGL(empty_kernel, dim3(c.wgs), dim3(c.threads), c.lds, 0, (int*)nullptr); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            for (int i = 0; i < 23; ++i) { if (i == 3) hipEventRecord(e0); hipLaunchKernelGGL(spin_kernel, dim3(c.wgs), dim3(c.threads), c.lds, 0, 48000LL, (int*)nullptr); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); bests = ms < bests ? ms : bests;
        }
        printf("%4d workgroups x %4d threads, %3d KiB LDS: empty %.2f us per launch, 48k-cycle spin %.2f us per launch\n", c.wgs, c.threads, c.lds / 1024, best * 50, bests * 50);
    }
    return 0;
}
