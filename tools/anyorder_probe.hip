// dev tool: is hipExtAnyOrderLaunch honoured on this device?  Kernel A (one workgroup, ~300 us of s_sleep) and, behind
// it on the same stream, kernel B (short) launched with and without the flag; B's own end time stamp tells.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void slow(long long* t, int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); if (threadIdx.x == 0) t[0] = wall_clock64(); }
__global__ void quick(long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64(); }
int main() {
    long long* t; hipMalloc(&t, 16); long long h[2];
    hipStream_t s; hipStreamCreate(&s);
    for (int flag = 0; flag < 2; ++flag) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(t, 0, 16);
            hipLaunchKernelGGL(slow, dim3(1), dim3(64), 0, s, t, 100);
            hipExtLaunchKernelGGL(quick, dim3(64), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, t);
            hipStreamSynchronize(s);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("flag %d: the short kernel ended %.1f us %s the slow one\n", flag, (h[1] > h[0] ? h[1] - h[0] : h[0] - h[1]) * 0.01, h[1] > h[0] ? "AFTER" : "BEFORE");
        }
    }
    return 0;
}
