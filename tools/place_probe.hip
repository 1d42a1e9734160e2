// dev tool: does a ONE-workgroup kernel on a second (high priority) stream get a CU while a big-LDS kernel that
// occupies one CU per workgroup is running with N < 256 workgroups?  Prints the start delay of the small kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void hog(long long* out, long long ticks) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    asm volatile("v_mov_b32 v127, 0" ::: "v127");          // 128 VGPRs x 16 waves: the whole register file of the CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(20);
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}
__global__ __launch_bounds__(512) void small(long long* out, int nw) {
    if (nw > 0) asm volatile("v_mov_b32 v223, 0" ::: "v223");   // 224 VGPRs x 8 waves, like potrf_reg_kernel
    if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}
int main(int argc, char** argv) {
    const int small_wgs = argc > 1 ? atoi(argv[1]) : 1;
    long long *a, *b; hipMalloc(&a, 8 * 512); hipMalloc(&b, 8 * 512);
    hipStream_t s1, s2; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int n : {224, 240, 248, 252}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(a, 0, 8 * 512); hipMemset(b, 0, 8 * 512); hipDeviceSynchronize();
            hipLaunchKernelGGL(small, dim3(1), dim3(64), 0, s1, b + 300, 0);       // marker: "previous kernel" on the main stream
            hipEventRecord(ev, s1);
            hipLaunchKernelGGL(hog, dim3(n), dim3(1024), 131072, s1, a, 20000LL);  // 200 us at 100 MHz
            hipStreamWaitEvent(s2, ev, 0);
            hipLaunchKernelGGL(small, dim3(small_wgs), dim3(512), 0, s2, b, small_wgs);
            hipDeviceSynchronize();
            long long ha[256], hb[512]; hipMemcpy(ha, a, 8 * 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 8 * 512, hipMemcpyDeviceToHost);
            long long amin = ha[0]; for (int i = 1; i < n; ++i) amin = ha[i] < amin ? ha[i] : amin;
            long long bmax = hb[0]; for (int i = 1; i < small_wgs; ++i) bmax = hb[i] > bmax ? hb[i] : bmax;
            printf("hog %3d WGs, small %3d WGs: small kernel's last WG started %.1f us after the hog's first\n", n, small_wgs, (bmax - amin) / 100.0);
        }
    }
    return 0;
}
