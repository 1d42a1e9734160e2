// dev tool: latency of a cross-stream hand-over by (a) hipEventRecord + hipStreamWaitEvent and
// (b) hipStreamWriteValue64 + hipStreamWaitValue64 on signal memory: time between the end of the producer
// kernel on stream A and the start of the consumer kernel on stream B.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void stamp(long long* out, long long spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d\n", can);
    long long* d; CK(hipMalloc(&d, 64)); long long h[4];
    hipStream_t a, b; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned long long* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1 && !can) break;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, d, 5000LL);          // producer: 50 us
            if (mode == 0) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
            else { CK(hipStreamWriteValue64(a, sig, (uint64_t)(rep + 1), 0)); CK(hipStreamWaitValue64(b, sig, (uint64_t)(rep + 1), hipStreamWaitValueGte, ~0ULL)); }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, d + 2, 100LL);        // consumer
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
            printf("%s: consumer started %.1f us after the producer ended\n", mode == 0 ? "event      " : "write/wait64", (h[2] - h[1]) / 100.0);
        }
    }
    return 0;
}
