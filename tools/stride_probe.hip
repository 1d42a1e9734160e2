// dev tool: does a power-of-two row stride cost HBM bandwidth?  Every workgroup writes (or reads) K3's output tile --
// 256 rows x 128 floats, 512-byte row segments -- into a (256, ld) row-major array, for ld = 65536 and padded strides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(float* out, long long ld, int mode, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const long long j = (long long)blockIdx.x * 128 + 4 * li;
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < 2; ++r)
        for (int e = 0; e < 16; ++e) {
            const int rb = r == 0 ? wave : 7 - wave;
            const int i = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            f4* p = reinterpret_cast<f4*>(out + (size_t)i * ld + j);
            if (mode == 0) __builtin_nontemporal_store(f4{(float)i, (float)j, 1.f, 2.f}, p);
            else if (mode == 1) *p = f4{(float)i, (float)j, 1.f, 2.f};
            else acc += *p;
        }
    if (mode == 2 && acc[0] == 12345.f) sink[0] = acc[1];
}
int main() {
    float *buf, *sink; hipMalloc(&sink, 16);
    const long long lds[] = {65536, 65536 + 128, 65536 + 512, 65536 + 4096 + 128};
    hipMalloc(&buf, (size_t)256 * (65536 + 8192) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (long long ld : lds) {
            for (int i = 0; i < 3; ++i) wr<<<512, 256>>>(buf, ld, mode, sink);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) wr<<<512, 256>>>(buf, ld, mode, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s ld=%lld: %.2f us per 67 MB = %.2f TB/s\n", mode == 0 ? "nt store" : mode == 1 ? "store   " : "load    ", ld, ms * 50, 67.1e6 / (ms * 50e-6) / 1e12);
        }
    return 0;
}
