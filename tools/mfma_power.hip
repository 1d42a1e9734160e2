// dev tool: sustained v_mfma_f32_32x32x2_f32 rate with constant vs random operand data
// (does operand toggling lower the sustained clock?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using acc_t = float __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void kr(const float* __restrict__ in, float* out, int iters) {
    acc_t acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = in[(i * 2) * 256 + threadIdx.x]; y[i] = in[(i * 2 + 1) * 256 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 8; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(a + u) & 7], y[(a * 3 + u) & 7], acc[a], 0, 0, 0);
        // keep accumulators bounded so the data stays "ordinary"
        if ((it & 63) == 63)
            for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] *= 0.001f;
    }
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *in, *out; hipMalloc(&in, 16 * 256 * 4); hipMalloc(&out, 512 * 256 * 4);
    std::vector<float> h(16 * 256);
    for (int mode = 0; mode < 3; ++mode) {
        for (size_t i = 0; i < h.size(); ++i)
            h[i] = mode == 0 ? 0.0f : mode == 1 ? 1.0f : (float)rand() / RAND_MAX * 2.f - 1.f;
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int iters = 16384;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kr<<<512, 256>>>(in, out, iters);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            kr<<<512, 256>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = 512.0 * 4 * iters * 32 * 4096.0;
            printf("operands %s: %.1f TF (%.2f ms)\n", mode == 0 ? "zero" : mode == 1 ? "one" : "random", flops / (ms * 1e-3) / 1e12, ms);
        }
    }
    return 0;
}
