// dev tool: sustained v_mfma_f32_32x32x2_f32 rate for different occupancies / dependency patterns
#include <hip/hip_runtime.h>
#include <cstdio>
using acc_t = float __attribute__((ext_vector_type(16)));
template <int NACC, int CHAIN>
__global__ void kc(float* out, int iters) {
    acc_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    float x = threadIdx.x * 0.001f, y = 1.0f - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / (NACC * CHAIN) + (16 % (NACC * CHAIN) ? 1 : 0); ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int c = 0; c < CHAIN; ++c) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int CHAIN> void runc(int wg_threads, int wgs_per_cu, const char* name) {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    int iters = 4096;
    dim3 grid(256 * wgs_per_cu), block(wg_threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kc<NACC, CHAIN><<<grid, block>>>(out, iters);
    hipEventRecord(e0);
    kc<NACC, CHAIN><<<grid, block>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per_iter = (16 / (NACC * CHAIN) + (16 % (NACC * CHAIN) ? 1 : 0)) * NACC * CHAIN;
    double flops = (double)grid.x * (wg_threads / 64) * iters * per_iter * 4096.0;
    printf("%-40s %.1f TF  (%.2f ms)\n", name, flops / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}
template <int NACC>
__global__ void k(float* out, int iters) {
    acc_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    float x = threadIdx.x * 0.001f, y = 1.0f - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int wg_threads, int wgs_per_cu, const char* name) {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    int iters = 4096;
    dim3 grid(256 * wgs_per_cu), block(wg_threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, block>>>(out, iters);
    hipEventRecord(e0);
    k<NACC><<<grid, block>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * (wg_threads / 64) * iters * 16 * 4096.0;
    printf("%-40s %.1f TF  (%.2f ms)\n", name, flops / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}
int main() {
    run<4>(256, 1, "1 wave/SIMD, 4 independent acc");
    run<4>(256, 2, "2 waves/SIMD, 4 independent acc");
    run<4>(256, 4, "4 waves/SIMD, 4 independent acc");
    run<1>(256, 1, "1 wave/SIMD, 1 acc (dependent chain)");
    run<1>(256, 2, "2 waves/SIMD, 1 acc (dependent chain)");
    run<1>(256, 4, "4 waves/SIMD, 1 acc (dependent chain)");
    run<2>(256, 2, "2 waves/SIMD, 2 acc");
    runc<4, 2>(256, 1, "1 w/SIMD, 4 acc, chain 2");
    runc<4, 4>(256, 1, "1 w/SIMD, 4 acc, chain 4");
    runc<8, 4>(256, 1, "1 w/SIMD, 8 acc, chain 4");
    runc<8, 8>(256, 1, "1 w/SIMD, 8 acc, chain 8");
    runc<4, 4>(256, 2, "2 w/SIMD, 4 acc, chain 4");
    runc<8, 4>(256, 2, "2 w/SIMD, 8 acc, chain 4");
    runc<8, 8>(256, 2, "2 w/SIMD, 8 acc, chain 8");
    runc<4, 1>(256, 2, "2 w/SIMD, 4 acc, chain 1");
    runc<4, 1>(256, 4, "4 w/SIMD, 4 acc, chain 1");
    runc<4, 4>(256, 4, "4 w/SIMD, 4 acc, chain 4");
    return 0;
}
