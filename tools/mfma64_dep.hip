// dev tool: does a chain of v_mfma_f64_16x16x4_f64 on the SAME accumulator issue as fast as one that rotates over
// several accumulators?  (the fp64 Gram kernel issues 4 dependent MFMAs per block and tile)
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = double __attribute__((ext_vector_type(4)));
template <int NACC, int CHAIN>      // CHAIN consecutive MFMAs on one accumulator, then the next of NACC
__global__ __launch_bounds__(1024) void k(const double* in, double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = in[threadIdx.x], b = in[threadIdx.x + 1024];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int NACC, int CHAIN> void run(const double* in, double* out, int threads) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, CHAIN><<<256, threads>>>(in, out, iters);
    hipEventRecord(e0);
    k<NACC, CHAIN><<<256, threads>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * NACC * CHAIN;          // MFMAs per wave
    const double waves_per_simd = threads / 64 / 4.0;
    printf("accumulators %d, chain %d, %4.1f waves/SIMD: %.1f TF, %.1f cycles per MFMA and SIMD at 2.4 GHz\n", NACC, CHAIN, waves_per_simd,
           n * (threads / 64) * 256 * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
}
int main() {
    double *in, *out; hipMalloc(&in, 4096 * 8); hipMalloc(&out, 256 * 1024 * 8); hipMemset(in, 0, 4096 * 8);
    for (int threads : {256, 512, 1024}) {
        run<8, 1>(in, out, threads); run<8, 2>(in, out, threads); run<8, 4>(in, out, threads); run<2, 4>(in, out, threads); run<1, 8>(in, out, threads);
    }
    return 0;
}
