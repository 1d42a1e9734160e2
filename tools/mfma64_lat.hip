#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = double __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0;
    double x = threadIdx.x * 0.001, y = 1.0 - x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    long long t1 = clock64();
    double s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8192); hipMalloc(&cyc, 8);
    long long h;
    k<1><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("f64 16x16x4 dependent chain: %.1f cycles/MFMA\n", h / 1000.0);
    k<4><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("f64 16x16x4 4 independent:    %.1f cycles/MFMA\n", h / 4000.0);
    k<8><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("f64 16x16x4 8 independent:    %.1f cycles/MFMA\n", h / 8000.0);
    return 0;
}
