// dev tool: time potrf_reg_kernel alone.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/potrf_bench.hip -o tools/potrf_bench
#include "../ces_amd/csrc/kernels_dense.hip"
#include <cstdio>
namespace cesx { int launch_noise(Engine&, uint64_t, void*, hipStream_t) { return 0; } }
#ifndef PB_SLOTS      // -DPB_SLOTS=2 / 5 / 10 / 17: the instantiation for n <= 64 / 128 / 192 / 256
#define PB_SLOTS 17
#endif
#ifndef PB_CHAINV     // 1: chained image + fp64 factor, 2: the image only
#define PB_CHAINV 2
#endif
#include <vector>
#include <random>
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 256;
    int np = (n + 31) / 32 * 32;
    std::vector<double> B((size_t)n * n), A((size_t)n * n);
    std::mt19937 g(1); std::normal_distribution<double> nd;
    for (auto& v : B) v = nd(g);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += B[i*n+k]*B[j*n+k]; A[i*n+j] = s / n + (i == j); }
    double *dA, *dL, *dLp; int* st;
    hipMalloc(&dA, n*n*8); hipMalloc(&dL, n*n*8); hipMalloc(&dLp, np*np*8); hipMalloc(&st, 4);
    hipMemcpy(dA, A.data(), n*n*8, hipMemcpyHostToDevice); hipMemset(st, 0, 4);
    const int npmax = PB_SLOTS <= 2 ? 64 : PB_SLOTS <= 5 ? 128 : PB_SLOTS <= 10 ? 192 : 256;
    size_t lds = (size_t)3 * 8 * (2 * npmax + 4) * 8;
    auto kern = cesx::potrf_reg_kernel<PB_SLOTS>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long long* dbg; hipMalloc(&dbg, 8 * 80); hipMemset(dbg, 0, 8 * 80);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(1), dim3(cesx::PRT), lds, 0, n, np, dA, dLp, st, (long long*)nullptr, 0, 0, (const double*)nullptr, (const double*)nullptr, 0, (unsigned long long*)nullptr, 0ull, (float*)nullptr, 0, 0, (const int*)nullptr, (const double*)nullptr);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(1), dim3(cesx::PRT), lds, 0, n, np, dA, dLp, st, (long long*)nullptr, 0, 0, (const double*)nullptr, (const double*)nullptr, 0, (unsigned long long*)nullptr, 0ull, (float*)nullptr, 0, 0, (const int*)nullptr, (const double*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> L((size_t)n * n); hipMemcpy2D(L.data(), n*8, dLp, np*8, n*8, n, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += L[i*n+k]*L[j*n+k]; err = fmax(err, fabs(s - A[i*n+j])); }
    int hst; hipMemcpy(&hst, st, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kern, dim3(1), dim3(cesx::PRT), lds, 0, n, np, dA, dLp, st, dbg, 0, 0, (const double*)nullptr, (const double*)nullptr, 0, (unsigned long long*)nullptr, 0ull, (float*)nullptr, 0, 0, (const int*)nullptr, (const double*)nullptr);
    long long hd[5]; hipMemcpy(hd, dbg, 40, hipMemcpyDeviceToHost);
    { long long pp[72]; hipMemcpy(pp, dbg + 8, 72 * 8, hipMemcpyDeviceToHost); printf("per panel (factor, trailing) cycles:"); for (int k = 0; k < np / 8; k += 1) printf(" %lld/%lld", pp[2 * k], pp[2 * k + 1]); printf("\n"); }
    printf("cycles (wave 0): init %lld | factor %lld barrier %lld | trailing %lld barrier %lld\n", hd[0], hd[1], hd[2], hd[3], hd[4]);
    printf("n=%d potrf %.1f us/call, max |LL^T - A| = %.3e, status %d\n", n, ms * 100.0, err, hst);
#if PB_SLOTS == 17
    if (n > 224) {      // the instantiation that also writes the chained image of kernels_update4.hip: time, and the image against L
        auto kc = cesx::potrf_reg_kernel<17, PB_CHAINV>;
        const size_t ldsc = lds + 2 * npmax * 8, img_floats = (size_t)(18 + 16) * 4096;
        hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
        float* wq; double* sinv; hipMalloc(&wq, img_floats * 4); hipMemset(wq, 0, img_floats * 4); hipMalloc(&sinv, n * 8);
        std::vector<double> hs(n); for (int i = 0; i < n; ++i) hs[i] = 0.01 * (1.0 + 0.3 * (i % 7));
        hipMemcpy(sinv, hs.data(), n * 8, hipMemcpyHostToDevice);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kc, dim3(1), dim3(cesx::PRT), ldsc, 0, n, np, dA, dLp, st, (long long*)nullptr, 0, 0, (const double*)nullptr, (const double*)nullptr, 0, (unsigned long long*)nullptr, 0ull, wq, 48, 256, (const int*)nullptr, (const double*)sinv);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kc, dim3(1), dim3(cesx::PRT), ldsc, 0, n, np, dA, dLp, st, (long long*)nullptr, 0, 0, (const double*)nullptr, (const double*)nullptr, 0, (unsigned long long*)nullptr, 0ull, wq, 48, 256, (const int*)nullptr, (const double*)sinv);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> img(img_floats); hipMemcpy(img.data(), wq, img_floats * 4, hipMemcpyDeviceToHost);
        hipMemcpy2D(L.data(), n*8, dLp, np*8, n*8, n, hipMemcpyDeviceToHost);
        double e1m = 0, e2m = 0; size_t nz = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
            e1m = fmax(e1m, fabs(img[cesx::wc_index_L(i, j)] - (float)L[(size_t)i*n+j]));
            e2m = fmax(e2m, fabs(img[cesx::wc_index_Lt(j, i)] - (float)(-L[(size_t)i*n+j] * hs[i])));
        }
        for (float v : img) nz += v != 0.f;
        printf("chained image: potrf %.1f us/call; max |img - L| = %.3g, max |img - (-L^T S^-1)| = %.3g, non-zeros %zu (expected <= %d)\n",
               ms * 100.0, e1m, e2m, nz, n * (n + 1));
    }
#endif
    return 0;
}
