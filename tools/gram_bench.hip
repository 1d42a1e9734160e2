// dev tool: time gram_kernel<float> alone (C2 shape) with ablation switches.
#include "../ces_amd/csrc/kernels_gram.hip"
#include <cstdio>
#include <vector>
#include <cmath>
int main(int argc, char** argv) {
    using namespace cesx;
    const int p = 256, n = 256, P = 512; const long long J = 65536;
    GramPlan pl = make_gram_plan(P, 32, GramCfg<float>::NBW, MAX_STAGE_ROWS, argc > 1 ? atoi(argv[1]) : 0, 8, 1,
                                 argc > 2 ? atoi(argv[2]) : 256, (J + 31) / 32);
    const int nslices = pl.total_wgs / pl.ntypes;
    float *U, *G, *shift, *slabs; double* rsp; int *th, *rows, *wblk;
    hipMalloc(&U, p * J * 4); hipMalloc(&G, n * J * 4); hipMalloc(&shift, P * 4);
    hipMalloc(&slabs, (size_t)pl.total_slabs * 1024 * 4); hipMalloc(&rsp, (size_t)pl.total_rs * P * 8);
    std::vector<float> h((size_t)p * J);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(U, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(G, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(shift, 0, P * 4);
    hipMalloc(&th, pl.type_hdr.size() * 4); hipMalloc(&rows, pl.rows.size() * 4); hipMalloc(&wblk, pl.wblk.size() * 4);
    hipMemcpy(th, pl.type_hdr.data(), pl.type_hdr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(rows, pl.rows.data(), pl.rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(wblk, pl.wblk.data(), pl.wblk.size() * 4, hipMemcpyHostToDevice);
    const int lds = 2 * pl.max_rb * pl.tile * ROW_STRIDE + pl.max_rb * pl.tile * 16;
    auto kern = gram_kernel<float, true>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(pl.total_wgs), block(GRAM_THREADS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs, rsp);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs, rsp);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef GRAM_CLOCKS
    {
        std::vector<long long> c(grid.x * 4);
        hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_gram_clk), c.size() * 8);
        double a0 = 0, a1 = 0, a2 = 0, w = 0, wmax = 0;
        for (unsigned i = 0; i < grid.x; ++i) { a0 += c[4 * i]; a1 += c[4 * i + 1]; a2 += c[4 * i + 2]; w += c[4 * i + 3]; wmax = fmax(wmax, (double)c[4 * i + 3]); }
        printf("wave 0 per WG: prologue %.0f, loop %.0f, epilogue(issue) %.0f cycles; WG wall %.1f us (max %.1f) -> %.0f MHz\n", a0 / grid.x, a1 / grid.x, a2 / grid.x,
               w / grid.x / 100, wmax / 100, (a0 + a1 + a2) / w * 100);
    }
#endif
    printf("GRAM_ABL=%d types %d slices %d: %.1f us/launch (%.1f TF executed, %.1f TF algorithmic)\n", GRAM_ABL, pl.ntypes, nslices,
           ms * 100.0, 2.0 * pl.nblocks * 1024 * J / (ms * 1e-4) / 1e12, (double)P * P * J / (ms * 1e-4) / 1e12);
    return 0;
}
