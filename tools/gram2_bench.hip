// dev tool: gram2_kernel (LDS-DMA) against gram_kernel on the C2 shape: bitwise comparison of the slabs
// and row sums, and time -- with the round-3 form of the LDS-DMA kernel (tools/gram2_r3.hip) timed beside it, the
// three interleaved round by round.   usage: gram2_bench [subset 0|1|2] [wg_budget] [f64]
#include "../ces_amd/csrc/kernels_gram.hip"
#include "../ces_amd/csrc/kernels_gram2.hip"
#include "gram2_r3.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
using namespace cesx;
template <typename T>
int run(int subset, int budget, int p, int n, long long J) {
    const int P = p + n, tile = Mfma<T>::TILE, KT = 128 / (int)sizeof(T);
    GramPlan pl = make_gram_plan(P, tile, GramCfg<T>::NBW, MAX_STAGE_ROWS, subset, (p + tile - 1) / tile, 1, budget, J / KT);
    T *U, *G, *shift, *slabs, *slabs2; double *rsp, *rsp2; int *th, *rows, *wblk;
    hipMalloc(&U, (size_t)p * J * sizeof(T)); hipMalloc(&G, (size_t)n * J * sizeof(T)); hipMalloc(&shift, P * sizeof(T));
    const size_t slab_elems = (size_t)pl.total_slabs * tile * tile;
    hipMalloc(&slabs, slab_elems * sizeof(T)); hipMalloc(&slabs2, slab_elems * sizeof(T));
    hipMalloc(&rsp, (size_t)pl.total_rs * P * 8); hipMalloc(&rsp2, (size_t)pl.total_rs * P * 8);
    hipMemset(rsp, 0, (size_t)pl.total_rs * P * 8); hipMemset(rsp2, 0, (size_t)pl.total_rs * P * 8);
    std::vector<T> h((size_t)p * J), hs(P);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (T)((i * 2654435761u) % 2001) / (T)1000 - (T)1 + (T)0.5;
    hipMemcpy(U, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (T)((i * 40503u + 17) % 1999) / (T)1000 - (T)1;
    hipMemcpy(G, h.data(), (size_t)n * J * sizeof(T), hipMemcpyHostToDevice);
    for (int i = 0; i < P; ++i) hs[i] = (T)(0.5 * (i < p) + 0.001 * (i % 13));
    hipMemcpy(shift, hs.data(), P * sizeof(T), hipMemcpyHostToDevice);
    hipMalloc(&th, pl.type_hdr.size() * 4); hipMalloc(&rows, pl.rows.size() * 4); hipMalloc(&wblk, pl.wblk.size() * 4);
    hipMemcpy(th, pl.type_hdr.data(), pl.type_hdr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(rows, pl.rows.data(), pl.rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(wblk, pl.wblk.data(), pl.wblk.size() * 4, hipMemcpyHostToDevice);
    const int nrows = pl.max_rb * tile;
    const int lds1 = 2 * nrows * ROW_STRIDE + nrows * 16;
    const bool imm = nrows * G2_ROWB <= G2_SLOT_IMM;
    const int lds2 = imm ? 2 * G2_SLOT_IMM : 2 * G2_SLOT;
    auto k1 = gram_kernel<T, true>;
    auto k2 = imm ? gram2_kernel<T, true, true> : gram2_kernel<T, true, false>;
    hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    dim3 grid(pl.total_wgs), block(1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0, ms3 = 0;
    const int lds3 = 2 * G2R3_SLOT + G2R3_MAX_ROWS * 8 + G2R3_MAX_ROWS * (int)sizeof(T);
    auto k3 = gram2r3_kernel<T>;
    hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
    for (int round = 0; round < 4; ++round)
    for (int which = 0; which < 3; ++which) {
        for (int i = 0; i < 13; ++i) {
            if (i == 3) hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k1, grid, block, lds1, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs, rsp);
            else if (which == 1) hipLaunchKernelGGL(k2, grid, block, lds2, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs2, rsp2);
            else hipLaunchKernelGGL(k3, grid, block, lds3, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs2, rsp2);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        float& dst = which == 0 ? ms1 : which == 1 ? ms2 : ms3;
        dst = round == 0 ? ms : fminf(dst, ms);
    }
    // (the compared slabs are the CURRENT kernel's: its launch is the last writer of slabs2 / rsp2)
    hipMemset(slabs2, 0, slab_elems * sizeof(T)); hipMemset(rsp2, 0, (size_t)pl.total_rs * P * 8);
    hipLaunchKernelGGL(k2, grid, block, lds2, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs2, rsp2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
    std::vector<T> a(slab_elems), b(slab_elems);
    hipMemcpy(a.data(), slabs, slab_elems * sizeof(T), hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), slabs2, slab_elems * sizeof(T), hipMemcpyDeviceToHost);
    size_t diff = 0; double maxd = 0, maxv = 0;
    for (size_t i = 0; i < slab_elems; ++i) { if (a[i] != b[i]) ++diff; maxd = fmax(maxd, fabs((double)a[i] - (double)b[i])); maxv = fmax(maxv, fabs((double)a[i])); }
    std::vector<double> r1((size_t)pl.total_rs * P), r2((size_t)pl.total_rs * P);
    hipMemcpy(r1.data(), rsp, r1.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), rsp2, r2.size() * 8, hipMemcpyDeviceToHost);
    double rd = 0, rv = 0;
    for (size_t i = 0; i < r1.size(); ++i) { rd = fmax(rd, fabs(r1[i] - r2[i])); rv = fmax(rv, fabs(r1[i])); }
    printf("%s subset %d types %d wgs %d max_rb %d: slabs differ in %zu of %zu elements (max |d| %.3g of %.3g); row sums max |d| %.3g of %.3g\n",
           sizeof(T) == 4 ? "f32" : "f64", subset, pl.ntypes, pl.total_wgs, pl.max_rb, diff, slab_elems, maxd, maxv, rd, rv);
    const double fl = (double)P * P * J, ex = 2.0 * pl.nblocks * tile * tile * J;
    printf("  staged kernel %.1f us (%.1f TF executed)   LDS-DMA kernel %.1f us (%.1f TF executed)   its round-3 form %.1f us (%.1f TF)   [best of 4 rounds of 10 launches; algorithmic share %.1f GF]\n",
           ms1 * 100, ex / (ms1 * 1e-4) / 1e12, ms2 * 100, ex / (ms2 * 1e-4) / 1e12, ms3 * 100, ex / (ms3 * 1e-4) / 1e12, fl / 1e9);
    for (int t = 0; t < pl.ntypes; ++t)
        printf("    type %2d: %3d blocks, %2d row blocks, %3d slices\n", t, pl.type_hdr[t * 8 + 3], pl.type_hdr[t * 8 + 0], pl.type_hdr[t * 8 + 5]);
#ifdef G2_CLOCKS
    {
        std::vector<long long> c(grid.x * 4);
        hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_gram2_clk), c.size() * 8);
        double a0 = 0, a1 = 0, a2 = 0, w = 0, wmax = 0;
        for (unsigned i = 0; i < grid.x; ++i) { a0 += c[4 * i]; a1 += c[4 * i + 1]; a2 += c[4 * i + 2]; w += c[4 * i + 3]; wmax = fmax(wmax, (double)c[4 * i + 3]); }
        std::vector<long long> bw(grid.x * 16);
        hipMemcpyFromSymbol(bw.data(), HIP_SYMBOL(g_gram2_bar), bw.size() * 8);
        printf("  barrier cycles per wave (avg over WGs):");
        for (int w2 = 0; w2 < 16; ++w2) { double sb = 0; for (unsigned i = 0; i < grid.x; ++i) sb += bw[i * 16 + w2]; printf(" %.0f", sb / grid.x); }
        printf("\n");
        for (int t = 0; t < pl.ntypes; ++t) {
            const int w0 = pl.type_hdr[t * 8 + 4], ns = pl.type_hdr[t * 8 + 5];
            double ww = 0, lp = 0; for (int i = w0; i < w0 + ns; ++i) { ww += c[4 * i + 3]; lp += c[4 * i + 1]; }
            printf("    type %2d: %3d blocks, %2d row blocks, %3d slices: WG wall %.1f us, loop %.0f cycles\n", t, pl.type_hdr[t * 8 + 3], pl.type_hdr[t * 8 + 0], ns, ww / ns / 100, lp / ns);
        }
        {
            std::vector<long long> pr(grid.x * 4);
            hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_gram2_pro), pr.size() * 8);
            double q[4] = {0, 0, 0, 0};
            for (unsigned i = 0; i < grid.x; ++i) for (int k = 0; k < 4; ++k) q[k] += pr[4 * i + k];
            printf("  prologue phases (avg cycles): tables+acc %.0f | row table+sync %.0f | first DMA wait %.0f | DMA+shift+barrier %.0f\n",
                   q[0] / grid.x, q[1] / grid.x, q[2] / grid.x, q[3] / grid.x);
        }
        printf("  wave 0 per WG: prologue %.0f, loop %.0f, epilogue %.0f cycles; WG wall %.1f us (max %.1f) -> %.0f MHz\n", a0 / grid.x, a1 / grid.x, a2 / grid.x,
               w / grid.x / 100, wmax / 100, (a0 + a1 + a2) / w * 100);
    }
#endif
    return 0;
}
int main(int argc, char** argv) {
    const int subset = argc > 1 ? atoi(argv[1]) : 0, budget = argc > 2 ? atoi(argv[2]) : 256;
    const bool f64 = argc > 3 && !strcmp(argv[3], "f64");
    if (f64) return run<double>(subset, budget, 512, 512, 32768);
    return run<float>(subset, budget, 256, 256, 65536);
}
