// dev tool: gram2_kernel (LDS-DMA) against gram_kernel on the C2 shape: bitwise comparison of the slabs
// and row sums, and time -- with the round-3 form of the LDS-DMA kernel (tools/gram2_r3.hip) timed beside it, the
// three interleaved round by round.   usage: gram2_bench [subset 0|1|2] [wg_budget] [f64]
#include "../ces_amd/csrc/kernels_gram.hip"
#include "../ces_amd/csrc/kernels_gram2.hip"
#include "gram2_r3.hip"
#include <algorithm>
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
    const int lds2 = (imm ? 2 * G2_SLOT_IMM : 2 * G2_SLOT) + G2_SHTAB;
    auto k1 = gram_kernel<T, true>;
    auto k2 = imm ? gram2_kernel<T, true, true> : gram2_kernel<T, true, false>;
    hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    dim3 grid(pl.total_wgs), block(1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0, ms3 = 0;
    const Gram2Tab tab2{th, pl.ntypes, rows, wblk, slabs2, rsp2, pl.total_wgs};
    const int lds3 = 2 * G2R3_SLOT + G2R3_MAX_ROWS * 8 + G2R3_MAX_ROWS * (int)sizeof(T);
    auto k3 = gram2r3_kernel<T>;
    hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
    for (int round = 0; round < 4; ++round)
    for (int which = 0; which < 3; ++which) {
        for (int i = 0; i < 13; ++i) {
            if (i == 3) hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k1, grid, block, lds1, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs, rsp);
            else if (which == 1) hipLaunchKernelGGL(k2, grid, block, lds2, 0, (const T*)U, (const T*)G, (const T*)shift, p, n, J, tab2);
            else hipLaunchKernelGGL(k3, grid, block, lds3, 0, U, G, shift, p, n, J, th, pl.ntypes, rows, wblk, slabs2, rsp2);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        float& dst = which == 0 ? ms1 : which == 1 ? ms2 : ms3;
        dst = round == 0 ? ms : fminf(dst, ms);
    }
    // (the compared slabs are the CURRENT kernel's: its launch is the last writer of slabs2 / rsp2)
    hipMemset(slabs2, 0, slab_elems * sizeof(T)); hipMemset(rsp2, 0, (size_t)pl.total_rs * P * 8);
    hipLaunchKernelGGL(k2, grid, block, lds2, 0, (const T*)U, (const T*)G, (const T*)shift, p, n, J, tab2);
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
// the fused launch (both parts in one kernel, kernels_gram2.hip): time, per-workgroup phase stamps (-DG2_FUSED_STAMPS), and
// the head of the moment buffer against gram_reduce_kernel's sums of the same slabs.  `waiter`: the side stream's
// wait kernel launched (on a high-priority stream) BEFORE each fused launch, as the engine's pipelined loop has it.
template <typename T>
int run_fused(int p, int n, long long J, int budget_b, int waiter) {
    const int P = p + n, tile = Mfma<T>::TILE, KT = 128 / (int)sizeof(T), pbU = (p + tile - 1) / tile;
    GramPlan pl[2] = {make_gram_plan(P, tile, GramCfg<T>::NBW, MAX_STAGE_ROWS, 1, pbU, 1, 256, J / KT),
                      make_gram_plan(P, tile, GramCfg<T>::NBW, MAX_STAGE_ROWS, 2, pbU, 1, budget_b, J / KT)};
    T *U, *G, *shift; hipMalloc(&U, (size_t)p * J * sizeof(T)); hipMalloc(&G, (size_t)n * J * sizeof(T)); hipMalloc(&shift, P * sizeof(T));
    std::vector<T> h((size_t)std::max(p, n) * J), hs(P);
    for (size_t i = 0; i < (size_t)p * J; ++i) h[i] = (T)((i * 2654435761u) % 2001) / (T)1000 - (T)1 + (T)0.5;
    hipMemcpy(U, h.data(), (size_t)p * J * sizeof(T), hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)n * J; ++i) h[i] = (T)((i * 40503u + 17) % 1999) / (T)1000 - (T)1;
    hipMemcpy(G, h.data(), (size_t)n * J * sizeof(T), hipMemcpyHostToDevice);
    for (int i = 0; i < P; ++i) hs[i] = (T)(0.5 * (i < p) + 0.001 * (i % 13));
    hipMemcpy(shift, hs.data(), P * sizeof(T), hipMemcpyHostToDevice);
    Gram2Tab tab[2]; int* d_blk_rc[2]; int* d_row_own[2];
    for (int k = 0; k < 2; ++k) {
        int *th, *rows, *wblk; T* slabs; double* rsp;
        hipMalloc(&th, pl[k].type_hdr.size() * 4); hipMalloc(&rows, pl[k].rows.size() * 4); hipMalloc(&wblk, pl[k].wblk.size() * 4);
        hipMemcpy(th, pl[k].type_hdr.data(), pl[k].type_hdr.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(rows, pl[k].rows.data(), pl[k].rows.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(wblk, pl[k].wblk.data(), pl[k].wblk.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&slabs, (size_t)pl[k].total_slabs * tile * tile * sizeof(T)); hipMalloc(&rsp, (size_t)pl[k].total_rs * P * 8);
        hipMemset(rsp, 0, (size_t)pl[k].total_rs * P * 8);
        hipMalloc(&d_blk_rc[k], pl[k].blk_rc.size() * 4); hipMemcpy(d_blk_rc[k], pl[k].blk_rc.data(), pl[k].blk_rc.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&d_row_own[k], pl[k].row_own.size() * 4); hipMemcpy(d_row_own[k], pl[k].row_own.data(), pl[k].row_own.size() * 4, hipMemcpyHostToDevice);
        tab[k] = Gram2Tab{th, pl[k].ntypes, rows, wblk, slabs, rsp, pl[k].total_wgs};
    }
    MomLayout ml{p, n};
    double *mom, *mom_ref; hipMalloc(&mom, ml.len() * 8); hipMalloc(&mom_ref, ml.len() * 8);
    hipMemset(mom, 0, ml.len() * 8); hipMemset(mom_ref, 0, ml.len() * 8);
    unsigned* sync; hipMalloc(&sync, 512); hipMemset(sync, 0, 512);
    Scalars* sc; hipMalloc(&sc, sizeof(Scalars)); hipMemset(sc, 0, sizeof(Scalars));
    const int lds = 2 * G2_SLOT_IMM + G2_SHTAB + G2_COMB + G2_CTL;
    auto kern = gram2_fused_kernel<T, true>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncAttributes fa{}; hipFuncGetAttributes(&fa, (const void*)kern);
    const int gram_wgs = std::max(pl[0].total_wgs, pl[1].total_wgs);
    int lo_ = 0, hi_ = 0; hipDeviceGetStreamPriorityRange(&lo_, &hi_);
    hipStream_t side; hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi_);
    hipStream_t mainS; hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking);
    const int row_lo = std::min(pl[0].own_lo * tile, P), row_hi = std::min(pl[0].own_hi * tile, P);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long seq = 0;
    float best = 1e9f;
    for (int round = 0; round < 4; ++round) {
        hipEventRecord(e0, mainS);
        for (int i = 0; i < 10; ++i) {
            ++seq;
            Gram2Fused f{sync + 16 * (seq % 4), sync + 16 * ((seq + 1) % 4), (unsigned long long*)(sync + 64), seq, (unsigned)pl[0].total_wgs,
                         d_blk_rc[0], d_row_own[0], pl[0].nblocks, tile, row_lo, row_hi, ml, J, mom};
            if (waiter) hipLaunchKernelGGL(gram2_wait_ready_kernel, dim3(1), dim3(64), 0, side, (const unsigned long long*)f.ready, seq, sc);
            hipLaunchKernelGGL(kern, dim3(gram_wgs), dim3(G2_THREADS), lds, mainS, (const T*)U, (const T*)G, (const T*)shift, p, n, J, tab[0], tab[1], f, gram_wgs, MetricFin{});
        }
        hipEventRecord(e1, mainS); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
    printf("fused %s: %d + %d workgroups, %d registers, LDS %d: %.1f us per launch (best of 4 x 10%s)\n", sizeof(T) == 4 ? "f32" : "f64",
           pl[0].total_wgs, pl[1].total_wgs, fa.numRegs, lds, best * 100, waiter ? ", wait kernel on a high-priority stream in front of each" : "");
    // reference head: gram_reduce_kernel over the slabs the last fused launch left
    {
        const long long ngroups = (long long)pl[0].nblocks * tile * tile / Mfma<T>::VEC;
        const long long wgs = (ngroups + RED_G - 1) / RED_G + std::max(1, (row_hi - row_lo + RED_G - 1) / RED_G);
        hipLaunchKernelGGL(gram_reduce_kernel<T>, dim3((unsigned)wgs), dim3(RED_G * RED_S), 0, mainS, (const T*)tab[0].slabs, d_blk_rc[0], d_row_own[0], pl[0].nblocks, tile, ml,
                           J, row_lo, row_hi, 1, tab[0].rowsum_part, (const double*)nullptr, mom_ref, MetricFin{});
        hipDeviceSynchronize();
        std::vector<double> a(ml.uu_len()), b(ml.uu_len());
        hipMemcpy(a.data(), mom, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), mom_ref, b.size() * 8, hipMemcpyDeviceToHost);
        double md = 0, mv = 0; for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs(a[i] - b[i])); mv = fmax(mv, fabs(b[i])); }
        printf("  head of the moment buffer against gram_reduce_kernel: max |d| %.3g of %.3g\n", md, mv);
    }
#ifdef G2_FUSED_STAMPS
    {
        std::vector<long long> st(1024 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_gram2_fst), st.size() * 8);
        long long t0 = st[0]; for (int w = 0; w < gram_wgs; ++w) t0 = std::min(t0, st[w * 8]);
        auto us = [&](long long v) { return v ? (v - t0) / 100.0 : -1.0; };
        double mx[6] = {0, 0, 0, 0, 0, 0}, mn[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9}; long long polls = 0, chunks = 0, maxch = 0; int noduty = 0;
        for (int w = 0; w < gram_wgs; ++w) {
            for (int k = 0; k < 6; ++k) { const double v = us(st[w * 8 + k]); if (v >= 0) { mx[k] = fmax(mx[k], v); mn[k] = fmin(mn[k], v); } }
            polls += st[w * 8 + 6]; chunks += st[w * 8 + 7]; maxch = std::max(maxch, st[w * 8 + 7]); if (!st[w * 8 + 3]) ++noduty;
        }
        const char* nm[6] = {"start", "first part done", "published", "duty begin", "duty end", "end"};
        for (int k = 0; k < 6; ++k) printf("  %-16s %8.1f .. %8.1f us\n", nm[k], mn[k], mx[k]);
        printf("  polls %lld, chunks %lld (max %lld per workgroup), workgroups without duty %d\n", polls, chunks, maxch, noduty);
        for (int w : {0, 1, 100, 244, 245, 250, 255}) if (w < gram_wgs)
            printf("    wg %3d: %.1f %.1f %.1f %.1f %.1f %.1f polls %lld chunks %lld\n", w, us(st[w*8]), us(st[w*8+1]), us(st[w*8+2]), us(st[w*8+3]), us(st[w*8+4]), us(st[w*8+5]), st[w*8+6], st[w*8+7]);
    }
#endif
    return 0;
}

int main(int argc, char** argv) {
    const int subset = argc > 1 ? atoi(argv[1]) : 0, budget = argc > 2 ? atoi(argv[2]) : 256;
    const bool f64 = argc > 3 && !strcmp(argv[3], "f64");
    if (argc > 1 && !strcmp(argv[1], "fused")) {
        const int waiter = argc > 4 ? atoi(argv[4]) : 0;
        return f64 ? run_fused<double>(512, 512, 32768, budget, waiter) : run_fused<float>(256, 256, 65536, budget, waiter);
    }
    if (f64) return run<double>(subset, budget, 512, 512, 32768);
    return run<float>(subset, budget, 256, 256, 65536);
}
