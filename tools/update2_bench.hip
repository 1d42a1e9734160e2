// dev tool: update2_kernel (LDS-DMA ring) against update_kernel on the C2 shape: results + time.
#include "../ces_amd/csrc/kernels_update.hip"
#include "../ces_amd/csrc/kernels_update2.hip"
#include "../ces_amd/csrc/kernels_update3.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
static bool flag(int argc, char** argv, const char* f) { for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], f)) return true; return false; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    using namespace cesx;
    int p = 256, n = 256; long long J = 65536;
    for (int i = 1; i + 1 < argc; ++i) {
        if (!strcmp(argv[i], "J")) J = atoll(argv[i + 1]);
        if (!strcmp(argv[i], "p")) p = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "n")) n = atoi(argv[i + 1]);
    }
    const bool f_mem = flag(argc, argv, "mem"), f_nomet = flag(argc, argv, "nomet"), f_notri = flag(argc, argv, "notri");
    const int kp = (p + 15) / 16 * 16, kn = (n + 15) / 16 * 16, ktot = 2 * kp + kn, rpad = (p + 255) / 256 * 256, nkt = ktot / 16;
    float *U, *G, *X, *W, *Wf, *bias, *out, *out2, *rowc; double *mpart, *mpart2;
    CK(hipMalloc(&U, (size_t)p * J * 4)); CK(hipMalloc(&G, (size_t)n * J * 4)); CK(hipMalloc(&X, (size_t)p * J * 4));
    CK(hipMalloc(&out, (size_t)p * J * 4)); CK(hipMalloc(&out2, (size_t)p * J * 4));
    CK(hipMalloc(&W, (size_t)rpad * ktot * 4)); CK(hipMalloc(&Wf, (size_t)rpad * ktot * 4)); CK(hipMalloc(&bias, rpad * 4));
    CK(hipMalloc(&rowc, kn * 16)); CK(hipMalloc(&mpart, 8192 * 16)); CK(hipMalloc(&mpart2, 6 * 8192 * 16));
    {
        std::vector<float> h((size_t)(p > n ? p : n) * J);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
        CK(hipMemcpy(U, h.data(), (size_t)p * J * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 1999) / 1000.f - 1.f;
        CK(hipMemcpy(G, h.data(), (size_t)n * J * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 69069u + 5) % 1997) / 1000.f - 1.f;
        CK(hipMemcpy(X, h.data(), (size_t)p * J * 4, hipMemcpyHostToDevice));
    }
    std::vector<float> hw((size_t)rpad * ktot, 0.f), hwf((size_t)rpad * ktot, 0.f), hb(rpad, 0.f), hr(kn * 4, 0.f);
    for (int i = 0; i < p; ++i) {
        for (int k = 0; k < ktot; ++k) {
            float v = (float)(((size_t)i * ktot + k) * 40503u % 1001) / 5000.f - 0.1f;
            if (k < kp) { if (k >= p) v = 0; }
            else if (k < kp + kn) { if (k - kp >= n) v = 0; }
            else { const int c = k - kp - kn; if (c >= p || c > i) v = 0; }
            hw[(size_t)i * ktot + k] = v;
        }
        hb[i] = 0.01f * i;
    }
    for (int i = 0; i < rpad; ++i) for (int k = 0; k < ktot; ++k) hwf[wf_index(i, k, nkt)] = hw[(size_t)i * ktot + k];
    for (int i = 0; i < n; ++i) { hr[4 * i] = 0.1f * (i % 7); hr[4 * i + 1] = -0.05f * (i % 5); hr[4 * i + 2] = 1.f + 0.01f * i; }
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), rpad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rowc, hr.data(), kn * 16, hipMemcpyHostToDevice));

    UpdArgs<float> a{};
    a.W = W; a.ktot = ktot; a.ldw = ktot; a.bias = bias; a.out_rows = p;
    a.src[0] = U; a.src[1] = G; a.src[2] = f_mem ? X : nullptr; a.src_rows[0] = p; a.src_rows[1] = n; a.src_rows[2] = p;
    a.src_k0[0] = 0; a.src_k0[1] = kp; a.src_k0[2] = kp + kn; a.src_kind[0] = 0; a.src_kind[1] = 0; a.src_kind[2] = f_mem ? 0 : 1;
    a.nsrc = 3; a.J = J; a.j_offset = 12345; a.out = out; a.rowc = rowc; a.metric_part = f_nomet ? nullptr : mpart; a.metric_seg = 1; a.tri_seg = f_notri ? -1 : 2;
    a.seed_lo = 1; a.seed_hi = 2; a.step = 3;
    Upd2Args b{};
    b.Wf = Wf; b.nkt = nkt; b.out_rows = p; b.bias = bias;
    b.src0 = a.src[0]; b.src1 = a.src[1]; b.src2 = a.src[2]; b.rows0 = p; b.rows1 = n; b.rows2 = p;
    b.kt1 = kp / 16; b.kt2 = (kp + kn) / 16; b.kind0 = 0; b.kind1 = 0; b.kind2 = a.src_kind[2];
    b.J = J; b.j_offset = a.j_offset; b.out = out2; b.rowc = rowc; b.metric_part = f_nomet ? nullptr : mpart2; b.metric_seg = 1; b.tri_seg = a.tri_seg;
    b.seed_lo = 1; b.seed_hi = 2; b.step = 3; b.stagger_from = flag(argc, argv, "nostagger") ? 0x7fffffff : 256; b.stagger_n = 2;

    using C = UpdCfg<float>;
    constexpr int RC = 4 * C::WR * 32, BN = C::WC * 32;
    dim3 grid((unsigned)((J + BN - 1) / BN), (unsigned)((p + RC - 1) / RC));
    const int lds = 2 * (RC * C::STRIDE_W + BK * (BN + C::XPAD)) * 4 + 64 + kn * 16;
    auto kern = (J % 4 == 0) ? update_kernel<float, true, UpdCfg<float>::WC> : update_kernel<float, false, UpdCfg<float>::WC>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int lds2 = U2_RING * (U2_WSLOT + U2_XSLOT) + kn * 16;
    auto kern2 = f_mem ? update2_kernel<false> : update2_kernel<true>;      // (mem: every segment from memory = the benchmark's launch)
    CK(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    CK(hipMemset(out, 0, (size_t)p * J * 4)); CK(hipMemset(out2, 0xff, (size_t)p * J * 4));
    hipLaunchKernelGGL(kern, grid, dim3(UPD_THREADS), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern2, grid, dim3(U2_THREADS), lds2, 0, b);
    CK(hipDeviceSynchronize());
    {
        std::vector<float> h1((size_t)p * J), h2((size_t)p * J);
        CK(hipMemcpy(h1.data(), out, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), out2, h2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t bad = 0, worst = 0;
        for (size_t i = 0; i < h1.size(); ++i) {
            const double d = fabs((double)h1[i] - h2[i]);
            if (!(d <= 1e-3)) { if (!bad) worst = i; ++bad; }
            if (d > md) md = d;
            if (fabs(h1[i]) > mx) mx = fabs(h1[i]);
        }
        printf("p=%d n=%d J=%lld xi=%s: max|old-new| = %.3g (max|old| %.3g), mismatches %zu", p, n, J, f_mem ? "mem" : "philox", md, mx, bad);
        if (bad) printf(" first at row %zu col %zu: %g vs %g", worst / J, worst % J, h1[worst], h2[worst]);
        printf("\n");
        if (!f_nomet) {
            std::vector<double> m1(grid.x * 2), m2(grid.x * 2);
            CK(hipMemcpy(m1.data(), mpart, m1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(m2.data(), mpart2, m2.size() * 8, hipMemcpyDeviceToHost));
            double rel = 0;
            for (size_t i = 0; i < m1.size(); ++i) rel = fmax(rel, fabs(m1[i] - m2[i]) / (fabs(m1[i]) + 1e-30));
            printf("metric partials: max rel diff %.3g (first %g vs %g)\n", rel, m1[0], m2[0]);
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern2, grid, dim3(U2_THREADS), lds2, 0, b);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern2, grid, dim3(U2_THREADS), lds2, 0, b);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef U2_CLOCKS
    if (!f_nomet) {
        std::vector<double> m2(grid.x * 2);
        CK(hipMemcpy(m2.data(), mpart2, m2.size() * 8, hipMemcpyDeviceToHost));
        double c = 0, w = 0, wmax = 0;
        for (unsigned i = 0; i < grid.x; ++i) { c += m2[2 * i]; w += m2[2 * i + 1]; wmax = fmax(wmax, m2[2 * i + 1]); }
        printf("per-WG: %.0f core cycles, %.2f us wall (max %.2f us) -> %.0f MHz\n", c / grid.x, w / grid.x / 100.0, wmax / 100.0, c / w * 100.0);
        std::vector<double> m3(grid.x * 2);
        CK(hipMemcpy(m3.data(), mpart2 + 8192, m3.size() * 8, hipMemcpyDeviceToHost));
        double bw = 0, lp = 0;
        for (unsigned i = 0; i < grid.x; ++i) { bw += m3[2 * i]; lp += m3[2 * i + 1]; }
        CK(hipMemcpy(m3.data(), mpart2 + 16384, m3.size() * 8, hipMemcpyDeviceToHost));
        double pro = 0, s0 = 1e30, s1 = 0;
        for (unsigned i = 0; i < grid.x; ++i) { pro += m3[2 * i]; s0 = fmin(s0, m3[2 * i + 1]); s1 = fmax(s1, m3[2 * i + 1]); }
        printf("prologue %.0f cycles; workgroup start times spread over %.2f us\n", pro / grid.x, (s1 - s0) / 100.0);
        CK(hipMemcpy(m3.data(), mpart2 + 24576, m3.size() * 8, hipMemcpyDeviceToHost));
        double g1 = 0, g2 = 0;
        for (unsigned i = 0; i < grid.x; ++i) { g1 += m3[2 * i]; g2 += m3[2 * i + 1]; }
        printf("segments: U %.0f cycles, G %.0f cycles (the rest is xi)\n", g1 / grid.x, g2 / grid.x);
        CK(hipMemcpy(m3.data(), mpart2 + 8192, m3.size() * 8, hipMemcpyDeviceToHost));
        printf("wave 0: K loop %.0f cycles, of which waiting at the barrier %.0f (%.1f per tile)\n", lp / grid.x, bw / grid.x, bw / grid.x / nkt);
        CK(hipMemcpy(m3.data(), mpart2 + 32768, m3.size() * 8, hipMemcpyDeviceToHost));
        double e1 = 0, e2 = 0;
        for (unsigned i = 0; i < grid.x; ++i) { e1 += m3[2 * i]; e2 += m3[2 * i + 1]; }
        printf("wave 0 epilogue: store issue %.0f cycles, + %.0f until acknowledged\n", e1 / grid.x, e2 / grid.x);
    }
#endif
    printf("update2: %.1f us/launch (%.1f TF algorithmic)\n", ms * 100.0, 2.0 * p * ktot * J / (ms * 1e-4) / 1e12);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, grid, dim3(UPD_THREADS), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("update1: %.1f us/launch (%.1f TF algorithmic)\n", ms * 100.0, 2.0 * p * ktot * J / (ms * 1e-4) / 1e12);
    return 0;
}
