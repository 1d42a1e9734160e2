// dev tool: update4_kernel (K3 through the Cholesky factor) against an fp64 host evaluation of the same formula on a
// sample of particles, and its time beside update2_kernel<false, true> (the hk-free form it replaces) at the same shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/update4_bench.hip -o tools/update4_bench && tools/update4_bench [J n] [p n]
#include "../ces_amd/csrc/kernels_update2.hip"
#include "../ces_amd/csrc/kernels_update4.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double rnd(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
int main(int argc, char** argv) {
    using namespace cesx;
    int p = 256, n = 256; long long J = 65536;
    for (int i = 1; i + 1 < argc; ++i) {
        if (!strcmp(argv[i], "J")) J = atoll(argv[i + 1]);
        if (!strcmp(argv[i], "p")) p = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "n")) n = atoi(argv[i + 1]);
    }
    const int kn = (n + 15) / 16 * 16, ng = kn / 16, ntiles = U4_TRI + ng;
    unsigned long long seed = 12345;
    std::vector<double> L((size_t)p * p, 0.0), sinv(p), K((size_t)p * n), bias(256, 0.0);
    for (int i = 0; i < p; ++i) {
        for (int j = 0; j < i; ++j) L[(size_t)i * p + j] = 0.1 * rnd(seed);
        L[(size_t)i * p + i] = 1.0 + 0.2 * rnd(seed);
        sinv[i] = 0.01 * (1.0 + 0.5 * rnd(seed));
        for (int c = 0; c < n; ++c) K[(size_t)i * n + c] = 0.05 * rnd(seed);
        bias[i] = rnd(seed);
    }
    const double hk = 0.0123, alpha = (p + 1.0) / J, s2 = sqrt(2.0 * hk);
    std::vector<float> img((size_t)ntiles * 4096, 0.f), hb(256, 0.f), hrc((size_t)kn * 4, 0.f);
    for (int i = 0; i < p; ++i)
        for (int j = 0; j <= i; ++j) {
            img[wc_index_L(i, j)] = (float)L[(size_t)i * p + j];
            img[wc_index_Lt(j, i)] = (float)(-L[(size_t)i * p + j] * sinv[i]);
        }
    for (int i = 0; i < p; ++i) for (int c = 0; c < n; ++c) img[wc_index_K(i, c)] = (float)(-K[(size_t)i * n + c]);
    for (int i = 0; i < p; ++i) hb[i] = (float)bias[i];
    for (int i = 0; i < n; ++i) { hrc[4 * i] = 0.1f * (i % 7); hrc[4 * i + 1] = -0.05f * (i % 5); hrc[4 * i + 2] = 1.f + 0.01f * i; }
    float *U, *G, *X, *Wc, *dbias, *out, *rowc; double *mpart, *scal;
    CK(hipMalloc(&U, (size_t)p * J * 4)); CK(hipMalloc(&G, (size_t)n * J * 4)); CK(hipMalloc(&X, (size_t)p * J * 4));
    CK(hipMalloc(&out, (size_t)p * J * 4)); CK(hipMalloc(&Wc, img.size() * 4)); CK(hipMalloc(&dbias, 1024));
    CK(hipMalloc(&rowc, kn * 16)); CK(hipMalloc(&mpart, (16384 + 8192 * 8) * 8)); CK(hipMalloc(&scal, 64));
    std::vector<float> hU((size_t)p * J), hG((size_t)n * J), hX((size_t)p * J);
    for (size_t i = 0; i < hU.size(); ++i) hU[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t i = 0; i < hG.size(); ++i) hG[i] = (float)((i * 40503u + 17) % 1999) / 1000.f - 1.f;
    for (size_t i = 0; i < hX.size(); ++i) hX[i] = (float)((i * 69069u + 5) % 1997) / 1000.f - 1.f;
    CK(hipMemcpy(U, hU.data(), hU.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wc, img.data(), img.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, hb.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(rowc, hrc.data(), kn * 16, hipMemcpyHostToDevice));
    const double hs[3] = {hk, s2, alpha};
    CK(hipMemcpy(scal, hs, 24, hipMemcpyHostToDevice));

    Upd4Args a{};
    a.Wc = Wc; a.ng = ng; a.p = p; a.n = n; a.U = U; a.G = G; a.xi = X; a.bias = dbias; a.J = J; a.out = out;
    a.rowc = rowc; a.metric_part = mpart; a.hkp = scal; a.s2p = scal + 1; a.alphap = scal + 2;
    a.stagger_from = 256; a.stagger_n = 2;
    const int lds = U4_RING * (U4_ASLOT + U4_XSLOT) + kn * 16 + 1024;
    CK(hipFuncSetAttribute((const void*)update4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((unsigned)((J + U4_BN - 1) / U4_BN));
    CK(hipMemset(out, 0xff, (size_t)p * J * 4));
    hipLaunchKernelGGL(update4_kernel, grid, dim3(U4_THREADS), lds, 0, a);
    CK(hipDeviceSynchronize());
    {
        std::vector<float> ho((size_t)p * J);
        std::vector<double> hm(grid.x * 2);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hm.data(), mpart, hm.size() * 8, hipMemcpyDeviceToHost));
        // fp64 evaluation on a sample of particles: the first 160, 96 in the middle, the last 160
        std::vector<long long> cols;
        for (long long j = 0; j < 160 && j < J; ++j) cols.push_back(j);
        for (long long j = J / 2; j < J / 2 + 96 && j < J; ++j) cols.push_back(j);
        for (long long j = J > 160 ? J - 160 : 0; j < J; ++j) cols.push_back(j);
        double maxerr = 0, maxref = 0; long long bad = 0;
        std::vector<double> v(p), w(p);
        for (long long j : cols) {
            for (int r = 0; r < p; ++r) {
                double s = 0;
                for (int k = r; k < p; ++k) s += L[(size_t)k * p + r] * sinv[k] * hU[(size_t)k * J + j];
                w[r] = s2 / hk * hX[(size_t)r * J + j] - s;
            }
            for (int i = 0; i < p; ++i) {
                double s = (1.0 / hk + alpha) * hU[(size_t)i * J + j];
                for (int k = 0; k <= i; ++k) s += L[(size_t)i * p + k] * w[k];
                for (int c = 0; c < n; ++c) s -= K[(size_t)i * n + c] * hG[(size_t)c * J + j];
                const double ref = hk * (s + bias[i]), got = ho[(size_t)i * J + j];
                const double err = fabs(ref - got);
                if (!(err <= 2e-4 * (1.0 + fabs(ref)))) { if (bad < 5) printf("  mismatch row %d col %lld: ref %.7g got %.7g\n", i, j, ref, got); ++bad; }
                if (err > maxerr) maxerr = err;
                if (fabs(ref) > maxref) maxref = fabs(ref);
            }
        }
        printf("p=%d n=%d J=%lld: max|ref-new| = %.3g (max|ref| %.3g) over %zu sampled particles, mismatches %lld\n", p, n, J, maxerr, maxref, cols.size(), bad);
        // metric partials of workgroup 0 and the last one
        for (unsigned wg : {0u, grid.x - 1}) {
            double sr = 0, se = 0;
            for (long long j = (long long)wg * U4_BN; j < (long long)(wg + 1) * U4_BN && j < J; ++j) {
                double qe = 0, qr = 0;
                for (int c = 0; c < n; ++c) {
                    const double x = hG[(size_t)c * J + j], be = x - hrc[4 * c], br = x - hrc[4 * c + 1];
                    qe += hrc[4 * c + 2] * be * be; qr += hrc[4 * c + 2] * br * br;
                }
                se += qe * qe; sr += qr * qr;
            }
            printf("metric partials of workgroup %u: rel err %.3g / %.3g\n", wg, fabs(hm[2 * wg] - sr) / sr, fabs(hm[2 * wg + 1] - se) / se);
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(update4_kernel, grid, dim3(U4_THREADS), lds, 0, a);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(update4_kernel, grid, dim3(U4_THREADS), lds, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("update4: %.1f us/launch (%.1f TF executed of %.2f GFLOP)\n", ms * 50.0, (2.0 * 32 * 32 * 32 * (72 + 4.0 * ng) * (J / 32.0)) / (ms * 5e-5) / 1e12,
               2.0 * 32 * 32 * 32 * (72 + 4.0 * ng) * (J / 32.0) / 1e9);
    }
#ifdef U4_CLOCKS
    {
        std::vector<double> m((size_t)grid.x * 8);
        CK(hipMemcpy(m.data(), mpart + 16384, m.size() * 8, hipMemcpyDeviceToHost));
        double s[6] = {0, 0, 0, 0, 0, 0}, w0 = 1e30, w1 = 0;
        for (unsigned i = 0; i < grid.x; ++i) { for (int k = 0; k < 6; ++k) s[k] += m[8 * i + k]; w0 = fmin(w0, m[8 * i + 6]); w1 = fmax(w1, m[8 * i + 6] + m[8 * i + 5]); }
        printf("wave 0, mean over workgroups: prologue %.0f | triangular tiles %.0f (%.0f per tile) | G tiles %.0f (%.0f per tile) | store issue %.0f | until acknowledged %.0f cycles; workgroup wall %.2f us; first start -> last end %.2f us\n",
               s[0] / grid.x, s[1] / grid.x, s[1] / grid.x / 18, s[2] / grid.x, s[2] / grid.x / ng, s[3] / grid.x, s[4] / grid.x, s[5] / grid.x / 100.0, (w1 - w0) / 100.0);
    }
#endif
    // the hk-free update2 launch at the same shape (timing only: a random fragment-major image)
    if (p == 256 && J % 128 == 0) {
        const int kp = 256, ktot = 2 * kp + kn, nkt = ktot / 16;
        float* Wf; CK(hipMalloc(&Wf, (size_t)256 * ktot * 4));
        std::vector<float> hw((size_t)256 * ktot, 0.f);
        for (int i = 0; i < p; ++i) for (int k = 0; k < ktot; ++k) { if (k < kp && k > i) continue; hw[wf_index(i, k, nkt)] = 0.01f * (float)rnd(seed); }
        CK(hipMemcpy(Wf, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        Upd2Args b{};
        b.Wf = Wf; b.nkt = nkt; b.out_rows = p; b.bias = dbias; b.src0 = X; b.src1 = U; b.src2 = G; b.rows0 = p; b.rows1 = p; b.rows2 = n;
        b.kt1 = kp / 16; b.kt2 = 2 * kp / 16; b.J = J; b.out = out; b.rowc = rowc; b.metric_part = mpart; b.metric_seg = 2; b.tri_seg = 0;
        b.stagger_from = 256; b.stagger_n = 2; b.hkp = scal; b.s2p = scal + 1;
        const int lds2 = U2_RING * (U2_WSLOT + U2_XSLOT) + kn * 16;
        auto k2 = update2_kernel<false, true>;
        CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
        dim3 grid2((unsigned)(J / 128), 1);
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k2, grid2, dim3(U2_THREADS), lds2, 0, b);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k2, grid2, dim3(U2_THREADS), lds2, 0, b);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("update2 (hk-free): %.1f us/launch\n", ms * 50.0);
        }
    }
    return 0;
}
