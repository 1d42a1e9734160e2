// dev tool: time update_kernel<float> alone (C2 shape) with ablation switches.
#include "../ces_amd/csrc/kernels_update.hip"
#include "../ces_amd/csrc/kernels_update2.hip"
#include <cstdio>
#include <vector>
#include <cstring>
#include <cstdlib>
static bool flag(int argc, char** argv, const char* f) { for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], f)) return true; return false; }
int main(int argc, char** argv) {
    using namespace cesx;
    const int p = 256, n = 256; long long J = 65536;
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], "J")) J = atoll(argv[i + 1]);
    const bool f_mem = flag(argc, argv, "mem"), f_nomet = flag(argc, argv, "nomet"), f_notri = flag(argc, argv, "notri");
    const int kp = 256, kn = 256, ktot = 768, rpad = 256;
    float *U, *G, *W, *bias, *out, *rowc; double* mpart;
    hipMalloc(&U, p * J * 4); hipMalloc(&G, n * J * 4); hipMalloc(&out, p * J * 4);
    hipMalloc(&W, rpad * ktot * 4); hipMalloc(&bias, rpad * 4); hipMalloc(&rowc, kn * 16); hipMalloc(&mpart, 4096 * 16);
    std::vector<float> h((size_t)p * J);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(U, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(G, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)rpad * ktot);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1001) / 5000.f - 0.1f;
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemset(bias, 0, rpad * 4); hipMemset(rowc, 0, kn * 16);
    UpdArgs<float> a{};
    a.W = W; a.ktot = ktot; a.ldw = ktot; a.bias = bias; a.out_rows = p;
    a.src[0] = U; a.src[1] = G; a.src[2] = nullptr; a.src_rows[0] = p; a.src_rows[1] = n; a.src_rows[2] = p;
    a.src_k0[0] = 0; a.src_k0[1] = kp; a.src_k0[2] = kp + kn; a.src_kind[0] = 0; a.src_kind[1] = 0; a.src_kind[2] = f_mem ? 0 : 1;
    if (f_mem) a.src[2] = U;
    a.nsrc = 3; a.J = J; a.j_offset = 0; a.out = out; a.rowc = rowc; a.metric_part = f_nomet ? nullptr : mpart; a.metric_seg = 1; a.tri_seg = f_notri ? -1 : 2;
    a.seed_lo = 1; a.seed_hi = 2; a.step = 3;
    using C = UpdCfg<float>;
    constexpr int RC = 4 * C::WR * 32, BN = C::WC * 32;
    dim3 grid((unsigned)((J + BN - 1) / BN), 1);
    const int lds = 2 * (RC * C::STRIDE_W + BK * (BN + C::XPAD)) * 4 + 64 + kn * 16;
    auto kern = update_kernel<float, true, UpdCfg<float>::WC>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(UPD_THREADS), lds, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, grid, dim3(UPD_THREADS), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ABL=%d J=%lld xi=%s metrics=%s tri=%s: %.1f us/launch  (%.1f TF algorithmic)\n", UPD_ABL, J, f_mem ? "mem" : "philox", f_nomet ? "off" : "on", f_notri ? "off" : "on",
           ms * 100.0, 2.0 * p * a.ktot * J / (ms * 1e-4) / 1e12);
    return 0;
}
