// Measurement support: what this device sustains on the bare matrix instruction the two O(J)
// kernels are built on, so that a roofline fraction can be read against the datasheet peak AND against the
// rate this box holds right now (devices of the pool differ by several percent in the clock they keep
// under an MFMA-dense load).  No product arithmetic here.
#include "cesx_internal.h"

namespace cesx {

// Bare MFMA loop: 4 independent accumulators per wave, operands in registers (random data: zero operands let
// the chip clock higher than real data does), one wave per SIMD and two workgroups per CU.  Wave 0 of
// workgroup 0 stamps the shader clock (s_memtime) and the 100 MHz reference clock (s_memrealtime) around its
// loop: clock = d(memtime) / d(memrealtime) x 100 MHz.
template <typename T>
__global__ __launch_bounds__(256, 2)
void mfma_calib_kernel(const T* __restrict__ in, T* __restrict__ out, long long* __restrict__ clk, int iters) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    const int lane = threadIdx.x & 63;
    acc_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) acc[c][r] = 0;
    T a[4], b[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { a[v] = in[(lane + 64 * v) & 1023]; b[v] = in[(lane + 64 * v + 256) & 1023]; }
    const bool stamp = blockIdx.x == 0 && threadIdx.x < 64;
    long long c0 = 0, r0 = 0;
    if (stamp) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = M::mma(a[v], b[(v + c) & 3], acc[c]);
    }
    if (stamp) {
        const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    }
    T s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) s += acc[c][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T>
static int calibrate_t(Engine& e, double target_ms, double* tflops, double* clock_ghz, hipStream_t s) {
    using M = Mfma<T>;
    const int wgs = 2 * e.num_cus;
    T *in = nullptr, *out = nullptr;
    long long* clk = nullptr;
    std::vector<T> h(1024);
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (T)((double)(x >> 8) / 8388608.0 - 1.0); }
    hipEvent_t ea = nullptr, eb = nullptr;
    int rc = CESX_OK;
    auto fail = [&](const char* what, hipError_t st) { e.err = std::string(what) + ": " + hipGetErrorString(st); rc = CESX_EHIP; };
    hipError_t st;
    // (every exit goes through the frees at the bottom: a failed allocation leaves the earlier ones to them)
    if ((st = hipMalloc(reinterpret_cast<void**>(&in), 1024 * sizeof(T))) != hipSuccess) fail("hipMalloc", st);
    if (rc == CESX_OK && (st = hipMalloc(reinterpret_cast<void**>(&out), (size_t)wgs * 256 * sizeof(T))) != hipSuccess) fail("hipMalloc", st);
    if (rc == CESX_OK && (st = hipMalloc(reinterpret_cast<void**>(&clk), 16)) != hipSuccess) fail("hipMalloc", st);
    if (rc == CESX_OK && (st = hipMemcpy(in, h.data(), 1024 * sizeof(T), hipMemcpyHostToDevice)) != hipSuccess) fail("hipMemcpy", st);
    if (rc == CESX_OK && ((st = hipEventCreate(&ea)) != hipSuccess || (st = hipEventCreate(&eb)) != hipSuccess)) fail("hipEventCreate", st);
    // flops of one loop iteration of the whole grid: 16 MFMAs per wave, 2 TILE^2 KSTEP flops each
    const double fl_iter = (double)wgs * 4 * 16 * 2.0 * M::TILE * M::TILE * M::KSTEP;
    int iters = 256;
    double ms = 0.0;
    for (int pass = 0; pass < 2 && rc == CESX_OK; ++pass) {
        if ((st = hipEventRecord(ea, s)) != hipSuccess) { fail("hipEventRecord", st); break; }
        hipLaunchKernelGGL(mfma_calib_kernel<T>, dim3(wgs), dim3(256), 0, s, (const T*)in, out, clk, iters);
        if ((st = hipGetLastError()) != hipSuccess) { fail("launch", st); break; }
        if ((st = hipEventRecord(eb, s)) != hipSuccess || (st = hipEventSynchronize(eb)) != hipSuccess) { fail("hipEventSynchronize", st); break; }
        float f = 0.f;
        if ((st = hipEventElapsedTime(&f, ea, eb)) != hipSuccess) { fail("hipEventElapsedTime", st); break; }
        ms = f;
        if (pass == 0) {        // size the measured pass from the short one
            const double per_iter = ms / iters;
            iters = (int)std::min(4.0e6, std::max(256.0, target_ms / std::max(per_iter, 1e-7)));
        }
    }
    if (rc == CESX_OK) {
        long long hc[2] = {0, 0};
        if ((st = hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost)) != hipSuccess) fail("hipMemcpy", st);
        if (tflops) *tflops = ms > 0.0 ? fl_iter * iters / (ms * 1e-3) / 1e12 : 0.0;
        if (clock_ghz) *clock_ghz = hc[1] > 0 ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0;
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    if (in) (void)hipFree(in);
    if (out) (void)hipFree(out);
    if (clk) (void)hipFree(clk);
    return rc;
}

int launch_calibrate(Engine& e, double target_ms, double* tflops, double* clock_ghz, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? calibrate_t<float>(e, target_ms, tflops, clock_ghz, s)
                                   : calibrate_t<double>(e, target_ms, tflops, clock_ghz, s);
}

}  // namespace cesx
