// dev tool: shader cycles per v_mfma_f32_32x32x2_f32 per SIMD as a function of waves per SIMD, operand
// source and accumulator dependence.  Cycles are s_memtime ticks (not wall time: DVFS does not enter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using acc_t = float __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
// OPS: 0 constant operand registers, 1 rotate over preloaded registers, 2 ds_read_b128 fragments (prefetched one
//      step ahead), 3 = 2 + shift subtraction (8 v_sub per 4 MFMAs)
// NACC: independent accumulators per wave (1 = fully dependent chain)
template <int OPS, int NACC>
__global__ __launch_bounds__(1024) void kr(const float* in, float* out, long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = in[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    acc_t acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    f4 a[2], b[2];
    for (int q = 0; q < 2; ++q) { a[q] = *(const f4*)&lds[(lane * 4 + q * 256) & 16383]; b[q] = *(const f4*)&lds[(lane * 4 + q * 256 + 512) & 16383]; }
    const float sa = in[lane], sb = in[64 + lane];
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // one "step" = 4 MFMAs on fragments (a[g&1], b[g&1])
                f4 fa = a[g & 1], fb = b[g & 1];
                if (OPS >= 2) {                    // prefetch the next step's fragments
                    const int o = ((it * 16 + c * 4 + g + 1) * 1024 + lane * 4) & 16383;
                    a[(g + 1) & 1] = *(const f4*)&lds[o];
                    b[(g + 1) & 1] = *(const f4*)&lds[(o + 8192) & 16383];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (OPS == 3) { for (int v = 0; v < 4; ++v) { fa[v] -= sa; fb[v] -= sb; } }
                if (OPS == 0) { for (int v = 0; v < 4; ++v) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc[c], 0, 0, 0); }
                else { for (int v = 0; v < 4; ++v) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[v], fb[v], acc[c], 0, 0, 0); }
                if (OPS >= 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a[0][0] + b[1][1];
    if (lane == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int OPS, int NACC> void run(const float* in, float* out, long long* clk, int threads) {
    const int iters = 4096 / NACC / 4 * 4, wgs = 256, waves = threads / 64;
    kr<OPS, NACC><<<wgs, threads>>>(in, out, clk, iters);
    kr<OPS, NACC><<<wgs, threads>>>(in, out, clk, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(wgs * 16);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0, av = 0;
    for (int w = 0; w < wgs; ++w) { long long m = 0; for (int q = 0; q < waves; ++q) m = std::max(m, h[w * 16 + q]); mx = std::max(mx, (double)m); av += m; }
    av /= wgs;
    const double mfma_per_simd = (double)iters * NACC * 16 * (waves / 4.0);
    printf("ops %d nacc %d waves/SIMD %d: %.1f cycles per MFMA per SIMD (avg WG), %.1f (slowest WG)\n", OPS, NACC, waves / 4, av / mfma_per_simd, mx / mfma_per_simd);
}
int main() {
    float *in, *out; long long* clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 16 * 8);
    std::vector<float> h(4096);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512, 1024}) {
        run<0, 1>(in, out, clk, threads); run<0, 4>(in, out, clk, threads);
        run<1, 1>(in, out, clk, threads); run<1, 4>(in, out, clk, threads);
        run<2, 1>(in, out, clk, threads); run<2, 4>(in, out, clk, threads);
        run<3, 1>(in, out, clk, threads); run<3, 4>(in, out, clk, threads);
    }
    return 0;
}
