// dev tool: which feature of the update kernel's K loop costs MFMA issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using acc_t = float __attribute__((ext_vector_type(16)));
struct Args { const float* in; float* out; int iters; int k0[3]; int kind[3]; int flagA, flagB; };
// MODE bit0: VALU operand generation between clusters; bit1: uniform branches around clusters;
// bit2: dynamic kernarg-array s_loads per iteration; bit3: LDS fragment reads (b128 + b32); bit4: barrier per iteration
template <int MODE>
__global__ __launch_bounds__(256, 2) void kr(const Args a) {
    __shared__ float lds[8192];
    acc_t acc[2][4];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[r][c][e] = 0;
    float x[2][4], y[4][4];
    for (int i = 0; i < 8; ++i) x[i / 4][i % 4] = a.in[i * 256 + threadIdx.x];
    for (int i = 0; i < 16; ++i) y[i / 4][i % 4] = a.in[(8 + i) * 256 + threadIdx.x];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a.in[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < a.iters; ++it) {
        bool need0 = true, need1 = true;
        if (MODE & 4) {
            int s = 0;
            for (int q = 1; q < 3; ++q) if (it * 16 >= a.k0[q]) s = q;
            need0 = a.kind[s] != 7; need1 = a.kind[(s + 1) % 3] != 9;
        } else if (MODE & 2) {
            need0 = a.flagA != it; need1 = a.flagB != it;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (MODE & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i / 4][i % 4] = (float)(lane + i + it) * 1e-3f;
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i / 4][i % 4] = (float)(lane - i + it) * 1e-3f;
            }
            if (MODE & 8) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(&lds[((it & 1) * 4096 + r * 640 + (lane & 31) * 20 + g * 8 + (lane >> 5) * 4) & 8191]);
                    x[r][0] = v.x; x[r][1] = v.y; x[r][2] = v.z; x[r][3] = v.w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int v = 0; v < 4; ++v) y[c][v] = lds[((it & 1) * 4096 + 2048 + (g * 8 + (lane >> 5) * 4 + v) * 128 + c * 32 + (lane & 31)) & 8191];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 0 ? need0 : need1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[r][v], y[c][v], acc[r][c], 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
        }
        if (MODE & 16) __syncthreads();
    }
    float s = 0;
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[r][c][e];
    a.out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(Args a, int wgs, int reps = 1) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kr<MODE><<<wgs, 256>>>(a);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) kr<MODE><<<wgs, 256>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    if (reps > 1) printf("  [short kernel, %d iters: %.1f us/launch, ideal %.1f us] ", a.iters, ms * 1e3, a.iters * 64.0 * 64 * (wgs / 256) / 2.4e3);
    double flops = (double)wgs * 4 * a.iters * 64 * 4096.0;
    printf("mode %2d wgs %d: %.1f TF  %.1f cycles/MFMA/SIMD @2.4GHz\n", MODE, wgs, flops / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / ((double)a.iters * 64 * (wgs / 256)));
}
int main() {
    Args a{};
    float* in; hipMalloc(&in, 8192 * 4); hipMalloc(&a.out, 512 * 256 * 4);
    std::vector<float> h(8192);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    a.in = in; a.iters = 4096; a.k0[0] = 0; a.k0[1] = 20000; a.k0[2] = 40000; a.kind[0] = a.kind[1] = a.kind[2] = 0; a.flagA = -1; a.flagB = -2;
    for (int wgs = 256; wgs <= 512; wgs += 256) {
        run<0>(a, wgs); run<1>(a, wgs); run<2>(a, wgs); run<3>(a, wgs); run<4>(a, wgs); run<7>(a, wgs);
        run<8>(a, wgs); run<10>(a, wgs); run<24>(a, wgs); run<30>(a, wgs);
    }
    a.iters = 48;
    run<0>(a, 512, 20); run<30>(a, 512, 20); run<0>(a, 256, 20); run<30>(a, 256, 20); run<0>(a, 1024, 20); run<30>(a, 1024, 20);
    a.iters = 96;
    run<0>(a, 512, 20); run<30>(a, 512, 20);
    return 0;
}
