// Small streaming kernels around the two MFMA kernels:
//   rowsum_kernel      sum_j x_ij: the centring shift of a fresh ensemble
//                      (U0.mean / Geval.mean, ces/calibrate.py:423, :427).  Inside a
//                      run the first moments come fused out of the Gram kernel.
//   (the data metrics q^e_j = e_j^T Gamma^{-1} e_j (:434/:466), q^r_j = r_j^T Gamma^{-1} r_j (:435/:467) are
//    accumulated by the update kernel K3 while the G rows stream by; a dense Gamma is whitened away first)
//   metric_final_kernel  fixed-order fp64 sum of the per-workgroup partials.
#include "cesx_internal.h"
#include <cstddef>

namespace cesx {

constexpr int ST_THREADS = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;   // valid on thread 0
}

// grid = (nslices, rows).  part[row * nslices + slice] = sum of x over the slice.
template <typename T>
__global__ __launch_bounds__(ST_THREADS)
void rowsum_kernel(const T* __restrict__ U, const T* __restrict__ G, int p, int n, long long J, int nslices,
                   double* __restrict__ part) {
    __shared__ double red[ST_THREADS / 64];
    const int row = blockIdx.y, slice = blockIdx.x;
    const T* x = row < p ? U + (size_t)row * J : G + (size_t)(row - p) * J;
    const long long per = (J + nslices - 1) / nslices;
    const long long j0 = slice * per, j1 = j0 + per < J ? j0 + per : J;
    double s = 0.0;
    for (long long j = j0 + threadIdx.x; j < j1; j += ST_THREADS) s += (double)x[j];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)row * nslices + slice] = tot;
}

// sums[0] = J, sums[1 + row] = sum over slices
__global__ void colsum_final_kernel(const double* __restrict__ part, int rows, int nslices,
                                    long long J, double* __restrict__ sums) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row == 0) sums[0] = (double)J;
    if (row >= rows) return;
    double s = 0.0;
    for (int k = 0; k < nslices; ++k) s += part[(size_t)row * nslices + k];
    sums[1 + row] = s;
}

// shift = sums / N, rounded to the engine dtype (the kernels subtract exactly
// this value, so K2 must add back exactly this value)
template <typename T>
__global__ void set_shift_kernel(const double* __restrict__ sums, int rows, T* __restrict__ shiftT,
                                 double* __restrict__ shift64) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const T v = (T)(sums[1 + row] / sums[0]);
    shiftT[row] = v;
    shift64[row] = (double)v;
}

// Dense Gamma (Engine::whiten): the G part of the centring sums of a fresh ensemble in the whitened coordinates the
// engine works in, sums_g <- L_Gamma^{-1} sums_g (the sum is linear: the caller summed the RAW rows, cesx_colsum)
__global__ void whiten_sums_kernel(const double* __restrict__ Li, int p, int n, const double* __restrict__ sums,
                                   double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 1 + p + n) return;
    if (i < 1 + p) { out[i] = sums[i]; return; }
    const int r = i - 1 - p;
    double t = 0.0;
    for (int k = 0; k <= r; ++k) t += Li[(size_t)r * n + k] * sums[1 + p + k];
    out[i] = t;
}

// sums[0..1] = sum over workgroups of {q_r^2, q_e^2}; scalars: this shard's contribution
// to the two data metrics (divided by the GLOBAL ensemble size) and, from the tail of the
// all-reduced moment buffer, the previous step's global values (multi-device runs)
__global__ void metric_final_kernel(MetricFin f) { metric_final_body(f); }

// ---------------------------------------------------------------------------
template <typename T>
static int colsum_t(Engine& e, const void* U, const void* G, double* sums, hipStream_t s) {
    const int P = e.p + e.n;
    hipLaunchKernelGGL(rowsum_kernel<T>, dim3(e.colsum_slices, P), dim3(ST_THREADS), 0, s, (const T*)U,
                       (const T*)G, e.p, e.n, (long long)e.J, e.colsum_slices, e.d_colsum_part);
    CESX_HIP(hipGetLastError());
    hipLaunchKernelGGL(colsum_final_kernel, dim3((P + 255) / 256), dim3(256), 0, s, e.d_colsum_part, P,
                       e.colsum_slices, (long long)e.J, sums);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}
int launch_colsum(Engine& e, const void* U, const void* G, double* sums, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? colsum_t<float>(e, U, G, sums, s) : colsum_t<double>(e, U, G, sums, s);
}

int launch_set_shift(Engine& e, const double* sums, hipStream_t s) {
    const int P = e.p + e.n;
    if (e.whiten) {
        hipLaunchKernelGGL(whiten_sums_kernel, dim3((1 + P + 255) / 256), dim3(256), 0, s, (const double*)e.d_Wh, e.p, e.n, sums, e.d_sums_w);
        CESX_HIP(hipGetLastError());
        sums = e.d_sums_w;
    }
    if (e.cfg.dtype == CESX_F32)
        hipLaunchKernelGGL(set_shift_kernel<float>, dim3((P + 255) / 256), dim3(256), 0, s, sums, P,
                           (float*)e.d_shiftT, e.d_shift64);
    else
        hipLaunchKernelGGL(set_shift_kernel<double>, dim3((P + 255) / 256), dim3(256), 0, s, sums, P,
                           (double*)e.d_shiftT, e.d_shift64);
    CESX_HIP(hipGetLastError());
    e.shift_valid = true;
    return CESX_OK;
}

// Results of a step go straight into pinned host memory; the sequence number is written
// last (system-scope fence in between) and the host polls it -- no D2H copy, no event wake-up.
__global__ void publish_kernel(const Scalars* __restrict__ sc, Scalars* host, unsigned long long seq) {
    const double* src = reinterpret_cast<const double*>(sc);
    double* dst = reinterpret_cast<double*>(host);
    constexpr int ND = offsetof(Scalars, seq) / 8;
    if (threadIdx.x < ND) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_publish(Engine& e, hipStream_t s) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, e.d_scal, e.h_scal_dev, ++e.seq);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// publish = true: this is the last kernel of the step and also writes the results to the host
MetricFin metric_fin_args(Engine& e, const double* mom, bool publish) {
    const int nparts = e.last_metric_parts;
    // mom == nullptr: {N, lag0, lag1} as K2's centring kernel copied them into the engine's d_lag
    return MetricFin{e.d_metric_part, nparts, mom ? mom : e.d_lag, mom ? e.ml.tail() : (size_t)1, e.d_metric_sums, e.d_scal,
                     publish ? e.h_scal_dev : (Scalars*)nullptr, publish ? ++e.seq : 0ull, 0.0};
}

int launch_metric_final(Engine& e, const double* mom, bool publish, hipStream_t s) {
    hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(ST_THREADS), 0, s, metric_fin_args(e, mom, publish));
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

}  // namespace cesx
