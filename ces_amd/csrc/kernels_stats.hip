// Small streaming kernels around the two MFMA kernels:
//   rowsum_kernel      sum_j x_ij: the centring shift of a fresh ensemble
//                      (U0.mean / Geval.mean, ces/calibrate.py:423, :427).  Inside a
//                      run the first moments come fused out of the Gram kernel.
//   particle_stats_dense_kernel   dense-Gamma data metrics
//                      q^e_j = e_j^T Gamma^{-1} e_j (:434/:466), q^r_j = r_j^T Gamma^{-1} r_j
//                      (:435/:467).  With diagonal Gamma (every reference example) the
//                      update kernel K3 accumulates them while the G rows stream by.
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

// Dense Gamma: q^e_j = e_j^T Ginv e_j (e_j = g_j - gbar) with a 64-particle tile of E
// staged in LDS; q^r_j = q^e_j + 2 wd^T e_j + c0 with d = gbar - y, wd = Ginv d, c0 = d^T wd
// (one quadratic form per particle instead of two).  VALU kernel: the general
// path, not the headline one (every reference example uses Gamma = gamma^2 I,
// examples/scripts/darcy-flow.py:33-34).
template <typename T>
__global__ __launch_bounds__(ST_THREADS)
void particle_stats_dense_kernel(const T* __restrict__ G, const T* __restrict__ shift_g,
                                 const T* __restrict__ Ginv, const T* __restrict__ wd, const double* __restrict__ c0p, int n,
                                 long long J, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* B = reinterpret_cast<T*>(smem);                 // [n][64]
    T* acc = B + (size_t)n * 64;                       // [4][64] partial q per row group
    T* lin = acc + 4 * 64;                             // [4][64] partial wd^T b
    __shared__ double red[ST_THREADS / 64];
    const int tj = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long j = (long long)blockIdx.x * 64 + tj;
    for (int i = grp; i < n; i += 4) B[(size_t)i * 64 + tj] = j < J ? G[(size_t)i * J + j] - shift_g[i] : (T)0;
    __syncthreads();
    T q = 0, l = 0;
    for (int i = grp; i < n; i += 4) {
        const T* gi = Ginv + (size_t)i * n;
        T t = 0;
        for (int k = 0; k < n; ++k) t += gi[k] * B[(size_t)k * 64 + tj];
        const T bi = B[(size_t)i * 64 + tj];
        q += bi * t;
        l += wd[i] * bi;
    }
    acc[grp * 64 + tj] = q;
    lin[grp * 64 + tj] = l;
    __syncthreads();
    T qe = 0, qr = 0;
    if (grp == 0) {
        qe = acc[tj] + acc[64 + tj] + acc[128 + tj] + acc[192 + tj];
        const T li = lin[tj] + lin[64 + tj] + lin[128 + tj] + lin[192 + tj];
        qr = qe + 2 * li + (T)(*c0p);
        if (j >= J) { qe = 0; qr = 0; }
    }
    const double a = block_sum((double)qr * (double)qr, red);
    const double b2 = block_sum((double)qe * (double)qe, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 2 + 0] = a;
        part[(size_t)blockIdx.x * 2 + 1] = b2;
    }
}

// d = gbar - y, wd = Ginv d, c0 = d^T wd  (dense Gamma only; one block)
template <typename T>
__global__ void dense_shift_terms_kernel(const double* __restrict__ shift_g, const double* __restrict__ y,
                                         const double* __restrict__ Ginv, int n, T* __restrict__ wd,
                                         double* __restrict__ c0) {
    __shared__ double red[ST_THREADS / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double t = 0.0;
        for (int k = 0; k < n; ++k) t += Ginv[(size_t)i * n + k] * (shift_g[k] - y[k]);
        wd[i] = (T)t;
        acc += t * (shift_g[i] - y[i]);
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) *c0 = tot;
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

template <typename T>
static int data_metrics_t(Engine& e, const void* G, hipStream_t s) {
    const int n = e.n;
    const int blocks = (int)((e.J + 63) / 64);
    hipLaunchKernelGGL(dense_shift_terms_kernel<T>, dim3(1), dim3(ST_THREADS), 0, s, e.d_gbar, e.d_y, e.d_Ginv, n,
                       (T*)e.d_wdT, e.d_c0);
    CESX_HIP(hipGetLastError());
    const size_t lds = ((size_t)n * 64 + 8 * 64) * sizeof(T);
    if (lds > 150 * 1024) { e.err = "dense-Gamma data metrics: n_obs too large for the LDS tile"; return CESX_EINVAL; }
    auto kern = particle_stats_dense_kernel<T>;
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(ST_THREADS), lds, s, (const T*)G, (const T*)e.d_gbarT,
                       (const T*)e.d_GinvT, (const T*)e.d_wdT, (const double*)e.d_c0, n, (long long)e.J,
                       e.d_metric_part);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}
// dense Gamma only: fills e.d_metric_part with one entry per 64 particles
int launch_data_metrics(Engine& e, const void* G, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? data_metrics_t<float>(e, G, s) : data_metrics_t<double>(e, G, s);
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
    const int nparts = e.diag_gamma ? e.last_metric_parts : (int)((e.J + 63) / 64);
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
