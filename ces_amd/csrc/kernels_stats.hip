// First moments and the two fourth-order data metrics (HBM-bound streaming
// kernels; they run beside the MFMA-bound Gram kernel).
//
//   rowsum_kernel      sum_j (x_ij - s_i)                [U0.mean / Geval.mean,
//                      and sum_j q_j (g_ij - s_i)          ces/calibrate.py:423, 427]
//   particle_stats_*   q^r_j = r_j^T Gamma^{-1} r_j,  r_j = g_j - y     (:435/:467)
//                      q^e_j = b_j^T Gamma^{-1} b_j,  b_j = g_j - s_g   (:434/:466,
//                      re-centred from s_g to the exact mean in K2)
//   All cross-particle sums are accumulated in fp64 and reduced in a fixed
//   order (two-stage, no atomics).
#include "cesx_internal.h"

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

// grid = (nslices, rows).  part[row * nslices + slice] = sum over the slice of
// (x - shift[row]); if qe != nullptr additionally partq[...] = sum qe_j (x - shift).
template <typename T>
__global__ __launch_bounds__(ST_THREADS)
void rowsum_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                   const T* __restrict__ qe, int p, int n, long long J, int nslices,
                   double* __restrict__ part, double* __restrict__ partq) {
    __shared__ double red[ST_THREADS / 64];
    const int row = blockIdx.y, slice = blockIdx.x;
    const T* x = row < p ? U + (size_t)row * J : G + (size_t)(row - p) * J;
    const T sh = shift ? shift[row] : (T)0;
    const long long per = (J + nslices - 1) / nslices;
    const long long j0 = slice * per, j1 = j0 + per < J ? j0 + per : J;
    const bool wq = qe != nullptr && row >= p;
    double s = 0.0, sq = 0.0;
    for (long long j = j0 + threadIdx.x; j < j1; j += ST_THREADS) {
        const T d = x[j] - sh;
        s += (double)d;
        if (wq) sq += (double)qe[j] * (double)d;
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)row * nslices + slice] = tot;
    if (partq != nullptr) {
        const double totq = block_sum(sq, red);
        if (threadIdx.x == 0) partq[(size_t)row * nslices + slice] = totq;
    }
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

// One thread per particle, diagonal Gamma (gw = 1 / diag(Gamma)).
template <typename T>
__global__ __launch_bounds__(ST_THREADS)
void particle_stats_diag_kernel(const T* __restrict__ G, const T* __restrict__ shift_g,
                                const T* __restrict__ y, const T* __restrict__ gw, int n,
                                long long J, T* __restrict__ qe_out, double* __restrict__ part) {
    __shared__ double red[ST_THREADS / 64];
    const long long j = (long long)blockIdx.x * ST_THREADS + threadIdx.x;
    T qe = 0, qr = 0;
    if (j < J) {
        for (int i = 0; i < n; ++i) {
            const T g = G[(size_t)i * J + j];
            const T b = g - shift_g[i], r = g - y[i], w = gw[i];
            qe += w * b * b;
            qr += w * r * r;
        }
        qe_out[j] = qe;
    }
    const double a = block_sum((double)qr * (double)qr, red);
    const double b2 = block_sum((double)qe * (double)qe, red);
    const double c = block_sum((double)qe, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 3 + 0] = a;
        part[(size_t)blockIdx.x * 3 + 1] = b2;
        part[(size_t)blockIdx.x * 3 + 2] = c;
    }
}

// Dense Gamma: q^e_j = b_j^T Ginv b_j with a 64-particle tile of B staged in
// LDS; q^r_j = q^e_j + 2 wd^T b_j + c0 with d = s_g - y, wd = Ginv d, c0 = d^T wd
// (one quadratic form per particle instead of two).  VALU kernel: the general
// path, not the headline one (every reference example uses Gamma = gamma^2 I,
// examples/scripts/darcy-flow.py:33-34).
template <typename T>
__global__ __launch_bounds__(ST_THREADS)
void particle_stats_dense_kernel(const T* __restrict__ G, const T* __restrict__ shift_g,
                                 const T* __restrict__ Ginv, const T* __restrict__ wd, const double* __restrict__ c0p, int n,
                                 long long J, T* __restrict__ qe_out, double* __restrict__ part) {
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
        if (j < J) qe_out[j] = qe; else { qe = 0; qr = 0; }
    }
    const double a = block_sum((double)qr * (double)qr, red);
    const double b2 = block_sum((double)qe * (double)qe, red);
    const double c = block_sum((double)qe, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 3 + 0] = a;
        part[(size_t)blockIdx.x * 3 + 1] = b2;
        part[(size_t)blockIdx.x * 3 + 2] = c;
    }
}

// d = s_g - y, wd = Ginv d, c0 = d^T wd  (dense Gamma only; one block)
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

// Gather all first-moment / metric partials into the packed moment buffer.
__global__ void stats_final_kernel(const double* __restrict__ rs_part, const double* __restrict__ rq_part,
                                   int rs_slices, const double* __restrict__ ps_part, int ps_blocks,
                                   int p, int n, long long J, double* __restrict__ mom) {
    const int P = p + n;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double* q = mom + 1 + P + (size_t)p * p + (size_t)p * n + (size_t)n * n;
    if (t == 0) mom[0] = (double)J;
    if (t < P) {
        double s = 0.0;
        for (int k = 0; k < rs_slices; ++k) s += rs_part[(size_t)t * rs_slices + k];
        mom[1 + t] = s;
        if (t >= p) {
            double v = 0.0;
            for (int k = 0; k < rs_slices; ++k) v += rq_part[(size_t)t * rs_slices + k];
            q[3 + (t - p)] = v;
        }
    } else if (t < P + 3) {
        const int c = t - P;
        double s = 0.0;
        for (int k = 0; k < ps_blocks; ++k) s += ps_part[(size_t)k * 3 + c];
        q[c] = s;
    }
}

// ---------------------------------------------------------------------------
template <typename T>
static int colsum_t(Engine& e, const void* U, const void* G, double* sums, hipStream_t s) {
    const int P = e.p + e.n;
    hipLaunchKernelGGL(rowsum_kernel<T>, dim3(e.colsum_slices, P), dim3(ST_THREADS), 0, s, (const T*)U,
                       (const T*)G, (const T*)nullptr, (const T*)nullptr, e.p, e.n, (long long)e.J,
                       e.colsum_slices, e.d_colsum_part, (double*)nullptr);
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
static int stats_t(Engine& e, const void* U, const void* G, double* mom, hipStream_t s) {
    const int P = e.p + e.n, n = e.n;
    const T* shift = (const T*)e.d_shiftT;
    T* qe = (T*)e.d_qe;
    int ps_blocks;
    if (e.diag_gamma) {
        ps_blocks = (int)((e.J + ST_THREADS - 1) / ST_THREADS);
        hipLaunchKernelGGL(particle_stats_diag_kernel<T>, dim3(ps_blocks), dim3(ST_THREADS), 0, s,
                           (const T*)G, shift + e.p, (const T*)e.d_yT, (const T*)e.d_gwT, n, (long long)e.J,
                           qe, e.d_stat_part);
    } else {
        ps_blocks = (int)((e.J + 63) / 64);
        hipLaunchKernelGGL(dense_shift_terms_kernel<T>, dim3(1), dim3(ST_THREADS), 0, s, e.d_shift64 + e.p,
                           e.d_y, e.d_Ginv, n, (T*)e.d_wdT, e.d_c0);
        CESX_HIP(hipGetLastError());
        const size_t lds = ((size_t)n * 64 + 8 * 64) * sizeof(T);
        auto kern = particle_stats_dense_kernel<T>;
        CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(ps_blocks), dim3(ST_THREADS), lds, s, (const T*)G, shift + e.p,
                           (const T*)e.d_GinvT, (const T*)e.d_wdT, (const double*)e.d_c0, n, (long long)e.J, qe,
                           e.d_stat_part);
    }
    CESX_HIP(hipGetLastError());
    hipLaunchKernelGGL(rowsum_kernel<T>, dim3(e.colsum_slices, P), dim3(ST_THREADS), 0, s, (const T*)U,
                       (const T*)G, shift, (const T*)qe, e.p, e.n, (long long)e.J, e.colsum_slices,
                       e.d_colsum_part, e.d_colsum_partq);
    CESX_HIP(hipGetLastError());
    hipLaunchKernelGGL(stats_final_kernel, dim3((P + 3 + 255) / 256), dim3(256), 0, s, e.d_colsum_part,
                       e.d_colsum_partq, e.colsum_slices, e.d_stat_part, ps_blocks, e.p, e.n,
                       (long long)e.J, mom);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}
int launch_stats(Engine& e, const void* U, const void* G, double* mom, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? stats_t<float>(e, U, G, mom, s) : stats_t<double>(e, U, G, mom, s);
}

}  // namespace cesx
