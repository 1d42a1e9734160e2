// K2 -- the small dense algebra between the two O(J) passes, all in fp64 and
// entirely on device (no host round trip inside a step):
//   centring of the summed moments             ces/calibrate.py:423-428, :459-460, :475-476
//   metrics                                     :432-435 / :464-467 / :506-509
//   time step                                   :243-267 (||D||_F and eig(D) from n x n moments)
//   C = cov(U) + 1e-8 I, L = chol(C)            :424/:476/:512, :446/:487/:526
//   K = C_ug Gamma^{-1} (or (hk C_gg + Gamma)^{-1})   :429/:461, :439-441/:470-473
//   M = C Sigma^{-1},  P = (I + hk M)^{-1}      :443, :485
//   assembly of W, b for the K3 update GEMM     :443-447, :484-488, :515-527
// p, n are a few hundred: these kernels are latency bound, not roofline bound.
#include "cesx_internal.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace cesx {

constexpr int NPB = 256;         // partial-sum blocks (engine.hip sizes d_part for 256)
int potrf_ld(int n);
constexpr int DT = 256;

__device__ __forceinline__ double dblock_sum(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;   // valid on every thread
}

// One entry of C = cov(U) + 1e-8 I (ces/calibrate.py:424/:476/:512) from the shifted raw moments, with the rounding
// sequence pinned (explicit fma): center_kernel and potrf_reg_kernel (centring fused into its load) must produce the
// same bits.  suu = S_ij - (sa_i sa_j) / N is returned through `suu` (the trace term of the metrics).
__device__ __forceinline__ double cov_entry(double S, double sai, double saj, double invN, double invdiv, bool diag, double* suu) {
    const double t = sai * saj;
    const double u = fma(-t, invN, S);
    if (suu) *suu = u;
    return fma(u, invdiv, diag ? 1e-8 : 0.0);
}

struct MomView {
    int p, n;
    const double* mom;
    __device__ MomLayout ml() const { return MomLayout{p, n}; }
    __device__ double N() const { return mom[0]; }
    __device__ const double* sa() const { return mom + ml().sa(); }
    __device__ const double* sb() const { return mom + ml().sb(); }
    __device__ const double* Saa() const { return mom + ml().Saa(); }
    __device__ const double* Sab() const { return mom + ml().Sab(); }
    __device__ const double* Sbb() const { return mom + ml().Sbb(); }
};

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void center_kernel(MomView mv, const double* __restrict__ shift, const double* __restrict__ y,
                   const double* __restrict__ ustar, const double* __restrict__ gw,
                   const double* __restrict__ sw, int unbiased, int what,
                   double* __restrict__ ubar, double* __restrict__ gbar, double* __restrict__ mvec,
                   double* __restrict__ dg, double* __restrict__ C, double* __restrict__ Cug,
                   double* __restrict__ See, double* __restrict__ Srr, double* __restrict__ K,
                   double* __restrict__ M, double* __restrict__ part, Scalars* __restrict__ sc,
                   double* __restrict__ lag,
                   // join != nullptr (the G part on the caller's stream): workgroup 0 ends only when *join >= join_want --
                   // chol(C) has signalled on the side stream -- so that the assembly launch behind needs no barrier packet
                   const unsigned long long* join = nullptr, unsigned long long join_want = 0,
                   // a poll that runs out (join_ticks of the 100-MHz wall clock) reports the step as failed AND marks it
                   // in join[1]: the assembly and update launches behind leave every result of the step untouched then
                   unsigned long long* fault = nullptr, unsigned long long join_ticks = 0) {
    __shared__ double red[1024 / 64];      // (launched with 256 or, for the U-only part beside a Gram launch, 1024 threads)
    // what & 1: the part that depends on U alone (C, M = C Sigma^{-1}, ubar, tr S_uu, |ubar - u*|^2):
    //           everything chol(C) needs, available before the rest of the Gram is finished
    // what & 2: the part that involves G
    // what & 4: the status word stays (chol(C) with the centring fused into its load already ran for these moments
    //           and may have reported CESX_ENOTPD: potrf_reg_kernel resets the word itself then)
    if ((what & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        if (!(what & 4)) sc->status = CESX_OK;
        sc->radspec = 0.0;
        sc->spare[0] = 0.0;
        sc->absmax = 0.0;
    }
    const int p = mv.p, n = mv.n;
    const double N = mv.N();
    // (reciprocals: potrf_reg_kernel forms the same C_ij while it loads S_aa -- one workgroup, 64 elements per thread --
    //  and two fp64 divisions per element cost it 18 us; both kernels use the SAME expression, so the matrix that is
    //  factored and the matrix M is built from agree bit for bit)
    const double invN = 1.0 / N;
    // what a deferred metric finalisation (Engine::met_deferred) needs of this buffer, copied into engine-owned
    // memory: the caller's moment buffer need not outlive cesx_apply
    if ((what & 2) && lag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        lag[0] = N; lag[1] = mv.mom[mv.ml().tail()]; lag[2] = mv.mom[mv.ml().tail() + 1];
    }
    const double div = unbiased ? N - 1.0 : N;
    const double invdiv = 1.0 / div;
    const double* sa = mv.sa();
    const double* sb = mv.sb();
    // 32-bit index arithmetic (p, n < 32768): a 64-bit divide per element used to dominate this kernel
    const unsigned pp = (unsigned)p * p, pn = (unsigned)p * n, nn = (unsigned)n * n;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    double tr = 0.0, b2 = 0.0, fr = 0.0;
    if (what & 1) {
        const double* Saa = mv.Saa();
        for (unsigned idx = gid; idx < pp; idx += gsz) {
            const unsigned i = idx / (unsigned)p, j = idx - i * (unsigned)p;
            double suu;
            const double c = cov_entry(Saa[idx], sa[i], sa[j], invN, invdiv, i == j, &suu);
            C[idx] = c;
            if (sw) M[idx] = c * sw[j];              // M = C Sigma^{-1}, diagonal Sigma
            if (i == j) tr += suu;
        }
    }
    if (what & 2) {
        const double* Sab = mv.Sab();
        const double* Sbb = mv.Sbb();
        for (unsigned k = gid; k < pn; k += gsz) {
            const unsigned i = k / (unsigned)n, j = k - i * (unsigned)n;
            const double cug = (Sab[k] - sa[i] * sb[j] / N) / N;
            Cug[k] = cug;
            if (gw) K[k] = cug * gw[j];              // K = C_ug Gamma^{-1}, diagonal Gamma
        }
        for (unsigned k = gid; k < nn; k += gsz) {
            const unsigned i = k / (unsigned)n, j = k - i * (unsigned)n;
            const double see = Sbb[k] - sb[i] * sb[j] / N;
            const double mi = shift[p + i] + sb[i] / N - y[i], mj = shift[p + j] + sb[j] / N - y[j];
            const double srr = see + N * mi * mj;
            See[k] = see;
            Srr[k] = srr;
            if (gw) fr += see * srr * gw[i] * gw[j];
        }
    }
    const long long vlo = (what & 1) ? 0 : p, vhi = (what & 2) ? p + n : p;
    for (long long i = vlo + gid; i < vhi; i += gsz) {
        if (i < p) {
            const double ub = shift[i] + sa[i] / N;
            ubar[i] = ub;
            b2 += (ub - ustar[i]) * (ub - ustar[i]);
        } else {
            const int k = (int)(i - p);
            const double d = sb[k] / N;
            gbar[k] = shift[p + k] + d;
            dg[k] = d;
            mvec[k] = shift[p + k] + d - y[k];
        }
    }
    tr = dblock_sum(tr, red);
    b2 = dblock_sum(b2, red);
    fr = dblock_sum(fr, red);
    if (threadIdx.x == 0) {
        if (what & 1) { part[blockIdx.x * 4 + 0] = tr; part[blockIdx.x * 4 + 1] = b2; }
        if (what & 2) part[blockIdx.x * 4 + 2] = fr;
    }
    // launched with fewer than NPB workgroups (the U-only part beside a Gram launch): the scalar kernel
    // sums all NPB slots, the ones no workgroup owns must read zero
    if (blockIdx.x == 0)
        for (int b = gridDim.x + threadIdx.x; b < NPB; b += blockDim.x) {
            if (what & 1) { part[b * 4 + 0] = 0.0; part[b * 4 + 1] = 0.0; }
            if (what & 2) part[b * 4 + 2] = 0.0;
        }
    // the U-only part runs on the side stream in front of chol(C), whose completion the caller's stream may take from a
    // polled word instead of a queue-level wait (launch_dense): its results are written back at agent scope here
    // (ONE write-back of the XCD's L2 per workgroup, behind its waves' acknowledged stores: 1024 threads each fencing
    //  took the kernel from 13 to 23 us)
    if (what & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __threadfence();
    }
    if (join != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {      // (one small workgroup waits: chol(C) needs a CU of its own)
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(join, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < join_want) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > join_ticks) {        // bounded in wall time (s_memrealtime), not in spins
                sc->status = CESX_EHIP;
                __hip_atomic_store(fault, join_want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}

// agent-scope (sc1) load of a double another stream's kernel wrote: coherent whatever the per-XCD L2 holds
__device__ __forceinline__ double ld_agent(const double* ptr) {
    return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void st_agent(double* ptr, double v) {
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// C(m x n) = alpha * A(m x k) * B(k x n) with arbitrary element strides, fp64, on the matrix pipe: one wave per
// 16 x 16 block of C (a workgroup = 2 x 2 blocks), v_mfma_f64_16x16x4_f64 with operands read straight from global
// memory -- the matrices of K2 are a few hundred KB and L2-resident, a lane's A operand is A[i][k + lane/16], its B
// operand B[k + lane/16][j] -- eight k-steps per batch, the next batch's loads in flight behind the current MFMAs.
// (The LDS-staged VALU kernel this replaces took 40 us for 256^3: load, barrier, multiply, barrier, eight times.)
using gemm_d4_t = double __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(DT)
void gemm_kernel(int m, int n, int k, double alpha, const double* __restrict__ A, long long a0, long long a1,
                 const double* __restrict__ B, long long b0, long long b1, double* Cm, int ldc,
                 const double* Cin = nullptr,         // Cin (may be Cm itself): C = Cin + alpha A B
                 const int* __restrict__ skip = nullptr) {      // *skip != 0: nothing to do (spd_inverse)
    if (skip != nullptr && *skip != 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = (blockIdx.y * 2 + (wave >> 1)) * 16, c0 = (blockIdx.x * 2 + (wave & 1)) * 16;
    if (r0 >= m || c0 >= n) return;
    const int i = r0 + (lane & 15), j = c0 + (lane & 15), kk = lane >> 4;
    const bool iok = i < m, jok = j < n;
    const double* pa = A + (long long)(iok ? i : 0) * a0;
    const double* pb = B + (long long)(jok ? j : 0) * b1;
    constexpr int UN = 8;
    gemm_d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double av[2][UN], bv[2][UN];
    auto load = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = k0 + 4 * u + kk;
            const bool kok = kc < k;
            av[buf][u] = (iok && kok) ? pa[(long long)kc * a1] : 0.0;
            bv[buf][u] = (jok && kok) ? pb[(long long)kc * b0] : 0.0;
        }
    };
    load(0, 0);
    int buf = 0;
    for (int k0 = 0; k0 < k; k0 += 4 * UN) {
        if (k0 + 4 * UN < k) {
            if (buf == 0) load(1, k0 + 4 * UN); else load(0, k0 + 4 * UN);
        }
        if (buf == 0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
        }
        buf ^= 1;
    }
    // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kk + 4 * r;
        if (row < m && jok) {
            const size_t o = (size_t)row * ldc + j;
            Cm[o] = (Cin ? Cin[o] : 0.0) + alpha * acc[r];
        }
    }
}

// The same product with K split over the four waves of a workgroup (one 16 x 16 block of C per workgroup): a block's
// chain of dependent operand loads is k / 32 round trips to L2 (8 at k = 256: the 20 us a 256^3 product took were
// latency, not arithmetic), a quarter of it per wave here; the four partial blocks are summed through LDS in a fixed
// order.  For the small n x p x p products of K2 (moments of a linear map, EKS gain, dense Gamma / Sigma).
__global__ __launch_bounds__(DT)
void gemm_splitk_kernel(int m, int n, int k, double alpha, const double* __restrict__ A, long long a0, long long a1,
                        const double* __restrict__ B, long long b0, long long b1, double* Cm, int ldc,
                        const int* __restrict__ skip = nullptr) {          // skip != nullptr and *skip != 0: nothing to do (spd_inverse)
    if (skip != nullptr && *skip != 0) return;
    __shared__ double part[3][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    const int i = r0 + (lane & 15), j = c0 + (lane & 15), kk = lane >> 4;
    const bool iok = i < m, jok = j < n;
    const double* pa = A + (long long)(iok ? i : 0) * a0;
    const double* pb = B + (long long)(jok ? j : 0) * b1;
    const int kper = ((k + 3) / 4 + 3) / 4 * 4;          // k-range of a wave, a multiple of the MFMA's 4
    const int kbeg = wave * kper, kend = kbeg + kper < k ? kbeg + kper : k;
    constexpr int UN = 8;
    gemm_d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double av[2][UN], bv[2][UN];
    auto load = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = k0 + 4 * u + kk;
            const bool kok = kc < kend;
            av[buf][u] = (iok && kok) ? pa[(long long)kc * a1] : 0.0;
            bv[buf][u] = (jok && kok) ? pb[(long long)kc * b0] : 0.0;
        }
    };
    if (kbeg < kend) load(0, kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 4 * UN) {
        if (k0 + 4 * UN < kend) {
            if (buf == 0) load(1, k0 + 4 * UN); else load(0, k0 + 4 * UN);
        }
        if (buf == 0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
        }
        buf ^= 1;
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave != 0) return;
    // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kk + 4 * r;
        if (row < m && jok) Cm[(size_t)row * ldc + j] = alpha * (((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane]);
    }
}

// out[r] = sum_c A[r][c] x[c]   (one wave per row)
__global__ __launch_bounds__(DT)
void matvec_kernel(int rows, int cols, const double* __restrict__ A, const double* __restrict__ x,
                   double* __restrict__ out) {
    const int row = blockIdx.x * (DT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    double s = 0.0;
    for (int c = lane; c < cols; c += 64) s += A[(size_t)row * cols + c] * x[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) out[row] = s;
}

// ---------------------------------------------------------------------------
// Register-resident Cholesky for np <= 256 (np = n rounded up to 32).  The
// lower triangle lives in the accumulator registers of 8 waves as 16 x 16
// tiles in v_mfma_f64_16x16x4_f64 C/D layout (np = 256: 136 tiles, 17 per
// wave, 136 VGPRs); only the current 8-column panel passes through LDS
// (k-major image PnT[k][row], double buffered).  No global traffic between
// the initial load and the store of each finished panel -- a single CU moves
// only ~10 B/clk, which is what bounds the global-memory version above.
// Per panel of 8 columns:
//   (b) every wave factors the 8 x 8 diagonal block redundantly (lane & 7 =
//       row, column broadcast by v_readlane, pivots by v_rsq_f64 + Newton):
//       no barrier, no LDS round trip
//   (c) each row below solves x L11^T = a with L11 in SGPRs
//   (d) rank-8 update of every tile right of the panel with two fp64 MFMAs
//       (operands are contiguous LDS reads), then the next panel's columns
//       are published into the other LDS buffer.
// Two barriers per panel.  A : n x n (lower part read); Lp : np x np, leading
// dimension np, entries above the diagonal are left undefined.
// ---------------------------------------------------------------------------
constexpr int PNB = 32;    // leading dimensions of Cholesky factors are rounded up to this
constexpr int PRT = 512;
constexpr int QNB = 8;
using d4_t = double __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);                   // v_rsq_f64: ~26 good bits
    y = y * (1.5 - 0.5 * a * y * y);
    y = y * (1.5 - 0.5 * a * y * y);
    return y;
}

template <int SLOTS, int CHAINV = 0>      // CHAINV: wq is the chained image of kernels_update4.hip (instantiations of their own: <17> keeps its registers);
                                           // 1: the fp64 factor is written back too, 2: the image only
__global__ __launch_bounds__(PRT, 2)
void potrf_reg_kernel(int n, int np, const double* __restrict__ A, double* __restrict__ Lp, int* status,
                      long long* dbg = nullptr,     // dbg: per-phase cycle counts (tools/potrf_bench only)
                      int lda = 0, int ldl = 0,     // row strides of A / Lp (0: n / np); a diagonal block of a larger matrix
                      // cen_sa != nullptr: A is the RAW second moment S_aa of the packed buffer and the covariance is
                      // formed while it is loaded, C_ij = (S_ij - sa_i sa_j (1/N)) (1/div) + 1e-8 [i == j] -- the arithmetic
                      // of center_kernel, element for element (ces/calibrate.py:424/:476/:512), so the U-only centring
                      // kernel no longer sits in front of the factorisation on the side stream; the status word is
                      // reset here then
                      const double* __restrict__ cen_sa = nullptr, const double* __restrict__ cen_N = nullptr,
                      int cen_unbiased = 0,
                      // done != nullptr: the factor (and the status word) written back at agent scope, then *done =
                      // done_val with release semantics -- what the caller's stream polls instead of waiting for an event
                      unsigned long long* done = nullptr, unsigned long long done_val = 0,
                      // wq != nullptr: every finished panel also goes, in fp32, into the first K segment (columns [0, wq_kp)) of
                      // the fragment-major coefficient image of the hk-free update (Engine::d_Wq, wf_index): a row's 8 panel
                      // entries are two 16-byte pieces of it (even / odd columns = the two k sub-blocks of the MFMA operand)
                      float* __restrict__ wq = nullptr, int wq_nkt = 0, int wq_kp = 0,
                      const int* __restrict__ skip = nullptr,        // *skip != 0: nothing to do (spd_inverse's warm start converged)
                      // wq_sinv != nullptr: wq is the CHAINED image of kernels_update4.hip (wc_index_L / wc_index_Lt): the panel goes into
                      // it twice -- as L, and transposed with its rows scaled by -1 / Sigma_kk (the diagonal prior covariance's inverse)
                      const double* __restrict__ wq_sinv = nullptr) {
    constexpr bool CHAIN = CHAINV != 0, WRITE_L = CHAINV != 2;
    if (skip != nullptr && *skip != 0) return;
    if (lda == 0) lda = n;
    if (ldl == 0) ldl = np;
    double cinvN = 1.0, cinvdiv = 1.0;
    if (cen_sa != nullptr) {
        const double cN = *cen_N;
        cinvN = 1.0 / cN;
        cinvdiv = 1.0 / (cen_unbiased ? cN - 1.0 : cN);
        if (threadIdx.x == 0) *status = CESX_OK;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Panel images, k-major: element (k, row relative to the panel's first row) at k * LDT + Z0 + row.  The Z0 = NPMAX
    // entries in front of every k-row stay ZERO: the rows / columns of a tile that lie left of the panel (finished,
    // no entry in the panel) have negative relative rows and read these zeros -- the operand address of a tile is
    // lane constant + 16 * (tile row or column) - kb, one scalar shift and one vector add, no compare / select.
    // LDT is a compile-time stride: the k + 4 half of an operand pair is an immediate offset of the ds_read.
    constexpr int NPMAX = SLOTS <= 2 ? 64 : SLOTS <= 5 ? 128 : SLOTS <= 10 ? 192 : 256;
    constexpr int LDT = 2 * NPMAX + 4, Z0 = NPMAX;
    double* PnT = reinterpret_cast<double*>(smem);        // [2][8][LDT]  the panel, double buffered
    double* NnT = PnT + 2 * QNB * LDT;                    // [8][LDT]     its negative: the A operand of the trailing update
    for (int i = threadIdx.x; i < 3 * QNB * LDT; i += PRT) PnT[i] = 0.0;
    const int tid = threadIdx.x, lane = tid & 63, i8 = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = np / 16;
    const int ntile = T * (T + 1) / 2;
    const int lc = lane & 15, lr = lane >> 4;             // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg

    d4_t Pt[SLOTS];
    // tile coordinates, 4 bits each, 8 slots per register: wave-uniform, extracted with scalar bit-field ops
    // (an int per slot spills out of the SGPR file and comes back through v_readlane in the trailing loop)
    constexpr int NPK = (SLOTS + 7) / 8;
    unsigned pkR[NPK], pkC[NPK];
#pragma clang loop unroll(full)
    for (int i = 0; i < NPK; ++i) { pkR[i] = 0; pkC[i] = 0; }
    int non = 0;                                          // slots of this wave that hold a tile (a prefix)
#pragma clang loop unroll(full)
    for (int s = 0; s < SLOTS; ++s) {
        // Tiles are dealt round-robin to the 8 waves in COLUMN-descending order (last tile column first,
        // rows top to bottom inside a column).  The tiles a panel still has to update are the tile columns
        // at or right of it, i.e. always the first N_act tiles of this order: every wave's share is a prefix
        // of its slots (balanced to within one tile at every panel) and the trailing update below runs as a
        // software-pipelined loop over that prefix instead of a branch per slot.
        const int q = wave + 8 * s;
        int m_ = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);            // columns to the right of this tile's column
        while ((m_ + 1) * (m_ + 2) / 2 <= q) ++m_;
        while (m_ * (m_ + 1) / 2 > q) --m_;
        const int Cc = T - 1 - m_;
        const int R = Cc + (q - m_ * (m_ + 1) / 2);
        const bool on = q < ntile;
        pkR[s / 8] |= (unsigned)__builtin_amdgcn_readfirstlane(on ? R : 0) << (4 * (s % 8));
        pkC[s / 8] |= (unsigned)__builtin_amdgcn_readfirstlane(on ? Cc : 0) << (4 * (s % 8));
        non += __builtin_amdgcn_readfirstlane(on ? 1 : 0);
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e) {
            int i = R * 16 + lr + 4 * e, j = Cc * 16 + lc;
            double v = (i == j) ? 1.0 : 0.0;              // identity padding
            if (j > i) { const int t = i; i = j; j = t; } // diagonal tiles are kept fully symmetric
            if (on && i < n) v = A[(size_t)i * lda + j];      // (cen_sa: the RAW second moment; centred below, from LDS)
            Pt[s][e] = v;
        }
    }
#define TROW(s_) ((int)((pkR[(s_) / 8] >> (4 * ((s_) % 8))) & 15u))
#define TCOL(s_) ((int)((pkC[(s_) / 8] >> (4 * ((s_) % 8))) & 15u))
    long long tph[5] = {0, 0, 0, 0, 0}, tl = clock64();
#define PH(i) if (dbg) { const long long t_ = clock64(); tph[i] += t_ - tl; if (tid == 0 && (i) >= 2) dbg[8 + (kb_dbg / QNB) * 2 + ((i) == 4)] = t_ - tl; tl = t_; }
    int kb_dbg = 0;
    // columns kbn .. kbn+7 of tile s -> buf[col - kbn][row - kbn] (rows at or below the panel's diagonal block)
    // (no per-lane branches: elements outside the panel are written to the 4 pad rows of column 0)
    auto publish = [&](int s, int kbn, double* buf) {
        const int col = TCOL(s) * 16 + lc - kbn;
        const bool cok = col >= 0 && col < QNB;
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e) {
            const int row = TROW(s) * 16 + lr + 4 * e - kbn;
            const int off = (cok && row >= 0) ? col * LDT + Z0 + row : Z0 + np - kbn + e;
            buf[off] = Pt[s][e];
        }
    };
    // centring fused into the load: the row sums go through LDS (one global load per matrix entry instead of
    // three: the dependent sa_i / sa_j loads made this one workgroup's load phase 20 us longer), the raw entries are
    // already on their way into the tile registers
    double* s_sa = PnT + 3 * QNB * LDT;                   // [np] (launched with 2 x NPMAX doubles more when cen_sa or wq_sinv is set)
    double* s_sinv = s_sa + NPMAX;                        // [np] -1 / Sigma_kk (chained image), zero from row n on
    if (cen_sa != nullptr)
        for (int i = threadIdx.x; i < n; i += PRT) s_sa[i] = cen_sa[i];
    if (CHAIN)
        for (int i = threadIdx.x; i < NPMAX; i += PRT) s_sinv[i] = i < n ? -wq_sinv[i] : 0.0;
    __syncthreads();                                      // (the zero fill above, the row sums)
    if (cen_sa != nullptr) {
#pragma clang loop unroll(full)
        for (int s = 0; s < SLOTS; ++s) {
            if (s < non) {
#pragma clang loop unroll(full)
                for (int e = 0; e < 4; ++e) {
                    int i = TROW(s) * 16 + lr + 4 * e, j = TCOL(s) * 16 + lc;
                    if (j > i) { const int t = i; i = j; j = t; }
                    if (i < n) Pt[s][e] = cov_entry(Pt[s][e], s_sa[i], s_sa[j], cinvN, cinvdiv, i == j, nullptr);
                }
            }
        }
    }
#pragma clang loop unroll(full)
    for (int s = 0; s < SLOTS; ++s)
        if (s < non && TCOL(s) == 0) publish(s, 0, PnT);
    __syncthreads();
    PH(0)

    // CHAIN: a finished panel goes into the chained image twice -- as L (wc_index_L) and transposed, its rows scaled by
    // -1 / Sigma_kk (wc_index_Lt) -- from its k-major LDS image, by waves 4..7: they hold no row of the factor phase (b), (c)
    // and share their SIMDs with waves 0..3, whose latency chains leave the issue slots free.  One panel BEHIND: the panel
    // factored in iteration k - 1 still sits in the other buffer during (b), (c) of iteration k (publish writes it in (d)).
    // 1024 sixteen-byte pieces per panel, four per thread.  (Stored by the row threads themselves the two images cost the
    // factorisation 7 - 10 us of 98, tools/potrf_bench.)
    auto image_pass = [&](int kbp, const double* buf) {
        typedef float f4w __attribute__((ext_vector_type(4)));
        const int mp = np - kbp, t4 = tid - PRT / 2;
#pragma clang loop unroll(full)
        for (int it = 0; it < 4; ++it) {
            const int item = t4 + (PRT / 2) * it;
            if (item < 2 * NPMAX) {                       // L[kbp + row][kbp + 4 half .. + 3]
                const int row = item >> 1, half = item & 1;
                if (row < mp && kbp + row < n) {
                    const double* src = buf + 4 * half * LDT + row;
                    *reinterpret_cast<f4w*>(wq + wc_index_L(kbp + row, kbp + 4 * half)) =
                        f4w{(float)src[0], (float)src[LDT], (float)src[2 * LDT], (float)src[3 * LDT]};
                }
            } else {                                      // -(L^T Sigma^{-1})[kbp + j][kbp + 4 c .. + 3] (zeros above the 8 x 8 block's diagonal)
                const int it2 = item - 2 * NPMAX, j = it2 & 7, c4 = (it2 >> 3) * 4;
                if (c4 < mp && kbp + c4 < n && kbp + j < n) {
                    const double* src = buf + j * LDT + c4;
                    const double* sv = s_sinv + kbp + c4;
                    *reinterpret_cast<f4w*>(wq + wc_index_Lt(kbp + j, kbp + c4)) =
                        f4w{(float)(src[0] * sv[0]), (float)(src[1] * sv[1]), (float)(src[2] * sv[2]), (float)(src[3] * sv[3])};
                }
            }
        }
    };
    static_assert(!CHAIN || (NPMAX == 256 && PRT == 512), "image_pass: 1024 pieces over waves 4..7");
    for (int kb = 0; kb < np; kb += QNB) {
        double* cur = PnT + ((kb / QNB) & 1) * QNB * LDT + Z0;            // (data origin: relative row 0)
        double* nxt = PnT + (((kb / QNB) & 1) ^ 1) * QNB * LDT;
        double* neg = NnT + Z0;
        const int m = np - kb;
        kb_dbg = kb;
        if (CHAIN && wave >= 4) {
            // (waves 4..7 hold no row of (b), (c) -- rows QNB + tid < m <= 256: the previous panel's image instead of their
            //  redundant copy of (b).  Without an image to write, dropping that copy alone changed nothing: 99.1 against 97.0 - 97.9 us)
            if (wq != nullptr && kb > 0 && kb - QNB < wq_kp) image_pass(kb - QNB, nxt + Z0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
        // (b) 8 x 8 diagonal block, redundantly per wave
        double d[QNB];
#pragma clang loop unroll(full)
        for (int j = 0; j < QNB; ++j) d[j] = cur[j * LDT + i8];
        double rinv[QNB];
        bool bad = false;
#pragma clang loop unroll(full)
        for (int j = 0; j < QNB; ++j) {
            double djj = readlane_d(d[j], j);
            if (!(djj > 0.0)) { bad = true; djj = 1.0; }
            const double rs = rsqrt_nr(djj);
            rinv[j] = rs;
            const double lij = (i8 == j) ? djj * rs : d[j] * rs;
            d[j] = lij;
#pragma clang loop unroll(full)
            for (int k = j + 1; k < QNB; ++k) d[k] -= lij * readlane_d(lij, k);
        }
        if (bad && tid == 0) *status = CESX_ENOTPD;
        // (c) rows below the block: x L11^T = a
        const int r = QNB + tid;
        if (r < m) {
            double x[QNB];
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) x[j] = cur[j * LDT + r];
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) {
                double sacc = x[j];
#pragma clang loop unroll(full)
                for (int k = 0; k < j; ++k) sacc -= x[k] * readlane_d(d[k], j);      // L11[j][k]
                x[j] = sacc * rinv[j];
            }
            // (CHAINV == 2: the fp64 factor is NOT written back -- the image is all a chained step reads, and the one
            //  CU's store path (~18 B/clk marginal) is what the image stores cost: 97.9 us without an image, 108.2 with both images
            //  AND the factor, tools/potrf_bench; whoever reads Engine::d_L afterwards re-factors C first, Engine::L_stale)
            double* dst = Lp + (size_t)(kb + r) * ldl + kb;
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) {
                cur[j * LDT + r] = x[j];
                neg[j * LDT + r] = -x[j];
                if (WRITE_L) dst[j] = x[j];
            }
            if (wq != nullptr && kb + r < n && kb < wq_kp) {
                typedef float f4w __attribute__((ext_vector_type(4)));
                if (!CHAIN) {          // (CHAIN: both images are written by the waves that hold no row, one panel behind: image_pass)
                    *reinterpret_cast<f4w*>(wq + wf_index(kb + r, kb, wq_nkt)) = f4w{(float)x[0], (float)x[2], (float)x[4], (float)x[6]};
                    *reinterpret_cast<f4w*>(wq + wf_index(kb + r, kb + 1, wq_nkt)) = f4w{(float)x[1], (float)x[3], (float)x[5], (float)x[7]};
                }
            }
        }
        if (tid < QNB) {                                   // the factored diagonal block itself
            double* dst = Lp + (size_t)(kb + tid) * ldl + kb;
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) {
                const double v = j <= tid ? d[j] : 0.0;
                cur[j * LDT + tid] = v;
                if (WRITE_L) dst[j] = v;
            }
            if (wq != nullptr && kb + tid < n && kb < wq_kp) {
                typedef float f4w __attribute__((ext_vector_type(4)));
                float z[QNB];
#pragma clang loop unroll(full)
                for (int j = 0; j < QNB; ++j) z[j] = j <= tid ? (float)d[j] : 0.f;
                if (!CHAIN) {
                    *reinterpret_cast<f4w*>(wq + wf_index(kb + tid, kb, wq_nkt)) = f4w{z[0], z[2], z[4], z[6]};
                    *reinterpret_cast<f4w*>(wq + wf_index(kb + tid, kb + 1, wq_nkt)) = f4w{z[1], z[3], z[5], z[7]};
                }
            }
        }
        // (raw barriers in this loop: __syncthreads() also waits for the global STORES of the panel -- vmcnt(0) -- and the
        //  16 - 24 KB a panel writes leave one CU at ~8 B/clk; nothing in the loop reads global memory)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        PH(2)
        // (d) rank-8 update of the tiles whose columns lie right of the panel (two MFMAs per
        //     tile: A = -L21 rows of the tile, B = L21 rows of the tile's columns); the tile
        //     that holds the next panel's columns is updated too, then published
        const int kn = kb + QNB;                          // first column of the next panel
        // active tiles = tile columns >= kn / 16: the first N_act tiles of the dealing order
        const int tact = T - kn / 16;
        const int n_act = tact > 0 ? tact * (tact + 1) / 2 : 0;
        const int nact = n_act > wave ? (n_act - wave + 7) / 8 : 0;       // ... of which this wave holds the first nact slots
        // Operands of one tile: A = -L21 rows of the tile (from the negated image), B = L21 rows of the tile's
        // columns; finished rows / columns read the zeros in front of the image.  Tiles go in PAIRS: the four
        // MFMAs of a pair alternate between its two accumulators (the two MFMAs of one tile depend on each other),
        // and the LDS reads of the next pair are in flight behind them.
        struct Ops { double av[2], bv[2]; };
        const double* nbase = neg + lr * LDT + lc - kb;
        const double* cbase = cur + lr * LDT + lc - kb;
        auto load_ops = [&](Ops& o, int s_) {
            const double* pa = nbase + TROW(s_) * 16;
            const double* pb = cbase + TCOL(s_) * 16;
#pragma clang loop unroll(full)
            for (int h = 0; h < 2; ++h) { o.av[h] = pa[4 * h * LDT]; o.bv[h] = pb[4 * h * LDT]; }
        };
        Ops oa[2], ob[2];                                 // [pair parity]: first / second tile of the pair
        if (nact > 0) load_ops(oa[0], 0);
        if (nact > 1) load_ops(ob[0], 1);
#pragma clang loop unroll(full)
        for (int s = 0; s < SLOTS; s += 2) {
            if (s < nact) {
                const int pp = (s >> 1) & 1;
                if (s + 2 < SLOTS) { if (s + 2 < nact) load_ops(oa[pp ^ 1], s + 2); }
                if (s + 3 < SLOTS) { if (s + 3 < nact) load_ops(ob[pp ^ 1], s + 3); }
                const bool two = s + 1 < SLOTS && s + 1 < nact;
                Pt[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[pp].av[0], oa[pp].bv[0], Pt[s], 0, 0, 0);
                if (s + 1 < SLOTS) { if (two) Pt[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[pp].av[0], ob[pp].bv[0], Pt[s + 1], 0, 0, 0); }
                Pt[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[pp].av[1], oa[pp].bv[1], Pt[s], 0, 0, 0);
                if (s + 1 < SLOTS) { if (two) Pt[s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[pp].av[1], ob[pp].bv[1], Pt[s + 1], 0, 0, 0); }
                if (kn < np && TCOL(s) == kn / 16) publish(s, kn, nxt);
                if (s + 1 < SLOTS) { if (two && kn < np && TCOL(s + 1) == kn / 16) publish(s + 1, kn, nxt); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PH(4)
    }
#undef TROW
#undef TCOL
    if (CHAIN && wave >= 4 && wq != nullptr && np - QNB < wq_kp)          // the last panel's image
        image_pass(np - QNB, PnT + (((np - QNB) / QNB) & 1) * QNB * LDT + Z0);
    if (dbg && tid == 0)
        for (int i = 0; i < 5; ++i) dbg[i] = tph[i];
#undef PH
    if (done != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores of L acknowledged, then one release (L2 write-back)
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// X = a * A + B with a = (*ap) / (*divp)  (both on device; divp may be null)
__global__ void axpb_kernel(long long len, const double* __restrict__ ap, const double* __restrict__ divp,
                            const double* __restrict__ A, const double* __restrict__ B,
                            double* __restrict__ X) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double a = divp ? (*ap) / (*divp) : (*ap);
    if (i < len) X[i] = a * A[i] + B[i];
}

// K <- Kp when the device-side flag says so (constant / late-mix recompute of D,
// ces/calibrate.py:439-441, :470-473)
__global__ void select_kernel(long long len, const Scalars* __restrict__ sc, const double* __restrict__ Kp,
                              double* __restrict__ K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len && sc->spare[0] != 0.0) K[i] = Kp[i];
}

// ---------------------------------------------------------------------------
// lambda_max of the symmetric PSD matrix B = Gamma^{-1/2} (See / N) Gamma^{-1/2} (Gamma diagonal, or whitened away), which has
// the non-zero spectrum of D = (1/J) E^T Gamma^{-1} E (ces/calibrate.py:250 takes eigvals of the J x J matrix)
// -- by REPEATED SQUARING, on the matrix pipe, with a two-sided bound instead of a convergence test:
//     M_0 = B,   M_{k+1} = (M_k / N_k)^2,   N_k = ||M_k||_F
//     log lambda_1(M_0) = sum_{k < K} 2^-k log N_k + 2^-K log lambda_1(M_K),      N_K / sqrt(n) <= lambda_1(M_K) <= N_K
// so with lambda_1(M_K) ~ N_K n^(-1/4) the relative error is at most 2^-K (ln n) / 4: 3e-11 at K = 36 for n = 16 384, whatever
// the spectrum looks like -- clustered leading eigenvalues (a Marchenko-Pastur bulk at C2: neighbours half a percent apart)
// included.  Rounds 1-4 ran a one-workgroup Lanczos iteration with full re-orthogonalisation and a residual stop criterion:
// rigorous too, but on such a spectrum it walks the whole Krylov space -- 256 steps of 8 - 20 us, 1.8 ms of a 2.2-ms step --
// and it can fail to converge under its step cap.  36 products of n x n x n on v_mfma_f64_16x16x4_f64 (~8 us each at n = 256):
// 0.3 ms, every CU busy, no failure mode.  Rounding: a squaring perturbs lambda_1(M_k) by O(n eps) relatively, weighted 2^-k in
// the sum: O(n eps) in all.  Deterministic: each product leaves the sum of squares of its output as per-workgroup partials,
// the next one adds them in a fixed order.
// ---------------------------------------------------------------------------
constexpr int SPEC_SQUARINGS = 36;

// acc[0] = sum_k 2^-k log N_k so far, acc[1] = 2^-k of the next term.  parts: npart partial sums of squares of M_k (fixed order).
__device__ __forceinline__ double spec_norm2(const double* __restrict__ parts, int npart, double* red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += DT) s += parts[i];
    return dblock_sum(s, red);          // (same value on every thread, the same in every workgroup)
}

// M_out = (M_in / N)^2 with N^2 = sum(parts_in); parts_out[workgroup] = sum of squares of this workgroup's 16 x 16 block of M_out.
// One 16 x 16 block per workgroup, K split over its four waves (the structure of gemm_splitk_kernel).
__global__ __launch_bounds__(DT)
void spec_square_kernel(int n, const double* __restrict__ Min, const double* __restrict__ parts_in, int npart,
                        double* __restrict__ Mout, double* __restrict__ parts_out, double* __restrict__ acc) {
    __shared__ double red[DT / 64];
    __shared__ double part[3][4][64];
    const double N2 = spec_norm2(parts_in, npart, red);
    const double inv = (N2 > 0.0 && N2 < 1e300) ? 1.0 / N2 : 0.0;          // (B = 0: everything stays zero, lambda = 0)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (N2 > 0.0 && N2 < 1e300) acc[0] += acc[1] * 0.5 * log(N2);
        else acc[2] = 1.0;                                                    // degenerate: reported as lambda = 0
        acc[1] *= 0.5;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    const int i = r0 + (lane & 15), j = c0 + (lane & 15), kk = lane >> 4;
    const bool iok = i < n, jok = j < n;
    const double* pa = Min + (size_t)(iok ? i : 0) * n;          // row i of M (symmetric: column j = row j)
    const double* pb = Min + (size_t)(jok ? j : 0) * n;
    const int kper = ((n + 3) / 4 + 3) / 4 * 4;
    const int kbeg = wave * kper, kend = kbeg + kper < n ? kbeg + kper : n;
    constexpr int UN = 8;
    gemm_d4_t a4 = {0.0, 0.0, 0.0, 0.0};
    double av[2][UN], bv[2][UN];
    auto load = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = k0 + 4 * u + kk;
            const bool kok = kc < kend;
            av[buf][u] = (iok && kok) ? pa[kc] : 0.0;
            bv[buf][u] = (jok && kok) ? pb[kc] : 0.0;
        }
    };
    if (kbeg < kend) load(0, kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 4 * UN) {
        if (k0 + 4 * UN < kend) {
            if (buf == 0) load(1, k0 + 4 * UN); else load(0, k0 + 4 * UN);
        }
        if (buf == 0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], a4, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], a4, 0, 0, 0);
        }
        buf ^= 1;
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave - 1][r][lane] = a4[r];
    }
    __syncthreads();
    double sq = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + kk + 4 * r;
            const double v = inv * (((a4[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane]);
            if (row < n && jok) { Mout[(size_t)row * n + j] = v; sq += v * v; }
        }
    }
    sq = dblock_sum(sq, red);          // (waves 1-3 contribute zeros; fixed order)
    if (threadIdx.x == 0) parts_out[blockIdx.y * gridDim.x + blockIdx.x] = sq;
}

// parts[0 .. npart) = per-workgroup sums of squares of A (len doubles); acc = {0, 1, 0}: the start of the sequence
__global__ __launch_bounds__(DT)
void spec_begin_kernel(long long len, const double* __restrict__ A, double* __restrict__ parts, int npart, double* __restrict__ acc) {
    __shared__ double red[DT / 64];
    double s = 0.0;
    const long long per = (len + npart - 1) / npart, lo = (long long)blockIdx.x * per, hi = lo + per < len ? lo + per : len;
    for (long long i = lo + threadIdx.x; i < hi; i += DT) s += A[i] * A[i];
    s = dblock_sum(s, red);
    if (threadIdx.x == 0) {
        parts[blockIdx.x] = s;
        if (blockIdx.x == 0) { acc[0] = 0.0; acc[1] = 1.0; acc[2] = 0.0; }
    }
}

// radspec = lambda_1(B) / N from the accumulated logarithms and the last norm (see the header above)
__global__ __launch_bounds__(DT)
void spec_end_kernel(int n, const double* __restrict__ parts, int npart, const double* __restrict__ acc, const double* __restrict__ divp,
                     Scalars* __restrict__ sc) {
    __shared__ double red[DT / 64];
    const double N2 = spec_norm2(parts, npart, red);
    if (threadIdx.x != 0) return;
    double lam = 0.0;
    if (acc[2] == 0.0 && N2 > 0.0 && N2 < 1e300) lam = exp(acc[0] + acc[1] * (0.5 * log(N2) - 0.25 * log((double)n)));
    sc->radspec = lam / (*divp) > 0.0 ? lam / (*divp) : 0.0;
}

// B = Wh See Wh^T for diagonal Gamma: B_ij = See_ij sqrt(gw_i gw_j)  (1/N applied by the caller)
__global__ void whiten_diag_kernel(int n, const double* __restrict__ See, const double* __restrict__ gw,
                                   double* __restrict__ B) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    B[idx] = See[idx] * sqrt(gw[i] * gw[j]);
}

// ---------------------------------------------------------------------------
// scalars: metrics, time step, pseudo-time
// ---------------------------------------------------------------------------
// the step's scalars from the three summed partials (trace of S_uu, |ubar - u*|^2, Frobenius term); returns hk
__device__ __forceinline__ double step_hk(const cesx_step_params& prm, double N, double fr, double radspec) {
    const double frob = sqrt(fr > 0.0 ? fr : 0.0) / N;
    switch (prm.time_step) {
        case CESX_TS_DEFAULT: return 1.0 / (frob + 1e-8);
        case CESX_TS_SPECTRAL: return 1.0 / radspec;
        case CESX_TS_CONSTANT: return prm.delta_t;
        case CESX_TS_MIX: return (prm.t_len == 0 || prm.t_last < prm.spinup) ? 1.0 / (frob + 1e-8) : prm.delta_t;
        default: return 0.0;
    }
}
__device__ __forceinline__ void write_scalars(const cesx_step_params& prm, int p, double N, double tr, double b2,
                                              double fr, Scalars* __restrict__ sc) {
    sc->tr_suu = tr;
    sc->self_bias = tr / N;
    sc->bias = tr / N + b2;
    sc->frob2 = fr;
    sc->alpha = (p + 1.0) / N;
    if (prm.update != CESX_UPDATE_ALDI_CONSTANT) {
        const double hk = step_hk(prm, N, fr, sc->radspec);
        sc->hk = hk;
        sc->sqrt2hk = sqrt(2.0 * hk);
        sc->t_new = prm.first_step ? hk : hk + prm.t_last;
        bool kp = prm.time_step == CESX_TS_CONSTANT;
        if (prm.update == CESX_UPDATE_ALDI && prm.time_step == CESX_TS_MIX && sc->t_new > 1.0) kp = true;
        sc->spare[0] = kp ? 1.0 : 0.0;
    }
}

// block 0: scalars; blocks 1..: the four matvecs K y, K gbar, M mu, M ubar (one wave per row)
__global__ __launch_bounds__(DT)
void scalar_kernel(MomView mv, cesx_step_params prm, const double* __restrict__ part,
                   Scalars* __restrict__ sc, const double* __restrict__ K, const double* __restrict__ M,
                   const double* __restrict__ y, const double* __restrict__ gbar,
                   const double* __restrict__ mu, const double* __restrict__ ubar, int mx,
                   double* __restrict__ mvs) {
    __shared__ double red[DT / 64];
    const int p = mv.p, tid = threadIdx.x;
    if (blockIdx.x > 0) {
        const int item = (blockIdx.x - 1) * (DT / 64) + (tid >> 6), lane = tid & 63;
        if (item >= 4 * p) return;
        const int which = item / p, row = item % p;
        const int cols = which < 2 ? mv.n : p;
        const double* A = (which < 2 ? K : M) + (size_t)row * cols;
        const double* x = which == 0 ? y : which == 1 ? gbar : which == 2 ? mu : ubar;
        double s = 0.0;
        for (int c = lane; c < cols; c += 64) s += A[c] * x[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) mvs[(size_t)which * mx + row] = s;
        return;
    }
    const double N = mv.N();
    double tr = 0.0, b2 = 0.0, fr = 0.0;
    for (int i = tid; i < NPB; i += DT) { tr += part[i * 4]; b2 += part[i * 4 + 1]; fr += part[i * 4 + 2]; }
    tr = dblock_sum(tr, red);
    b2 = dblock_sum(b2, red);
    fr = dblock_sum(fr, red);
    if (tid != 0) return;
    write_scalars(prm, p, N, tr, b2, fr, sc);
}

// hk = 0.1 / max|drift| (ces/calibrate.py:519-523)
__global__ void constant_hk_kernel(cesx_step_params prm, const double* __restrict__ absmax,
                                   Scalars* __restrict__ sc) {
    const double hk = 0.1 / absmax[0];
    sc->absmax = absmax[0];
    sc->hk = hk;
    sc->sqrt2hk = sqrt(2.0 * hk);
    sc->t_new = prm.first_step ? hk : hk + prm.t_last;
}

// ---------------------------------------------------------------------------
// assembly of W (rpad x ktot, zero padded) and bias in the engine dtype
//   mode 0 ALDI        W = [ (1 + hk a) I - hk M | -hk K | sqrt(2hk) L ],  b = hk (Ky + M mu - a ubar)
//   mode 1 EKS         W = [ P | -hk PK | sqrt(2hk) L ],                   b = P hk (Ky + M mu)   (= Pv)
//   mode 2 ALDI-const drift   W = [ sw a I - M | -K ],                     b = Ky + M mu - sw a ubar
//   mode 3 ALDI-const noise   W = [ sqrt(2hk) L ]
// and the next centring shift (predicted mean for ALDI, current mean otherwise).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void assemble_kernel(int mode, int p, int n, int kp, int kn, int rpad, int ktot, double sw,
                                const Scalars* __restrict__ sc, const double* __restrict__ M,
                                const double* __restrict__ K, const double* __restrict__ L, int ldl,
                                const double* __restrict__ P, const double* __restrict__ PK,
                                const double* __restrict__ mvs, int mx, const double* __restrict__ ubar,
                                const double* __restrict__ gbar, const double* __restrict__ y,
                                const double* __restrict__ gw, T* __restrict__ W, T* __restrict__ bias,
                                T* __restrict__ shiftT, double* __restrict__ shift64, T* __restrict__ rowc,
                                float* __restrict__ Wf) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double hk = sc->hk, s2 = sc->sqrt2hk, al = sc->alpha;
    const double* Ky = mvs;            // K y
    const double* Kg = mvs + mx;       // K gbar
    const double* Mm = mvs + 2 * mx;   // M mu
    const double* Mu = mvs + 3 * mx;   // M ubar
    const double* Pv = mvs + 4 * mx;   // P (hk (Ky + M mu))
    if (idx < (long long)rpad * ktot) {
        const int i = (int)(idx / ktot), k = (int)(idx % ktot);
        double v = 0.0;
        if (i < p) {
            if (mode == 3) {
                if (k < p && k <= i) v = s2 * L[(size_t)i * ldl + k];
            } else if (k < kp) {
                if (k < p) {
                    if (mode == 0) v = (i == k ? 1.0 + hk * al : 0.0) - hk * M[(size_t)i * p + k];
                    else if (mode == 1) v = P[(size_t)i * p + k];
                    else v = (i == k ? sw * al : 0.0) - M[(size_t)i * p + k];
                }
            } else if (k < kp + kn) {
                const int c = k - kp;
                if (c < n) {
                    if (mode == 0) v = -hk * K[(size_t)i * n + c];
                    else if (mode == 1) v = -hk * PK[(size_t)i * n + c];
                    else v = -K[(size_t)i * n + c];
                }
            } else {
                const int c = k - kp - kn;
                if (c < p && c <= i && mode != 2) v = s2 * L[(size_t)i * ldl + c];
            }
        }
        W[idx] = (T)v;
        if (Wf) {                                             // fragment-major copy for the LDS-DMA kernels
            if (sizeof(T) == 4) Wf[wf_index(i, k, ktot / 16)] = (float)v;
            else reinterpret_cast<double*>(Wf)[wd_index(i, k, ktot / 16)] = v;
        }
    }
    if (idx < rpad) {
        const int i = (int)idx;
        double b = 0.0;
        if (i < p) {
            if (mode == 0) b = hk * (Ky[i] + Mm[i] - al * ubar[i]);
            else if (mode == 1) b = Pv[i];
            else if (mode == 2) b = Ky[i] + Mm[i] - sw * al * ubar[i];
        }
        bias[i] = (T)b;
    }
    if (idx < kn && mode != 3) {
        // per-row constants of the K3 data metrics; padded rows get weight 0
        const int i = (int)idx;
        rowc[i * 4 + 0] = (T)(i < n ? gbar[i] : 0.0);
        rowc[i * 4 + 1] = (T)(i < n ? y[i] : 0.0);
        rowc[i * 4 + 2] = (T)((i < n && gw != nullptr) ? gw[i] : 0.0);
        rowc[i * 4 + 3] = (T)0;
    }
    if (idx < p + n && mode != 3) {
        const int i = (int)idx;
        double s;
        if (i < p) {
            s = ubar[i];
            if (mode == 0) s += -hk * (Mu[i] - Mm[i]) - hk * (Kg[i] - Ky[i]);
        } else {
            s = gbar[i - p];
        }
        const T st = (T)s;
        shiftT[i] = st;
        shift64[i] = (double)st;
    }
}

// ---------------------------------------------------------------------------
// ALDI with the default / spectral time step: nothing sits between the scalar kernel and the assembly, so both run
// as ONE launch (a kernel boundary costs 5-7 us here, more beside the noise draw).  Every workgroup sums the same
// NPB partials in the same order -- the same hk everywhere, no grid-wide dependency; workgroups [0, nwb) write W
// (and the vectors that need no matvec), the rest take one row i < p per wave: K_i . y, K_i . gbar, M_i . mu,
// M_i . ubar -> bias_i and the next centring shift.  Same arithmetic, in the same order, as scalar_kernel +
// assemble_kernel<T>(mode 0).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
template <typename T, bool POLLED>
__global__ __launch_bounds__(DT)
void finish_aldi_kernel(MomView mv, cesx_step_params prm, const double* part, Scalars* __restrict__ sc,
                        int nwb, int kp, int kn, int rpad, int ktot, const double* M,
                        const double* K, const double* L, int ldl,
                        const double* __restrict__ y, const double* gbar, const double* __restrict__ mu,
                        const double* ubar, const double* __restrict__ gw, int mx, double* __restrict__ mvs,
                        T* __restrict__ W, T* __restrict__ bias, T* __restrict__ shiftT, double* __restrict__ shift64,
                        T* __restrict__ rowc, float* __restrict__ Wf,
                        const unsigned long long* fault, unsigned long long fault_seq) {
    static_assert(DT == NPB, "one partial per thread");
    __shared__ double red[DT / 64];
    const int p = mv.p, n = mv.n, tid = threadIdx.x;
    // POLLED and the poll in front of this launch ran out: chol(C) is not known to be complete -- nothing of this step is
    // written (W, the next centring shift, the scalars), the status word says why (the U-only centring of the side stream
    // may reset it when it runs at last: set again here)
    if (POLLED && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fault_seq) {
        if (blockIdx.x == 0 && tid == 0) sc->status = CESX_EHIP;
        return;
    }
    const double N = mv.N();
    // POLLED: the side stream was joined through chol(C)'s signal word (a workgroup of the launch in front waited for
    // it), not through a barrier packet: what that stream wrote is read with agent-scope loads
    auto side = [&](const double* q) -> double { return POLLED ? ld_agent(q) : *q; };
    const double tr = dblock_sum(side(part + tid * 4), red);
    const double b2 = dblock_sum(side(part + tid * 4 + 1), red);
    const double fr = dblock_sum(part[tid * 4 + 2], red);
    const double hk = step_hk(prm, N, fr, sc->radspec), s2 = sqrt(2.0 * hk), al = (p + 1.0) / N;
    if (blockIdx.x == 0 && tid == 0) write_scalars(prm, p, N, tr, b2, fr, sc);
    if ((int)blockIdx.x >= nwb) {
        const int i = ((int)blockIdx.x - nwb) * (DT / 64) + (tid >> 6), lane = tid & 63;
        if (i >= p) return;
        double ky = 0.0, kg = 0.0, mm = 0.0, mu_ = 0.0;
        const double* Mi = M + (size_t)i * p;
        const double* Ki = K + (size_t)i * n;
        for (int c = lane; c < n; c += 64) { ky += Ki[c] * y[c]; kg += Ki[c] * gbar[c]; }
        for (int c = lane; c < p; c += 64) { const double m_ = side(Mi + c); mm += m_ * mu[c]; mu_ += m_ * side(ubar + c); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ky += __shfl_down(ky, o, 64); kg += __shfl_down(kg, o, 64);
            mm += __shfl_down(mm, o, 64); mu_ += __shfl_down(mu_, o, 64);
        }
        if (lane == 0) {
            const double ub = side(ubar + i);
            mvs[i] = ky; mvs[(size_t)mx + i] = kg; mvs[(size_t)2 * mx + i] = mm; mvs[(size_t)3 * mx + i] = mu_;
            bias[i] = (T)(hk * (ky + mm - al * ub));
            const T st = (T)(ub + (-hk * (mu_ - mm) - hk * (kg - ky)));
            shiftT[i] = st;
            shift64[i] = (double)st;
        }
        return;
    }
    const long long idx = (long long)blockIdx.x * blockDim.x + tid;
    if (idx < (long long)rpad * ktot) {
        const int i = (int)(idx / ktot), k = (int)(idx % ktot);
        double v = 0.0;
        if (i < p) {
            if (k < kp) {
                if (k < p) v = (i == k ? 1.0 + hk * al : 0.0) - hk * side(M + (size_t)i * p + k);
            } else if (k < kp + kn) {
                const int c = k - kp;
                if (c < n) v = -hk * K[(size_t)i * n + c];
            } else {
                const int c = k - kp - kn;
                if (c < p && c <= i) v = s2 * side(L + (size_t)i * ldl + c);
            }
        }
        W[idx] = (T)v;
        if (Wf) {
            if (sizeof(T) == 4) Wf[wf_index(i, k, ktot / 16)] = (float)v;
            else reinterpret_cast<double*>(Wf)[wd_index(i, k, ktot / 16)] = v;
        }
    }
    if (idx >= p && idx < rpad) bias[idx] = (T)0;
    if (idx < kn) {
        const int i = (int)idx;
        const double gb = i < n ? gbar[i] : 0.0;
        rowc[i * 4 + 0] = (T)gb;
        rowc[i * 4 + 1] = (T)(i < n ? y[i] : 0.0);
        rowc[i * 4 + 2] = (T)((i < n && gw != nullptr) ? gw[i] : 0.0);
        rowc[i * 4 + 3] = (T)0;
    }
    if (idx >= p && idx < p + n) {
        const T st = (T)gbar[idx - p];
        shiftT[idx] = st;
        shift64[idx] = (double)st;
    }
}


// ---------------------------------------------------------------------------
// ALDI, default time step, fp32, diagonal Gamma / Sigma, the update through the LDS-DMA kernel: the G part of the
// centring AND what is left of the assembly once hk is out of the coefficient matrix, as ONE launch behind the second
// reduce (center_kernel(what = 2) + finish_aldi_kernel took two, 8 + 9-17 us beside the noise draw, with W rewritten
// in full between them).  The factorisation has stored (or is storing) L into Engine::d_Wq on the side stream; here:
//   every workgroup : one row i of K = C_ug Gamma^{-1} and of M = C Sigma^{-1} (straight from the moments: nothing of it
//                     waits for the side stream): -K_i and a I - M_i into the image, K_i . y, K_i . gbar, M_i . mu, M_i . ubar;
//                     S_ee, S_rr and its partial of the Frobenius term, gbar and the data-metric constants;
//   the LAST one to arrive (ticket): joins the side stream (polled word, as center_kernel did), sums the partials in the
//                     fixed order of finish_aldi_kernel -> hk, t, metrics (bit-identical scalars), then b' = K y + M mu - a ubar,
//                     the p diagonal entries a - M_ii + 1/hk of the image and the next centring shift.
// Nothing of the step is written when the poll runs out (the update launch checks the same fault word).
// Launched with NPB workgroups of DT threads.
// ---------------------------------------------------------------------------
template <bool DSIG, bool CHAIN = false>        // DSIG: a dense prior covariance (its own instantiation: the benchmark's keeps its registers -- 121 VGPRs, 4 spilled SGPRs)
__global__ __launch_bounds__(DT)                 // CHAIN: wq is the chained image of kernels_update4.hip: -K goes to its G tiles (wc_index_K), a I - M and the
void tail_aldi_kernel(MomView mv, cesx_step_params prm, const double* shift, const double* __restrict__ y,
                      const double* __restrict__ gw, double* __restrict__ gbar, double* __restrict__ mvec,
                      double* __restrict__ dg, double* __restrict__ Cug, double* __restrict__ See,
                      double* __restrict__ Srr, double* __restrict__ K, double* part, Scalars* sc,
                      double* __restrict__ lag, double* mvs, int mx, const double* __restrict__ sw,
                      // Sinv != nullptr: a dense prior covariance -- row i of M = C Sigma^{-1} is C_i . Sigma^{-1} (the C row goes
                      // through LDS, a thread walks its column of Sigma^{-1}: p fp64 fmas and as many L2-resident loads)
                      const double* __restrict__ Sinv,
                      const double* __restrict__ mu,
                      // self_u: no U-only centring ran for these moments (the factorisation formed C while it loaded S_aa):
                      // C, M, ubar and the trace / bias sums are formed here
                      int self_u, const double* __restrict__ ustar, double* __restrict__ ubar, double* __restrict__ Cm,
                      double* __restrict__ Mm,
                      float* wq, int nkt, int kp, int kn, float* __restrict__ bias, float* shiftT, double* shift64,
                      float* __restrict__ rowc, unsigned* ticket,
                      const unsigned long long* join, unsigned long long join_want, unsigned long long* fault,
                      unsigned long long join_ticks) {      // diagonal are not part of it (M only feeds b' and the next shift here)
    constexpr bool chain = CHAIN;
    static_assert(DT == NPB, "one partial per thread");
    __shared__ double red[DT / 64];
    __shared__ int s_flag;
    const int p = mv.p, n = mv.n, tid = threadIdx.x, lane = tid & 63;
    const double N = mv.N();
    const double* sa = mv.sa();
    const double* sb = mv.sb();
    const double* Sab = mv.Sab();
    const double* Sbb = mv.Sbb();
    const unsigned gid = blockIdx.x * DT + tid, gsz = gridDim.x * DT;
    if (lag != nullptr && gid == 0) { lag[0] = N; lag[1] = mv.mom[mv.ml().tail()]; lag[2] = mv.mom[mv.ml().tail() + 1]; }
    // rows of K and of M = C Sigma^{-1} (the arithmetic of center_kernel and of finish_aldi_kernel's matvecs, element for
    // element; M straight from the moments' head: nothing here waits for the side stream).  One WORKGROUP per row: a
    // thread holds one entry of each (every load of the row is in flight at once -- a wave per row walked its 2 x 4
    // dependent iterations in ~8 us), the four matvec sums keep finish_aldi_kernel's order (lane l adds its entries
    // c = l, l + 64, ... in turn, then the shuffle tree) through one LDS exchange per 256 columns.
    const double* Saa = mv.Saa();
    // this thread's first S_ee / S_rr element: its operands are fetched HERE, beside the row's, and used behind the rows
    // (two dependent round trips to L2 / HBM in a row cost this latency-bound launch ~1.5 us)
    const unsigned nn = (unsigned)n * n;
    double e_S = 0.0, e_sbi = 0.0, e_sbj = 0.0, e_shi = 0.0, e_shj = 0.0, e_yi = 0.0, e_yj = 0.0, e_gwi = 0.0, e_gwj = 0.0;
    if (gid < nn) {
        const unsigned i = gid / (unsigned)n, j = gid - i * (unsigned)n;
        e_S = Sbb[gid]; e_sbi = sb[i]; e_sbj = sb[j]; e_shi = shift[p + i]; e_shj = shift[p + j];
        e_yi = y[i]; e_yj = y[j]; e_gwi = gw[i]; e_gwj = gw[j];
    }
    const double invN = 1.0 / N, invdiv = 1.0 / (N - 1.0), al0 = (p + 1.0) / N;
    __shared__ double xch[4][DT];
    __shared__ double crow[DSIG ? DT : 1];
    const int wv = tid >> 6;
    // One row per workgroup and one column pass (p, n <= 256: the benchmark's shapes): the bulk stores of the row -- the
    // image entries, the fp64 copies of C_ug / K (and C / M) -- go out BEHIND the ticket; only what the last workgroup reads
    // (agent-scope stores below) stands between this workgroup's arithmetic and its arrival.
    const bool defer = p <= (int)gridDim.x && p <= DT && n <= DT;
    double d_c = 0.0, d_m = 0.0, d_cug = 0.0, d_kk = 0.0, d_see = 0.0, d_srr = 0.0;
    for (int i = blockIdx.x; i < p; i += gridDim.x) {
        double ky = 0.0, kg = 0.0, mm = 0.0, mu_ = 0.0;
        const int cmax = p > n ? p : n;
        for (int c0 = 0; c0 < cmax; c0 += DT) {
            const int c = c0 + tid;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
            double c_ = 0.0, suu = 0.0;
            if (c < p) c_ = cov_entry(Saa[(size_t)i * p + c], sa[i], sa[c], invN, invdiv, i == c, &suu);
            if (DSIG && c0 == 0) {          // (p <= DT on this path: the whole row of C is in this column pass)
                __syncthreads();
                crow[tid] = c_;
                __syncthreads();
            }
            if (c < p) {
                double m_;
                if (!DSIG) m_ = c_ * sw[c];
                else {
                    // (32 loads of the column in flight at a time: walked one dependent load after the other the column cost the
                    //  launch 100 us -- 256 round trips to L2)
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    for (int k0 = 0; k0 < p; k0 += 32) {
                        double v[32];
#pragma unroll
                        for (int u = 0; u < 32; ++u) v[u] = k0 + u < p ? Sinv[(size_t)(k0 + u) * p + c] : 0.0;
#pragma unroll
                        for (int u = 0; u < 32; u += 4) {
                            a0 = fma(crow[k0 + u], v[u], a0); a1 = fma(crow[k0 + u + 1], v[u + 1], a1);
                            a2 = fma(crow[k0 + u + 2], v[u + 2], a2); a3 = fma(crow[k0 + u + 3], v[u + 3], a3);
                        }
                    }
                    m_ = (a0 + a1) + (a2 + a3);
                }
                if (self_u && i == c) st_agent(mvs + (size_t)5 * mx + i, suu);
                if (i == c) st_agent(mvs + (size_t)4 * mx + i, al0 - m_);      // (the diagonal entry gets 1/hk from the last workgroup)
                if (defer) { d_c = c_; d_m = m_; }
                else {
                    if (self_u) Cm[(size_t)i * p + c] = c_;
                    if (self_u || DSIG) Mm[(size_t)i * p + c] = m_;
                    if (i != c && !chain) wq[wf_index(i, kp + c, nkt)] = (float)(-m_);
                }
                v2 = m_ * mu[c];
                v3 = m_ * (shift[c] + sa[c] / N);
            }
            if (c < n) {
                const size_t k = (size_t)i * n + c;
                const double cug = (Sab[k] - sa[i] * sb[c] / N) / N;
                const double kk = cug * gw[c];
                if (defer) { d_cug = cug; d_kk = kk; }
                else {
                    Cug[k] = cug;
                    K[k] = kk;
                    wq[chain ? wc_index_K(i, c) : wf_index(i, 2 * kp + c, nkt)] = (float)(-kk);
                }
                v0 = kk * y[c];
                v1 = kk * (shift[p + c] + sb[c] / N);
            }
            __syncthreads();
            xch[0][tid] = v0; xch[1][tid] = v1; xch[2][tid] = v2; xch[3][tid] = v3;
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int w = 0; w < DT / 64; ++w) {
                    ky += xch[0][w * 64 + lane]; kg += xch[1][w * 64 + lane];
                    mm += xch[2][w * 64 + lane]; mu_ += xch[3][w * 64 + lane];
                }
            }
        }
        if (wv == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ky += __shfl_down(ky, o, 64); kg += __shfl_down(kg, o, 64);
                mm += __shfl_down(mm, o, 64); mu_ += __shfl_down(mu_, o, 64);
            }
            if (lane == 0) { st_agent(mvs + i, ky); st_agent(mvs + (size_t)mx + i, kg); st_agent(mvs + (size_t)2 * mx + i, mm); st_agent(mvs + (size_t)3 * mx + i, mu_); }
        }
    }
    double fr = 0.0;
    if (gid < nn) {
        const double see = e_S - e_sbi * e_sbj / N;
        const double mi = e_shi + e_sbi / N - e_yi, mj = e_shj + e_sbj / N - e_yj;
        const double srr = see + N * mi * mj;
        d_see = see; d_srr = srr;                 // (stored behind the ticket)
        fr += see * srr * e_gwi * e_gwj;
    }
    for (unsigned k = gid + gsz; k < nn; k += gsz) {          // (n^2 > 65 536 only: the first element of every thread went ahead of the rows)
        const unsigned i = k / (unsigned)n, j = k - i * (unsigned)n;
        const double see = Sbb[k] - sb[i] * sb[j] / N;
        const double mi = shift[p + i] + sb[i] / N - y[i], mj = shift[p + j] + sb[j] / N - y[j];
        const double srr = see + N * mi * mj;
        See[k] = see;
        Srr[k] = srr;
        fr += see * srr * gw[i] * gw[j];
    }
    for (unsigned i = gid; i < (unsigned)kn; i += gsz) {
        double gb = 0.0;
        if (i < (unsigned)n) {
            const double d = sb[i] / N;
            gb = shift[p + i] + d;
            gbar[i] = gb;
            dg[i] = d;
            mvec[i] = gb - y[i];
        }
        rowc[i * 4 + 0] = (float)gb;
        rowc[i * 4 + 1] = (float)(i < (unsigned)n ? y[i] : 0.0);
        rowc[i * 4 + 2] = (float)(i < (unsigned)n ? gw[i] : 0.0);
        rowc[i * 4 + 3] = 0.f;
    }
    fr = dblock_sum(fr, red);
    // arrival.  What the last workgroup reads of this one (the Frobenius partial, the row's four sums and diagonal base)
    // is stored at agent scope (write-through, by the threads that hold it) and waited for; NO write-back of the L2 -- 256
    // of them beside a noise draw that has tens of MB of dirty lines in flight took this kernel from ~12 to 29 us.
    // Everything else written above is read by LATER launches only.
    if (tid == 0) st_agent(part + blockIdx.x * 4 + 2, fr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = t == gridDim.x - 1 ? 1 : 0;
    }
    // the bulk stores held back above: read by LATER launches only
    if (gid < nn) { See[gid] = d_see; Srr[gid] = d_srr; }
    if (defer && (int)blockIdx.x < p) {
        const int i = blockIdx.x, c = tid;
        if (c < p) {
            if (self_u) Cm[(size_t)i * p + c] = d_c;
            if (self_u || DSIG) Mm[(size_t)i * p + c] = d_m;
            if (i != c && !chain) wq[wf_index(i, kp + c, nkt)] = (float)(-d_m);
        }
        if (c < n) {
            const size_t k = (size_t)i * n + c;
            Cug[k] = d_cug;
            K[k] = d_kk;
            wq[chain ? wc_index_K(i, c) : wf_index(i, 2 * kp + c, nkt)] = (float)(-d_kk);
        }
    }
    __syncthreads();
    if (s_flag == 0) return;
    // ---- the last workgroup ----
    if (tid == 0) {
        int ok = 1;
        if (join != nullptr) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(join, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < join_want) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > join_ticks) {
                    sc->status = CESX_EHIP;
                    __hip_atomic_store(fault, join_want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        *ticket = 0u;
        s_flag = ok ? 2 : 3;
    }
    __syncthreads();
    if (s_flag == 3) return;
    // (every agent-scope load of this phase goes out before the first sum: one round trip, not four)
    const double q0 = ld_agent(part + tid * 4), q1 = ld_agent(part + tid * 4 + 1), q2 = ld_agent(part + tid * 4 + 2);
    const int i0 = tid < p ? tid : 0;
    const double ky0 = ld_agent(mvs + i0), kg0 = ld_agent(mvs + (size_t)mx + i0), mm0 = ld_agent(mvs + (size_t)2 * mx + i0),
                 mu0 = ld_agent(mvs + (size_t)3 * mx + i0), db0 = ld_agent(mvs + (size_t)4 * mx + i0);
    double t0 = q0, t1 = q1;
    if (self_u) {          // the trace of S_uu and |ubar - u*|^2, a thread per row
        t0 = 0.0; t1 = 0.0;
        for (int i = tid; i < p; i += DT) {
            const double ub = shift[i] + sa[i] / N, du = ub - ustar[i];
            ubar[i] = ub;
            t0 += ld_agent(mvs + (size_t)5 * mx + i);
            t1 += du * du;
        }
        if (tid == 0) { sc->radspec = 0.0; sc->absmax = 0.0; }
    }
    const double tr = dblock_sum(t0, red);
    const double b2 = dblock_sum(t1, red);
    const double frs = dblock_sum(q2, red);
    const double hk = step_hk(prm, N, frs, 0.0), al = (p + 1.0) / N;
    if (tid == 0) write_scalars(prm, p, N, tr, b2, frs, sc);
    for (int i = tid; i < p; i += DT) {
        const double ub = shift[i] + sa[i] / N;
        const bool first = i == tid;
        const double ky = first ? ky0 : ld_agent(mvs + i), kg = first ? kg0 : ld_agent(mvs + (size_t)mx + i);
        const double mm = first ? mm0 : ld_agent(mvs + (size_t)2 * mx + i), mu_ = first ? mu0 : ld_agent(mvs + (size_t)3 * mx + i),
                     db = first ? db0 : ld_agent(mvs + (size_t)4 * mx + i);
        bias[i] = (float)(ky + mm - al * ub);
        const float st = (float)(ub + (-hk * (mu_ - mm) - hk * (kg - ky)));
        shiftT[i] = st;
        shift64[i] = (double)st;
        if (!chain) wq[wf_index(i, kp + i, nkt)] = (float)(db + 1.0 / hk);
    }
    for (int i = tid; i < n; i += DT) {
        const float st = (float)(shift[p + i] + sb[i] / N);
        shiftT[p + i] = st;
        shift64[p + i] = (double)st;
    }
}

// forward map A (n x p) -> zero-padded rpad x kp row-major image + the fragment-major image the LDS-DMA update
// kernels read (wf_index / wd_index), b -> padded offset vector
template <typename T>
__global__ void stage_forward_kernel(int n, int p, int rpad, int kp, const T* __restrict__ A, const T* __restrict__ b,
                                     T* __restrict__ W, T* __restrict__ Wf, T* __restrict__ bias,
                                     double* __restrict__ A64, double* __restrict__ b64) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long long)rpad * kp) {
        const int i = (int)(idx / kp), k = (int)(idx % kp);
        const T v = (i < n && k < p) ? A[(size_t)i * p + k] : (T)0;
        W[idx] = v;
        if (sizeof(T) == 4) Wf[wf_index(i, k, kp / 16)] = v;
        else Wf[wd_index(i, k, kp / 16)] = v;
        if (i < n && k < p) A64[(size_t)i * p + k] = (double)v;      // the SAME map in fp64 (cesx_moments_rest_lineal)
    }
    if (idx < rpad) bias[idx] = (b != nullptr && idx < n) ? b[idx] : (T)0;
    if (idx < n) b64[idx] = b != nullptr ? (double)b[idx] : 0.0;
}

int launch_stage_forward(Engine& e, const void* A, const void* b, hipStream_t s) {
    const long long len = (long long)e.rpad * e.kp;
    if (e.cfg.dtype == CESX_F32)
        hipLaunchKernelGGL(stage_forward_kernel<float>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, e.n, e.p, e.rpad, e.kp,
                           (const float*)A, (const float*)b, (float*)e.d_Wfwd, (float*)e.d_Wfwd_f, (float*)e.d_bfwd, e.d_A64, e.d_b64);
    else
        hipLaunchKernelGGL(stage_forward_kernel<double>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, e.n, e.p, e.rpad, e.kp,
                           (const double*)A, (const double*)b, (double*)e.d_Wfwd, (double*)e.d_Wfwd_f, (double*)e.d_bfwd, e.d_A64, e.d_b64);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// ---------------------------------------------------------------------------
// G-dependent moments of a LINEAR forward map without a pass over G ("K1 can then fuse G = A U
// into the moments pass").  With g_j = A u_j + b (utils.lineal, ces/utils.py:25-31), a_j = u_j - s_u and
// c = A s_u + b - s_g:   g_j - s_g = A a_j + c, so from the U-only head (N, sa = sum a_j, S_aa = sum a_j a_j^T):
//     sum (g_j - s_g)           = A sa + N c
//     S_ab = sum a_j (g_j-s_g)^T = S_aa A^T + sa c^T
//     S_bb                       = A S_aa A^T + (A sa) c^T + c (A sa)^T + N c c^T
// -- the moments ces/calibrate.py:459-461 / :472 take from Geval, exactly (in fp64, from the fp64 head), for the G the
// engine's own forward kernel produces from this U (cesx_forward_apply).  Two n x p x p fp64 GEMMs instead of the
// second Gram launch (100 of the 136 blocks) and its reduce.
// ---------------------------------------------------------------------------
// lv[0][j] = c_j, lv[1][j] = (A sa)_j: one wave per row of A
__global__ __launch_bounds__(DT)
void lineal_vec_kernel(int n, int p, const double* __restrict__ A, const double* __restrict__ b,
                       const double* __restrict__ shift, const double* __restrict__ sa, double* __restrict__ lv) {
    const int j = blockIdx.x * (DT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    double su = 0.0, as = 0.0;
    for (int k = lane; k < p; k += 64) { const double a = A[(size_t)j * p + k]; su += a * shift[k]; as += a * sa[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { su += __shfl_down(su, o, 64); as += __shfl_down(as, o, 64); }
    if (lane == 0) { lv[j] = su + b[j] - shift[p + j]; lv[n + j] = as; }
}

// tail of the moment buffer from T = S_aa A^T (p x n), B0 = A T (n x n), lv and the head
__global__ void lineal_fix_kernel(MomLayout ml, const double* __restrict__ T, const double* __restrict__ B0,
                                  const double* __restrict__ lv, const double* __restrict__ tail_src, double* __restrict__ mom) {
    const int p = ml.p, n = ml.n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double N = mom[0];
    const double* sa = mom + ml.sa();
    const double* c = lv;
    const double* As = lv + n;
    if (idx < (long long)p * n) {
        const int i = (int)(idx / n), j = (int)(idx % n);
        mom[ml.Sab() + idx] = T[idx] + sa[i] * c[j];
    }
    if (idx < (long long)n * n) {
        const int i = (int)(idx / n), j = (int)(idx % n);
        mom[ml.Sbb() + idx] = B0[idx] + As[i] * c[j] + c[i] * As[j] + N * c[i] * c[j];
    }
    if (idx < n) mom[ml.sb() + idx] = As[idx] + N * c[idx];
    if (idx == 0 && tail_src) { mom[ml.tail()] = tail_src[0]; mom[ml.tail() + 1] = tail_src[1]; }
}

static int gemm(Engine& e, hipStream_t s, int m, int n, int k, double alpha, const double* A, long long a0,
                long long a1, const double* B, long long b0, long long b1, double* C);

int launch_moments_lineal(Engine& e, double* mom, hipStream_t s) {
    const int p = e.p, n = e.n;
    int rc;
    hipLaunchKernelGGL(lineal_vec_kernel, dim3((n + DT / 64 - 1) / (DT / 64)), dim3(DT), 0, s, n, p, e.d_A64, e.d_b64,
                       e.d_shift64, mom + e.ml.sa(), e.d_lvec);
    CESX_HIP(hipGetLastError());
    // T = S_aa A^T (p x n);  B0 = A T (n x n)
    if ((rc = gemm(e, s, p, n, p, 1.0, mom + e.ml.Saa(), p, 1, e.d_A64, 1, p, e.d_t1))) return rc;
    if ((rc = gemm(e, s, n, n, p, 1.0, e.d_A64, p, 1, e.d_t1, n, 1, e.d_t2))) return rc;
    const long long len = (long long)(p > n ? p : n) * n;
    hipLaunchKernelGGL(lineal_fix_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, e.ml, e.d_t1, e.d_t2, e.d_lvec,
                       e.d_metric_sums, mom);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// v = hk * (a + b)
__global__ void hk_sum_kernel(int len, const Scalars* __restrict__ sc, const double* __restrict__ a,
                              const double* __restrict__ b, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) out[i] = sc->hk * (a[i] + b[i]);
}

// ---------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------
static inline dim3 g1(long long len, int bs = 256) { return dim3((unsigned)((len + bs - 1) / bs)); }

static int gemm(Engine& e, hipStream_t s, int m, int n, int k, double alpha, const double* A, long long a0,
                long long a1, const double* B, long long b0, long long b1, double* C) {
    // (small output, long k: K split over the waves of a workgroup -- a 256^3 product 20 -> 8 us, NOTEBOOK.md section 3; else 2 x 2 blocks per workgroup)
    if (k >= 64 && (long long)((m + 15) / 16) * ((n + 15) / 16) <= 4096)
        hipLaunchKernelGGL(gemm_splitk_kernel, dim3((n + 15) / 16, (m + 15) / 16), dim3(DT), 0, s, m, n, k, alpha, A, a0,
                           a1, B, b0, b1, C, n, e.gate);
    else
    hipLaunchKernelGGL(gemm_kernel, dim3((n + 31) / 32, (m + 31) / 32), dim3(DT), 0, s, m, n, k, alpha, A, a0,
                       a1, B, b0, b1, C, n, (const double*)nullptr, e.gate);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

struct PotrfCen { const double* sa = nullptr; const double* N = nullptr; int unbiased = 0; };    // centring fused into the load

template <int SLOTS>
static int potrf_reg_launch(Engine& e, hipStream_t s, int n, int np, const double* A, double* Lp, int lda = 0, int ldl = 0,
                            hipEvent_t stop = nullptr,        // stop: event bound to this kernel's own completion signal
                            PotrfCen cen = PotrfCen(),
                            unsigned long long* done = nullptr, unsigned long long done_val = 0, float* wq = nullptr) {
    constexpr int NPMAX = SLOTS <= 2 ? 64 : SLOTS <= 5 ? 128 : SLOTS <= 10 ? 192 : 256;
    const double* wq_sinv = (wq != nullptr && e.chain) ? (const double*)e.d_sw : (const double*)nullptr;      // the chained image (kernels_update4.hip)
    const size_t lds = (size_t)3 * QNB * (2 * NPMAX + 4) * 8 + ((cen.sa || wq_sinv) ? (size_t)2 * NPMAX * 8 : 0);      // panel x 2, its negative (each k-row behind NPMAX zeros), the row sums of a fused centring, -1 / Sigma_kk
    auto kern = potrf_reg_kernel<SLOTS, 0>;
    // the chained image is all a chained step reads: the fp64 factor is written back only when the step may turn out NOT to be one
    // (Engine::skip_L_hint: the caller knows, or the previous step was chained -- a wrong guess costs one in-line factorisation)
    const bool skip_L = wq_sinv != nullptr && Lp == e.d_L && e.skip_L_hint;
    if (Lp == e.d_L) e.L_stale = skip_L;
    if constexpr (SLOTS == 17) { if (wq_sinv) kern = skip_L ? potrf_reg_kernel<SLOTS, 2> : potrf_reg_kernel<SLOTS, 1>; }      // (Engine::chain implies 224 < p <= 256)
    else if (wq_sinv) { e.err = "potrf: the chained image needs 224 < p <= 256"; return CESX_EINVAL; }
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (stop)
        hipExtLaunchKernelGGL(kern, dim3(1), dim3(PRT), (unsigned)lds, s, nullptr, stop, 0, n, np, A, Lp,
                              &e.d_scal->status, (long long*)nullptr, lda, ldl, cen.sa, cen.N, cen.unbiased, done, done_val,
                              wq, e.ktot / 16, e.kp, e.gate, wq_sinv);
    else
    hipLaunchKernelGGL(kern, dim3(1), dim3(PRT), lds, s, n, np, A, Lp, &e.d_scal->status, (long long*)nullptr,
                       lda, ldl, cen.sa, cen.N, cen.unbiased, done, done_val, wq, e.ktot / 16, e.kp, e.gate, wq_sinv);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// ---------------------------------------------------------------------------
// Rows below a factored diagonal block of the blocked Cholesky (p > 256):  X L^T = A, L the
// nc x nc lower-triangular block just factored (nc <= 256, a multiple of 32), A the nr x nc
// block under it, X -> the same rows of the factor.  Rows are independent, so a workgroup
// takes 64 of them; like potrf_reg_kernel it keeps its 64 x nc block in MFMA accumulator
// registers (4 x nc/16 tiles over 8 waves), walks 8-column panels, solves the panel against
// the 8 x 8 diagonal block (one row per thread) and applies the rank-8 update to the columns
// right of it with v_mfma_f64_16x16x4_f64.  Two barriers per panel.
// ---------------------------------------------------------------------------
template <int SLOTS>
__global__ __launch_bounds__(PRT, 2)
void trsm_reg_kernel(int nr, int nc, const double* __restrict__ A, int lda, const double* __restrict__ L, int ldl,
                     double* __restrict__ X, int ldx, const int* __restrict__ skip = nullptr) {
    if (skip != nullptr && *skip != 0) return;
    __shared__ double XP[QNB][64 + 4];        // the panel: current values, then the solution (column major)
    __shared__ double LBs[2][QNB][256 + 4];   // L[r][kb + j] for r >= kb + j, 0 above the diagonal; double buffered:
                                              // the next panel's columns are fetched from global memory while this
                                              // panel is solved and applied (a ~2 us round trip per panel otherwise)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = nc / 16;                    // tile columns
    const int lc = lane & 15, lr = lane >> 4; // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
    const int rw0 = blockIdx.x * 64;

    d4_t Pt[SLOTS];
    int tR[SLOTS], tC[SLOTS];
#pragma clang loop unroll(full)
    for (int s = 0; s < SLOTS; ++s) {
        const int q = wave + 8 * s;           // tiles dealt round-robin: every wave owns tiles in every column band
        const bool on = q < 4 * T;
        tR[s] = __builtin_amdgcn_readfirstlane(on ? q / T : -1);
        tC[s] = __builtin_amdgcn_readfirstlane(on ? q % T : 0);
#pragma clang loop unroll(full)
        for (int e = 0; e < 4; ++e) {
            const int i = rw0 + (q / T) * 16 + lr + 4 * e, j = (q % T) * 16 + lc;
            // A == nullptr: the right-hand side is the identity (columns ioff .. of it): X = L^{-T}
            Pt[s][e] = (on && i < nr) ? (A ? A[(size_t)i * lda + j] : (i == j + lda ? 1.0 : 0.0)) : 0.0;
        }
    }
    // panel columns of L: thread -> (column j = tid & 7, rows (kbn & ~15) + (tid >> 3) + 64 u), nc <= 256: u < 4
    const int lj = tid & 7;
    double lreg[4];
    auto load_l = [&](int kbn) {
#pragma clang loop unroll(full)
        for (int u = 0; u < 4; ++u) {
            const int r = (kbn & ~15) + (tid >> 3) + 64 * u;
            lreg[u] = (kbn < nc && r < nc && r >= kbn + lj) ? L[(size_t)r * ldl + kbn + lj] : 0.0;
        }
    };
    auto store_l = [&](int kbn, int buf) {
#pragma clang loop unroll(full)
        for (int u = 0; u < 4; ++u) {
            const int r = (kbn & ~15) + (tid >> 3) + 64 * u;
            if (kbn < nc && r < nc) LBs[buf][lj][r] = lreg[u];
        }
    };
    load_l(0);
    store_l(0, 0);
    for (int kb = 0; kb < nc; kb += QNB) {
        const int kn = kb + QNB;
        double (*LB)[256 + 4] = LBs[(kb / QNB) & 1];
        // (a) the panel's columns of the tiles that hold them -> XP; the NEXT panel's columns of L on their way
#pragma clang loop unroll(full)
        for (int s = 0; s < SLOTS; ++s) {
            if (tR[s] >= 0 && tC[s] == kb / 16) {
                const int col = lc - (kb & 15);
                if (col >= 0 && col < QNB) {
#pragma clang loop unroll(full)
                    for (int e = 0; e < 4; ++e) XP[col][tR[s] * 16 + lr + 4 * e] = Pt[s][e];
                }
            }
        }
        load_l(kn);
        __syncthreads();
        // (b) one row per thread: x L_kk^T = a
        if (tid < 64) {
            double x[QNB];
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) {
                double sacc = XP[j][tid];
#pragma clang loop unroll(full)
                for (int k = 0; k < j; ++k) sacc -= x[k] * LB[k][kb + j];       // L_kk[j][k]
                x[j] = sacc / LB[j][kb + j];
            }
            const int i = rw0 + tid;
#pragma clang loop unroll(full)
            for (int j = 0; j < QNB; ++j) {
                XP[j][tid] = x[j];
                if (i < nr) X[(size_t)i * ldx + kb + j] = x[j];
            }
        }
        __syncthreads();
        // (c) rank-8 update of the tiles with columns right of the panel (the tile that holds the
        //     panel is updated too: its remaining columns need it, the solved ones are never read again)
#pragma clang loop unroll(full)
        for (int s = 0; s < SLOTS; ++s) {
            if (tR[s] >= 0 && tC[s] * 16 + 15 >= kn) {
                double av[2], bv[2];
#pragma clang loop unroll(full)
                for (int h = 0; h < 2; ++h) {
                    av[h] = XP[4 * h + lr][tR[s] * 16 + lc];
                    bv[h] = LB[4 * h + lr][tC[s] * 16 + lc];
                }
#pragma clang loop unroll(full)
                for (int h = 0; h < 2; ++h)
                    Pt[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[h], bv[h], Pt[s], 0, 0, 0);
            }
        }
        store_l(kn, ((kb / QNB) & 1) ^ 1);
        __syncthreads();
    }
}

// W (np x np) = A (n x n) bordered by the identity: the padded problem the blocked factorisation works on
__global__ void pad_copy_kernel(int n, int np, const double* __restrict__ A, double* __restrict__ W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)np * np) return;
    const int i = (int)(idx / np), j = (int)(idx % np);
    W[idx] = (i < n && j < n) ? A[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
}

// Cholesky factor of the n x n SPD matrix A into Lp (leading dimension
// potrf_ld(n) = n rounded up to 32; entries above the diagonal are undefined).
int potrf_ld(int n) { return (n + PNB - 1) / PNB * PNB; }

static int potrf_reg_any(Engine& e, hipStream_t s, int n, int np, const double* A, double* Lp, int lda, int ldl,
                         hipEvent_t stop = nullptr, PotrfCen cen = PotrfCen(),
                         unsigned long long* done = nullptr, unsigned long long done_val = 0, float* wq = nullptr) {
    const int T = np / 16, ntile = T * (T + 1) / 2, slots = (ntile + 7) / 8;
    if (slots <= 2) return potrf_reg_launch<2>(e, s, n, np, A, Lp, lda, ldl, stop, cen, done, done_val, wq);       // np <= 64
    if (slots <= 5) return potrf_reg_launch<5>(e, s, n, np, A, Lp, lda, ldl, stop, cen, done, done_val, wq);       // np <= 128
    if (slots <= 10) return potrf_reg_launch<10>(e, s, n, np, A, Lp, lda, ldl, stop, cen, done, done_val, wq);     // np <= 192
    if (slots <= 17) return potrf_reg_launch<17>(e, s, n, np, A, Lp, lda, ldl, stop, cen, done, done_val, wq);     // np <= 256
    e.err = "potrf: diagonal block too large for the register kernel";
    return CESX_EINVAL;
}

static int trsm_reg(Engine& e, hipStream_t s, int nr, int nc, const double* A, int lda, const double* L, int ldl,
                    double* X, int ldx) {
    const dim3 grid((nr + 63) / 64), block(PRT);
    const int slots = (4 * (nc / 16) + 7) / 8;
    if (slots <= 2) hipLaunchKernelGGL(trsm_reg_kernel<2>, grid, block, 0, s, nr, nc, A, lda, L, ldl, X, ldx, e.gate);
    else if (slots <= 4) hipLaunchKernelGGL(trsm_reg_kernel<4>, grid, block, 0, s, nr, nc, A, lda, L, ldl, X, ldx, e.gate);
    else if (slots <= 6) hipLaunchKernelGGL(trsm_reg_kernel<6>, grid, block, 0, s, nr, nc, A, lda, L, ldl, X, ldx, e.gate);
    else hipLaunchKernelGGL(trsm_reg_kernel<8>, grid, block, 0, s, nr, nc, A, lda, L, ldl, X, ldx, e.gate);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// stop (optional): an event to complete with the factorisation.  One-kernel factorisations bind it to the kernel's own
// completion signal (hipExtLaunchKernel: no separate marker packet on the stream -- a marker costs ~6 us before the
// next kernel of the stream starts); the blocked path records it behind its last kernel.
static int potrf(Engine& e, hipStream_t s, int n, const double* A, double* Lp, hipEvent_t stop = nullptr,
                 unsigned long long* done = nullptr, unsigned long long done_val = 0,      // done: stored by the chain's last kernel
                 float* wq = nullptr) {                                                      // wq: the hk-free update's image (one-kernel factorisations)
    const int np = potrf_ld(n);
    if (np <= 256) {
        return potrf_reg_any(e, s, n, np, A, Lp, 0, 0, stop, PotrfCen(), done, done_val, wq);
    }
    // Blocked right-looking factorisation with 256-wide diagonal blocks (p > 256): register
    // Cholesky of the diagonal block, register TRSM of the rows below it (64 rows per
    // workgroup), fp64 GEMM for the trailing update -- on a work copy bordered by the identity.
    if (!e.d_Lwork) { e.err = "potrf: no workspace for the blocked factorisation"; return CESX_EINVAL; }
    double* W = e.d_Lwork;
    hipLaunchKernelGGL(pad_copy_kernel, g1((long long)np * np), dim3(256), 0, s, n, np, A, W);
    CESX_HIP(hipGetLastError());
    int rc;
    for (int k0 = 0; k0 < np; k0 += 256) {
        const int nb = std::min(256, np - k0), below = np - k0 - nb;
        double* Lkk = Lp + (size_t)k0 * np + k0;
        // (the LAST diagonal block's factorisation is the chain's last kernel: it carries the completion word)
        const bool last = below == 0;
        if ((rc = potrf_reg_any(e, s, nb, nb, W + (size_t)k0 * np + k0, Lkk, np, np, nullptr, PotrfCen(),
                                last ? done : nullptr, done_val))) return rc;
        if (below > 0) {
            double* X = Lp + (size_t)(k0 + nb) * np + k0;
            if ((rc = trsm_reg(e, s, below, nb, W + (size_t)(k0 + nb) * np + k0, np, Lkk, np, X, np))) return rc;
            double* W22 = W + (size_t)(k0 + nb) * np + k0 + nb;
            hipLaunchKernelGGL(gemm_kernel, dim3((below + 31) / 32, (below + 31) / 32), dim3(DT), 0, s, below, below, nb, -1.0,
                               X, (long long)np, 1LL, X, 1LL, (long long)np, W22, np, W22);
            CESX_HIP(hipGetLastError());
        }
    }
    if (stop) CESX_HIP(hipEventRecord(stop, s));
    return CESX_OK;
}

// Ainv = A^{-1} for SPD A (n x n); uses t1 (chol), t2 (tri inverse)
// R (nr x nb, row stride nb) = columns c0 .. c0+nb-1 of A (row stride lda), or of the identity (A == nullptr)
__global__ void block_copy_kernel(int nr, int nb, const double* __restrict__ A, int lda, int c0, double* __restrict__ R) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nr * nb) return;
    const int i = (int)(idx / nb), j = (int)(idx % nb);
    R[idx] = A ? A[(size_t)i * lda + c0 + j] : (i == c0 + j ? 1.0 : 0.0);
}

// X L^T = A (A == nullptr: the identity, i.e. X = L^{-T}) for the np x np lower-triangular L (np a
// multiple of 32, padding = identity), nr rows.  np <= 256: one register TRSM; larger: column blocks
// of 256, X_k = (A_k - sum_{j<k} X_j L_kj^T) L_kk^{-T}, the sum by fp64 GEMMs into the workspace.
static int trsm_right_lt(Engine& e, hipStream_t s, int nr, int np, const double* A, int lda, const double* L, int ldl,
                         double* X, int ldx) {
    if (np <= 256) return trsm_reg(e, s, nr, np, A, A ? lda : 0, L, ldl, X, ldx);
    if (!e.d_Lwork) { e.err = "trsm: no workspace"; return CESX_EINVAL; }
    double* R = e.d_Lwork;
    int rc;
    for (int k0 = 0; k0 < np; k0 += 256) {
        const int nb = std::min(256, np - k0);
        hipLaunchKernelGGL(block_copy_kernel, g1((long long)nr * nb), dim3(256), 0, s, nr, nb, A, lda, k0, R);
        CESX_HIP(hipGetLastError());
        for (int j0 = 0; j0 < k0; j0 += 256) {
            // R -= X[:, j0 .. j0+255] * L[k0 .. k0+nb-1, j0 .. j0+255]^T
            hipLaunchKernelGGL(gemm_kernel, dim3((nb + 31) / 32, (nr + 31) / 32), dim3(DT), 0, s, nr, nb, 256, -1.0,
                               X + j0, (long long)ldx, 1LL, L + (size_t)k0 * ldl + j0, 1LL, (long long)ldl, R, nb, R);
            CESX_HIP(hipGetLastError());
        }
        if ((rc = trsm_reg(e, s, nr, nb, R, nb, L + (size_t)k0 * ldl + k0, ldl, X + k0, ldx))) return rc;
    }
    return CESX_OK;
}

// ---------------------------------------------------------------------------
// The hk-dependent SPD inverses of K2 -- (Sigma + hk C)^{-1} of the EKS rule (ces/calibrate.py:443), (hk C_gg + Gamma)^{-1} of the
// recomputed gain (:440-441 / :472-473) -- sit on the step's critical path with nothing to hide behind: hk comes from the
// Frobenius term of the COMPLETE Gram.  Factored from scratch they are a 97-us column-sequential Cholesky + a 77-us triangular
// inverse + a product (n = 256).  Inside a run the matrix changes little from step to step, so the previous step's inverse X is
// a good start for Newton-Schulz:   R = I - A X,   X <- X + X R,   R <- R R   (the residual squares every sweep; both products of
// a sweep are independent: one launch).  Tried when ||R_0||_F < 4 (a sufficient condition is rho(R_0) < 1; the Frobenius norm of
// an n x n residual whose every direction is off by 10 % is 0.1 sqrt(n)): five sweeps when the PREVIOUS step's start was close
// (||R_0||_F < 0.3: 0.3^32 = 2e-17), seven otherwise (rho = 0.7: 0.7^128 = 1e-20).  Then the TRUE residual I - A X is formed and
// checked (||.||_F < 1e-10): only then the factorisation chain is skipped (its kernels read one word and return).  Anything else
// -- the first step, an ensemble that moved a lot (the sweeps diverge: the check fails), an ill-conditioned A whose residual floor
// is higher, a NaN -- takes the factorisation, as before.  80 - 100 us instead of ~180 at n = 256; every product on
// v_mfma_f64_16x16x4_f64.
// One 16 x 16 block of the output per workgroup, K split over its four waves (gemm_splitk_kernel's structure).
// ---------------------------------------------------------------------------
__device__ __forceinline__ gemm_d4_t ns_block(int n, const double* __restrict__ A, const double* __restrict__ B, int r0, int c0,
                                              double (*part)[4][64]) {
    // C(r0.., c0..) = A (n x n, row-major) . B (n x n, row-major); valid in wave 0 (C/D map: col = lane & 15, row = (lane >> 4) + 4 reg)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = r0 + (lane & 15), j = c0 + (lane & 15), kk = lane >> 4;
    const bool iok = i < n, jok = j < n;
    const double* pa = A + (size_t)(iok ? i : 0) * n;
    const double* pb = B + (jok ? j : 0);
    const int kper = ((n + 3) / 4 + 3) / 4 * 4;
    const int kbeg = wave * kper, kend = kbeg + kper < n ? kbeg + kper : n;
    constexpr int UN = 8;
    gemm_d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double av[2][UN], bv[2][UN];
    auto load = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = k0 + 4 * u + kk;
            const bool kok = kc < kend;
            av[buf][u] = (iok && kok) ? pa[kc] : 0.0;
            bv[buf][u] = (jok && kok) ? pb[(size_t)kc * n] : 0.0;
        }
    };
    if (kbeg < kend) load(0, kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 4 * UN) {
        if (k0 + 4 * UN < kend) {
            if (buf == 0) load(1, k0 + 4 * UN); else load(0, k0 + 4 * UN);
        }
        if (buf == 0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
        }
        buf ^= 1;
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane];
    }
    return acc;
}

constexpr double NS_START2 = 16.0;      // ||I - A X_prev||_F^2 below which the warm start is tried
constexpr double NS_DONE2 = 1e-20;      // ||I - A X||_F^2 below which the result is accepted

// R = I - A X, parts[workgroup] = sum of squares of its block of R.  gate != nullptr: the closing residual -- formed only when
// sum(gate[0..npart)) < NS_START2 (the warm start was tried), and the LAST workgroup to arrive writes the verdict:
// skip[0] = 1 -- the warm start was taken AND its true residual passed: the factorisation chain behind has nothing to do;
// sc->spare[3] = ||R_0||^2, published with the step's result (the host sizes the next step's sweeps with it; accum: the
// larger of the step's two inverses).  skip[1] is the arrival counter (zero between launches).
__global__ __launch_bounds__(DT)
void ns_resid_kernel(int n, const double* __restrict__ A, const double* __restrict__ X, double* __restrict__ R,
                     double* __restrict__ parts, const double* __restrict__ gate, int npart, int* __restrict__ skip,
                     Scalars* __restrict__ sc, int accum) {
    __shared__ double red[DT / 64];
    __shared__ double part[3][4][64];
    __shared__ int s_last;
    double g0 = 0.0;
    if (gate != nullptr) {
        g0 = spec_norm2(gate, npart, red);
        if (!(g0 < NS_START2)) {
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
                skip[0] = 0;
                sc->spare[3] = accum ? fmax(sc->spare[3], g0) : g0;
            }
            return;
        }
    }
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, lane = threadIdx.x & 63, kk = lane >> 4, j = c0 + (lane & 15);
    const gemm_d4_t acc = ns_block(n, A, X, r0, c0, part);
    double sq = 0.0;
    if ((threadIdx.x >> 6) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + kk + 4 * r;
            if (row < n && j < n) { const double v = (row == j ? 1.0 : 0.0) - acc[r]; R[(size_t)row * n + j] = v; sq += v * v; }
        }
    }
    sq = dblock_sum(sq, red);
    if (gate == nullptr) {
        if (threadIdx.x == 0) parts[blockIdx.y * gridDim.x + blockIdx.x] = sq;
        return;
    }
    // arrival: the partial goes out at agent scope (the last workgroup may sit on another XCD), then the ticket
    if (threadIdx.x == 0) {
        st_agent(parts + blockIdx.y * gridDim.x + blockIdx.x, sq);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(skip) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x * gridDim.y - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    double rf = 0.0;
    for (int i = threadIdx.x; i < npart; i += DT) rf += ld_agent(parts + i);
    rf = dblock_sum(rf, red);          // (fixed order: the same verdict whichever workgroup arrives last)
    if (threadIdx.x == 0) {
        skip[0] = rf < NS_DONE2 ? 1 : 0;
        sc->spare[3] = accum ? fmax(sc->spare[3], g0) : g0;
        __hip_atomic_store(reinterpret_cast<unsigned*>(skip) + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one sweep, both products in one launch: blockIdx.z == 0: Xn = X + X R;  1: Rn = R R
__global__ __launch_bounds__(DT)
void ns_sweep_kernel(int n, const double* __restrict__ X, const double* __restrict__ R, double* __restrict__ Xn,
                     double* __restrict__ Rn, const double* __restrict__ gate, int npart) {
    __shared__ double red[DT / 64];
    __shared__ double part[3][4][64];
    if (!(spec_norm2(gate, npart, red) < NS_START2)) return;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, lane = threadIdx.x & 63, kk = lane >> 4, j = c0 + (lane & 15);
    const bool isx = blockIdx.z == 0;
    const gemm_d4_t acc = ns_block(n, isx ? X : R, R, r0, c0, part);
    if ((threadIdx.x >> 6) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + kk + 4 * r;
            if (row < n && j < n) {
                const size_t o = (size_t)row * n + j;
                if (isx) Xn[o] = X[o] + acc[r]; else Rn[o] = acc[r];
            }
        }
    }
}

// rows of X scaled by a diagonal: P = diag(d) X  (the EKS rule's Sigma (Sigma + hk C)^{-1} with a diagonal prior covariance)
__global__ void scale_rows_kernel(int n, const double* __restrict__ D, int ldd, const double* __restrict__ X, double* __restrict__ P) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    P[idx] = D[(size_t)(idx / n) * ldd] * X[idx];
}

// *Ainv = A^{-1} for SPD A (n x n).  which: 0 the recomputed gain's matrix, 1 the EKS rule's.  Each keeps two buffers: the
// previous step's inverse (the warm start) and this step's result, which is the next step's start -- no copy.
// Cold (or not converged): t1 (chol), t2 (X = L^{-T}),  A^{-1} = L^{-T} L^{-1} = X X^T.
static int ns_sweeps(double r0sq) {
    // the residual squares every sweep: r^(2^s) <= 1e-11 with r = 1.5 ||R_0||_F of the last step (the ensemble keeps moving)
    const double r = 1.5 * std::sqrt(r0sq);
    if (!(r < 0.9)) return 7;
    if (!(r > 1e-6)) return 2;
    const int s = (int)std::ceil(std::log2(std::log(1e-11) / std::log(r)));
    return s < 2 ? 2 : s > 7 ? 7 : s;
}

static int spd_inverse(Engine& e, hipStream_t s, int n, const double* A, const double** Ainv, int which, bool accum) {
    int rc;
    const int np = potrf_ld(n);
    const bool keep = np <= 256 && e.d_ns_x[which][0] != nullptr;          // (larger: blocked factorisation, no warm start)
    const bool warm = keep && e.ns_ok;
    double* out = keep ? e.d_ns_x[which][e.ns_cur[which] ^ 1] : e.d_t4;
    if (warm) {
        const int nb = (n + 15) / 16, npart = nb * nb;
        const double* Xp = e.d_ns_x[which][e.ns_cur[which]];
        double* parts0 = e.d_ns_parts, *partsf = e.d_ns_parts + npart;
        double* Rb[2] = {e.d_ns_r[0], e.d_ns_r[1]};
        double* Xb[2] = {out, e.d_ns_r[2]};
        hipLaunchKernelGGL(ns_resid_kernel, dim3(nb, nb), dim3(DT), 0, s, n, A, Xp, Rb[0], parts0, (const double*)nullptr, npart,
                           (int*)nullptr, (Scalars*)nullptr, 0);
        const double* Xc = Xp;
        const int sweeps = ns_sweeps(e.ns_r0_last);
        for (int it = 0; it < sweeps; ++it) {          // X_sweeps lands in out
            double* Xn = Xb[(sweeps - 1 - it) & 1];
            hipLaunchKernelGGL(ns_sweep_kernel, dim3(nb, nb, 2), dim3(DT), 0, s, n, Xc, (const double*)Rb[it & 1], Xn, Rb[(it & 1) ^ 1],
                               (const double*)parts0, npart);
            Xc = Xn;
        }
        hipLaunchKernelGGL(ns_resid_kernel, dim3(nb, nb), dim3(DT), 0, s, n, A, (const double*)out, Rb[0], partsf, (const double*)parts0, npart,
                           e.d_ns_skip, e.d_scal, accum ? 1 : 0);
        CESX_HIP(hipGetLastError());
        e.gate = e.d_ns_skip;
    }
    rc = potrf(e, s, n, A, e.d_t1);
    if (rc == CESX_OK) rc = trsm_right_lt(e, s, n, np, nullptr, 0, e.d_t1, np, e.d_t2, np);
    // X is upper triangular: X[i][k] = 0 for k < i; columns >= n of the rows < n are zero
    // (K split over the waves of a workgroup for the sizes of K2: 21 -> 7 us at n = 256)
    if (rc == CESX_OK) rc = gemm(e, s, n, n, n, 1.0, e.d_t2, (long long)np, 1LL, e.d_t2, 1LL, (long long)np, out);
    e.gate = nullptr;
    if (rc != CESX_OK) return rc;
    if (keep) e.ns_cur[which] ^= 1;
    *Ainv = out;
    return CESX_OK;
}

template <typename T>
static int assemble(Engine& e, hipStream_t s, int mode, int ktot, double sw) {
    const int mx = e.p > e.n ? e.p : e.n;
    const long long len = (long long)e.rpad * ktot;
    hipLaunchKernelGGL(assemble_kernel<T>, g1(len), dim3(256), 0, s, mode, e.p, e.n, e.kp, e.kn, e.rpad, ktot,
                       sw, e.d_scal, e.d_M, e.d_K, e.d_L, potrf_ld(e.p), e.d_P, e.d_PK, e.d_mv, mx, e.d_ubar, e.d_gbar,
                       e.d_y, (const double*)e.d_gw, (T*)e.d_W, (T*)e.d_bias,
                       (T*)e.d_shiftT, e.d_shift64, (T*)e.d_rowc,
                       (float*)e.d_Wf);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

// the fp64 factor of the last step's covariance, for callers that read Engine::d_L (cesx_debug_dense): the chained factorisation
// keeps it in the image only
int refresh_factor(Engine& e, hipStream_t s) {
    if (!e.L_stale) return CESX_OK;
    return potrf(e, s, e.p, e.d_C, e.d_L);
}

// phase 0: everything for eks / aldi.  phase 1: aldi_constant drift coefficients.
// phase 2: aldi_constant noise coefficients after hk is known.
bool stream_below_side(Engine& e, hipStream_t s) {
    if (!e.side_has_prio || s == e.side) return false;
    // asked of the runtime at every cesx_apply (a cheap host call): a verdict cached by stream HANDLE would be inherited by
    // another stream created at the same address with another priority (torch's stream pools, tests)
    int pr = 0;
    return hipStreamGetPriority(s, &pr) == hipSuccess && pr > e.side_prio;
}

int launch_dense(Engine& e, const cesx_step_params& prm, const double* mom, int phase, hipStream_t s, bool upd2_ok) {
    const int p = e.p, n = e.n, mx = p > n ? p : n;
    const bool f32 = e.cfg.dtype == CESX_F32;
    int rc;
    if (phase == 2) {
        hipLaunchKernelGGL(constant_hk_kernel, dim3(1), dim3(1), 0, s, prm, e.d_absmax, e.d_scal);
        CESX_HIP(hipGetLastError());
        return f32 ? assemble<float>(e, s, 3, e.kp, 0.0) : assemble<double>(e, s, 3, e.kp, 0.0);
    }
    MomView mv{p, n, mom};
    const int unbiased = prm.update == CESX_UPDATE_EKS ? 0 : 1;
    // If cesx_chol_async already ran for these moments, the U-only part of K2 (C, M, ubar, chol(C))
    // is done or in flight on the side stream; otherwise do it here, in line.
    // The side stream (U-only centring, chol(C), the prefetched noise block) is joined ONCE, as late as the
    // data flow allows: the G part of the centring needs the moments only; the scalar kernel is the first to
    // read what the side stream wrote (trace / bias partials, later L).  One event each way per step -- every
    // record / wait pair costs ~6 us of idle GPU.
    const bool early = e.chol_inflight;
    const bool fused_finish = phase == 0 && prm.update == CESX_UPDATE_ALDI &&
        (prm.time_step == CESX_TS_DEFAULT || prm.time_step == CESX_TS_SPECTRAL);
    // (early, centring fused into the Cholesky's load: the U part is done HERE, with the G part, and leaves the
    //  status word alone -- the side stream carried nothing but the factorisation)
    const int what = !early ? 3 : e.chol_fused_center ? (3 | 4) : 2;
    // hk kept out of the coefficient matrix (cesx_internal.h, Engine::d_Wq): the side stream wrote L, a I - M, M mu, M ubar
    // for this factorisation, ONE launch adds the rest and the update kernel takes hk at run time
    const bool img_ok = e.hkfree_ok && e.d_Wq != nullptr && f32 && e.update_v2 && potrf_ld(p) <= 256;
    const bool hkfree = upd2_ok && fused_finish && prm.time_step == CESX_TS_DEFAULT && img_ok && (early ? e.side_img : true);
    e.last_hkfree = false;
    if (hkfree) {
        // the side stream is joined by the LAST workgroup of that launch (a polled word, under the conditions of the polled
        // join below), else by the event in front of it; no factorisation in flight: the U part runs here, in line
        const bool polled = early && e.poll_join_ok && e.chol_signals && e.J == e.Jg && s != e.side && stream_below_side(e, s);
        // (no factorisation in flight: in line, the same kernels the side stream would have run -- with CESX_FUSE_CENTER=1 the
        //  factorisation forms C while it loads S_aa and the tail launch forms the rest of the U part itself)
        const int self_u = (early ? e.chol_fused_center : (e.fuse_center_ok || (e.fuse_center_auto && e.gram_b_short))) ? 1 : 0;
        e.skip_L_hint = true;          // (an in-line factorisation of a step that IS hk-free)
        if (!early && self_u) {
            PotrfCen cen{mv.mom + e.ml.sa(), mv.mom, unbiased};
            if ((rc = potrf_reg_any(e, s, p, potrf_ld(p), mv.mom + e.ml.Saa(), e.d_L, p, 0, nullptr, cen, nullptr, 0, (float*)e.d_Wq))) return rc;
        } else if (!early) {
            hipLaunchKernelGGL(center_kernel, dim3(NPB), dim3(DT), 0, s, mv, e.d_shift64, e.d_y, e.d_ustar,
                               (const double*)e.d_gw, e.diag_sigma ? (const double*)e.d_sw : (const double*)nullptr, unbiased, 1, e.d_ubar, e.d_gbar,
                               e.d_m, e.d_dg, e.d_C, e.d_Cug, e.d_See, e.d_Srr, e.d_K, e.d_M, e.d_part, e.d_scal, (double*)nullptr);
            CESX_HIP(hipGetLastError());
            if ((rc = potrf(e, s, p, e.d_C, e.d_L, nullptr, nullptr, 0, (float*)e.d_Wq))) return rc;
        } else if (!polled) CESX_HIP(hipStreamWaitEvent(s, e.ev_b, 0));
        auto tail_kern = !e.diag_sigma ? tail_aldi_kernel<true, false> : e.chain ? tail_aldi_kernel<false, true> : tail_aldi_kernel<false, false>;
        hipLaunchKernelGGL(tail_kern, dim3(NPB), dim3(DT), 0, s, mv, prm, (const double*)e.d_shift64, (const double*)e.d_y,
                           (const double*)e.d_gw, e.d_gbar, e.d_m, e.d_dg, e.d_Cug, e.d_See, e.d_Srr, e.d_K, e.d_part, e.d_scal,
                           e.d_lag, e.d_mv, mx, (const double*)e.d_sw, e.diag_sigma ? (const double*)nullptr : (const double*)e.d_Sinv,
                           (const double*)e.d_mu, self_u, (const double*)e.d_ustar,
                           e.d_ubar, e.d_C, e.d_M, (float*)e.d_Wq, e.ktot / 16, e.kp,
                           e.kn, (float*)e.d_bias, (float*)e.d_shiftT, e.d_shift64, (float*)e.d_rowc,
                           e.d_ticket, polled ? (const unsigned long long*)e.d_cholflag : (const unsigned long long*)nullptr,
                           (unsigned long long)e.chol_seq, e.d_cholflag + 1, e.poll_ticks);
        CESX_HIP(hipGetLastError());
        e.last_join_polled = polled;
        if (early) {
            e.evb_waited_seq = e.chol_seq;
            e.evb_waited_stream = s;
            e.chol_inflight = false;
        }
        e.last_hkfree = true;
        return CESX_OK;
    }
    // The side stream joined WITHOUT a barrier packet (6-8 us of the caller's stream even when the event completed long
    // before): workgroup 0 of the G-part centring launch -- the launch in front of the assembly launch -- ends only when
    // chol(C) has stored its sequence number, and the assembly launch reads what that stream wrote with agent-scope
    // loads (no queue-level acquire stands between that stream's kernels and it).  Only where nothing else sits between the two and reads those results (ALDI, default time
    // step, diagonal Gamma / Sigma, one device), and only for the one-kernel factorisation that signals.
    const bool can_poll = fused_finish && prm.time_step == CESX_TS_DEFAULT && early && !e.chol_fused_center && e.poll_join_ok &&
        e.chol_signals && e.diag_sigma && e.J == e.Jg && s != e.side && stream_below_side(e, s);
    const bool polled = can_poll && !(early && e.L_stale);      // (a factor that must be re-formed in line: joined with the event)
    hipLaunchKernelGGL(center_kernel, dim3(NPB), dim3(DT), 0, s, mv, e.d_shift64, e.d_y, e.d_ustar,
                       (const double*)e.d_gw,
                       e.diag_sigma ? e.d_sw : (const double*)nullptr, unbiased, what, e.d_ubar, e.d_gbar,
                       e.d_m, e.d_dg, e.d_C, e.d_Cug, e.d_See, e.d_Srr, e.d_K, e.d_M, e.d_part, e.d_scal, e.d_lag,
                       polled ? (const unsigned long long*)e.d_cholflag : (const unsigned long long*)nullptr,
                       (unsigned long long)e.chol_seq, e.d_cholflag + 1, e.poll_ticks);
    CESX_HIP(hipGetLastError());
    e.last_join_polled = polled;
    if (!early)
        if ((rc = potrf(e, s, p, e.d_C, e.d_L))) return rc;
    if (early) {
        if (!polled) CESX_HIP(hipStreamWaitEvent(s, e.ev_b, 0));
        // the factorisation in flight expected a chained step and kept L in the image only (the time-step rule changed, or the
        // ensembles of this call do not qualify): factor C again, in line
        if (e.L_stale && (rc = refresh_factor(e, s))) return rc;
        // (polled: no queue-level wait was issued, but the launches behind are ordered behind chol(C) all the same -- the
        //  poll ended on its word, or it ran out and they leave the step untouched (the update launch checks the same
        //  fault word).  A noise block drawn BEFORE this chol(C) on the side stream is therefore complete: take_noise
        //  relies on exactly that, as it does behind the event.)
        e.evb_waited_seq = e.chol_seq;
        e.evb_waited_stream = s;
        e.chol_inflight = false;
    }
    if (!e.diag_sigma)
        if ((rc = gemm(e, s, p, p, p, 1.0, e.d_C, p, 1, e.d_Sinv, p, 1, e.d_M))) return rc;
    if (prm.time_step == CESX_TS_SPECTRAL && prm.update != CESX_UPDATE_ALDI_CONSTANT) {
        // B = Gamma^{-1/2} See Gamma^{-1/2} (Gamma diagonal, or whitened away), symmetric PSD, same non-zero spectrum as D (times N)
        hipLaunchKernelGGL(whiten_diag_kernel, g1((long long)n * n), dim3(256), 0, s, n, e.d_See, e.d_gw, e.d_t3);
        // lambda_max(B) by repeated squaring (spec_square_kernel above): B in d_t3, the squares alternate between d_t1 and d_t2
        const int nb16 = (n + 15) / 16, npart = nb16 * nb16;
        double* acc = e.d_spec;                          // {sum of 2^-k log N_k, 2^-k, degenerate flag}
        double* parts[2] = {e.d_spec + 4, e.d_spec + 4 + npart};
        hipLaunchKernelGGL(spec_begin_kernel, dim3(npart), dim3(DT), 0, s, (long long)n * n, (const double*)e.d_t3, parts[0], npart, acc);
        const double* src = e.d_t3;
        double* dst = e.d_t1;
        for (int q = 0; q < SPEC_SQUARINGS; ++q) {
            hipLaunchKernelGGL(spec_square_kernel, dim3(nb16, nb16), dim3(DT), 0, s, n, src, (const double*)parts[q & 1], npart, dst,
                               parts[(q & 1) ^ 1], acc);
            src = dst;
            dst = dst == e.d_t1 ? e.d_t2 : e.d_t1;
        }
        hipLaunchKernelGGL(spec_end_kernel, dim3(1), dim3(DT), 0, s, n, (const double*)parts[SPEC_SQUARINGS & 1], npart, (const double*)acc,
                           mom, e.d_scal);
        CESX_HIP(hipGetLastError());
    }
    if (fused_finish) {
        const int nwb = (int)(((long long)e.rpad * e.ktot + DT - 1) / DT), nvb = (p + DT / 64 - 1) / (DT / 64);
        auto go = [&](auto tag, auto poll_tag) {
            using T = decltype(tag);
            constexpr bool POLLED = decltype(poll_tag)::value;
            hipLaunchKernelGGL((finish_aldi_kernel<T, POLLED>), dim3(nwb + nvb), dim3(DT), 0, s, mv, prm, (const double*)e.d_part, e.d_scal, nwb,
                               e.kp, e.kn, e.rpad, e.ktot, (const double*)e.d_M, (const double*)e.d_K, (const double*)e.d_L, potrf_ld(p),
                               (const double*)e.d_y, (const double*)e.d_gbar, (const double*)e.d_mu,
                               (const double*)e.d_ubar, (const double*)e.d_gw, mx, e.d_mv, (T*)e.d_W,
                               (T*)e.d_bias, (T*)e.d_shiftT, e.d_shift64, (T*)e.d_rowc, (float*)e.d_Wf,
                               (const unsigned long long*)(e.d_cholflag + 1), (unsigned long long)e.chol_seq);
        };
        auto pick = [&](auto tag) {
            if (polled) go(tag, std::true_type{});
            else go(tag, std::false_type{});
        };
        if (f32) pick(float{}); else pick(double{});
        CESX_HIP(hipGetLastError());
        return CESX_OK;
    }
    auto scalars_and_matvecs = [&]() {
        hipLaunchKernelGGL(scalar_kernel, dim3(1 + (4 * p + DT / 64 - 1) / (DT / 64)), dim3(DT), 0, s, mv, prm,
                           e.d_part, e.d_scal, e.d_K, e.d_M, e.d_y, e.d_gbar, e.d_mu, e.d_ubar, mx, e.d_mv);
    };
    scalars_and_matvecs();
    CESX_HIP(hipGetLastError());

    bool gain_inverse = false;
    if (phase == 0 && (prm.time_step == CESX_TS_CONSTANT || (prm.time_step == CESX_TS_MIX && prm.update == CESX_UPDATE_ALDI))) {
        // K' = C_ug (hk C_gg + Gamma)^{-1},  C_gg = See / N   (:440-441, :472-473)
        hipLaunchKernelGGL(axpb_kernel, g1((long long)n * n), dim3(256), 0, s, (long long)n * n, &e.d_scal->hk,
                           mom, e.d_See, e.d_Gamma, e.d_t3);
        CESX_HIP(hipGetLastError());
        const double* inv;
        if ((rc = spd_inverse(e, s, n, e.d_t3, &inv, 0, false))) return rc;
        gain_inverse = true;
        if ((rc = gemm(e, s, p, n, n, 1.0, e.d_Cug, n, 1, inv, n, 1, e.d_Kp))) return rc;
        hipLaunchKernelGGL(select_kernel, g1((long long)p * n), dim3(256), 0, s, (long long)p * n, e.d_scal, e.d_Kp, e.d_K);
        CESX_HIP(hipGetLastError());
        scalars_and_matvecs();          // K y and K gbar with the selected gain (scalars are recomputed identically)
        CESX_HIP(hipGetLastError());
    }

    int mode = 0;
    if (phase == 1) mode = 2;
    else if (prm.update == CESX_UPDATE_EKS) {
        mode = 1;
        // P = Sigma (Sigma + hk C)^{-1}  ( = (I + hk C Sigma^{-1})^{-1}, :443 )
        hipLaunchKernelGGL(axpb_kernel, g1((long long)p * p), dim3(256), 0, s, (long long)p * p, &e.d_scal->hk,
                           (const double*)nullptr, e.d_C, e.d_Sigma, e.d_t3);
        CESX_HIP(hipGetLastError());
        const double* inv;
        if ((rc = spd_inverse(e, s, p, e.d_t3, &inv, 1, gain_inverse))) return rc;
        if (e.diag_sigma) {
            hipLaunchKernelGGL(scale_rows_kernel, g1((long long)p * p), dim3(256), 0, s, p, (const double*)e.d_Sigma, p + 1, inv, e.d_P);
            CESX_HIP(hipGetLastError());
        } else if ((rc = gemm(e, s, p, p, p, 1.0, e.d_Sigma, p, 1, inv, p, 1, e.d_P))) return rc;
        if ((rc = gemm(e, s, p, n, p, 1.0, e.d_P, p, 1, e.d_K, n, 1, e.d_PK))) return rc;
        hipLaunchKernelGGL(hk_sum_kernel, g1(p), dim3(256), 0, s, p, e.d_scal, e.d_mv, e.d_mv + 2 * mx, e.d_mv + 5 * mx);
        hipLaunchKernelGGL(matvec_kernel, g1(p, 4), dim3(DT), 0, s, p, p, e.d_P, e.d_mv + 5 * mx, e.d_mv + 4 * mx);
        CESX_HIP(hipGetLastError());
    }
    const int ktot = mode == 2 ? e.kp + e.kn : e.ktot;
    return f32 ? assemble<float>(e, s, mode, ktot, prm.switch_mult) : assemble<double>(e, s, mode, ktot, prm.switch_mult);
}

// U-only part of K2, started as soon as the U x U part of the moments is complete (and, on
// several devices, all-reduced): C = S_uu / div + 1e-8 I, M, ubar, then chol(C) on the engine's
// side stream.  cesx_apply joins it right before W is assembled.
int launch_chol_async(Engine& e, int update, const double* mom, hipStream_t s, bool ev_a_bound) {
    const int p = e.p, n = e.n;
    MomView mv{p, n, mom};
    const int unbiased = update == CESX_UPDATE_EKS ? 0 : 1;
    // the whole U-only part of K2 (centre, C, M, then chol(C)) goes to the side stream: the main
    // stream continues with the second Gram launch straight after the U x U reduce
    if (s != e.side) {                             // (a caller already on the side stream needs no hand-over)
        if (!ev_a_bound) {          // (bound: ev_a is the U x U reduce kernel's own signal, the side stream waits for it already)
            CESX_HIP(hipEventRecord(e.ev_a, s));
            CESX_HIP(hipStreamWaitEvent(e.side, e.ev_a, 0));
        }
    }
    int rc;
    // (round 4: also measured at C4 -- p = 64, where the separate centring launch is a third of the side chain: 0.0844
    //  against 0.0805 ms/step with it fused, the one workgroup's load phase is the longer way; stays opt-in)
    // the hk-free update's share of the side stream (cesx_internal.h, Engine::d_Wq): whether the step takes that path is
    // decided in cesx_apply (time-step rule, alignment of the ensembles); storing L into the image costs the factorisation ~1 us
    const bool img = e.hkfree_ok && e.d_Wq != nullptr && e.cfg.dtype == CESX_F32 && update == CESX_UPDATE_ALDI && e.update_v2 &&
        potrf_ld(p) <= 256;
    e.chol_fused_center = potrf_ld(p) <= 256 && (e.fuse_center_ok || (e.fuse_center_auto && img && e.gram_b_short));
    e.side_img = false;
    e.skip_L_hint = e.last_hkfree;      // (whether THIS step is hk-free is decided in cesx_apply: expect what the last one was)
    if (e.chol_fused_center) {
        e.side_img = img;
        // p <= 256 (one register-resident factorisation): the covariance is formed while the kernel loads the raw
        // second moments -- no centring launch (13 us + a kernel boundary) in front of the 100-us Cholesky, which is
        // what the caller's stream ends up waiting for; cesx_apply's own centring launch does the U part with the G
        // part (C, M, ubar, the trace / bias partials: nothing the factorisation needs)
        const int np = potrf_ld(p);
        PotrfCen cen{mv.mom + e.ml.sa(), mv.mom, unbiased};
        float* wq = img ? (float*)e.d_Wq : (float*)nullptr;
        // (CESX_TEST_DROP_CHOL_SIGNAL=k, tests only: the k-th factorisation does not store its word)
        unsigned long long* flag = e.test_drop_signal_at == e.chol_seq + 1 ? nullptr : e.d_cholflag;
        if ((rc = potrf_reg_any(e, e.side, p, np, mv.mom + e.ml.Saa(), e.d_L, p, 0, e.ev_b, cen, flag, e.chol_seq + 1, wq))) return rc;
    } else {
    e.side_img = img;
    float* wq = e.side_img ? (float*)e.d_Wq : (float*)nullptr;
    // (few workgroups -> 1024 threads each: 8 x 256 threads took 25 us for the 65 k elements of C, latency bound)
    hipLaunchKernelGGL(center_kernel, dim3(std::min(NPB, e.center_u_wgs)), dim3(e.center_u_wgs < NPB ? 1024 : DT), 0, e.side, mv, e.d_shift64, e.d_y, e.d_ustar,
                       (const double*)e.d_gw,
                       e.diag_sigma ? e.d_sw : (const double*)nullptr, unbiased, 1, e.d_ubar, e.d_gbar,
                       e.d_m, e.d_dg, e.d_C, e.d_Cug, e.d_See, e.d_Srr, e.d_K, e.d_M, e.d_part, e.d_scal, (double*)nullptr);
    CESX_HIP(hipGetLastError());
    // (CESX_TEST_DROP_CHOL_SIGNAL=k, tests only: the k-th factorisation does not store its word -- the polled join of that step runs out)
    unsigned long long* flag = e.test_drop_signal_at == e.chol_seq + 1 ? nullptr : e.d_cholflag;
    if ((rc = potrf(e, e.side, p, e.d_C, e.d_L, e.ev_b, flag, e.chol_seq + 1, wq))) return rc;      // ev_b: C, M, ubar, L -- what K2's scalar and assemble kernels read
    }
    e.chol_signals = true;      // (the one-kernel factorisation, or the last diagonal block of the blocked one, stores the word)
    ++e.chol_seq;
    if (e.xi_want >= 0 && e.d_xi[0]) {
        // noise blocks asked for by cesx_prefetch_noise (cesx_internal.h): this step's, unless the lookahead of an
        // earlier step drew it, then the next step's.  Nothing but the update kernel that reads a block waits for it.
        auto draw = [&](long long step, int b) -> int {
            if ((rc = launch_noise(e, (uint64_t)step, e.d_xi[b], e.side))) return rc;
            CESX_HIP(hipEventRecord(e.ev_x[b], e.side));
            e.xi_step[b] = step;
            e.xi_seq[b] = e.chol_seq;
            return CESX_OK;
        };
        int have = e.xi_step[0] == e.xi_want ? 0 : (e.d_xi[1] && e.xi_step[1] == e.xi_want) ? 1 : -1;
        if (have < 0) {
            have = 0;
            if ((rc = draw(e.xi_want, 0))) return rc;
        }
        if (e.xi_lookahead && e.d_xi[1] && e.xi_step[have ^ 1] != e.xi_want + 1)
            if ((rc = draw(e.xi_want + 1, have ^ 1))) return rc;
        e.xi_want = -1;
    }
    e.chol_inflight = true;
    return CESX_OK;
}

}  // namespace cesx
