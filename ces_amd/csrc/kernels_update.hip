// K3 -- fused drift + diffusion update of the ensemble shard as ONE GEMM
// (replaces ces/calibrate.py:443-447, :484-488, :515-527):
//
//     U_next = W . [U ; G ; xi] + b 1^T          W in R^{p x (2p+n)}
//
// with, for ALDI,  W = [ (1 + hk a_J) I - hk C Sigma^{-1} | -hk K | sqrt(2 hk) L ],
// K = C_ug Gamma^{-1}, L = chol(C), b = hk (K y + C Sigma^{-1} mu - a_J ubar)
// (W and b are assembled by K2 in kernels_dense.hip).  The
// J x J product (U0 - Umean) @ D of :484 never appears.
//
// One workgroup = 4 waves = all (up to 256) output rows x BN particles; the
// K-loop walks the stacked rows of [U; G; xi] in 16-row tiles, staged through
// LDS (register-staged double buffer).  xi tiles are either read from memory
// (parity runs inject the reference's np.random.normal block, :447/:488/:527)
// or drawn in-kernel with Philox4x32-10 + Box-Muller keyed by the GLOBAL
// particle index, so a particle's noise does not depend on how the ensemble
// is sharded.  MFMA: v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64.
// The same kernel evaluates the linear forward map G = A U + b
// (ces/utils.py:25-31) with a single K-segment.
// Bound: MFMA.
#include "cesx_internal.h"
#include <cstdlib>

namespace cesx {

constexpr int UPD_THREADS = 256;
constexpr int BK = 16;
#ifndef UPD_ABL      // timing ablations of tools/update_bench.hip (results are wrong when set)
#define UPD_ABL 0
#endif

template <typename T> struct UpdCfg;
template <> struct UpdCfg<float> {
    static constexpr int WR = 2, WC = 4;          // 32x32 blocks per wave: 64 rows x 128 particles
    static constexpr int STRIDE_W = BK + 4;       // floats; conflict-free ds_read_b128
    static constexpr int XPAD = 0;
};
template <> struct UpdCfg<double> {
    static constexpr int WR = 4, WC = 4;          // 16x16 blocks per wave: 64 rows x 64 particles
    static constexpr int STRIDE_W = BK + 2;
    static constexpr int XPAD = 8;
};

template <typename T>
struct UpdArgs {
    const T* W; int ktot; int ldw; const T* bias; int out_rows;   // W: out_rows x ktot window of a row-major matrix with row stride ldw
    const T* src[3]; int src_rows[3]; int src_k0[3]; int src_kind[3]; int nsrc;
    long long J, j_offset;
    T* out;
    const T* add1; const double* c1p; double c1i;
    const T* add2; const double* c2p; double c2i;
    double* absmax_part;
    unsigned int seed_lo, seed_hi, step;
    // data metrics (ces/calibrate.py:434-435 / :466-467) accumulated while the G rows stream by:
    // rowc[i] = {gbar_i, y_i, 1/Gamma_ii, 0}; metric_part[block] = {sum q_r^2, sum q_e^2}
    const T* rowc; double* metric_part; int metric_seg;
    int tri_seg;      // K-segment whose W columns are lower triangular (sqrt(2hk) L), -1 if none
    const unsigned long long* fault; unsigned long long fault_seq;   // fault != nullptr and *fault == fault_seq: leave `out` untouched (UpdateOpt)
};

template <typename T, bool ALIGNED, int WCT>
__global__ __launch_bounds__(UPD_THREADS, sizeof(T) == 4 ? 2 : 1)
void update_kernel(const UpdArgs<T> a) {
    // a polled join of the side stream that ran out in front of this launch (kernels_dense.hip): W is stale, the output stays as it was
    if (a.fault != nullptr && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.fault_seq) return;
    using M = Mfma<T>;
    using vec_t = typename M::vec_t;
    using acc_t = typename M::acc_t;
    using C = UpdCfg<T>;
    constexpr int TILE = M::TILE, VEC = M::VEC, WR = C::WR, WC = WCT;   // WCT blocks of particles per wave
    constexpr int RC = 4 * WR * TILE;                 // output rows per workgroup (256)
    constexpr int BN = WC * TILE;                     // particles per workgroup
    constexpr int SW = C::STRIDE_W, SX = BN + C::XPAD;
    constexpr int WCH = RC * BK / VEC / UPD_THREADS;  // W chunks per thread
    constexpr int XCH = BK * BN / VEC / UPD_THREADS;  // X chunks per thread (2)
    constexpr int CPR = BN / VEC;                     // X chunks per row (32)
    constexpr int WPR = BK / VEC;                     // W chunks per row
    constexpr int NQ = 4 * BN / UPD_THREADS;          // philox items per thread

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T (*sW)[RC * SW] = reinterpret_cast<T (*)[RC * SW]>(smem);
    T (*sX)[BK * SX] = reinterpret_cast<T (*)[BK * SX]>(smem + 2 * RC * SW * sizeof(T));
    double* red = reinterpret_cast<double*>(smem + 2 * (RC * SW + BK * SX) * sizeof(T));
    // per-row constants of the data metrics {gbar, y, 1/Gamma_ii, 0}, staged once: a global
    // load inside the K loop would make its vmcnt wait drain the tile prefetch as well
    T* sRowc = reinterpret_cast<T*>(smem + 2 * (RC * SW + BK * SX) * sizeof(T) + 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long jt0 = (long long)blockIdx.x * BN;
    const int rc0 = blockIdx.y * RC;
    // Row blocks of this wave.  Blocks are dealt in mirrored pairs (w, 2*NW-1-w) so that every
    // wave does the same amount of work in the lower-triangular noise segment, where block rb
    // only needs the k-tiles with k <= its last row.
    constexpr int NW = UPD_THREADS / 64;
    int rbk[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r)
        rbk[r] = (r / 2) * 2 * NW + ((r & 1) ? 2 * NW - 1 - wave : wave);
    if (WR == 1) rbk[0] = wave;
    const int nkt = a.ktot / BK;

    acc_t acc[WR][WC];
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int c = 0; c < WC; ++c)
#pragma unroll
            for (int e = 0; e < M::NACC; ++e) acc[r][c][e] = 0;

    vec_t wst[WCH], xst[XCH];
    int xkind = 0, xq0 = 0;
    T mq_e = 0, mq_r = 0;          // per-particle quadratic forms (partial over this thread's rows)
    const bool do_metrics = a.metric_part != nullptr && blockIdx.y == 0;
    if (do_metrics) {
        const int sg = a.metric_seg;
        const int kend = sg + 1 < a.nsrc ? a.src_k0[sg + 1] : a.ktot;
        for (int i = tid; i < (kend - a.src_k0[sg]) * 4; i += UPD_THREADS) sRowc[i] = a.rowc[i];
    }

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        // W tile: rows rc0 .. rc0+RC-1, columns k0 .. k0+15 (W is zero padded)
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + UPD_THREADS * i;
            const int row = c / WPR, part = c % WPR;
            wst[i] = *reinterpret_cast<const vec_t*>(a.W + (size_t)(rc0 + row) * a.ldw + k0 + part * VEC);
        }
        // which K-segment does this tile belong to?
        int s = 0;
#pragma unroll
        for (int q = 1; q < 3; ++q)
            if (q < a.nsrc && k0 >= a.src_k0[q]) s = q;
        xkind = a.src_kind[s];
        if (xkind == 0) {
            const int r0 = k0 - a.src_k0[s];
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int c = tid + UPD_THREADS * i;
                const int row = r0 + c / CPR;
                const long long j = jt0 + (c % CPR) * VEC;
                vec_t v;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = 0;
                if (row < a.src_rows[s]) {
                    const T* ptr = a.src[s] + (size_t)row * a.J;
                    if (ALIGNED && j + VEC <= a.J) {
                        v = *reinterpret_cast<const vec_t*>(ptr + j);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (j + e < a.J) v[e] = ptr[j + e];
                    }
                }
                xst[i] = v;
            }
        } else {
            xq0 = (k0 - a.src_k0[s]) / 4;      // first row quad of this noise tile
        }
    };
    // half = 0 / 1: the staged tile is written in two halves, one after each MFMA group of the
    // current tile (so that staging and noise generation interleave with the matrix pipe);
    // half = -1 writes everything
    auto store_tile = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            if (half >= 0 && (i * 2 / WCH) != half) continue;
            const int c = tid + UPD_THREADS * i;
            const int row = c / WPR, part = c % WPR;
            *reinterpret_cast<vec_t*>(&sW[buf][row * SW + part * VEC]) = wst[i];
        }
        if (xkind == 0) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                if (half >= 0 && (XCH > 1 ? (i * 2 / XCH) : 0) != half) continue;
                const int c = tid + UPD_THREADS * i;
                *reinterpret_cast<vec_t*>(&sX[buf][(c / CPR) * SX + (c % CPR) * VEC]) = xst[i];
            }
        } else {
            const unsigned q0 = (unsigned)xq0;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (half >= 0 && (NQ > 1 ? (i * 2 / NQ) : 1) != half) continue;
                const int item = tid + UPD_THREADS * i;
                const int jl = item % BN, ql = item / BN;
                const unsigned long long gj = (unsigned long long)(a.j_offset + jt0 + jl);
                const uint4x r = philox4x32_10((uint32_t)gj, (uint32_t)(gj >> 32), q0 + ql, a.step,
                                               a.seed_lo, a.seed_hi);
                T z[4];
                normal4(r, z);
#pragma unroll
                for (int e = 0; e < 4; ++e) sX[buf][(4 * ql + e) * SX + jl] = z[e];
            }
        }
    };

    // skip row blocks that are entirely padding (small p)
    bool rb_on[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r) rb_on[r] = (rc0 + rbk[r] * TILE) < a.out_rows;
    bool any_on = false;
#pragma unroll
    for (int r = 0; r < WR; ++r) any_on = any_on || rb_on[r];

    const int li = lane % TILE, lh = lane / TILE;

    load_tile(0);
    store_tile(0, -1);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (!(UPD_ABL & 1) && kt + 1 < nkt) load_tile(kt + 1);
        if (do_metrics) {
            // is the tile in sX[cur] a tile of G rows?
            const int k0 = kt * BK;
            const int sg = a.metric_seg;
            const int kend = sg + 1 < a.nsrc ? a.src_k0[sg + 1] : a.ktot;
            if (k0 >= a.src_k0[sg] && k0 < kend) {
                const int r0 = k0 - a.src_k0[sg];
                const int jl = tid % BN, grp = tid / BN;           // grp in [0, 256/BN)
                constexpr int RPT = BK / (UPD_THREADS / BN);       // rows per thread per tile
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int rr = grp * RPT + q;
                    const T x = sX[cur][rr * SX + jl];
                    const T* rc = sRowc + (size_t)(r0 + rr) * 4;
                    const T b = x - rc[0], r = x - rc[1], w = rc[2];
                    mq_e += w * b * b;
                    mq_r += w * r * r;
                }
            }
        }
        // lower-triangular segment: a row block needs this k-tile only if its last row >= k0
        bool need[WR];
        {
            const int k0 = kt * BK;
            const int ts = a.tri_seg;
            const int tend = ts >= 0 ? (ts + 1 < a.nsrc ? a.src_k0[ts + 1] : a.ktot) : 0;
            const bool intri = ts >= 0 && k0 >= a.src_k0[ts] && k0 < tend;
            const int k0l = intri ? k0 - a.src_k0[ts] : 0;
#pragma unroll
            for (int r = 0; r < WR; ++r) need[r] = rb_on[r] && (!intri || rc0 + rbk[r] * TILE + TILE - 1 >= k0l);
        }
        bool any_need = false;
#pragma unroll
        for (int r = 0; r < WR; ++r) any_need = any_need || need[r];
        auto compute_group = [&](int g) {
                vec_t af[WR];
                T xf[WC][VEC];
                if (UPD_ABL & 4) {
                    // ablation: no LDS fragment reads (operands from registers)
#pragma unroll
                    for (int r = 0; r < WR; ++r)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) af[r][v] = (T)(lane + r + v + kt);
#pragma unroll
                    for (int c = 0; c < WC; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) xf[c][v] = (T)(lane - c - v + kt);
                } else {
#pragma unroll
                    for (int r = 0; r < WR; ++r)
                        af[r] = *reinterpret_cast<const vec_t*>(
                            &sW[cur][(rbk[r] * TILE + li) * SW + g * GROUP + lh * VEC]);
#pragma unroll
                    for (int c = 0; c < WC; ++c)
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            xf[c][v] = sX[cur][(g * GROUP + lh * VEC + v) * SX + c * TILE + li];
                }
#pragma unroll
                for (int r = 0; r < WR; ++r) {
                    if (need[r]) {
#pragma unroll
                        for (int c = 0; c < WC; ++c)
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                acc[r][c] = M::mma(af[r][v], xf[c][v], acc[r][c]);
                    }
                }
            asm volatile("" ::: "memory");
        };
        static_assert(BK / GROUP == 2, "two MFMA groups per k-tile");
        const bool stage_next = !(UPD_ABL & 1) && kt + 1 < nkt;
        if (any_need) compute_group(0);
        if (stage_next) store_tile(cur ^ 1, 0);
        if (any_need) compute_group(1);
        if (stage_next) store_tile(cur ^ 1, 1);
        if (!(UPD_ABL & 2)) __syncthreads();
    }

    // epilogue
    const double c1 = a.add1 ? (a.c1p ? *a.c1p * a.c1i : a.c1i) : 0.0;
    const double c2 = a.add2 ? (a.c2p ? *a.c2p * a.c2i : a.c2i) : 0.0;
    T amax = 0;
#pragma unroll
    for (int r = 0; r < WR; ++r) {
#pragma unroll
        for (int e = 0; e < M::NACC; ++e) {
            const int i = rc0 + rbk[r] * TILE + M::crow(lane, e);
            if (i < a.out_rows) {
                const T bi = a.bias ? a.bias[i] : (T)0;
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    const long long j = jt0 + c * TILE + M::ccol(lane);
                    if (j < a.J) {
                        T v = acc[r][c][e] + bi;
                        const size_t o = (size_t)i * a.J + j;
                        if (a.add1) v += (T)c1 * a.add1[o];
                        if (a.add2) v += (T)c2 * a.add2[o];
                        if (!(UPD_ABL & 8) || v == (T)123456.789) a.out[o] = v;
                        const T av = v < 0 ? -v : v;
                        amax = av > amax ? av : amax;
                    }
                }
            }
        }
    }
    if (do_metrics) {
        // combine the row groups of each particle through LDS (the K loop is over), square, reduce
        T* comb = reinterpret_cast<T*>(smem);
        __syncthreads();
        comb[tid] = mq_e;
        comb[UPD_THREADS + tid] = mq_r;
        __syncthreads();
        double se = 0.0, sr = 0.0;
        if (tid < BN && jt0 + tid < a.J) {
            T qe = 0, qr = 0;
#pragma unroll
            for (int g = 0; g < UPD_THREADS / BN; ++g) { qe += comb[g * BN + tid]; qr += comb[UPD_THREADS + g * BN + tid]; }
            se = (double)qe * (double)qe;
            sr = (double)qr * (double)qr;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sr += __shfl_down(sr, o, 64); }
        __syncthreads();
        double* redm = reinterpret_cast<double*>(smem);
        if (lane == 0) { redm[wave] = sr; redm[4 + wave] = se; }
        __syncthreads();
        if (tid == 0) {
            a.metric_part[blockIdx.x * 2 + 0] = redm[0] + redm[1] + redm[2] + redm[3];
            a.metric_part[blockIdx.x * 2 + 1] = redm[4] + redm[5] + redm[6] + redm[7];
        }
        __syncthreads();
    }
    if (a.absmax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const T other = __shfl_down(amax, o, 64);
            amax = other > amax ? other : amax;
        }
        if (lane == 0) red[wave] = (double)amax;
        __syncthreads();
        if (tid == 0) {
            double m = red[0];
            for (int w = 1; w < UPD_THREADS / 64; ++w) m = red[w] > m ? red[w] : m;
            a.absmax_part[blockIdx.y * gridDim.x + blockIdx.x] = m;
        }
    }
}

__global__ void absmax_final_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    __shared__ double red[256];
    double m = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = part[i] > m ? part[i] : m;
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x + s] > red[threadIdx.x] ? red[threadIdx.x + s] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// xi block alone (the block drawn ahead by cesx_prefetch_noise; tests of the generator): xi[r][j] for r < p.
// A thread draws the 4 x 4 block of rows 4q..4q+3 x particles j0..j0+3 -- four independent Philox chains in flight,
// one 16-byte store per row -- when VEC4 (J a multiple of 4, 16-byte aligned rows); one particle otherwise.  The
// numbers are a function of (seed, step, row quad, global particle index) alone.
template <typename T, bool VEC4>
__global__ __launch_bounds__(256)
void noise_kernel(T* __restrict__ xi, int p, long long J, long long j_offset, unsigned seed_lo,
                  unsigned seed_hi, unsigned step) {
    noise_body<T, VEC4>(xi, p, J, j_offset, seed_lo, seed_hi, step, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------
// opt.ldw    row stride of W (0: = ktot)
template <typename T, int WCT>
static int update_launch(Engine& e, UpdArgs<T>& a, bool aligned, int out_rows, int prof_which, hipStream_t s) {
    using C = UpdCfg<T>;
    constexpr int RC = 4 * C::WR * Mfma<T>::TILE, BN = WCT * Mfma<T>::TILE;
    dim3 grid((unsigned)((e.J + BN - 1) / BN), (unsigned)((out_rows + RC - 1) / RC));
    const int lds = 2 * (RC * C::STRIDE_W + BK * (BN + C::XPAD)) * (int)sizeof(T) + 64 + e.kn * 4 * (int)sizeof(T);
    auto kern = aligned ? update_kernel<T, true, WCT> : update_kernel<T, false, WCT>;
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    e.last_update_grid_x = (int)grid.x;
    e.last_update_grid = (int)(grid.x * grid.y);
    {
        ProfScope prof(e, prof_which, s);
        hipLaunchKernelGGL(kern, grid, dim3(UPD_THREADS), lds, s, a);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

template <typename T>
static int update_t(Engine& e, int out_rows, const void* W, int ktot, const void* bias,
                    const UpdateSrc* src, int nsrc, const void* add1, const double* c1, double c1_imm,
                    const void* add2, const double* c2, double c2_imm, void* out, double* absmax_part,
                    uint64_t step_index, bool metrics, const UpdateOpt& opt, hipStream_t s) {
    UpdArgs<T> a{};
    a.W = (const T*)W; a.ktot = ktot; a.ldw = opt.ldw ? opt.ldw : ktot; a.bias = (const T*)bias; a.out_rows = out_rows;
    int k0 = 0;
    bool aligned = (e.J % Mfma<T>::VEC == 0) && ((uintptr_t)W % 16 == 0) && (a.ldw % Mfma<T>::VEC == 0);
    for (int i = 0; i < 3; ++i) {
        a.src[i] = nullptr; a.src_rows[i] = 0; a.src_k0[i] = 0x7fffffff; a.src_kind[i] = 0;
    }
    for (int i = 0; i < nsrc; ++i) {
        a.src[i] = (const T*)src[i].ptr;
        a.src_rows[i] = src[i].rows;
        a.src_k0[i] = k0;
        a.src_kind[i] = src[i].kind;
        k0 += (src[i].rows + BK - 1) / BK * BK;
        if (src[i].kind == 0 && (uintptr_t)src[i].ptr % 16 != 0) aligned = false;
    }
    if (k0 != ktot) { e.err = "update: K segments do not add up to ktot"; return CESX_EINVAL; }
    a.nsrc = nsrc;
    a.J = e.J; a.j_offset = e.cfg.j_offset;
    a.out = (T*)out;
    a.add1 = (const T*)add1; a.c1p = c1; a.c1i = c1_imm;
    a.add2 = (const T*)add2; a.c2p = c2; a.c2i = c2_imm;
    a.absmax_part = absmax_part;
    a.rowc = (const T*)e.d_rowc;
    a.metric_part = metrics ? e.d_metric_part : nullptr;
    a.metric_seg = opt.metric_seg;
    a.fault = opt.fault; a.fault_seq = opt.fault_seq;
    a.tri_seg = -1;
    for (int i = 0; i < nsrc; ++i)
        if (src[i].tri) a.tri_seg = i;
    a.seed_lo = (unsigned)e.cfg.seed; a.seed_hi = (unsigned)(e.cfg.seed >> 32); a.step = (unsigned)step_index;
    return update_launch<T, UpdCfg<T>::WC>(e, a, aligned, out_rows, opt.prof, s);
}

int launch_update(Engine& e, int out_rows, const void* W, int ktot, const void* bias,
                  const UpdateSrc* src, int nsrc, const void* add1, const double* c1, double c1_imm,
                  const void* add2, const double* c2, double c2_imm, void* out, double* absmax_part,
                  uint64_t step_index, bool metrics, const UpdateOpt& opt, hipStream_t s) {
    if (opt.wf && e.update_v2) {
        // LDS-DMA fast paths (kernels_update2.hip fp32, kernels_update3.hip fp64); -1 = this launch does not qualify
        const int rc = e.cfg.dtype == CESX_F32
            ? launch_update2(e, out_rows, opt.wf, ktot, bias, src, nsrc, add1, c1, c1_imm, add2, c2, c2_imm, out,
                             absmax_part, step_index, metrics, opt, s)
            : launch_update3(e, out_rows, opt.wf, ktot, bias, src, nsrc, add1, c1, c1_imm, add2, c2, c2_imm, out,
                             absmax_part, metrics, opt, s);
        if (rc != -1) return rc;
    }
    if (opt.hkp) { e.err = "update: the hk-free coefficient image has no fallback kernel"; return CESX_EINVAL; }
    return e.cfg.dtype == CESX_F32
        ? update_t<float>(e, out_rows, W, ktot, bias, src, nsrc, add1, c1, c1_imm, add2, c2, c2_imm, out, absmax_part, step_index, metrics, opt, s)
        : update_t<double>(e, out_rows, W, ktot, bias, src, nsrc, add1, c1, c1_imm, add2, c2, c2_imm, out, absmax_part, step_index, metrics, opt, s);
}

// upper bound of the number of workgroups of any update launch (sizes the partial-result buffers)
int update_grid_blocks(Engine& e, int out_rows) {
    int bn = e.cfg.dtype == CESX_F32 ? UpdCfg<float>::WC * 32 : UpdCfg<double>::WC * 16;
    if (bn > 32) bn = 32;          // (update2s_kernel / update3s_kernel: 64 / 32 particles per workgroup)
    const int rc = 256;
    return (int)((e.J + bn - 1) / bn) * ((out_rows + rc - 1) / rc);
}

int launch_absmax_final(Engine& e, int nparts, double* absmax_out, hipStream_t s) {
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, s, e.d_absmax_part, nparts, absmax_out);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

int launch_noise(Engine& e, uint64_t step_index, void* xi, hipStream_t s) {
    const bool vec4 = e.J % 4 == 0 && ((uintptr_t)xi % (4 * e.esz)) == 0;
    const long long per = vec4 ? 4 : 1;
    dim3 grid((unsigned)((e.J / per + 255) / 256), (unsigned)((e.p + 3) / 4));
    auto go = [&](auto kern, auto* ptr) {
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, ptr, e.p, (long long)e.J, (long long)e.cfg.j_offset,
                           (unsigned)e.cfg.seed, (unsigned)(e.cfg.seed >> 32), (unsigned)step_index);
    };
    if (e.cfg.dtype == CESX_F32) {
        if (vec4) go(noise_kernel<float, true>, (float*)xi); else go(noise_kernel<float, false>, (float*)xi);
    } else {
        if (vec4) go(noise_kernel<double, true>, (double*)xi); else go(noise_kernel<double, false>, (double*)xi);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

}  // namespace cesx
