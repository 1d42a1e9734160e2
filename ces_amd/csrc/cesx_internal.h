// Internal declarations shared by the HIP translation units of libcesx.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/cesx.h"

namespace cesx {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// MFMA traits.  A/B operand lane map for both shapes: index within the tile =
// lane % TILE, k within the instruction = lane / TILE (one element per lane).
// ---------------------------------------------------------------------------
template <typename T> struct Mfma;

template <> struct Mfma<float> {
    static constexpr int TILE = 32;   // v_mfma_f32_32x32x2_f32
    static constexpr int KSTEP = 2;   // k values consumed per instruction
    static constexpr int NACC = 16;   // accumulator elements per lane
    static constexpr int VEC = 4;     // elements per 16-byte LDS read
    using acc_t = float __attribute__((ext_vector_type(16)));
    using vec_t = float __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    static __device__ __forceinline__ int crow(int lane, int reg) {
        return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
    static __device__ __forceinline__ int ccol(int lane) { return lane & 31; }
};

template <> struct Mfma<double> {
    static constexpr int TILE = 16;   // v_mfma_f64_16x16x4_f64
    static constexpr int KSTEP = 4;
    static constexpr int NACC = 4;
    static constexpr int VEC = 2;
    using acc_t = double __attribute__((ext_vector_type(4)));
    using vec_t = double __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D map differs from the f32 family: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int crow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
    static __device__ __forceinline__ int ccol(int lane) { return lane & 15; }
};

// Gram kernels: blocks per wave (accumulator budget: 5 x 16 / 9 x 8 VGPRs of the 128 a wave has at
// 4 waves per SIMD); the host's work partition and both Gram kernels use the same constant.
template <typename T> struct GramCfg;
template <> struct GramCfg<float>  { static constexpr int NBW = 4; };
template <> struct GramCfg<double> { static constexpr int NBW = 8; };

// one LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to lds_dst + 16 lane (M0 = the
// wave-uniform LDS byte address; hipcc keeps nothing live in M0 across a statement).  The compiler's
// s_waitcnt bookkeeping does not see these loads: callers wait with an explicit vmcnt.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}

// The same piece from a wave-uniform 64-bit base (an SGPR pair) plus a 32-bit per-lane byte offset: the address
// arithmetic of a piece is scalar, no vector ALU instruction is spent on it (on gfx950 every VALU instruction in an
// f32 MFMA loop costs the matrix pipe ~4 cycles, a 64-bit v_mul four times that).
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(sbase), "v"(voff), "s"(lds_dst) : "memory");
}

// Slab layout of one partial Gram block ("accumulator-major"): element ((q * 64 + lane) * VEC + e)
// is accumulator register VEC * q + e of lane `lane`, so a wave stores a block with 16-byte stores
// of consecutive lanes (1 KiB per instruction).  (row, col) of the first element of group `grp`
// (= q * 64 + lane); the VEC elements of a group are rows row0 + e (f32) / row0 + 4 e (f64).
template <typename T> __host__ __device__ inline void slab_group_rc(int grp, int& row0, int& col, int& rstep);
template <> __host__ __device__ inline void slab_group_rc<float>(int grp, int& row0, int& col, int& rstep) {
    const int q = grp >> 6, lane = grp & 63;
    row0 = 8 * q + 4 * (lane >> 5); col = lane & 31; rstep = 1;          // reg = 4 q + e: (reg & 3) + 8 (reg >> 2) + 4 lh
}
template <> __host__ __device__ inline void slab_group_rc<double>(int grp, int& row0, int& col, int& rstep) {
    const int q = grp >> 6, lane = grp & 63;
    row0 = (lane >> 4) + 8 * q; col = lane & 15; rstep = 4;              // reg = 2 q + e: (lane >> 4) + 4 reg
}

// j (or k) values covered by one 16-byte fragment read of every lane: the
// KSTEP lane groups each take VEC consecutive values -> KSTEP * VEC = 8.
constexpr int GROUP = 8;

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  counter = (particle_lo, particle_hi,
// row_quad, step), key = seed.  oracle/philox.py restates this in numpy.
// ---------------------------------------------------------------------------
struct uint4x { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint4x philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                         uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return {c0, c1, c2, c3};
}

// Box-Muller radius sqrt(-2 ln u), u in [2^-25, 1): the bare v_log_f32 / v_sqrt_f32 (1 ulp each; no denormal scaling,
// no Newton step around the square root -- the library forms cost 14 % of the noise draw's vector instructions, and the
// draw is paid for in full wherever it runs: round 4).  Shared by normal4 and the in-kernel draw of kernels_update2.hip.
__device__ __forceinline__ float bm_radius(float u) {
    return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));      // -2 ln 2 * log2(u)
}

// Four N(0,1) draws for rows 4q..4q+3 of one particle (Box-Muller on two pairs).
__device__ __forceinline__ void normal4(uint4x r, float out[4]) {
    const float s = 5.9604644775390625e-08f;   // 2^-24
    float u0 = ((float)(r.x >> 8) + 0.5f) * s, u1 = ((float)(r.y >> 8) + 0.5f) * s;
    float u2 = ((float)(r.z >> 8) + 0.5f) * s, u3 = ((float)(r.w >> 8) + 0.5f) * s;
    float ra = bm_radius(u0), rb = bm_radius(u2);
    // v_sin_f32 / v_cos_f32 take their argument in revolutions
    out[0] = ra * __builtin_amdgcn_cosf(u1);
    out[1] = ra * __builtin_amdgcn_sinf(u1);
    out[2] = rb * __builtin_amdgcn_cosf(u3);
    out[3] = rb * __builtin_amdgcn_sinf(u3);
}
__device__ __forceinline__ void normal4(uint4x r, double out[4]) {
    const double s = 2.3283064365386963e-10;   // 2^-32
    const double twopi = 6.283185307179586476925286766559;
    double u0 = ((double)r.x + 0.5) * s, u1 = ((double)r.y + 0.5) * s;
    double u2 = ((double)r.z + 0.5) * s, u3 = ((double)r.w + 0.5) * s;
    double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
    double sa, ca, sb, cb;
    sincos(twopi * u1, &sa, &ca);
    sincos(twopi * u3, &sb, &cb);
    out[0] = ra * ca; out[1] = ra * sa; out[2] = rb * cb; out[3] = rb * sb;
}

// One workgroup's share of a noise block xi[p][J] (256 threads; thread = rows 4 by .. 4 by + 3 of NP consecutive
// particles of column block bx): the body of noise_kernel (kernels_update.hip).
template <typename T, bool VEC4>
__device__ __forceinline__ void noise_body(T* __restrict__ xi, int p, long long J, long long j_offset, unsigned seed_lo,
                                           unsigned seed_hi, unsigned step, unsigned bx, unsigned by) {
    constexpr int NP = VEC4 ? 4 : 1;
    const long long j0 = ((long long)bx * 256 + threadIdx.x) * NP;
    const int q = (int)by;
    if (j0 >= J) return;
    T z[NP][4];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const unsigned long long gj = (unsigned long long)(j_offset + j0 + c);
        const uint4x r = philox4x32_10((uint32_t)gj, (uint32_t)(gj >> 32), (uint32_t)q, step, seed_lo, seed_hi);
        normal4(r, z[c]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * q + e >= p) break;
        T* dst = xi + (size_t)(4 * q + e) * J + j0;
        if (VEC4) {
            typedef T v4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4*>(dst) = v4{z[0][e], z[NP > 1 ? 1 : 0][e], z[NP > 2 ? 2 : 0][e], z[NP > 3 ? 3 : 0][e]};
        } else {
            dst[0] = z[0][e];
        }
    }
}

// ---------------------------------------------------------------------------
// Packed fp64 moment buffer (the only data that crosses GPUs).  The part that
// depends on U alone comes first and is contiguous, so that it can be reduced
// (and chol(C) started) before the rest of the Gram is finished:
//   [ N | sum a (p) | S_aa (p x p) ][ sum b (n) | S_ab (p x n) | S_bb (n x n) | lagged q sums (2) ]
// ---------------------------------------------------------------------------
struct MomLayout {
    int p, n;
    __host__ __device__ size_t sa() const { return 1; }
    __host__ __device__ size_t Saa() const { return 1 + (size_t)p; }
    __host__ __device__ size_t uu_len() const { return 1 + (size_t)p + (size_t)p * p; }
    __host__ __device__ size_t sb() const { return uu_len(); }
    __host__ __device__ size_t Sab() const { return sb() + n; }
    __host__ __device__ size_t Sbb() const { return Sab() + (size_t)p * n; }
    __host__ __device__ size_t tail() const { return Sbb() + (size_t)n * n; }
    __host__ __device__ size_t len() const { return tail() + 2; }
};

// ---------------------------------------------------------------------------
// Gram work partition (host builds it once per engine).
// ---------------------------------------------------------------------------
struct GramPlan {
    int tile;            // Mfma<T>::TILE
    int nbr;             // block rows = ceil((p+n)/tile)
    int nblocks;         // lower-triangular blocks
    int ntypes;          // workgroup types
    int max_rb;          // max staged row blocks over types
    int nbw;             // per-wave block capacity (compile-time constant of the kernel)
    // Every type gets a number of J-slices (workgroups) proportional to its block count, so that
    // all workgroups of a launch carry the same amount of MFMA work whatever the shape of the types.
    // flattened tables uploaded to the device:
    //  type_hdr[type*8 + {0: nrb, 1: rows_off, 2: blocks_off, 3: blocks of the type, 4: first workgroup,
    //                     5: slices, 6: first slab (in blocks), 7: first row-sum slot}]
    //  rows[rows_off + i]            = global block row of compact row i (| 1<<16: this type reports its sums)
    //  wblk[(blocks_off + wave*nbw + b)*3 + {0,1,2}] = {ia, ib, block index inside the type} or ia = -1
    //  blk_rc[out*5 + {0..4}] = (R, C) of output block, its first slab, slab stride (blocks), slices
    //  row_own[br*2 + {0,1}]  = first row-sum slot and slices of the type that owns block row br (slices 0: not owned)
    std::vector<int> type_hdr, rows, wblk, blk_rc, row_own;
    int total_wgs = 0, total_slabs = 0, total_rs = 0;
    int own_lo = 0, own_hi = 0;   // block rows [own_lo, own_hi) whose row sums this plan reports
};
// subset: 0 = all lower-triangular blocks, 1 = the blocks that only involve the first pbU
// block rows (U x U: everything chol(C) needs), 2 = all the others
GramPlan make_gram_plan(int P, int tile, int nbw, int max_rows_lds, int subset = 0, int pbU = 0, int min_types = 1,
                        int wg_budget = 256, long long ntiles = 1 << 30);

// one Gram launch: plan + its device tables and partial-result buffers
struct GramPart {
    GramPlan plan;
    int *d_type_hdr = nullptr, *d_rows = nullptr, *d_wblk = nullptr, *d_blk_rc = nullptr, *d_row_own = nullptr;
    void* d_slabs = nullptr;            // per type [slices][blocks of the type][tile*tile] engine dtype
    double* d_rowsum_part = nullptr;    // [row-sum slots][p+n]
};

// ---------------------------------------------------------------------------
// Engine state
// ---------------------------------------------------------------------------
struct Scalars {              // device-resident fp64 scalars written by K2
    double hk, t_new, sqrt2hk, alpha;
    double self_bias, self_bias_data, bias_data, bias;
    double radspec, frob2, tr_suu, absmax;
    double spare[4];
    int    status, pad;
    unsigned long long seq;   // host copy only: step counter written last by publish_kernel
};

// Finalisation of a step's data metrics and publication of its scalars to the host: the fixed-order fp64 sum of the
// update kernel's per-workgroup partials (sums[0..1] = sum over workgroups of {q_r^2, q_e^2}), this shard's
// contribution to the two data metrics (divided by the GLOBAL ensemble size), from the tail of the all-reduced moment
// buffer the previous step's global values (multi-device runs), then -- host != nullptr -- the copy of the scalars
// into the host-mapped result block, its sequence number last.  Run by ONE workgroup of 256 threads: the
// metric_final kernel, or (single-device fast path) an extra workgroup of the NEXT step's U x U reduce launch, so
// that no one-workgroup kernel sits between the update kernel and the next Gram launch.
struct MetricFin {
    const double* part; int nparts; const double* mom; size_t tail_off; double* sums; Scalars* sc;
    Scalars* host; unsigned long long seq;
    double N;            // > 0: the global ensemble size (else read from mom[0])
};
__device__ __forceinline__ void metric_final_body(const MetricFin& f) {
    __shared__ double mf_red[2][4];
    __shared__ double mf_out[4];
    constexpr int ND = offsetof(Scalars, seq) / 8;
    // 256 threads do the work whatever the launch's block size is (the summation order is part of the result);
    // the other waves of a larger block leave (a finished wave does not hold up the barriers below)
    constexpr unsigned MF_THREADS = 256;
    if (threadIdx.x >= MF_THREADS) return;
    // the scalars earlier kernels of the step wrote (hk, t, bias ...): read FIRST, beside the partial sums -- the
    // chain of dependent round trips of this one workgroup is what a launch that carries it waits for
    double mine = 0.0;
    if (f.host != nullptr && threadIdx.x < ND)
        mine = __hip_atomic_load(reinterpret_cast<const double*>(f.sc) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double lag0 = 0.0, lag1 = 0.0, Nm = f.N;          // (thread 0's inputs from the moment buffer: also up front)
    if (threadIdx.x == 0) {
        lag0 = f.mom[f.tail_off]; lag1 = f.mom[f.tail_off + 1];
        if (!(Nm > 0.0)) Nm = f.mom[0];
    }
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < f.nparts; i += MF_THREADS) { a += f.part[(size_t)i * 2]; b += f.part[(size_t)i * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4) { mf_red[0][threadIdx.x >> 6] = a; mf_red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int i = 0; i < (int)(MF_THREADS >> 6); ++i) { a += mf_red[0][i]; b += mf_red[1][i]; }
        const double N = Nm;
        f.sums[0] = a; f.sums[1] = b;
        mf_out[0] = f.sc->bias_data = a / N;
        mf_out[1] = f.sc->self_bias_data = b / N;
        mf_out[2] = f.sc->spare[1] = lag0 / N;          // lagged global bias-data
        mf_out[3] = f.sc->spare[2] = lag1 / N;          // lagged global self-bias-data
    }
    if (f.host == nullptr) return;
    // last kernel of an eks / aldi step: publish the results (see publish_kernel)
    __syncthreads();
    if (threadIdx.x < ND) {
        const int i = threadIdx.x;
        const double v = i == (int)(offsetof(Scalars, bias_data) / 8) ? mf_out[0]
                       : i == (int)(offsetof(Scalars, self_bias_data) / 8) ? mf_out[1]
                       : i == (int)(offsetof(Scalars, spare) / 8) + 1 ? mf_out[2]
                       : i == (int)(offsetof(Scalars, spare) / 8) + 2 ? mf_out[3] : mine;
        reinterpret_cast<double*>(f.host)[i] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&f.host->seq, f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Engine {
    cesx_config cfg{};
    std::string err;
    bool problem_set = false, shift_valid = false;
    bool chol_inflight = false;    // cesx_chol_async ran for the current moments; cesx_apply joins the side stream
    bool chol_fused_center = false;   // ... with the centring fused into the factorisation's load (no U-only centring launch on the side stream)
    bool fuse_center_auto = true;     // no CESX_FUSE_CENTER given: fused where the step takes the hk-free form AND the second Gram launch is
                                      // short (Engine::gram_b_short: small ensembles -- the side chain is then the step's critical path and
                                      // the host's launches its floor; C4 0.0790 -> 0.0728 ms/step, round 4), not at C2 (see below)
    bool gram_b_short = false;        // the second Gram launch's MFMA time is below ~60 us (cesx_create)
    bool fuse_center_ok = false;      // CESX_FUSE_CENTER=1 switches that on (round 4: also with the hk-free form -- the tail launch then forms C, M, ubar
                                      // and the trace / bias sums itself; measured 0.3988 against 0.3952 ms/step at C2: the side chain ends 15 us
                                      // earlier, the noise draw behind it meets the end of the second Gram launch, and the tail launch is longer).  Measured at C2 (round 3, row sums staged through LDS): the factorisation
                                      // takes 107 us instead of 13 + 99, the side chain ends 4.5 us earlier, the step gains 0.2-1 % over 400 steps and
                                      // nothing in the 20-step line (the caller's stream -- second Gram launch, reduce, centring -- is the critical path)
    bool overlap_chol = true;      // run chol(C) on the side stream beside the second (non U x U) part of the Gram
                                   // (CESX_OVERLAP=0 disables)
    int p = 0, n = 0, P = 0;
    int64_t J = 0, Jg = 0;
    size_t esz = 4;               // sizeof(T)
    bool diag_sigma = false;
    // Dense Gamma (the reference's pde examples use a sample covariance, examples/notebooks/lorenz63.ipynb): the engine
    // WHITENS the data once per step -- G~ = L_Gamma^{-1} G with Gamma = L L^T (one triangular n x n x J product on the matrix
    // pipe, the update kernel's own code, ~4 GFLOP at C2), y~ = L^{-1} y, Gamma~ = I -- and every kernel behind it runs the
    // diagonal-Gamma path: D = (1/J) E^T Gamma^{-1} R = (1/J) E~^T R~ (ces/calibrate.py:429/:461/:503), the data metrics
    // (:434-435), the recomputed gain C_ug (hk C_gg + Gamma)^{-1} (g - y) (:440-441/:472-473) and K (g - y) are all invariant.
    // (Round 4 kept Gamma^{-1} in the small algebra -- fp64 GEMMs for the Frobenius term and K -- and ran a VALU pass over G for
    //  the per-particle metrics: 1.23 ms per step against 0.39 for diagonal Gamma.)
    bool whiten = false;
    void *d_Wwh = nullptr, *d_Wwh_f = nullptr;   // L_Gamma^{-1} zero padded [rpad][kn], row-major and in the update kernels' fragment-major order
    void* d_Gw = nullptr;                        // [n][J] the whitened data of the current step (allocated with the first dense problem)
    const void* gw_src = nullptr;                // ... of this caller array, whitened on gw_stream
    hipStream_t gw_stream = nullptr;
    unsigned long long gw_calls = 0;             // Engine::moments_calls when it was whitened
    std::vector<double> h_LG, h_Li;              // host copies of L_Gamma and its inverse (cesx_debug_dense reports in the caller's coordinates)

    // problem data (fp64) on device
    double *d_y = nullptr, *d_mu = nullptr, *d_ustar = nullptr;
    double *d_Gamma = nullptr, *d_gw = nullptr;    // the engine's (diagonal, or whitened: identity) Gamma and gw = diag(Gamma^{-1})
    double *d_Sigma = nullptr, *d_Sinv = nullptr, *d_sw = nullptr;
    // centring shift
    double* d_shift64 = nullptr;   // [p+n]
    void*   d_shiftT = nullptr;    // [p+n] engine dtype
    void*   d_yT = nullptr;        // y in engine dtype
    void*   d_gwT = nullptr;       // diag(Gamma^{-1}) in engine dtype
    // gram: part 0 = U x U blocks (all chol(C) needs), part 1 = the rest
    GramPart gp[2];
    MomLayout ml{};
    // stats partials
    int colsum_slices = 0;
    double* d_colsum_part = nullptr;
    // moments
    size_t mom_len = 0;
    double* d_mom = nullptr;
    double* d_sums = nullptr;      // [1+p+n]
    double* d_sums_w = nullptr;    // [1+p+n] the same with the G part whitened (dense Gamma)
    // dense workspace (fp64)
    double *d_ubar = nullptr, *d_gbar = nullptr, *d_m = nullptr;
    double *d_C = nullptr, *d_L = nullptr, *d_Cug = nullptr, *d_See = nullptr, *d_Srr = nullptr;
    double *d_dg = nullptr;
    double *d_K = nullptr, *d_Kp = nullptr, *d_M = nullptr, *d_P = nullptr, *d_PK = nullptr;
    double *d_t1 = nullptr, *d_t2 = nullptr, *d_t3 = nullptr, *d_t4 = nullptr;   // max(p,n)^2 each
    double *d_Wh = nullptr;        // L_Gamma^{-1} in fp64 (dense Gamma: the centring sums of a fresh ensemble are whitened with it)
    // warm-started SPD inverses of K2 (kernels_dense.hip, spd_inverse): the previous step's inverse of the gain matrix (0, n x n) and
    // of the EKS matrix (1, p x p), three n_max^2 scratch matrices, 2 x ceil(n_max/16)^2 residual partials, the verdict word
    double *d_ns_x[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [which][ns_cur[which]]: the previous inverse; the other one: this step's
    int ns_cur[2] = {0, 0};
    double *d_ns_r[3] = {nullptr, nullptr, nullptr}, *d_ns_parts = nullptr;
    int* d_ns_skip = nullptr;
    bool ns_ok = true;             // CESX_NS_WARM=0: always the factorisation
    double ns_r0_last = 1e300;     // ||I - A X_prev||_F^2 of the last step's warm-start attempt (cesx_result reads it from the result block)
    const int* gate = nullptr;     // != nullptr while spd_inverse enqueues its factorisation chain: those kernels return when *gate != 0
    double *d_spec = nullptr;      // spectral rule: {sum 2^-k log N_k, 2^-k, degenerate flag, pad} + 2 x ceil(n/16)^2 partial sums of squares
    double *d_absmax = nullptr;    // [1]
    void   *d_rowc = nullptr;        // [kn][4] {gbar_i, y_i, 1/Gamma_ii, 0} engine dtype (K3 data metrics)
    double *d_metric_part = nullptr; // [blocks][2] per-workgroup {sum q_r^2, sum q_e^2}
    double *d_metric_sums = nullptr; // [2] this shard's sums of the last apply
    double *d_mv = nullptr;        // matvec results [6][max(p,n)]
    double *d_part = nullptr;      // reduction partials
    Scalars* d_scal = nullptr;
    double* d_absmax_part = nullptr;
    // update coefficients (engine dtype)
    int bk = 16, kp = 0, kn = 0, ktot = 0, rpad = 0;
    double* d_Lwork = nullptr;     // p > 256: work copy of C for the blocked Cholesky [potrf_ld(p)]^2
    void* d_W = nullptr;           // [rpad][ktot]
    void* d_Wf = nullptr;          // the same matrix in the fragment-major order of kernels_update2.hip (fp32) / kernels_update3.hip (fp64)
    bool update_v2 = true;         // fp32 K3 through the LDS-DMA kernel (CESX_UPDATE_V1=1 switches back)
    int  center_u_wgs = 256;       // workgroups of the U-only centring on the side stream (see cesx_create)
    bool gram_v2 = true;           // K1 through the LDS-DMA kernel when the shapes allow (CESX_GRAM_V1=1 switches back)
    int num_cus = 256;
    void* d_bias = nullptr;        // [rpad]
    // ---- ALDI with the time step kept OUT of the update coefficients (round 4; launch_dense, "hk-free") ----
    // U_next = hk ( sqrt(2/hk) L xi + (a I - M + I/hk) U - K G + b' ),  b' = K y + M mu - a ubar: the image
    // Wq = [ L | a I - M (+ 1/hk on the diagonal) | -K ] needs hk in p diagonal entries only.  L is written by the
    // factorisation itself (it stores its panels into the image as it finishes them, on the side stream), everything else
    // by ONE launch of the caller's stream behind the second reduce (tail_aldi_kernel: -K and a I - M rows straight from the
    // moments, the four matvecs, the Frobenius partials; its last workgroup, by ticket, joins the side stream, sums the
    // partials and writes hk, b', the diagonal and the next shift), and K3 takes hk and sqrt(2hk) from the scalar block at
    // run time: xi segment first, accumulators rescaled once by sqrt(2/hk), the result times hk in the epilogue.
    void* d_Wq = nullptr;          // [rpad][ktot] fragment-major (wf_index), fp32; zero outside what the kernels above write
    unsigned* d_ticket = nullptr;  // arrival counter of tail_aldi_kernel
    bool hkfree_ok = true;         // CESX_HKFREE=0 switches the path off
    bool update_small = true;      // CESX_UPDATE_SMALL=0: update2_kernel also for out_rows <= 64 (dev A/B)
    bool side_img = false;         // the factorisation in flight stores L into d_Wq (launch_chol_async)
    bool last_hkfree = false;      // the last launch_dense took the path: the update launch reads d_Wq in the order [xi; U; G]
    // ---- K3 through the Cholesky factor (round 6; kernels_update4.hip) ----
    // With a diagonal Sigma, C Sigma^{-1} (U - mu) = L (L^T Sigma^{-1} U) - M mu: two triangular products instead of the dense M U.
    // d_Wq then holds the CHAINED image (wc_index_L / _Lt / _K): the factorisation stores every panel twice (L, and transposed and
    // scaled), the tail launch stores -K, nothing stores a I - M.  Decided by the problem and the shape alone (cesx_set_problem):
    // every call flow of a problem runs the same kernels.
    bool chain = false;            // d_Wq is in the chained layout and the hk-free step launches update4_kernel
    bool chain_ok = true;          // CESX_CHAIN=0: the hk-free form of round 4 (update2_kernel<., true>) also where the chained one qualifies
    bool skip_L_hint = false;      // set by the callers of the factorisation: the step it belongs to is (expected to be) a chained one
    bool L_stale = false;          // the last factorisation wrote the chained image only: d_L does not hold its factor (refresh_factor)
    void* d_xi_tmp = nullptr;      // [p][J] a noise block drawn right in front of update4_kernel when none was prefetched or injected
    void* d_Wfwd = nullptr;        // forward-map staging [npad][kp]
    void* d_Wfwd_f = nullptr;      // the same map in the fragment-major order of the LDS-DMA update kernels (cesx_forward_set_lineal)
    void* d_bfwd = nullptr;        // [rpad] its offset b
    bool  fwd_set = false, fwd_has_b = false;
    double *d_A64 = nullptr, *d_b64 = nullptr;   // the installed map in fp64 (n x p, n): cesx_moments_rest_lineal
    double *d_lvec = nullptr;                    // [2][n] c = A s_u + b - s_g and A sa
    // per-kernel profiling (cesx_profile_*)
    int prof_part = 0;                 // which moments launch (0: U x U, 1: the rest) the next profiled Gram launch is
    unsigned long long prof_step = 0;  // bumped by every first-half entry point (cesx_moments_uu*): the step the next profiled launches belong to
    bool profile = false;
    int  profile_only = -1;            // cesx_profile_enable(h, 3 / 4): only the update (1) / the moments (0) launches carry events
    bool profile_gap_only = false;     // cesx_profile_enable(h, 2): bind ONLY the stop of the second moments launch and the start of the
                                       // update launch (the gap between them, cesx_profile_gap): a step whose launches carry start AND stop
                                       // events runs ~70 us longer and its gap reads anywhere between 40 and 150 us
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[2];
    std::vector<unsigned long long> prof_tag[2];       // the step (Engine::moments_calls at the launch) each pair belongs to: cesx_profile_gap pairs events of ONE step
    std::vector<hipEvent_t> prof_pool;
    long long* d_clk = nullptr;    // [4] {s_memtime, s_memrealtime} ticks of the last PROFILED update launch (workgroup 0, wave 0)
    // results
    Scalars* h_scal = nullptr;           // pinned, device-mapped: the GPU writes results straight into it
    Scalars* h_scal_dev = nullptr;       // device address of h_scal
    unsigned long long seq = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;      // hand-over main -> side (U x U moments reduced) / side -> main (chol(C) done)
    hipStream_t side = nullptr;      // side stream (high priority): U x U Gram, chol(C), the step's last small kernel
    hipEvent_t ev_x[2] = {nullptr, nullptr};   // prefetched noise block b is complete (side stream)
    // Noise blocks drawn ahead by cesx_prefetch_noise, [p][J] engine dtype each.  Two of them: behind chol(C) of the
    // step that asked for block s the side stream draws block s (unless an earlier step already did) AND block s + 1
    // (the lookahead), so that from the second step on the update kernel finds its block complete -- ordered before
    // this step's chol(C) on the side stream, whose event the caller's stream waits for anyway -- and the draw of the
    // next one runs beside K2's latency-bound kernels with nothing waiting for it.
    void* d_xi[2] = {nullptr, nullptr};
    long long xi_step[2] = {-1, -1};                 // step index block b holds (-1: none)
    unsigned long long xi_seq[2] = {0, 0};           // chol_seq of the cesx_chol_async call that drew block b
    long long xi_want = -1;          // step index asked for by cesx_prefetch_noise, drawn behind the next chol(C)
    bool xi_lookahead = true;        // CESX_NOISE_LOOKAHEAD=0 switches the second draw off
    unsigned long long chol_seq = 0;              // cesx_chol_async calls so far
    // ---- the side stream joined through a polled word instead of a barrier packet (round 3, launch_dense) ----
    bool poll_join_ok = true;                     // CESX_POLL_JOIN=0: always join the side stream with the event
    unsigned long long* d_cholflag = nullptr;     // chol_seq of the last factorisation that completed on the side stream, stored by the kernel itself
    bool chol_signals = false;                    // the factorisation in flight stores that word (its last kernel does)
    unsigned long long evb_waited_seq = 0;        // ... the last one whose ev_b a stream has waited for,
    hipStream_t evb_waited_stream = nullptr;      // and that stream
    // ---- the polled join made safe (round 4) ----
    int side_prio = 0; bool side_has_prio = false;   // priority of the side stream (cesx_create)
    unsigned long long poll_ticks = 200000000ull; // bound of the poll in 100-MHz wall-clock ticks (2 s; CESX_POLL_TIMEOUT_MS)
    bool last_join_polled = false;                // the last launch_dense joined the side stream through the polled word
    unsigned long long poll_recoveries = 0;       // steps whose poll ran out and that cesx_result re-ran with chol(C) in line
    bool in_retry = false;
    unsigned long long test_drop_signal_at = 0;   // CESX_TEST_DROP_CHOL_SIGNAL (tests): that factorisation does not store its word
    unsigned long long moments_calls = 0;         // cesx_moments* calls so far (a re-run step tells whether a later one read an unwritten ensemble)
    struct LastApply { bool valid = false; cesx_step_params prm{}; const double* mom = nullptr; const void *U = nullptr, *G = nullptr, *xi = nullptr;
                       void* Unext = nullptr; hipStream_t s = nullptr; unsigned long long moments_calls = 0;
                       bool mom_reused = false;      // a later cesx_moments* call was given the same buffer before the result was read: no re-run
                     } last_apply;
    // ---- RCCL communicator of a sharded ensemble (comm.hip; nullptr: none) ----
    void* comm = nullptr;
    void* comm_side = nullptr;     // the side stream's own communicator (ncclCommSplit of `comm`; nullptr: both streams share `comm`)
    int comm_nranks = 0, comm_rank = 0;
    unsigned long long comm_calls = 0, comm_doubles = 0;
    int last_update_grid_x = 0, last_update_grid = 0, last_metric_parts = 0;
    bool pending = false;
    // single-device fast path: the metric finalisation + publication of the last update rides on the next
    // U x U reduce launch (cesx_moments_uu_chol); every other entry point flushes it as a kernel of its own first
    bool met_deferred = false;
    hipStream_t met_stream = nullptr;
    double* d_lag = nullptr;       // [3] {N, lagged sum q_r^2, lagged sum q_e^2} of the moment buffer of the last cesx_apply (K2 copies
                                   // them here: the deferred finalisation reads engine-owned memory, not the caller's buffer)
    cesx_step_params last_prm{};
};

// ---------------------------------------------------------------------------
// kernel launchers (defined in the .hip files)
// ---------------------------------------------------------------------------
struct UpdateSrc {            // one K-segment of the update GEMM
    const void* ptr;          // (rows x J) array, or nullptr for on-device noise
    int rows;                 // real rows
    int kind;                 // 0 = memory, 1 = philox noise
    int tri;                  // 1: the W columns of this segment are lower triangular (sqrt(2hk) L)
};

int launch_colsum(Engine& e, const void* U, const void* G, double* sums, hipStream_t s);
int launch_set_shift(Engine& e, const double* sums, hipStream_t s);
int launch_gram(Engine& e, int part, const void* U, const void* G, double* mom, hipStream_t s, bool no_reduce = false);   // part 0 / 1
int launch_gram_reduce(Engine& e, int part, double* mom, hipStream_t s, hipEvent_t stop = nullptr, const MetricFin* fin = nullptr);   // the fp64 slab reduce of that launch (stop: bound to its completion)
// kernels_gram2.hip (LDS-DMA Gram): CESX_OK, an error, or -1 when the launch does not qualify (caller falls back)
int launch_gram2(Engine& e, int part, const void* U, const void* G, hipStream_t s);
int launch_dense(Engine& e, const cesx_step_params& prm, const double* mom, int phase, hipStream_t s, bool upd2_ok = false);
// Kernels of the caller's stream and of the side stream may WAIT for each other inside a launch (the polled join of
// launch_dense) only when the two streams cannot share a hardware queue: HIP maps the streams of one priority level
// onto a few queues, and a waiter in front of what it waits for in one in-order queue never ends.  True when `s` has a
// strictly lower priority than the side stream (a numerically greater one).
bool stream_below_side(Engine& e, hipStream_t s);
int launch_chol_async(Engine& e, int update, const double* mom, hipStream_t s, bool ev_a_bound = false);
int refresh_factor(Engine& e, hipStream_t s);
// whether an update launch [xi; U; G] -> Unext of this engine qualifies for the LDS-DMA fp32 kernel (kernels_update2.hip)
bool update2_qualifies(const Engine& e, const void* U, const void* G, const void* xi, const void* Unext);
struct UpdateOpt {
    const unsigned long long* fault = nullptr;   // != nullptr: the launch leaves `out` untouched when *fault == fault_seq (a polled join that ran out)
    unsigned long long fault_seq = 0;
    int ldw = 0;          // row stride of W (0: = ktot)
    const double* hkp = nullptr;   // != nullptr (LDS-DMA fp32 kernel, triangular segment FIRST): W carries no time step -- the accumulators
    const double* s2p = nullptr;   // are scaled by *s2p / *hkp behind the first segment and the result (+ bias) by *hkp in the epilogue
    int metric_seg = 1;   // K-segment that holds G (data metrics)
    const void* wf = nullptr;  // fragment-major copy of the WHOLE W (fp32): enables the LDS-DMA kernel
    int prof = -1;        // profiling slot (1 = K3) or -1
};
int launch_update(Engine& e, int out_rows, const void* W, int ktot, const void* bias,
                  const UpdateSrc* src, int nsrc,
                  const void* add1, const double* c1, double c1_imm,
                  const void* add2, const double* c2, double c2_imm,
                  void* out, double* absmax_part, uint64_t step_index, bool metrics,
                  const UpdateOpt& opt, hipStream_t s);
int update_grid_blocks(Engine& e, int out_rows);
// kernels_update2.hip: returns CESX_OK, an error, or -1 when the launch does not qualify (caller falls back)
int launch_update2(Engine& e, int out_rows, const void* Wf, int ktot, const void* bias,
                   const UpdateSrc* src, int nsrc,
                   const void* add1, const double* c1, double c1_imm,
                   const void* add2, const double* c2, double c2_imm,
                   void* out, double* absmax_part, uint64_t step_index, bool metrics,
                   const UpdateOpt& opt, hipStream_t s);
// index of W[i][k] (row i of the zero-padded rpad x ktot matrix, nkt = ktot / 16) in the fragment-major
// image: for every 256-row chunk y and k-tile kt, 16 pieces (g, rb) of 64 lanes x 4 floats; lane
// (lh, li) of piece (g, rb) holds W[256 y + 32 rb + li][16 kt + 2 (4 g + v) + lh], v = 0..3
__host__ __device__ inline size_t wf_index(int i, int k, int nkt) {
    const int y = i >> 8, rb = (i >> 5) & 7, li = i & 31;
    const int kt = k >> 4, kk = k & 15, m = kk >> 1, lh = kk & 1, g = m >> 2, v = m & 3;
    return ((((size_t)y * nkt + kt) * 16 + g * 8 + rb) * 64 + lh * 32 + li) * 4 + v;
}
// ---- the "chained" coefficient image of kernels_update4.hip (K3 through the Cholesky factor) ----
// (18 + ng) tiles of 16 pieces (g, b) of 64 lanes x 4 floats.  k order inside a 16-column tile: lane (lh, m) of piece (g, b)
// holds columns 8 g + 4 lh + v, v = 0..3 -- the accumulator row order of v_mfma_f32_32x32x2_f32 (wf_index keeps 2 (4 g + v) + lh).
__host__ __device__ inline size_t wc_slot(int t, int b, int m, int kl) {          // tile t, row block b, row m of it, column kl (0..15) of the tile
    const int g = kl >> 3, lh = (kl >> 2) & 1, v = kl & 3;
    return ((((size_t)t * 16 + g * 8 + b) * 64) + lh * 32 + m) * 4 + v;
}
// L[i][j], i >= j: column block cb = j / 32 of the lower factor sits in tiles 2 (8 - cb) + h, rows b >= cb
__host__ __device__ inline size_t wc_index_L(int i, int j) {
    return wc_slot(2 * (8 - (j >> 5)) + ((j >> 4) & 1), i >> 5, i & 31, j & 15);
}
// -(L^T Sigma^{-1})[r][k] = -L[k][r] / Sigma_kk, k >= r: column block kb = k / 32 sits in tiles 2 (7 - kb) + h, rows b <= kb
__host__ __device__ inline size_t wc_index_Lt(int r, int k) {
    return wc_slot(2 * (7 - (k >> 5)) + ((k >> 4) & 1), r >> 5, r & 31, k & 15);
}
// -K[i][c]: tiles 18 + c / 16
__host__ __device__ inline size_t wc_index_K(int i, int c) {
    return wc_slot(18 + (c >> 4), i >> 5, i & 31, c & 15);
}
// kernels_update4.hip: CESX_OK, an error, or -1 when the launch does not qualify
int launch_update4(Engine& e, const void* U, const void* G, const void* xi, void* out, bool metrics, const UpdateOpt& opt, hipStream_t s);
bool update4_shape_ok(const Engine& e);
// kernels_update3.hip (fp64 LDS-DMA update): CESX_OK, an error, or -1 when the launch does not qualify
int launch_update3(Engine& e, int out_rows, const void* Wd, int ktot, const void* bias,
                   const UpdateSrc* src, int nsrc,
                   const void* add1, const double* c1, double c1_imm,
                   const void* add2, const double* c2, double c2_imm,
                   void* out, double* absmax_part, bool metrics, const UpdateOpt& opt, hipStream_t s);
// index of W[i][k] in the fp64 fragment-major image read by update3_kernel: for every 256-row chunk y, k-tile kt
// and 16-row block rb two pieces (sp) of 64 lanes x 2 doubles; lane (lr, li) = 16 lr + li of piece sp holds
// W[256 y + 16 rb + li][16 kt + 4 (2 sp + e) + lr], e = 0, 1
__host__ __device__ inline size_t wd_index(int i, int k, int nkt) {
    const int y = i >> 8, rb = (i >> 4) & 15, li = i & 15;
    const int kt = k >> 4, kk = k & 15, s = kk >> 2, lr = kk & 3, sp = s >> 1, e = s & 1;
    return ((((size_t)y * nkt + kt) * 16 + rb) * 2 + sp) * 128 + (size_t)(lr * 16 + li) * 2 + e;
}
// dense Gamma: the whitened copy of G this step's kernels read (G itself when Gamma is diagonal); force: whiten again even
// when the same array was whitened last (the first moments call of a step: its contents may be new)
const void* whitened_G(Engine& e, const void* G, hipStream_t s, bool force, int* rc);
int launch_metric_final(Engine& e, const double* mom, bool publish, hipStream_t s);
MetricFin metric_fin_args(Engine& e, const double* mom, bool publish);     // (publish: takes the next sequence number; mom == nullptr: the engine's own copy d_lag)
int launch_publish(Engine& e, hipStream_t s);
int launch_absmax_final(Engine& e, int nparts, double* absmax_out, hipStream_t s);
int potrf_ld(int n);
int gram_nbw(int dtype);
int gram_tile(int dtype);
int gram_kt(int dtype);
int gram_max_stage_rows();
int launch_noise(Engine& e, uint64_t step_index, void* xi, hipStream_t s);
int launch_stage_forward(Engine& e, const void* A, const void* b, hipStream_t s);   // A, b -> d_Wfwd, d_Wfwd_f, d_bfwd
int launch_moments_lineal(Engine& e, double* mom, hipStream_t s);                   // G part of the moments from the head + the installed linear map
int launch_calibrate(Engine& e, double target_ms, double* tflops, double* clock_ghz, hipStream_t s);   // kernels_calib.hip

// Event pair for one profiled launch (cesx_profile_*).  bound = false: the pair is RECORDED around the launch (two
// marker packets: they delay the stream by ~6 us each and the interval includes that).  bound = true: the caller
// passes a / b to hipExtLaunchKernel, which ties them to the kernel's own start / end time stamps -- the interval
// is the kernel's duration as rocprofv3 reports it, and no marker sits in front of the next launch.
struct ProfScope {
    Engine& e; int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool bound;
    ProfScope(Engine& e_, int which_, hipStream_t s_, bool bound_ = false) : e(e_), which(which_), s(s_), bound(bound_) {
        if (!e.profile || which < 0) return;
        if (e.profile_only >= 0 && which != e.profile_only) return;
        if (e.profile_gap_only && !bound) return;         // (recorded pairs are markers on the stream: not in a gap-only step)
        auto get = [&]() { hipEvent_t ev = nullptr; if (!e.prof_pool.empty()) { ev = e.prof_pool.back(); e.prof_pool.pop_back(); } else (void)hipEventCreate(&ev); return ev; };
        // gap-only: the moments launch binds its stop alone, the update launch its start alone (the other stays null)
        if (!e.profile_gap_only || which == 1) a = get();
        if (!e.profile_gap_only || which == 0) b = get();
        if (!bound) (void)hipEventRecord(a, s);
    }
    bool on() const { return a != nullptr || b != nullptr; }
    ~ProfScope() {
        if (!on()) return;
        if (!bound) (void)hipEventRecord(b, s);
        e.prof_ev[which].push_back({a, b});
        e.prof_tag[which].push_back(e.prof_step * 2 + (which == 0 ? (unsigned long long)e.prof_part : 1ull));
    }
};

// the message cesx_last_error(NULL) returns (per thread): entry points that have no handle (cesx_create, cesx_comm_unique_id)
void set_global_error(const std::string& msg);

#define CESX_HIP(call)                                                              \
    do {                                                                            \
        hipError_t _e = (call);                                                     \
        if (_e != hipSuccess) {                                                     \
            e.err = std::string(#call) + ": " + hipGetErrorString(_e);              \
            return CESX_EHIP;                                                       \
        }                                                                           \
    } while (0)

}  // namespace cesx
