// K3, fp32 fast path -- the same GEMM  U_next = W . [U ; G ; xi] + b 1^T  as kernels_update.hip
// (ces/calibrate.py:443-447, :484-488, :515-527), restructured around the gfx950 LDS-DMA:
//
//  * W is read in "fragment-major" order (written that way by K2's assemble kernel): for every
//    16-column k-tile, 16 pieces of 1 KiB; piece (g, rb) holds, lane by lane, the four A operands
//    of k-steps 4g..4g+3 of row block rb.  One global_load_lds_dwordx4 per piece drops it into LDS
//    exactly as the waves read it back (ds_read_b128, lane-linear, conflict-free).  A wave loads
//    only the pieces it consumes itself.
//  * [U; G] tiles (16 rows x 128 particles, 512-B row segments) go global -> LDS the same way,
//    2 pieces per wave; no staging registers, no ds_write pass.
//  * particle 4 li + c of the tile is column li of MFMA block c: one ds_read_b128 yields the B
//    operands of all four blocks of a k-step, and the epilogue stores float4s.
//  * 3-slot LDS ring, loads two tiles ahead, ONE raw s_barrier per k-tile with a counted
//    s_waitcnt vmcnt(N) (the DMAs of the tile after next stay in flight across the barrier).
//  * xi tiles: Philox4x32-10 + Box-Muller into the ring slot one tile ahead; the lower-triangular
//    sqrt(2hk) L segment skips row blocks above the diagonal.
//
// All global -> LDS traffic is inline asm, so hipcc's own s_waitcnt bookkeeping never sees a DMA
// (it would otherwise drain the ring with vmcnt(0) at the first ds_read of every tile).
// Bound: MFMA (v_mfma_f32_32x32x2_f32).
#include "cesx_internal.h"
#include <hip/hip_ext.h>

namespace cesx {

constexpr int U2_THREADS = 256;
constexpr int U2_BK = 16;            // k-tile
constexpr int U2_BN = 128;           // particles per workgroup
constexpr int U2_RC = 256;           // output rows per workgroup
constexpr int U2_WSLOT = U2_RC * U2_BK * 4;      // 16 KiB
constexpr int U2_XSLOT = U2_BK * U2_BN * 4;      // 8 KiB
constexpr int U2_RING = 3;
#ifndef U2_ABL      // timing ablations (tools/update2_bench.hip); results are wrong when set
#define U2_ABL 0
#endif

struct Upd2Args {
    const float* Wf; int nkt; int out_rows; const float* bias;
    // K segments (no arrays: an indexable kernarg array sends the whole struct to scratch)
    const float *src0, *src1, *src2; int rows0, rows1, rows2; int kt1, kt2;      // first k-tile of segments 1, 2 (INT_MAX: absent)
    int kind0, kind1, kind2;
    long long J, j_offset;
    float* out;
    const float* add1; const double* c1p; double c1i;
    const float* add2; const double* c2p; double c2i;
    double* absmax_part;
    unsigned int seed_lo, seed_hi, step;
    const float* rowc; double* metric_part; int metric_seg;
    int tri_seg;
    int stagger_from;     // workgroups with a linear index >= this start late (see the kernel)
    int stagger_n;        // ... by this many s_sleep(100) = 6.4k cycles each
    long long* clk;       // profiled launches only: wave 0 of workgroup (0, 0) writes its {s_memtime, s_memrealtime} ticks
    const unsigned long long* fault; unsigned long long fault_seq;   // fault != nullptr and *fault == fault_seq: leave `out` untouched (UpdateOpt)
    const double* hkp; const double* s2p;      // HKF instantiations: the time step and sqrt(2 hk), read at run time (UpdateOpt)
};

// wait until at most `n` of this wave's DMAs are outstanding, retire its LDS traffic, barrier
__device__ __forceinline__ void ring_barrier(int n) {
    if (n >= 6)      asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else             asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// What the K loop needs to know about one k-tile, all wave-uniform (SGPRs); three of them (tiles
// kt, kt+1, kt+2) rotate through the loop, so the segment arithmetic runs once per tile.
struct TileD {
    int valid;           // tile index < nkt
    int noise;           // drawn in-kernel (else read from memory)
    int r0;              // first row of the tile inside its segment
    int rows;            // rows of the segment (rows >= r0 + 16 except in a ragged last tile)
    const float* base;   // segment base pointer
    int ndma;            // DMAs one wave issues for it
};

template <bool NOISE>
__device__ __forceinline__ TileD make_tile(int t, const Upd2Args& a) {
    // branch-free selects (a ternary chain over loaded values becomes a lookup table in scratch)
    const int b1 = t >= a.kt1 ? 1 : 0, b2 = t >= a.kt2 ? 1 : 0;
    TileD d;
    d.valid = t < a.nkt ? 1 : 0;
    d.noise = NOISE && (a.kind0 + b1 * (a.kind1 - a.kind0) + b2 * (a.kind2 - a.kind1)) != 0 ? 1 : 0;
    d.r0 = (t - (b1 * a.kt1 + b2 * (a.kt2 - a.kt1))) * U2_BK;
    d.rows = a.rows0 + b1 * (a.rows1 - a.rows0) + b2 * (a.rows2 - a.rows1);
    const long long p0 = (long long)a.src0, p1 = (long long)a.src1, p2 = (long long)a.src2;
    d.base = (const float*)(p0 + b1 * (p1 - p0) + b2 * (p2 - p1));
    d.ndma = d.valid ? (((!d.noise && !(U2_ABL & 1)) ? 2 : 0) + ((U2_ABL & 2) ? 0 : 4)) : 0;
    return d;
}

// One xi item (4 normals: rows 4q..4q+3 of one particle) generated in 7 stages, so that the
// stages can sit between MFMA groups and run in their shadow (a wave issues in order: 32
// back-to-back MFMAs followed by 150 VALU instructions overlap nothing).
struct NoiseItem { uint32_t c0, c1, c2, c3, k0, k1; };
__device__ __forceinline__ void noise_rounds2(NoiseItem& s) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint64_t p0 = (uint64_t)M0 * s.c0, p1 = (uint64_t)M1 * s.c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c1 ^ s.k0, n2 = (uint32_t)(p0 >> 32) ^ s.c3 ^ s.k1;
        s.c1 = (uint32_t)p1; s.c3 = (uint32_t)p0; s.c0 = n0; s.c2 = n2;
        s.k0 += W0; s.k1 += W1;
    }
}
// Box-Muller on one pair of the counter words (same arithmetic as normal4 in cesx_internal.h)
__device__ __forceinline__ void noise_pair(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float sc = 5.9604644775390625e-08f;   // 2^-24
    const float u0 = ((float)(a >> 8) + 0.5f) * sc, u1 = ((float)(b >> 8) + 0.5f) * sc;
    const float ra = bm_radius(u0);
    z0 = ra * __builtin_amdgcn_cosf(u1);
    z1 = ra * __builtin_amdgcn_sinf(u1);
}

// NOISE = false: every K segment is read from memory (xi injected or drawn ahead); the Philox stages and their
// branches are compiled out of the K loop.
// HKF: W carries no time step (Engine::d_Wq, segments [xi | U | G] = [L | a I - M + I/hk | -K]): behind the first
// segment the accumulators hold L xi and are scaled once by sqrt(2 hk) / hk, the other segments add (a I - M + I/hk) U - K G,
// and the epilogue multiplies (sum + b') by hk:  U_next = sqrt(2hk) L xi + U + hk ((a I - M) U - K G + b')
// (ces/calibrate.py:484-488 with the step size factored out of the coefficient matrix).
template <bool NOISE, bool HKF = false>
__global__ __launch_bounds__(U2_THREADS, 2)
void update2_kernel(const Upd2Args a) {
    // a polled join of the side stream that ran out in front of this launch (kernels_dense.hip): W is stale, the output stays as it was
    if (a.fault != nullptr && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.fault_seq) return;
    using acc_t = Mfma<float>::acc_t;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [ W ring 3 x 16 KiB | X ring 3 x 8 KiB | rowc kn x 16 B ]
    float* const sRowc = reinterpret_cast<float*>(smem + U2_RING * (U2_WSLOT + U2_XSLOT));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);

#ifdef U2_CLOCKS
    const long long clk0 = clock64(), wclk0 = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the clock this launch ran at: shader-clock and 100 MHz reference time stamps of one wave at its start and end
    // (cesx_profile_clock; the start stamps go straight to memory: nothing stays live in SGPRs across the K loop)
    if (a.clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
        const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c0; a.clk[1] = r0; }
    }
    float hkf = 1.f, cres = 1.f;
    if (HKF) {
        const double hk = *a.hkp;
        hkf = (float)hk;
        cres = (float)(*a.s2p / hk);
    }
    const int li = lane & 31, lh = lane >> 5;
    const long long jt0 = (long long)blockIdx.x * U2_BN;
    const int rc0 = blockIdx.y * U2_RC;
    const int nkt = a.nkt;
    // mirrored pair of row blocks (w, 7 - w): equal work for every wave in the triangular segment
    const int rb0 = wave, rb1 = 7 - wave;
    const bool on0 = rc0 + rb0 * 32 < a.out_rows, on1 = rc0 + rb1 * 32 < a.out_rows;
    const int tri_t0 = a.tri_seg == 0 ? 0 : a.tri_seg == 1 ? a.kt1 : a.tri_seg == 2 ? a.kt2 : 0x7fffffff;
    const int tri_t1 = a.tri_seg == 0 ? a.kt1 : a.tri_seg == 1 ? a.kt2 : a.tri_seg == 2 ? nkt : 0x7fffffff;
    // last row of each block, relative to the triangular segment's first column
    const int last0 = rc0 + rb0 * 32 + 31, last1 = rc0 + rb1 * 32 + 31;
    const bool do_metrics = a.metric_part != nullptr && blockIdx.y == 0;
    const int met_t0 = !do_metrics ? 0x7fffffff : a.metric_seg == 0 ? 0 : a.metric_seg == 1 ? a.kt1 : a.kt2;
    const int met_t1 = !do_metrics ? 0x7fffffff : a.metric_seg == 0 ? (a.kt1 < nkt ? a.kt1 : nkt)
                                                : a.metric_seg == 1 ? (a.kt2 < nkt ? a.kt2 : nkt) : nkt;

    // this lane's column inside the tile, clamped for the ragged last workgroup (J % 4 == 0)
    long long colc = jt0 + 4 * li;
    if (colc > a.J - 4) colc = a.J - 4;
    // fragment-major W: piece 0 of tile 0 (wave-uniform); + t * 16 KiB + piece * 1 KiB + lane * 16
    const char* const wlane = reinterpret_cast<const char*>(a.Wf) + ((size_t)blockIdx.y * nkt * 16) * 1024;
    // per-lane byte offsets of the two [U; G; xi] pieces this wave issues per tile, from the tile's first row
    const bool wide = a.J >= (1ll << 26);
    const unsigned xoff0 = wide ? 0u : (unsigned)(((long long)(2 * wave + lh) * a.J + colc) * 4);
    const unsigned xoff1 = wide ? 0u : (unsigned)(((long long)(2 * (wave + 4) + lh) * a.J + colc) * 4);
    const unsigned long long gj0 = (unsigned long long)(a.j_offset + jt0);

    // LDS byte addresses of the ring slots (wave-uniform), rotated with the tiles
    unsigned wsl0 = lds0, wsl1 = lds0 + U2_WSLOT, wsl2 = lds0 + 2 * U2_WSLOT;
    unsigned xsl0 = lds0 + U2_RING * U2_WSLOT, xsl1 = xsl0 + U2_XSLOT, xsl2 = xsl0 + 2 * U2_XSLOT;
    // the same slots as generic pointers for the ds_reads / noise stores
#define U2_LDSP(addr) (smem + ((addr) - lds0))

    // piece i of a tile (descriptor d, W image at wt) into ring slots (wsl, xsl):
    // i = 0, 1: rows 2q, 2q+1 of the [U; G] tile for q = wave + 4 i; i = 2..5: the W pieces
    // (g, rb) = (0, rb0), (0, rb1), (1, rb0), (1, rb1) that this wave alone consumes.
    // Addresses are scalar: a wave-uniform base (SGPR pair) + a per-lane byte offset fixed for the whole kernel
    // (xoff0 / xoff1 / lane * 16); only a ragged last tile of a segment (rows clamped per lane) and ensembles
    // of 2^26 particles or more take the per-lane 64-bit form.
#define U2_PIECE(i, d, wt, wsl, xsl) do {                                                                   \
        if ((d).valid) {                                                                                    \
            if ((i) < 2) {                                                                                  \
                if (!(d).noise && !(U2_ABL & 1)) {                                                          \
                    const int q_ = wave + 4 * (i);                                                          \
                    if ((d).r0 + U2_BK <= (d).rows && !wide) {                                              \
                        glds16s((d).base + (size_t)(d).r0 * a.J, (i) == 0 ? xoff0 : xoff1, (xsl) + q_ * 1024); \
                    } else {                                                                                \
                        int row_ = (d).r0 + 2 * q_ + lh;                                                    \
                        row_ = row_ < (d).rows ? row_ : (d).rows - 1;   /* padded rows meet zero columns of W */ \
                        glds16((d).base + (size_t)row_ * a.J + colc, (xsl) + q_ * 1024);                    \
                    }                                                                                       \
                }                                                                                           \
            } else if (!(U2_ABL & 2)) {                                                                     \
                const int piece_ = (((i) - 2) >> 1) * 8 + ((((i) - 2) & 1) ? rb1 : rb0);                    \
                glds16s((wt) + piece_ * 1024, lane * 16, (wsl) + piece_ * 1024);                            \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
    // xi tile by Philox4x32-10 + Box-Muller into ring slot xsl: item = (row quad, particle), 2 per thread
#define U2_NOISE(half, d, xsl) do {                                                                         \
        float* X_ = reinterpret_cast<float*>(U2_LDSP(xsl));                                                 \
        const int item_ = tid + U2_THREADS * (half);                                                        \
        const int jl_ = item_ % U2_BN, ql_ = item_ / U2_BN;                                                 \
        const unsigned long long gj_ = gj0 + jl_;                                                           \
        const uint4x r_ = philox4x32_10((uint32_t)gj_, (uint32_t)(gj_ >> 32), (unsigned)((d).r0 / 4 + ql_), \
                                        a.step, a.seed_lo, a.seed_hi);                                      \
        float z_[4];                                                                                        \
        normal4(r_, z_);                                                                                    \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) X_[(4 * ql_ + e_) * U2_BN + jl_] = z_[e_];         \
    } while (0)

    // stage k (0..6) of the item of half `half` of the xi tile d (ring slot xsl), when `on`
#define U2_NSTAGE(k, on, half, d, xsl, it) do { if (NOISE && (on)) {                                                   \
        const int item_ = tid + U2_THREADS * (half);                                                        \
        const int jl_ = item_ % U2_BN, ql_ = item_ / U2_BN;                                                 \
        if ((k) == 0) {                                                                                     \
            const unsigned long long gj_ = gj0 + jl_;                                                       \
            (it).c0 = (uint32_t)gj_; (it).c1 = (uint32_t)(gj_ >> 32);                                       \
            (it).c2 = (unsigned)((d).r0 / 4 + ql_); (it).c3 = a.step;                                       \
            (it).k0 = a.seed_lo; (it).k1 = a.seed_hi;                                                       \
        }                                                                                                   \
        if ((k) < 5) noise_rounds2(it);                                                                     \
        else {                                                                                              \
            float* X_ = reinterpret_cast<float*>(U2_LDSP(xsl)) + (4 * ql_ + ((k) == 5 ? 0 : 2)) * U2_BN + jl_; \
            float z0_, z1_;                                                                                 \
            if ((k) == 5) noise_pair((it).c0, (it).c1, z0_, z1_); else noise_pair((it).c2, (it).c3, z0_, z1_); \
            X_[0] = z0_; X_[U2_BN] = z1_;                                                                   \
        }                                                                                                   \
    } } while (0)

    acc_t acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0;

    if (do_metrics) {
        const int rows4 = (met_t1 - met_t0) * U2_BK * 4;
        for (int i = tid; i < rows4; i += U2_THREADS) sRowc[i] = a.rowc[i];
    }
    float mq_e[4] = {0, 0, 0, 0}, mq_r[4] = {0, 0, 0, 0};
#ifdef U2_CLOCKS
    long long clk_bar = 0, clk_loop0 = 0, clk_s1 = 0, clk_s2 = 0;
#endif

    // Software pipeline (one barrier B_kt per k-tile, in the MIDDLE of its MFMA stream):
    //   first half of iteration kt : read F1 = fragments of k-steps 4..7 of tile kt; MFMAs of k-steps
    //                                0..3 from F0, the DMAs of tile kt+2 issued one by one in their shadow
    //   B_kt                       : counted vmcnt (tile kt+1 landed, tile kt+2 stays in flight), barrier
    //   second half                : read F0 = fragments of k-steps 0..3 of tile kt+1; MFMAs 4..7 from F1
    // so every ds_read has 32 MFMAs between issue and use.  Slot (kt+2) % 3 was last read before
    // B_{kt-1}; tile kt+1 is read only after B_kt.
    // The two workgroups that share a CU start ~12k cycles apart: all workgroups have the same
    // amount of work, so without this they all reach the 128-KiB store epilogue together and the
    // matrix pipes idle while HBM drains; staggered, one workgroup's stores (and barrier / staging
    // stalls) are covered by the other's MFMAs (-3 % at C2).
    if ((int)(blockIdx.y * gridDim.x + blockIdx.x) >= a.stagger_from)
        for (int i = 0; i < a.stagger_n; ++i) __builtin_amdgcn_s_sleep(100);
    TileD d0 = make_tile<NOISE>(0, a), d1 = make_tile<NOISE>(1, a);
    const char* wt = wlane;                       // W image of the tile being issued
#pragma unroll
    for (int i = 0; i < 6; ++i) U2_PIECE(i, d0, wt, wsl0, xsl0);
    if (NOISE && d0.valid && d0.noise) { U2_NOISE(0, d0, xsl0); U2_NOISE(1, d0, xsl0); }
    wt += 16 * 1024;
#pragma unroll
    for (int i = 0; i < 6; ++i) U2_PIECE(i, d1, wt, wsl1, xsl1);
    ring_barrier(d1.ndma);
    f4 f0b[4], f0a0, f0a1;
    {
        const float* X0 = reinterpret_cast<const float*>(U2_LDSP(xsl0));
#pragma unroll
        for (int v = 0; v < 4; ++v) f0b[v] = *reinterpret_cast<const f4*>(X0 + (2 * v + lh) * U2_BN + 4 * li);
        f0a0 = *reinterpret_cast<const f4*>(U2_LDSP(wsl0) + rb0 * 1024 + lane * 16);
        f0a1 = *reinterpret_cast<const f4*>(U2_LDSP(wsl0) + rb1 * 1024 + lane * 16);
    }
    if (NOISE && d1.valid && d1.noise) U2_NOISE(0, d1, xsl1);

#ifdef U2_CLOCKS
    clk_loop0 = clock64();
#endif
#define U2_MFMA4(R, A, B) do { _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) \
        acc[R][c_] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, (B)[c_], acc[R][c_], 0, 0, 0); } while (0)
    for (int kt = 0; kt < nkt; ++kt) {
#ifdef U2_CLOCKS
        if (kt == a.kt1) clk_s1 = clock64();
        if (kt == a.kt2) clk_s2 = clock64();
#endif
        if (HKF && kt == a.kt1) {          // (wave-uniform, once per kernel: L xi -> sqrt(2/hk) L xi)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][c][e] *= cres;
        }
        wt += 16 * 1024;
        const TileD d2 = make_tile<NOISE>(kt + 2, a);
        const bool intri = kt >= tri_t0 && kt < tri_t1;
        const bool need0 = on0 && (!intri || last0 >= d0.r0);
        const bool need1 = on1 && (!intri || last1 >= d0.r0);
        const char* Wt = U2_LDSP(wsl0);
        const float* Xt = reinterpret_cast<const float*>(U2_LDSP(xsl0));

        f4 f1b[4], f1a0, f1a1;
        if (U2_ABL & 4) {
#pragma unroll
            for (int v = 0; v < 4; ++v) f1b[v] = f0b[v] + 1.f;
            f1a0 = f0a0 - 1.f; f1a1 = f0a1 + 2.f;
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) f1b[v] = *reinterpret_cast<const f4*>(Xt + (2 * (4 + v) + lh) * U2_BN + 4 * li);
            f1a0 = *reinterpret_cast<const f4*>(Wt + (8 + rb0) * 1024 + lane * 16);
            f1a1 = *reinterpret_cast<const f4*>(Wt + (8 + rb1) * 1024 + lane * 16);
        }
        // data metrics of the G tile in this slot: thread = (row t/32 + 8h, 4 particles).  The operands are read
        // here, with the fragments; the arithmetic sits behind the first 16 MFMAs (their 1k cycles cover the LDS
        // latency, so the wave does not stall on it with an idle matrix pipe)
        const bool met = kt >= met_t0 && kt < met_t1;
        f4 mx[2], mrc[2];
        if (met) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = (tid >> 5) + 8 * h;
                mx[h] = *reinterpret_cast<const f4*>(Xt + rr * U2_BN + 4 * (tid & 31));
                mrc[h] = *reinterpret_cast<const f4*>(sRowc + (size_t)(d0.r0 + rr) * 4);
            }
        }
        // first half: MFMAs of k-steps 0..3; in their shadow the DMAs of tile kt+2 and, stage by
        // stage, the second xi item of tile kt+1
        const bool g1 = d1.valid && d1.noise;
        NoiseItem it;
        if (need0) {
            U2_MFMA4(0, f0a0[0], f0b[0]); U2_PIECE(0, d2, wt, wsl2, xsl2); U2_NSTAGE(0, g1, 1, d1, xsl1, it);
            U2_MFMA4(0, f0a0[1], f0b[1]); U2_PIECE(1, d2, wt, wsl2, xsl2); U2_NSTAGE(1, g1, 1, d1, xsl1, it);
            U2_MFMA4(0, f0a0[2], f0b[2]); U2_PIECE(2, d2, wt, wsl2, xsl2); U2_NSTAGE(2, g1, 1, d1, xsl1, it);
            U2_MFMA4(0, f0a0[3], f0b[3]); U2_PIECE(3, d2, wt, wsl2, xsl2); U2_NSTAGE(3, g1, 1, d1, xsl1, it);
        } else {
            U2_PIECE(0, d2, wt, wsl2, xsl2); U2_PIECE(1, d2, wt, wsl2, xsl2);
            U2_PIECE(2, d2, wt, wsl2, xsl2); U2_PIECE(3, d2, wt, wsl2, xsl2);
            U2_NSTAGE(0, g1, 1, d1, xsl1, it); U2_NSTAGE(1, g1, 1, d1, xsl1, it);
            U2_NSTAGE(2, g1, 1, d1, xsl1, it); U2_NSTAGE(3, g1, 1, d1, xsl1, it);
        }
        if (met) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float b = mx[h][c] - mrc[h][0], r = mx[h][c] - mrc[h][1];
                    mq_e[c] += mrc[h][2] * b * b;
                    mq_r[c] += mrc[h][2] * r * r;
                }
        }
        if (need1) {
            U2_MFMA4(1, f0a1[0], f0b[0]); U2_PIECE(4, d2, wt, wsl2, xsl2); U2_NSTAGE(4, g1, 1, d1, xsl1, it);
            U2_MFMA4(1, f0a1[1], f0b[1]); U2_PIECE(5, d2, wt, wsl2, xsl2); U2_NSTAGE(5, g1, 1, d1, xsl1, it);
            U2_MFMA4(1, f0a1[2], f0b[2]); U2_NSTAGE(6, g1, 1, d1, xsl1, it);
            U2_MFMA4(1, f0a1[3], f0b[3]);
        } else {
            U2_PIECE(4, d2, wt, wsl2, xsl2); U2_PIECE(5, d2, wt, wsl2, xsl2);
            U2_NSTAGE(4, g1, 1, d1, xsl1, it); U2_NSTAGE(5, g1, 1, d1, xsl1, it); U2_NSTAGE(6, g1, 1, d1, xsl1, it);
        }
#ifdef U2_CLOCKS
        const long long tb0 = clock64();
#endif
        if (!(U2_ABL & 8)) ring_barrier(d2.ndma);
#ifdef U2_CLOCKS
        clk_bar += clock64() - tb0;
#endif
        // second half
        if (d1.valid && !(U2_ABL & 4)) {
            const float* Xn = reinterpret_cast<const float*>(U2_LDSP(xsl1));
#pragma unroll
            for (int v = 0; v < 4; ++v) f0b[v] = *reinterpret_cast<const f4*>(Xn + (2 * v + lh) * U2_BN + 4 * li);
            f0a0 = *reinterpret_cast<const f4*>(U2_LDSP(wsl1) + rb0 * 1024 + lane * 16);
            f0a1 = *reinterpret_cast<const f4*>(U2_LDSP(wsl1) + rb1 * 1024 + lane * 16);
        }
        // ... MFMAs of k-steps 4..7, and the first xi item of tile kt+2
        const bool g2 = d2.valid && d2.noise;
        if (need0) {
            U2_MFMA4(0, f1a0[0], f1b[0]); U2_NSTAGE(0, g2, 0, d2, xsl2, it);
            U2_MFMA4(0, f1a0[1], f1b[1]); U2_NSTAGE(1, g2, 0, d2, xsl2, it);
            U2_MFMA4(0, f1a0[2], f1b[2]); U2_NSTAGE(2, g2, 0, d2, xsl2, it);
            U2_MFMA4(0, f1a0[3], f1b[3]); U2_NSTAGE(3, g2, 0, d2, xsl2, it);
        } else {
            U2_NSTAGE(0, g2, 0, d2, xsl2, it); U2_NSTAGE(1, g2, 0, d2, xsl2, it);
            U2_NSTAGE(2, g2, 0, d2, xsl2, it); U2_NSTAGE(3, g2, 0, d2, xsl2, it);
        }
        if (need1) {
            U2_MFMA4(1, f1a1[0], f1b[0]); U2_NSTAGE(4, g2, 0, d2, xsl2, it);
            U2_MFMA4(1, f1a1[1], f1b[1]); U2_NSTAGE(5, g2, 0, d2, xsl2, it);
            U2_MFMA4(1, f1a1[2], f1b[2]); U2_NSTAGE(6, g2, 0, d2, xsl2, it);
            U2_MFMA4(1, f1a1[3], f1b[3]);
        } else {
            U2_NSTAGE(4, g2, 0, d2, xsl2, it); U2_NSTAGE(5, g2, 0, d2, xsl2, it); U2_NSTAGE(6, g2, 0, d2, xsl2, it);
        }
        // rotate tiles and ring slots
        d0 = d1; d1 = d2;
        { const unsigned t_ = wsl0; wsl0 = wsl1; wsl1 = wsl2; wsl2 = t_; }
        { const unsigned t_ = xsl0; xsl0 = xsl1; xsl1 = xsl2; xsl2 = t_; }
    }
#ifdef U2_CLOCKS
    const long long clk_loop1 = clock64();
#endif
    // the last iteration's barrier had nothing in flight; one more so that the ring can be reused
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#ifdef U2_CLOCKS
    const long long clk_ep0 = clock64();
#endif
    // epilogue: lane holds, for row (e&3) + 8 (e>>2) + 4 lh of each of its blocks, particles 4 li .. 4 li + 3
    const double c1 = a.add1 ? (a.c1p ? *a.c1p * a.c1i : a.c1i) : 0.0;
    const double c2 = a.add2 ? (a.c2p ? *a.c2p * a.c2i : a.c2i) : 0.0;
    const long long j = jt0 + 4 * li;
    float amax = 0.f;
    if (j < a.J) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rb = r == 0 ? rb0 : rb1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = rc0 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (i < a.out_rows) {
                    const float bi = a.bias ? a.bias[i] : 0.f;
                    const size_t o = (size_t)i * a.J + j;
                    f4 v = {acc[r][0][e] + bi, acc[r][1][e] + bi, acc[r][2][e] + bi, acc[r][3][e] + bi};
                    if (HKF) v *= hkf;
                    if (a.add1) v += (float)c1 * *reinterpret_cast<const f4*>(a.add1 + o);
                    if (a.add2) v += (float)c2 * *reinterpret_cast<const f4*>(a.add2 + o);
                    // (non-temporal: the 67 MB of U_next are not read again by this kernel; the lines leave L2 as they
                    //  are written instead of at the kernel boundary -- 0.4 % of the step at C2, bit-identical)
                    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(a.out + o));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float av = v[c] < 0 ? -v[c] : v[c];
                        amax = av > amax ? av : amax;
                    }
                }
            }
        }
    }
#ifdef U2_CLOCKS
    const long long clk_ep1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long clk_ep2 = clock64();
#endif
    if (do_metrics) {
        // combine the 8 row groups of every particle through LDS (the ring is idle now)
        float* comb = reinterpret_cast<float*>(smem);            // [2][8][128]
        const int grp = tid >> 5;
        *reinterpret_cast<f4*>(comb + grp * U2_BN + 4 * (tid & 31)) = f4{mq_e[0], mq_e[1], mq_e[2], mq_e[3]};
        *reinterpret_cast<f4*>(comb + 8 * U2_BN + grp * U2_BN + 4 * (tid & 31)) = f4{mq_r[0], mq_r[1], mq_r[2], mq_r[3]};
        __syncthreads();
        double se = 0.0, sr = 0.0;
        if (tid < U2_BN && jt0 + tid < a.J) {
            float qe = 0, qr = 0;
#pragma unroll
            for (int g = 0; g < 8; ++g) { qe += comb[g * U2_BN + tid]; qr += comb[8 * U2_BN + g * U2_BN + tid]; }
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
            const float other = __shfl_down(amax, o, 64);
            amax = other > amax ? other : amax;
        }
        double* red = reinterpret_cast<double*>(smem) + 16;
        if (lane == 0) red[wave] = (double)amax;
        __syncthreads();
        if (tid == 0) {
            double m = red[0];
            for (int w = 1; w < U2_THREADS / 64; ++w) m = red[w] > m ? red[w] : m;
            a.absmax_part[blockIdx.y * gridDim.x + blockIdx.x] = m;
        }
    }
    if (a.clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
        const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[2] = c1; a.clk[3] = r1; }
    }
#ifdef U2_CLOCKS
    if (tid == 0 && a.metric_part) {      // dev instrumentation: core-clock and 100 MHz wall-clock ticks of this workgroup
        a.metric_part[blockIdx.x * 2 + 0] = (double)(clock64() - clk0);
        a.metric_part[blockIdx.x * 2 + 1] = (double)(wall_clock64() - wclk0);
        a.metric_part[8192 + blockIdx.x * 2 + 0] = (double)clk_bar;
        a.metric_part[8192 + blockIdx.x * 2 + 1] = (double)(clk_loop1 - clk_loop0);
        a.metric_part[16384 + blockIdx.x * 2 + 0] = (double)(clk_loop0 - clk0);
        a.metric_part[16384 + blockIdx.x * 2 + 1] = (double)(wclk0 % 100000);
        a.metric_part[24576 + blockIdx.x * 2 + 0] = (double)(clk_s1 - clk_loop0);
        a.metric_part[24576 + blockIdx.x * 2 + 1] = (double)(clk_s2 - clk_s1);
        a.metric_part[32768 + blockIdx.x * 2 + 0] = (double)(clk_ep1 - clk_ep0);      // store issue
        a.metric_part[32768 + blockIdx.x * 2 + 1] = (double)(clk_ep2 - clk_ep1);      // ... until the stores are acknowledged
    }
#endif
#undef U2_PIECE
#undef U2_NSTAGE
#undef U2_NOISE
#undef U2_MFMA4
#undef U2_LDSP
}

// ---------------------------------------------------------------------------
// K3 for SMALL coefficient matrices (out_rows <= 64, ktot <= 192: the reference's own problem sizes -- Darcy p = 64,
// n = 50, examples/scripts/darcy-flow.py:97-105; BASELINE config C4).  update2_kernel walks 16-column k-tiles through a
// 3-slot ring with one barrier per tile and two of its eight row blocks per wave pair: with 64 output rows two of its four
// waves hold no block at all and every one of the 12 tiles is a DMA round trip with 32 MFMAs to hide behind (25 us at C4).
// Here the WHOLE problem of a workgroup -- 64 rows x 64 particles x ktot -- is LDS resident: every DMA of W and [U; G; xi]
// is issued up front (24 per wave), ONE wait, ONE barrier, then the MFMAs; wave (rb, cb) owns one 32 x 32 block.
// Same W image (wf_index), same arguments, same Philox counters as update2_kernel; the triangular segment's zero blocks
// are multiplied (0 x finite = 0).  Chosen by the SHAPE alone (launch_update2), so every call flow of a problem runs it.
// ---------------------------------------------------------------------------
constexpr int U2S_BN = 64;            // particles per workgroup
constexpr int U2S_MAX_KT = 12;        // ktot <= 192
template <bool NOISE, bool HKF>
__global__ __launch_bounds__(U2_THREADS)
void update2s_kernel(const Upd2Args a) {
    using acc_t = Mfma<float>::acc_t;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkt = a.nkt;
    // [ W: nkt x 4 pieces (g, rb) of 1 KiB | X: ktot rows x 64 particles | rowc ktot x 16 B | bias 64 | comb 2 x 4 x 64 floats ]
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned ldsx = lds0 + (unsigned)nkt * 4096u;
    const char* const sW = smem;
    float* const sX = reinterpret_cast<float*>(smem + (size_t)nkt * 4096);
    float* const sRowc = reinterpret_cast<float*>(smem + (size_t)nkt * 8192);
    float* const sBias = sRowc + (size_t)nkt * U2_BK * 4;
    float* const comb = sBias + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const long long jt0 = (long long)blockIdx.x * U2S_BN;
    const int rb = wave & 1, cb = wave >> 1;
    // What the workgroup reads into REGISTERS goes first (the fault word, hk, the row constants, the bias), then every DMA:
    // one batch in flight, one wait -- a load issued behind the DMAs would be waited for with them (vmcnt retires in order),
    // and five dependent round trips in front of the MFMAs were a third of this kernel's first version.
    // (a polled join of the side stream that ran out in front of this launch: W is stale, the output stays as it was)
    unsigned long long fword = 0;
    if (a.fault != nullptr) fword = __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double hk = 1.0, s2 = 1.0;
    if (HKF) { hk = *a.hkp; s2 = *a.s2p; }
    const double c1v = a.add1 && a.c1p ? *a.c1p : 1.0, c2v = a.add2 && a.c2p ? *a.c2p : 1.0;
    const bool do_metrics = a.metric_part != nullptr;
    const int met_t0 = a.metric_seg == 0 ? 0 : a.metric_seg == 1 ? a.kt1 : a.kt2;
    const int met_t1 = a.metric_seg == 0 ? (a.kt1 < nkt ? a.kt1 : nkt) : a.metric_seg == 1 ? (a.kt2 < nkt ? a.kt2 : nkt) : nkt;
    const int nrow_m = do_metrics ? (met_t1 - met_t0) * U2_BK : 0;          // <= 192 rows: at most 3 floats of rowc per thread
    float rcv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) rcv[q] = tid + q * U2_THREADS < nrow_m * 4 ? a.rowc[tid + q * U2_THREADS] : 0.f;
    const float biasv = (a.bias && tid < a.out_rows && tid < 64) ? a.bias[tid] : 0.f;
    // W: piece pw = (kt, g, rb') -> image piece (kt * 16 + g * 8 + rb')
    for (int pw = wave; pw < nkt * 4; pw += 4) {
        const int kt = pw >> 2, g = (pw >> 1) & 1, rbp = pw & 1;
        glds16s(reinterpret_cast<const char*>(a.Wf) + (size_t)(kt * 16 + g * 8 + rbp) * 1024, lane * 16, lds0 + pw * 1024);
    }
    // [U; G; xi]: piece px = 4 rows x 64 particles (256 B per row): lane = (row lane >> 4, 16-byte chunk lane & 15)
    long long colc = jt0 + 4 * (lane & 15);
    if (colc > a.J - 4) colc = a.J - 4;
    for (int px = wave; px < nkt * 4; px += 4) {
        const TileD d = make_tile<NOISE>(px >> 2, a);
        if (d.noise) continue;
        int row = d.r0 + (px & 3) * 4 + (lane >> 4);
        row = row < d.rows ? row : d.rows - 1;          // padded rows meet zero columns of W
        glds16(d.base + (size_t)row * a.J + colc, ldsx + px * 1024);
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c0; a.clk[1] = r0; }
    }
    if (NOISE) {
        // xi rows by Philox4x32-10 + Box-Muller: item = (row quad of the tile, particle), the counters of update2_kernel
        const unsigned long long gj0 = (unsigned long long)(a.j_offset + jt0);
        for (int kt = 0; kt < nkt; ++kt) {
            const TileD d = make_tile<NOISE>(kt, a);
            if (!d.noise) continue;
            const int jl = tid & 63, ql = tid >> 6;
            const unsigned long long gj = gj0 + jl;
            const uint4x r = philox4x32_10((uint32_t)gj, (uint32_t)(gj >> 32), (unsigned)(d.r0 / 4 + ql), a.step, a.seed_lo, a.seed_hi);
            float z[4];
            normal4(r, z);
#pragma unroll
            for (int e = 0; e < 4; ++e) sX[(size_t)(kt * 16 + 4 * ql + e) * U2S_BN + jl] = z[e];
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (tid + q * U2_THREADS < nrow_m * 4) sRowc[tid + q * U2_THREADS] = rcv[q];
    if (tid < 64) sBias[tid] = biasv;
    const bool faulted = a.fault != nullptr && fword == a.fault_seq;
    const float hkf = (float)hk, cres = (float)(s2 / hk);
    const double c1 = a.add1 ? c1v * a.c1i : 0.0, c2 = a.add2 ? c2v * a.c2i : 0.0;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (faulted) return;

    acc_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    const float* xb = sX + lh * U2S_BN + 32 * cb + li;
    const char* wb = sW + rb * 1024 + lane * 16;
    // operands of tile kt + 1 are read before the MFMAs of tile kt (one wave per SIMD: nothing else covers the LDS latency)
    f4 a0 = *reinterpret_cast<const f4*>(wb), a1 = *reinterpret_cast<const f4*>(wb + 2048);
    float b[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) b[v] = xb[2 * v * U2S_BN];
    for (int kt = 0; kt < nkt; ++kt) {
        if (HKF && kt == a.kt1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] *= cres;
        }
        const int kn = kt + 1 < nkt ? kt + 1 : kt;
        const f4 n0 = *reinterpret_cast<const f4*>(wb + (size_t)kn * 4096);
        const f4 n1 = *reinterpret_cast<const f4*>(wb + (size_t)kn * 4096 + 2048);
        const float* xk = xb + (size_t)kn * 16 * U2S_BN;
        float nb[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) nb[v] = xk[2 * v * U2S_BN];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[v], b[v], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[v], b[4 + v], acc, 0, 0, 0);
        a0 = n0; a1 = n1;
#pragma unroll
        for (int v = 0; v < 8; ++v) b[v] = nb[v];
    }
    // data metrics of the G rows (the whole tile is resident): thread = (particle tid & 63, row group tid >> 6)
    float mq_e = 0.f, mq_r = 0.f;
    if (do_metrics) {
        const float* xm = sX + (size_t)met_t0 * 16 * U2S_BN + (tid & 63);
        for (int rr = tid >> 6; rr < nrow_m; rr += 4) {
            const f4 rc = *reinterpret_cast<const f4*>(sRowc + (size_t)rr * 4);
            const float x = xm[(size_t)rr * U2S_BN];
            const float be = x - rc[0], br = x - rc[1];
            mq_e += rc[2] * be * be;
            mq_r += rc[2] * br * br;
        }
    }
    // epilogue: lane holds rows (e & 3) + 8 (e >> 2) + 4 lh of its block for particle 32 cb + li
    const long long j = jt0 + 32 * cb + li;
    float amax = 0.f;
    if (j < a.J) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (i < a.out_rows) {
                const size_t o = (size_t)i * a.J + j;
                float v = acc[e] + sBias[i];
                if (HKF) v *= hkf;
                if (a.add1) v += (float)c1 * a.add1[o];
                if (a.add2) v += (float)c2 * a.add2[o];
                a.out[o] = v;
                const float av = v < 0 ? -v : v;
                amax = av > amax ? av : amax;
            }
        }
    }
    if (do_metrics) {
        comb[(tid >> 6) * U2S_BN + (tid & 63)] = mq_e;
        comb[4 * U2S_BN + (tid >> 6) * U2S_BN + (tid & 63)] = mq_r;
        __syncthreads();
        if (wave == 0) {
            double se = 0.0, sr = 0.0;
            if (jt0 + lane < a.J) {
                float qe = 0, qr = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) { qe += comb[g * U2S_BN + lane]; qr += comb[4 * U2S_BN + g * U2S_BN + lane]; }
                se = (double)qe * (double)qe;
                sr = (double)qr * (double)qr;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sr += __shfl_down(sr, o, 64); }
            if (lane == 0) { a.metric_part[blockIdx.x * 2 + 0] = sr; a.metric_part[blockIdx.x * 2 + 1] = se; }
        }
        __syncthreads();
    }
    if (a.absmax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_down(amax, o, 64);
            amax = other > amax ? other : amax;
        }
        if (lane == 0) comb[wave] = amax;
        __syncthreads();
        if (tid == 0) {
            float m = comb[0];
            for (int w = 1; w < U2_THREADS / 64; ++w) m = comb[w] > m ? comb[w] : m;
            a.absmax_part[blockIdx.x] = (double)m;
        }
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c1k = __builtin_amdgcn_s_memtime(), r1k = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[2] = c1k; a.clk[3] = r1k; }
    }
}

bool update2_qualifies(const Engine& e, const void* U, const void* G, const void* xi, const void* Unext) {
    if (e.cfg.dtype != CESX_F32 || !e.update_v2) return false;
    if (e.J % 4 != 0 || e.J < 4 || e.ktot % U2_BK != 0) return false;
    if (U2_RING * (U2_WSLOT + U2_XSLOT) + e.kn * 16 > 80 * 1024) return false;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return U && G && Unext && al16(U) && al16(G) && al16(Unext) && (!xi || al16(xi));
}

int launch_update2(Engine& e, int out_rows, const void* Wf, int ktot, const void* bias,
                   const UpdateSrc* src, int nsrc,
                   const void* add1, const double* c1, double c1_imm,
                   const void* add2, const double* c2, double c2_imm,
                   void* out, double* absmax_part, uint64_t step_index, bool metrics,
                   const UpdateOpt& opt, hipStream_t s) {
    // qualifies: fp32, whole W, 16-byte aligned rows everywhere, LDS budget for two workgroups per CU
    if (e.cfg.dtype != CESX_F32 || !Wf || opt.ldw != 0 || nsrc < 1 || nsrc > 3) return -1;
    if (e.J % 4 != 0 || e.J < 4 || ktot % U2_BK != 0) return -1;
    const int lds = U2_RING * (U2_WSLOT + U2_XSLOT) + e.kn * 16;
    if (lds > 80 * 1024) return -1;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(Wf) || !al16(out) || (add1 && !al16(add1)) || (add2 && !al16(add2))) return -1;
    Upd2Args a{};
    a.Wf = (const float*)Wf; a.nkt = ktot / U2_BK; a.out_rows = out_rows; a.bias = (const float*)bias;
    const float* sp[3] = {nullptr, nullptr, nullptr};
    int rows[3] = {1, 1, 1}, kind[3] = {0, 0, 0}, kt0[3] = {0, 0x7fffffff, 0x7fffffff};
    int k0 = 0;
    a.tri_seg = -1;
    for (int i = 0; i < nsrc; ++i) {
        if (src[i].kind == 0 && (!src[i].ptr || !al16(src[i].ptr))) return -1;
        sp[i] = (const float*)src[i].ptr; rows[i] = src[i].rows; kind[i] = src[i].kind; kt0[i] = k0 / U2_BK;
        if (src[i].tri) a.tri_seg = i;
        k0 += (src[i].rows + U2_BK - 1) / U2_BK * U2_BK;
    }
    if (k0 != ktot) { e.err = "update: K segments do not add up to ktot"; return CESX_EINVAL; }
    a.src0 = sp[0]; a.src1 = sp[1]; a.src2 = sp[2];
    a.rows0 = rows[0]; a.rows1 = rows[1]; a.rows2 = rows[2];
    a.kind0 = kind[0]; a.kind1 = kind[1]; a.kind2 = kind[2];
    a.kt1 = kt0[1]; a.kt2 = kt0[2];
    a.J = e.J; a.j_offset = e.cfg.j_offset;
    a.out = (float*)out;
    a.add1 = (const float*)add1; a.c1p = c1; a.c1i = c1_imm;
    a.add2 = (const float*)add2; a.c2p = c2; a.c2i = c2_imm;
    a.absmax_part = absmax_part;
    a.seed_lo = (unsigned)e.cfg.seed; a.seed_hi = (unsigned)(e.cfg.seed >> 32); a.step = (unsigned)step_index;
    a.rowc = (const float*)e.d_rowc;
    a.metric_part = metrics ? e.d_metric_part : nullptr;
    a.metric_seg = opt.metric_seg;
    a.fault = opt.fault; a.fault_seq = opt.fault_seq;
    a.hkp = opt.hkp; a.s2p = opt.s2p;
    if (opt.hkp && (!opt.s2p || nsrc != 3 || a.tri_seg != 0 || add1 || add2)) { e.err = "update: the hk-free form needs [xi | U | G] with the triangular segment first"; return CESX_EINVAL; }
    const bool noise = kind[0] != 0 || kind[1] != 0 || kind[2] != 0;
    if (out_rows <= 64 && a.nkt <= U2S_MAX_KT && e.J < (1ll << 26) && e.update_small) {
        // small coefficient matrix: the whole problem of a workgroup LDS resident (update2s_kernel)
        const int lds_s = a.nkt * 8192 + (a.nkt * U2_BK * 4 + 64 + 2 * 4 * U2S_BN) * 4;
        dim3 grid_s((unsigned)((e.J + U2S_BN - 1) / U2S_BN));
        auto kern = opt.hkp ? (noise ? update2s_kernel<true, true> : update2s_kernel<false, true>)
                            : (noise ? update2s_kernel<true, false> : update2s_kernel<false, false>);
        CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
        e.last_update_grid_x = (int)grid_s.x;
        e.last_update_grid = (int)grid_s.x;
        {
            ProfScope prof(e, opt.prof, s, true);
            a.clk = (prof.a && prof.b) ? e.d_clk : nullptr;
            if (prof.on()) hipExtLaunchKernelGGL(kern, grid_s, dim3(U2_THREADS), (unsigned)lds_s, s, prof.a, prof.b, 0, a);
            else hipLaunchKernelGGL(kern, grid_s, dim3(U2_THREADS), lds_s, s, a);
        }
        CESX_HIP(hipGetLastError());
        return CESX_OK;
    }
    dim3 grid((unsigned)((e.J + U2_BN - 1) / U2_BN), (unsigned)((out_rows + U2_RC - 1) / U2_RC));
    // the dispatcher gives every CU one workgroup before any CU gets its second: from there on start late
    a.stagger_from = (long long)grid.x * grid.y > e.num_cus ? e.num_cus : 0x7fffffff;
    a.stagger_n = 2;
    auto kern = opt.hkp ? (noise ? update2_kernel<true, true> : update2_kernel<false, true>)
                        : (noise ? update2_kernel<true, false> : update2_kernel<false, false>);
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    e.last_update_grid_x = (int)grid.x;
    e.last_update_grid = (int)(grid.x * grid.y);
    {
        ProfScope prof(e, opt.prof, s, true);
        a.clk = (prof.a && prof.b) ? e.d_clk : nullptr;
        if (prof.on()) hipExtLaunchKernelGGL(kern, grid, dim3(U2_THREADS), (unsigned)lds, s, prof.a, prof.b, 0, a);
        else hipLaunchKernelGGL(kern, grid, dim3(U2_THREADS), lds, s, a);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

}  // namespace cesx
