// K3, fp64 fast path -- the same GEMM  U_next = W . [U ; G ; xi] + b 1^T  as kernels_update.hip
// (ces/calibrate.py:443-447, :484-488, :515-527) for fp64 engines (BASELINE.json configs[4], and the
// drop-in class's default dtype), fed the way the fp32 kernel of kernels_update2.hip is fed, adapted
// to v_mfma_f64_16x16x4_f64 (64 cycles per instruction and SIMD, one double per lane and operand):
//
//  * [U; G; xi] tiles (16 rows x 64 particles, 512-B row segments) go global -> LDS by DMA
//    (global_load_lds_dwordx4), 2 pieces of 2 rows per wave and tile, into a 3-slot ring.  Particle
//    4 li + c of the tile is column li of MFMA block c, so two ds_read_b128 yield the B operands of
//    all four blocks of a k value and the epilogue stores 32 contiguous bytes per lane and row.  The
//    16-byte chunks of ODD tile rows are swapped pairwise at the source (chunk ^ 1): the 16-lane
//    groups of a ds_read_b128 then hit 16 different bank quads (rows 512 B apart would collide).
//  * W does NOT pass through LDS (a 256 x 16 fp64 tile is 32 KiB; three slots of it would leave room
//    for one workgroup per CU): every wave reads the A fragments of its own four row blocks straight
//    from a fragment-major image of W (written by K2's assemble kernel) with 16-byte loads -- W is
//    3 MB, L2-resident and shared by all workgroups -- one k-tile ahead, into a second register set.
//  * the noise segment is read from memory (the block drawn ahead by cesx_prefetch_noise, or the
//    injected one): fp64 Philox + Box-Muller (log, sincos) inside the loop would cost the matrix pipe
//    far more than it does in fp32.  A launch that has to draw its noise in-kernel takes the
//    register-staged kernel.
//  * one barrier per k-tile; two workgroups per CU cover each other's barrier and epilogue.
// Bound: MFMA.
#include "cesx_internal.h"
#include <hip/hip_ext.h>

namespace cesx {

constexpr int U3_THREADS = 256;
constexpr int U3_BK = 16;            // k-tile
constexpr int U3_BN = 64;            // particles per workgroup
constexpr int U3_RC = 256;           // output rows per workgroup
constexpr int U3_XSLOT = U3_BK * U3_BN * 8;      // 8 KiB
constexpr int U3_RING = 3;

struct Upd3Args {
    const double* Wd; int nkt; int out_rows; const double* bias;
    const double *src0, *src1, *src2; int rows0, rows1, rows2; int kt1, kt2;      // first k-tile of segments 1, 2 (INT_MAX: absent)
    long long J;
    double* out;
    const double* add1; const double* c1p; double c1i;
    const double* add2; const double* c2p; double c2i;
    double* absmax_part;
    const double* rowc; double* metric_part; int metric_seg;
    int tri_seg;
    long long* clk;       // profiled launches only: wave 0 of workgroup (0, 0) writes its {s_memtime, s_memrealtime} ticks
    const unsigned long long* fault; unsigned long long fault_seq;   // fault != nullptr and *fault == fault_seq: leave `out` untouched (UpdateOpt)
};

__global__ __launch_bounds__(U3_THREADS, 2)
void update3_kernel(const Upd3Args a) {
    // a polled join of the side stream that ran out in front of this launch (kernels_dense.hip): W is stale, the output stays as it was
    if (a.fault != nullptr && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.fault_seq) return;
    using d4 = double __attribute__((ext_vector_type(4)));
    using d2 = double __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [ X ring 3 x 8 KiB | per-thread metric sums 256 x 64 B | rowc kn x 32 B ]
    // (the eight per-particle metric sums of a thread live in LDS, not in registers: 16 of the kernel's 255 registers --
    //  it sat one register from spilling; they are touched in the 32 of 96 k-tiles that carry G rows only)
    double* const sMq = reinterpret_cast<double*>(smem + U3_RING * U3_XSLOT);
    double* const sRowc = sMq + U3_THREADS * 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (cesx_profile_clock; the start stamps go straight to memory: nothing stays live in SGPRs across the K loop)
    if (a.clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
        const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c0; a.clk[1] = r0; }
    }
    const int li = lane & 15, lr = lane >> 4;
    const long long jt0 = (long long)blockIdx.x * U3_BN;
    const int rc0 = blockIdx.y * U3_RC;
    const int nkt = a.nkt;
    // row blocks (16 rows) of this wave in mirrored pairs (w, 7-w, 8+w, 15-w): equal work in the triangular segment
    int rbk[4];
    rbk[0] = wave; rbk[1] = 7 - wave; rbk[2] = 8 + wave; rbk[3] = 15 - wave;
    bool rb_on[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb_on[r] = rc0 + rbk[r] * 16 < a.out_rows;
    const int tri_t0 = a.tri_seg == 0 ? 0 : a.tri_seg == 1 ? a.kt1 : a.tri_seg == 2 ? a.kt2 : 0x7fffffff;
    const int tri_t1 = a.tri_seg == 0 ? a.kt1 : a.tri_seg == 1 ? a.kt2 : a.tri_seg == 2 ? nkt : 0x7fffffff;
    const bool do_metrics = a.metric_part != nullptr && blockIdx.y == 0;
    const int met_t0 = !do_metrics ? 0x7fffffff : a.metric_seg == 0 ? 0 : a.metric_seg == 1 ? a.kt1 : a.kt2;
    const int met_t1 = !do_metrics ? 0x7fffffff : a.metric_seg == 0 ? (a.kt1 < nkt ? a.kt1 : nkt)
                                                : a.metric_seg == 1 ? (a.kt2 < nkt ? a.kt2 : nkt) : nkt;

    // X DMA: piece q (0..7) = tile rows 2q, 2q+1; this wave issues q = wave and wave + 4.  Lane d fills the 16-byte
    // chunk d % 32 of row 2q + d / 32; the odd row fetches chunk ^ 1 (see the header).  Ragged last workgroup
    // (J % 4 == 0): columns are clamped, the clamped lanes' results are never stored.
    const int drow = lane >> 5, dchunk = (lane & 31) ^ drow;
    long long colc = jt0 + 2 * dchunk;
    if (colc > a.J - 2) colc = a.J - 2;
    auto issue_x = [&](int t, int slot) {
        if (t >= nkt) return;
        const int b1 = t >= a.kt1 ? 1 : 0, b2 = t >= a.kt2 ? 1 : 0;
        const int r0 = (t - (b1 * a.kt1 + b2 * (a.kt2 - a.kt1))) * U3_BK;
        const int rows = a.rows0 + b1 * (a.rows1 - a.rows0) + b2 * (a.rows2 - a.rows1);
        const long long p0 = (long long)a.src0, p1 = (long long)a.src1, p2 = (long long)a.src2;
        const double* base = (const double*)(p0 + b1 * (p1 - p0) + b2 * (p2 - p1));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = wave + 4 * i;
            int row = r0 + 2 * q + drow;
            row = row < rows ? row : rows - 1;          // padded rows meet zero columns of W
            glds16(base + (size_t)row * a.J + colc, lds0 + slot * U3_XSLOT + q * 1024);
        }
    };
    // A fragments of k-tile t: for every row block of this wave two 16-byte loads (k-steps 0,1 and 2,3)
    const double* const wbase = a.Wd + ((size_t)blockIdx.y * nkt * 16) * 256 + lane * 2;     // 256 doubles per (rb, kt)
    auto load_a = [&](d2 (&af)[4][2], int t) {
        const int tt = t < nkt ? t : nkt - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
                af[r][sp] = *reinterpret_cast<const d2*>(wbase + ((size_t)tt * 16 + rbk[r]) * 256 + sp * 128);
    };

    d4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][c][e] = 0;

    if (do_metrics) {
        const int rows4 = (met_t1 - met_t0) * U3_BK * 4;
        for (int i = tid; i < rows4; i += U3_THREADS) sRowc[i] = a.rowc[i];
    }
    if (do_metrics) {
#pragma unroll
        for (int c = 0; c < 8; ++c) sMq[c * U3_THREADS + tid] = 0.0;       // (thread-private, [c][tid]: conflict-free)
    }

    d2 afA[4][2], afB[4][2];
    issue_x(0, 0);
    issue_x(1, 1);
    load_a(afA, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // byte offset of this lane's 32 bytes (particles 4 li .. 4 li + 3) in an even / odd tile row
    const int xoff = li * 32;
    auto tile_body = [&](int kt, int slot, d2 (&af)[4][2], d2 (&afn)[4][2]) {
        load_a(afn, kt + 1);
        const char* X = smem + slot * U3_XSLOT;
        const bool intri = kt >= tri_t0 && kt < tri_t1;
        const int k0l = (kt - tri_t0) * U3_BK;
        bool need[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) need[r] = rb_on[r] && (!intri || rc0 + rbk[r] * 16 + 15 >= k0l);
        if (kt >= met_t0 && kt < met_t1) {
            // data metrics of the G tile in this slot: thread = (row tid / 16, 4 particles)
            const int rr = tid >> 4;
            const char* rowp = X + rr * 512;
            const int sw = (rr & 1) * 16;
            const d2 xa = *reinterpret_cast<const d2*>(rowp + (((tid & 15) * 32 + 0) ^ sw));
            const d2 xb = *reinterpret_cast<const d2*>(rowp + (((tid & 15) * 32 + 16) ^ sw));
            const double xv[4] = {xa[0], xa[1], xb[0], xb[1]};
            const double* rc = sRowc + (size_t)((kt - met_t0) * U3_BK + rr) * 4;
            const double gb = rc[0], yy = rc[1], w = rc[2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double b = xv[c] - gb, r = xv[c] - yy;
                sMq[c * U3_THREADS + tid] += w * b * b;
                sMq[(4 + c) * U3_THREADS + tid] += w * r * r;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // The DMAs of tile kt + 2 are issued in the MIDDLE of the tile (its slot was last read during tile
            // kt - 1): the compiler's own vmcnt wait for the A fragments of the next tile also covers every
            // older vector-memory operation, so a DMA issued at the top of a tile would be waited for at once.
            if (s == 2) issue_x(kt + 2, (slot + 2) % U3_RING);
            // B operands of k value 4 s + lr for the four particle blocks
            const int krow = 4 * s + lr;
            const char* rowp = X + krow * 512;
            const int sw = (krow & 1) * 16;
            const d2 b01 = *reinterpret_cast<const d2*>(rowp + ((xoff + 0) ^ sw));
            const d2 b23 = *reinterpret_cast<const d2*>(rowp + ((xoff + 16) ^ sw));
            const double bv[4] = {b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (need[r]) {
                    const double av = af[r][s >> 1][s & 1];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[c], acc[r][c], 0, 0, 0);
                }
            }
        }
        // tile kt + 1 has landed for this wave once at most the 2 DMAs of tile kt + 2 and the 8 A loads of
        // tile kt + 1 (all younger) are outstanding; then every wave has also finished reading this slot
        if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else              asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile_body(kt, kt % U3_RING, afA, afB);
        if (kt + 1 < nkt) tile_body(kt + 1, (kt + 1) % U3_RING, afB, afA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // The epilogue takes its arguments from the kernarg segment AGAIN, through an opaque copy of the segment pointer: held in SGPRs
    // across the K loop they were 15 spilled registers (v_writelane in front of the loop, v_readlane behind it) and a nominal
    // 36-byte stack frame (profiles/r05_resource_usage.txt); the K loop itself keeps what it needs.
    typedef const __attribute__((address_space(4))) Upd3Args* kargp_t;
    kargp_t ea = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ea));
    // epilogue: lane holds, for row (lane >> 4) + 4 e of each of its blocks, particles 4 li .. 4 li + 3
    const double c1 = ea->add1 ? (ea->c1p ? *ea->c1p * ea->c1i : ea->c1i) : 0.0;
    const double c2 = ea->add2 ? (ea->c2p ? *ea->c2p * ea->c2i : ea->c2i) : 0.0;
    const long long j = jt0 + 4 * li;
    double amax = 0.0;
    if (j < ea->J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = rc0 + rbk[r] * 16 + lr + 4 * e;
                if (i < ea->out_rows) {
                    const double bi = ea->bias ? ea->bias[i] : 0.0;
                    const size_t o = (size_t)i * ea->J + j;
                    d2 v0 = {acc[r][0][e] + bi, acc[r][1][e] + bi}, v1 = {acc[r][2][e] + bi, acc[r][3][e] + bi};
                    if (ea->add1) { v0 += c1 * *reinterpret_cast<const d2*>(ea->add1 + o); v1 += c1 * *reinterpret_cast<const d2*>(ea->add1 + o + 2); }
                    if (ea->add2) { v0 += c2 * *reinterpret_cast<const d2*>(ea->add2 + o); v1 += c2 * *reinterpret_cast<const d2*>(ea->add2 + o + 2); }
                    *reinterpret_cast<d2*>(ea->out + o) = v0;
                    *reinterpret_cast<d2*>(ea->out + o + 2) = v1;
                    amax = fmax(amax, fmax(fmax(fabs(v0[0]), fabs(v0[1])), fmax(fabs(v1[0]), fabs(v1[1]))));
                }
            }
        }
    }
    if (do_metrics) {
        // combine the 16 row groups of every particle through LDS (the ring is idle now)
        __syncthreads();
        double* comb = reinterpret_cast<double*>(smem);           // [2][16][64]
        const int grp = tid >> 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            comb[grp * U3_BN + 4 * (tid & 15) + c] = sMq[c * U3_THREADS + tid];
            comb[16 * U3_BN + grp * U3_BN + 4 * (tid & 15) + c] = sMq[(4 + c) * U3_THREADS + tid];
        }
        __syncthreads();
        double se = 0.0, sr = 0.0;
        if (tid < U3_BN && jt0 + tid < ea->J) {
            double qe = 0, qr = 0;
#pragma unroll
            for (int g = 0; g < 16; ++g) { qe += comb[g * U3_BN + tid]; qr += comb[16 * U3_BN + g * U3_BN + tid]; }
            se = qe * qe;
            sr = qr * qr;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sr += __shfl_down(sr, o, 64); }
        __syncthreads();
        if (tid == 0) {                                            // (tid < 64: only wave 0 holds particles)
            ea->metric_part[blockIdx.x * 2 + 0] = sr;
            ea->metric_part[blockIdx.x * 2 + 1] = se;
        }
        __syncthreads();
    }
    if (ea->absmax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, 64));
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        if (tid == 0) ea->absmax_part[blockIdx.y * gridDim.x + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
    if (ea->clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
        const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { ea->clk[2] = c1; ea->clk[3] = r1; }
    }
}

// ---------------------------------------------------------------------------
// K3 in fp64 for SMALL coefficient matrices (out_rows <= 64, ktot <= 192; see update2s_kernel in kernels_update2.hip: the
// reference's own problem sizes, and fp64 is the drop-in class's default dtype).  update3_kernel gives each wave ONE of its
// four row blocks there and walks 12 k-tiles through the ring: 23 us at C4.  Here a workgroup owns 64 rows x 32 particles:
// the whole [U; G; xi] tile (ktot x 32 doubles <= 48 KiB) is LDS resident -- every DMA issued up front, one wait, one
// barrier --, wave w keeps ALL A fragments of row block w in registers (24 16-byte loads from the fragment-major image)
// and owns two 16 x 16 blocks.  Same image (wd_index) and arguments as update3_kernel; chosen by the shape alone.
// ---------------------------------------------------------------------------
constexpr int U3S_BN = 32;
constexpr int U3S_MAX_KT = 12;
__global__ __launch_bounds__(U3_THREADS)
void update3s_kernel(const Upd3Args a) {
    using d4 = double __attribute__((ext_vector_type(4)));
    using d2 = double __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkt = a.nkt;
    // [ X: ktot rows x 32 particles (256 B per row) | rowc ktot x 32 B | bias 64 | comb 2 x 8 x 32 ]
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const double* const sX = reinterpret_cast<const double*>(smem);
    double* const sRowc = reinterpret_cast<double*>(smem + (size_t)nkt * 4096);
    double* const sBias = sRowc + (size_t)nkt * U3_BK * 4;
    double* const comb = sBias + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lr = lane >> 4;
    const long long jt0 = (long long)blockIdx.x * U3S_BN;
    // what the workgroup reads into REGISTERS goes first (fault word, scalars, row constants, bias, the A fragments), then
    // every DMA: one batch in flight, one wait (see update2s_kernel)
    unsigned long long fword = 0;
    if (a.fault != nullptr) fword = __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double c1v = a.add1 && a.c1p ? *a.c1p : 1.0, c2v = a.add2 && a.c2p ? *a.c2p : 1.0;
    const bool do_metrics = a.metric_part != nullptr;
    const int met_t0 = a.metric_seg == 0 ? 0 : a.metric_seg == 1 ? a.kt1 : a.kt2;
    const int met_t1 = a.metric_seg == 0 ? (a.kt1 < nkt ? a.kt1 : nkt) : a.metric_seg == 1 ? (a.kt2 < nkt ? a.kt2 : nkt) : nkt;
    const int nrow_m = do_metrics ? (met_t1 - met_t0) * U3_BK : 0;          // <= 192 rows: at most 3 doubles of rowc per thread
    double rcv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) rcv[q] = tid + q * U3_THREADS < nrow_m * 4 ? a.rowc[tid + q * U3_THREADS] : 0.0;
    const double biasv = (a.bias && tid < a.out_rows && tid < 64) ? a.bias[tid] : 0.0;
    // A fragments of row block `wave`, every k-tile: two 16-byte loads per tile (k-steps 0,1 and 2,3)
    d2 af[U3S_MAX_KT][2];
    const double* const wbase = a.Wd + lane * 2 + (size_t)wave * 256;          // 256 doubles per (rb, kt)
#pragma unroll
    for (int t = 0; t < U3S_MAX_KT; ++t) {
        const int tt = t < nkt ? t : nkt - 1;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) af[t][sp] = *reinterpret_cast<const d2*>(wbase + (size_t)tt * 16 * 256 + sp * 128);
    }
    // [U; G; xi]: piece px = 4 rows x 32 particles; lane = (row lane >> 4, 16-byte chunk lane & 15)
    long long colc = jt0 + 2 * (lane & 15);
    if (colc > a.J - 2) colc = a.J - 2;
    for (int px = wave; px < nkt * 4; px += 4) {
        const int t = px >> 2;
        const int b1 = t >= a.kt1 ? 1 : 0, b2 = t >= a.kt2 ? 1 : 0;
        const int r0 = (t - (b1 * a.kt1 + b2 * (a.kt2 - a.kt1))) * U3_BK;
        const int rows = a.rows0 + b1 * (a.rows1 - a.rows0) + b2 * (a.rows2 - a.rows1);
        const long long p0 = (long long)a.src0, p1 = (long long)a.src1, p2 = (long long)a.src2;
        const double* base = (const double*)(p0 + b1 * (p1 - p0) + b2 * (p2 - p1));
        int row = r0 + (px & 3) * 4 + (lane >> 4);
        row = row < rows ? row : rows - 1;          // padded rows meet zero columns of W
        glds16(base + (size_t)row * a.J + colc, lds0 + px * 1024);
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c0; a.clk[1] = r0; }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (tid + q * U3_THREADS < nrow_m * 4) sRowc[tid + q * U3_THREADS] = rcv[q];
    if (tid < 64) sBias[tid] = biasv;
    const bool faulted = a.fault != nullptr && fword == a.fault_seq;
    const double c1 = a.add1 ? c1v * a.c1i : 0.0, c2 = a.add2 ? c2v * a.c2i : 0.0;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (faulted) return;

    d4 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = 0;
    const double* xb = sX + (size_t)lr * U3S_BN + li;
    // B operands of tile t + 1 are read before the MFMAs of tile t (the LDS latency has nothing else to hide behind)
    double bc[4][2], bn[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) { bc[s][0] = xb[(size_t)(4 * s) * U3S_BN]; bc[s][1] = xb[(size_t)(4 * s) * U3S_BN + 16]; }
#pragma unroll
    for (int t = 0; t < U3S_MAX_KT; ++t) {
        if (t < nkt) {
            const int tn = t + 1 < nkt ? t + 1 : t;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double* xk = xb + (size_t)(tn * 16 + 4 * s) * U3S_BN;
                bn[s][0] = xk[0]; bn[s][1] = xk[16];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double av = af[t][s >> 1][s & 1];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bc[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bc[s][1], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) { bc[s][0] = bn[s][0]; bc[s][1] = bn[s][1]; }
        }
    }
    // data metrics of the G rows: thread = (particle tid & 31, row group tid >> 5)
    double mq_e = 0.0, mq_r = 0.0;
    if (do_metrics) {
        const double* xm = sX + (size_t)met_t0 * 16 * U3S_BN + (tid & 31);
        for (int rr = tid >> 5; rr < nrow_m; rr += 8) {
            const double* rc = sRowc + (size_t)rr * 4;
            const double x = xm[(size_t)rr * U3S_BN];
            const double be = x - rc[0], br = x - rc[1];
            mq_e += rc[2] * be * be;
            mq_r += rc[2] * br * br;
        }
    }
    // epilogue: lane holds rows lr + 4 e of row block `wave` for particles 16 c + li
    double amax = 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const long long j = jt0 + 16 * c + li;
        if (j < a.J) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = wave * 16 + lr + 4 * e;
                if (i < a.out_rows) {
                    const size_t o = (size_t)i * a.J + j;
                    double v = acc[c][e] + sBias[i];
                    if (a.add1) v += c1 * a.add1[o];
                    if (a.add2) v += c2 * a.add2[o];
                    a.out[o] = v;
                    amax = fmax(amax, fabs(v));
                }
            }
        }
    }
    if (do_metrics) {
        comb[(tid >> 5) * U3S_BN + (tid & 31)] = mq_e;
        comb[8 * U3S_BN + (tid >> 5) * U3S_BN + (tid & 31)] = mq_r;
        __syncthreads();
        if (wave == 0) {
            double se = 0.0, sr = 0.0;
            if (lane < U3S_BN && jt0 + lane < a.J) {
                double qe = 0, qr = 0;
#pragma unroll
                for (int g = 0; g < 8; ++g) { qe += comb[g * U3S_BN + lane]; qr += comb[8 * U3S_BN + g * U3S_BN + lane]; }
                se = qe * qe;
                sr = qr * qr;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sr += __shfl_down(sr, o, 64); }
            if (lane == 0) { a.metric_part[blockIdx.x * 2 + 0] = sr; a.metric_part[blockIdx.x * 2 + 1] = se; }
        }
        __syncthreads();
    }
    if (a.absmax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, 64));
        if (lane == 0) comb[wave] = amax;
        __syncthreads();
        if (tid == 0) a.absmax_part[blockIdx.x] = fmax(fmax(comb[0], comb[1]), fmax(comb[2], comb[3]));
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c1k = __builtin_amdgcn_s_memtime(), r1k = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[2] = c1k; a.clk[3] = r1k; }
    }
}

// returns CESX_OK, an error, or -1 when the launch does not qualify (caller falls back to update_kernel)
int launch_update3(Engine& e, int out_rows, const void* Wd, int ktot, const void* bias,
                   const UpdateSrc* src, int nsrc,
                   const void* add1, const double* c1, double c1_imm,
                   const void* add2, const double* c2, double c2_imm,
                   void* out, double* absmax_part, bool metrics, const UpdateOpt& opt, hipStream_t s) {
    if (e.cfg.dtype != CESX_F64 || !Wd || opt.ldw != 0 || nsrc < 1 || nsrc > 3) return -1;
    if (e.J % 4 != 0 || e.J < 4 || ktot % U3_BK != 0) return -1;
    const int lds = U3_RING * U3_XSLOT + U3_THREADS * 64 + e.kn * 32;      // (the epilogue's 16 KiB metric scratch reuses the ring)
    if (lds > 78 * 1024) return -1;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(Wd) || !al16(out) || (add1 && !al16(add1)) || (add2 && !al16(add2))) return -1;
    Upd3Args a{};
    a.Wd = (const double*)Wd; a.nkt = ktot / U3_BK; a.out_rows = out_rows; a.bias = (const double*)bias;
    const double* sp[3] = {nullptr, nullptr, nullptr};
    int rows[3] = {1, 1, 1}, kt0[3] = {0, 0x7fffffff, 0x7fffffff};
    int k0 = 0;
    a.tri_seg = -1;
    for (int i = 0; i < nsrc; ++i) {
        if (src[i].kind != 0 || !src[i].ptr || !al16(src[i].ptr)) return -1;      // in-kernel noise: register-staged kernel
        sp[i] = (const double*)src[i].ptr; rows[i] = src[i].rows; kt0[i] = k0 / U3_BK;
        if (src[i].tri) a.tri_seg = i;
        k0 += (src[i].rows + U3_BK - 1) / U3_BK * U3_BK;
    }
    if (k0 != ktot) { e.err = "update: K segments do not add up to ktot"; return CESX_EINVAL; }
    for (int i = nsrc; i < 3; ++i) sp[i] = sp[0];
    a.src0 = sp[0]; a.src1 = sp[1]; a.src2 = sp[2];
    a.rows0 = rows[0]; a.rows1 = rows[1]; a.rows2 = rows[2];
    a.kt1 = kt0[1]; a.kt2 = kt0[2];
    a.J = e.J;
    a.out = (double*)out;
    a.add1 = (const double*)add1; a.c1p = c1; a.c1i = c1_imm;
    a.add2 = (const double*)add2; a.c2p = c2; a.c2i = c2_imm;
    a.absmax_part = absmax_part;
    a.rowc = (const double*)e.d_rowc;
    a.metric_part = metrics ? e.d_metric_part : nullptr;
    a.metric_seg = opt.metric_seg;
    a.fault = opt.fault; a.fault_seq = opt.fault_seq;
    if (out_rows <= 64 && a.nkt <= U3S_MAX_KT && e.update_small) {
        // small coefficient matrix: the whole tile of a workgroup LDS resident, W in registers (update3s_kernel)
        const int lds_s = a.nkt * 4096 + (a.nkt * U3_BK * 4 + 64 + 2 * 8 * U3S_BN) * 8;
        dim3 grid_s((unsigned)((e.J + U3S_BN - 1) / U3S_BN));
        CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(update3s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
        e.last_update_grid_x = (int)grid_s.x;
        e.last_update_grid = (int)grid_s.x;
        {
            ProfScope prof(e, opt.prof, s, true);
            a.clk = (prof.a && prof.b) ? e.d_clk : nullptr;
            if (prof.on()) hipExtLaunchKernelGGL(update3s_kernel, grid_s, dim3(U3_THREADS), (unsigned)lds_s, s, prof.a, prof.b, 0, a);
            else hipLaunchKernelGGL(update3s_kernel, grid_s, dim3(U3_THREADS), lds_s, s, a);
        }
        CESX_HIP(hipGetLastError());
        return CESX_OK;
    }
    dim3 grid((unsigned)((e.J + U3_BN - 1) / U3_BN), (unsigned)((out_rows + U3_RC - 1) / U3_RC));
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(update3_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    e.last_update_grid_x = (int)grid.x;
    e.last_update_grid = (int)(grid.x * grid.y);
    {
        ProfScope prof(e, opt.prof, s, true);
        a.clk = (prof.a && prof.b) ? e.d_clk : nullptr;
        if (prof.on()) hipExtLaunchKernelGGL(update3_kernel, grid, dim3(U3_THREADS), (unsigned)lds, s, prof.a, prof.b, 0, a);
        else hipLaunchKernelGGL(update3_kernel, grid, dim3(U3_THREADS), lds, s, a);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

}  // namespace cesx
