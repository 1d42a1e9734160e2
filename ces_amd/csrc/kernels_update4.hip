// K3 through the Cholesky factor (fp32, ALDI with the default time step, diagonal Sigma, 224 < p <= 256):
//
//     U_next = hk ( (1/hk + a) U  +  L ( sqrt(2/hk) xi - L^T Sigma^{-1} U )  -  K G  +  b' ),   b' = K y + M mu - a ubar
//
// which is ces/calibrate.py:484-488 with C = L L^T as factored at :476-478:  C Sigma^{-1} (U - mu) = L (L^T Sigma^{-1} U) - M mu.
// Two TRIANGULAR products (p^2 J flop each) replace the dense M U (2 p^2 J) of the hk-free form in kernels_update2.hip:
// 18.25 GFLOP executed at C2 instead of 22.0.
//
// The intermediate V = sqrt(2/hk) xi - L^T Sigma^{-1} U never leaves the registers.  A wave owns ALL eight 32-row blocks
// of ONE 32-particle block, and the k order of every product is the accumulator's own row order -- k-step i of a 32-row block
// is row 8 (i >> 2) + 4 (lane >> 5) + (i & 3), the C/D map of v_mfma_f32_32x32x2_f32 -- so accumulator register i of a finished
// block IS the B operand of k-step i of the second product: no LDS round trip, no shuffle.  Walking the 32-row blocks of U from
// the last to the first (kb = 7 .. 0):
//     P1(kb): V_r += (-L^T Sigma^{-1})_{r,kb} U_kb   for r <= kb      -> V_kb is complete (it needs the blocks kb .. 7 only)
//     P2(kb): O_r += L_{r,kb} V_kb                   for r >= kb
// P1(kb) and P2(kb + 1) are independent, and together they are 8 block products whatever kb is: tile (kb, h) = one 16-row half
// of both, 64 MFMAs per wave, 16 coefficient pieces of 1 KiB -- the shape of every other tile of this kernel (the -K G tiles
// follow).  Live accumulators: V_0..V_kb, O_kb..O_7 and V_kb+1 = 10 blocks = 160 registers; two workgroups per CU as before.
// Every wave does the same work in every tile (the row-block pairing of kernels_update2.hip left 24 against 18 block
// tiles in lockstep in its triangular segment).
//
// Coefficient image (Engine::d_Wq in the "chained" layout, wc_index_* in cesx_internal.h): tile t = 2 (8 - cb) + h holds
// column block cb of the 256 x 288 matrix  [ -L^T Sigma^{-1} (blocks above cb) ; L (blocks from cb down) ], written by the
// factorisation itself as it finishes its panels; the -K tiles by tail_aldi_kernel.  U / G tiles go global -> LDS by DMA
// (swizzled: rows with bit 2 set keep their 16-byte chunks rotated by 128 B, so the two half-waves of a fragment read hit
// disjoint banks); xi goes straight into registers in the accumulator layout (it is never an MFMA operand by itself).
// Ring: 3 slots, loads two tiles ahead, one barrier per tile in the middle of its MFMA stream; the second half of a slot's
// coefficient pieces is re-filled BEHIND that barrier (its readers are past it), the first half in front of it.
// Bound: MFMA (v_mfma_f32_32x32x2_f32).
#include "cesx_internal.h"
#include <hip/hip_ext.h>
#include <utility>

namespace cesx {

constexpr int U4_THREADS = 256;
constexpr int U4_BN = 128;                 // particles per workgroup (4 waves x 32)
constexpr int U4_ASLOT = 16 * 1024;        // 16 coefficient pieces (g, b) of 1 KiB
constexpr int U4_XSLOT = 8 * 1024;         // 16 rows x 128 particles
constexpr int U4_RING = 3;
constexpr int U4_TRI = 18;                 // tiles of the two triangular products
#ifndef U4_ABL      // timing ablations (tools/update4_bench.hip); results are wrong when set
#define U4_ABL 0
#endif

struct Upd4Args {
    const float* Wc; int ng;               // chained image, (18 + ng) tiles of 16 KiB; ng = k-tiles of the G segment
    int p, n;
    const float *U, *G, *xi;
    const float* bias;                     // b' [256]
    long long J;
    float* out;
    const float* rowc; double* metric_part;
    const double *hkp, *s2p, *alphap;
    int stagger_from, stagger_n;
    long long* clk;
    const unsigned long long* fault; unsigned long long fault_seq;
};

// one 4-byte global load per lane from a wave-uniform base + a per-lane byte offset, invisible to the compiler's
// s_waitcnt bookkeeping like the LDS-DMA pieces (a load the compiler tracks would be waited for with a vmcnt that does
// not count the DMAs issued behind it): the value is valid only behind u4_tie8 behind an explicit vmcnt wait
__device__ __forceinline__ void gld32s(float& dst, const void* sbase, unsigned voff) {
    asm volatile("global_load_dword %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gld32(float& dst, const void* gsrc) {
    asm volatile("global_load_dword %0, %1, off" : "=&v"(dst) : "v"(gsrc) : "memory");
}
__device__ __forceinline__ void u4_tie8(float* x) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
template <int N> __device__ __forceinline__ void u4_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

template <class F, int... Is>
__device__ __forceinline__ void u4_unroll(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// loads a wave issues for tile tt in front of the barrier (coefficient pieces of k-steps 0..3, the U / G rows, xi)
__host__ __device__ constexpr int u4_n0(int tt, int ntiles) {
    return tt >= ntiles ? 0 : tt < 16 ? 2 + 2 + 8 : tt < U4_TRI ? 2 : 4;
}

__global__ __launch_bounds__(U4_THREADS, 2)
void update4_kernel(const Upd4Args a) {
    // a polled join of the side stream that ran out in front of this launch (kernels_dense.hip): the image is stale, the output stays as it was
    if (a.fault != nullptr && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.fault_seq) return;
    using acc_t = Mfma<float>::acc_t;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [ A ring 3 x 16 KiB | X ring 3 x 8 KiB | rowc ng x 16 rows x 16 B | bias 256 x 4 B ]
    float* const sRowc = reinterpret_cast<float*>(smem + U4_RING * (U4_ASLOT + U4_XSLOT));
    float* const sBias = sRowc + (size_t)a.ng * 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned ldsx = lds0 + U4_RING * U4_ASLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
#ifdef U4_CLOCKS
    const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[0] = c0; a.clk[1] = r0; }
    }
    const double hk = *a.hkp;
    const float hkf = (float)hk, cxi = (float)(*a.s2p / hk), cd = (float)(1.0 / hk + *a.alphap);
    const long long jt0 = (long long)blockIdx.x * U4_BN;
    const int ng = a.ng, ntiles = U4_TRI + ng;
    const bool do_metrics = a.metric_part != nullptr;

    // ---- addressing, all of it fixed for the kernel ----
    // X pieces (1 KiB = rows 2 q, 2 q + 1 of the tile, q = wave and wave + 4): lane = (row lane >> 5, PHYSICAL chunk lane & 31),
    // which holds logical chunk (lane & 31) ^ 8 [row bit 2] -- bit 2 of rows 2 q, 2 q + 1 is bit 1 of q = bit 1 of the wave
    const int swz = ((wave >> 1) & 1) * 8;
    long long colx = jt0 + 4 * ((lane & 31) ^ swz);
    if (colx > a.J - 4) colx = a.J - 4;                        // ragged last workgroup (J % 4 == 0): columns that are not stored
    const unsigned xoff0 = (unsigned)(((long long)(2 * wave + lh) * a.J + colx) * 4);
    const unsigned xoff1 = (unsigned)(((long long)(2 * (wave + 4) + lh) * a.J + colx) * 4);
    // this lane's particle: column li of the wave's block; rows 8 q + 4 lh + j of every 16-row tile (q = 0, 1; j = 0..3)
    long long colp = jt0 + 32 * wave + li;
    const bool col_ok = colp < a.J;
    if (!col_ok) colp = a.J - 1;
    const unsigned poff = (unsigned)(((long long)(4 * lh) * a.J + colp) * 4);
    // fragment reads of the X slots: byte offset of (row 4 lh, this lane's particle), chunk rotated for the rows with bit 2 set
    const unsigned xrd = (unsigned)(4 * lh * 512 + (((8 * wave + (li >> 2)) ^ (8 * lh)) * 16) + (li & 3) * 4);
    const char* const wbase = reinterpret_cast<const char*>(a.Wc);
    const long long rowJ = a.J * 4;                           // bytes per ensemble row

    // ---- this wave's share of the loads of tile tt (ring slot sl) ----
    // in front of the tile barrier: coefficient pieces (0, wave), (0, wave + 4); the two X pieces; 8 xi registers
    auto issue_a = [&](int tt, int sl, int g) __attribute__((always_inline)) {
        if (U4_ABL & 2) return;
        const char* src = wbase + (size_t)tt * U4_ASLOT + (g * 8 + wave) * 1024;
        const unsigned dst = lds0 + sl * U4_ASLOT + (g * 8 + wave) * 1024;
        glds16s(src, lane * 16, dst);
        glds16s(src + 4096, lane * 16, dst + 4096);
    };
    auto issue_x = [&](int tt, int sl) __attribute__((always_inline)) {
        if (U4_ABL & 1) return;
        const float* base; int r0, rows;
        if (tt < 16) { base = a.U; r0 = (7 - (tt >> 1)) * 32 + (tt & 1) * 16; rows = a.p; }
        else { base = a.G; r0 = (tt - U4_TRI) * 16; rows = a.n; }
        const unsigned dst = ldsx + sl * U4_XSLOT;
        if ((tt >= 2 && tt < 16) || r0 + 16 <= rows) {        // (p > 224: only the last 32-row block of U can be ragged)
            const char* rb = reinterpret_cast<const char*>(base) + (long long)r0 * rowJ;
            glds16s(rb, xoff0, dst + wave * 1024);
            glds16s(rb, xoff1, dst + (wave + 4) * 1024);
        } else {                                             // ragged last tile of a segment: padded rows meet zero columns of the image
            int ra = r0 + 2 * wave + lh, rb_ = r0 + 2 * (wave + 4) + lh;
            ra = ra < rows ? ra : rows - 1; rb_ = rb_ < rows ? rb_ : rows - 1;
            glds16(base + (size_t)ra * a.J + colx, dst + wave * 1024);
            glds16(base + (size_t)rb_ * a.J + colx, dst + (wave + 4) * 1024);
        }
    };
    auto issue_xi = [&](int tt, float* x, int q) __attribute__((always_inline)) {      // registers 4 q .. 4 q + 3 of the tile's eight
        const int r0 = (7 - (tt >> 1)) * 32 + (tt & 1) * 16 + 8 * q;
        if (tt >= 2 || r0 + 8 <= a.p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gld32s(x[4 * q + j], reinterpret_cast<const char*>(a.xi) + (long long)(r0 + j) * rowJ, poff);
        } else {                                             // ragged last block: rows >= p meet zero columns of L, any finite value will do
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int r = r0 + j + 4 * lh;
                r = r < a.p ? r : a.p - 1;
                gld32(x[4 * q + j], a.xi + (size_t)r * a.J + colp);
            }
        }
    };

    acc_t V[9], O[8];
    float af[4][4];          // coefficient fragments of four consecutive block steps
    float uf[8], ufn[8];     // B fragments (U / G rows) of this tile / the next
    float xr[2][8];          // xi registers of the tiles of either parity
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) V[b][e] = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[b][e] = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { uf[e] = 0; ufn[e] = 0; xr[0][e] = 0; xr[1][e] = 0; }

    // coefficient fragment of block step (tile slot sl, group g, block b)
    auto read_a = [&](float* dst, int sl, int g, int b) __attribute__((always_inline)) {
        const f4 v = *reinterpret_cast<const f4*>(smem + sl * U4_ASLOT + (g * 8 + b) * 1024 + lane * 16);
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    };
    auto read_x = [&](float* dst, int sl) __attribute__((always_inline)) {
        const char* xb = smem + U4_RING * U4_ASLOT + sl * U4_XSLOT + xrd;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[4 * q + j] = *reinterpret_cast<const float*>(xb + (8 * q + j) * 512);
    };

    // the two workgroups of a CU start apart (kernels_update2.hip: one's stores and barrier stalls under the other's MFMAs)
    if ((int)blockIdx.x >= a.stagger_from)
        for (int i = 0; i < a.stagger_n; ++i) __builtin_amdgcn_s_sleep(100);

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed ----
    issue_a(0, 0, 0); issue_x(0, 0); issue_xi(0, xr[0], 0); issue_xi(0, xr[0], 1); issue_a(0, 0, 1);
    issue_a(1, 1, 0); issue_x(1, 1); issue_xi(1, xr[1], 0); issue_xi(1, xr[1], 1); issue_a(1, 1, 1);
    // (the row constants and the bias BEHIND the first tiles' loads: one round trip to memory in front of the first MFMA, not two)
    if (do_metrics)
        for (int i = tid; i < ng * 64; i += U4_THREADS) sRowc[i] = a.rowc[i];
    sBias[tid] = a.bias[tid];
    u4_barrier<(U4_ABL & 3) ? 0 : 14>();
    u4_tie8(xr[0]);
    read_a(af[0], 0, 0, 0); read_a(af[1], 0, 0, 1); read_a(af[2], 0, 0, 2);
    read_x(uf, 0);

#ifdef U4_CLOCKS
    const long long ck1 = clock64();
#endif
    // ---- the two triangular products: tiles 0 .. 17, fully unrolled (the accumulators change roles) ----
    auto tri_tile = [&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        constexpr int CB = 8 - T / 2, KB = CB - 1, H = T & 1;
        constexpr int SL = T % 3, SLN = (T + 1) % 3, SL2 = (T + 2) % 3;
        float* const xcur = xr[T & 1];
        if constexpr (KB >= 0) {
            // xi of this tile's rows into V_kb (sqrt(2/hk) xi - L^T Sigma^{-1} U), the diagonal term (1/hk + a) U into O_kb
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                V[KB][8 * H + e] = __builtin_fmaf(cxi, xcur[e], V[KB][8 * H + e]);
                O[KB][8 * H + e] = cd * uf[e];
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int s = T * 16 + g * 8 + b;
                // the fragment three block steps ahead (the next tile's slot once this tile's barrier is passed)
                {
                    const int s3 = s + 3, t3 = s3 >> 4, g3 = (s3 >> 3) & 1, b3 = s3 & 7;
                    if (!(U4_ABL & 4)) read_a(af[s3 & 3], t3 == T ? SL : SLN, g3, b3);
                }
                float* const fa = af[s & 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (b <= KB) V[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], uf[4 * g + j], V[b], 0, 0, 0);
                    else         O[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], V[CB][8 * H + 4 * g + j], O[b], 0, 0, 0);
                }
                // loads of tile T + 2, one by one in the MFMAs' shadow
                if (T + 2 < ntiles) {
                    if (g == 0) {
                        if (b == 0) issue_a(T + 2, SL2, 0);
                        if (b == 2 && T + 2 < 16) issue_x(T + 2, SL2);
                        if (b == 2 && T + 2 >= U4_TRI) issue_x(T + 2, SL2);
                        if (b == 4 && T + 2 < 16) issue_xi(T + 2, xcur, 0);
                        if (b == 6 && T + 2 < 16) issue_xi(T + 2, xcur, 1);
                    } else if (b == 1) issue_a(T + 2, SL2, 1);
                }
            }
            if (g == 0) {
                // everything of tile T + 1 has landed (only what this tile issued above may still be in flight)
                if (U4_ABL & 3) u4_barrier<0>();
                else if (T + 2 >= ntiles) u4_barrier<0>();
                else u4_barrier<u4_n0(T + 2, 1 << 30)>();
                u4_tie8(xr[(T + 1) & 1]);
                if (T + 1 < 16 || T + 1 >= U4_TRI) read_x(ufn, SLN);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) uf[e] = ufn[e];
    };
    u4_unroll(tri_tile, std::make_integer_sequence<int, U4_TRI>{});

#ifdef U4_CLOCKS
    const long long ck2 = clock64();
#endif
    // ---- - K G: ng dense tiles; the data metrics ride on the G fragments ----
    f2 mq = {0.f, 0.f};          // {sum w (g - gbar)^2, sum w (g - y)^2} over this lane's rows of its particle
    int sl = 0;                  // ring slot of tile 18 + gt: (18 + gt) % 3 = gt % 3
    for (int gt = 0; gt < ng; ++gt) {
        const int sln = sl == 2 ? 0 : sl + 1, sl2 = sln == 2 ? 0 : sln + 1;
        const int tt = U4_TRI + gt;
        const bool more2 = gt + 2 < ng;
        f4 rc[8];
        if (do_metrics) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) rc[4 * q + j] = *reinterpret_cast<const f4*>(sRowc + (size_t)(16 * gt + 8 * q + 4 * lh + j) * 4);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int s = g * 8 + b, s3 = s + 3;
                if (!(U4_ABL & 4)) read_a(af[s3 & 3], (s3 >> 4) ? sln : sl, (s3 >> 3) & 1, s3 & 7);
                float* const fa = af[s & 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) O[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], uf[4 * g + j], O[b], 0, 0, 0);
                if (more2) {
                    if (g == 0 && b == 0) issue_a(tt + 2, sl2, 0);
                    if (g == 0 && b == 2) issue_x(tt + 2, sl2);
                    if (g == 1 && b == 1) issue_a(tt + 2, sl2, 1);
                }
                if (g == 1 && do_metrics) {          // behind MFMAs: one G row of this lane per block step
                    const float x = uf[b];
                    const f2 d = f2{x, x} - f2{rc[b][0], rc[b][1]};
                    mq += (d * d) * rc[b][2];
                }
            }
            if (g == 0) {
                if ((U4_ABL & 3) || !more2) u4_barrier<0>(); else u4_barrier<4>();
                if (gt + 1 < ng) read_x(ufn, sln);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) uf[e] = ufn[e];
        sl = sln;
    }
    // (the last tile's barrier had nothing in flight; one more so that the ring can be reused)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#ifdef U4_CLOCKS
    const long long ck3 = clock64();
#endif
    // ---- epilogue: U_next = hk (sum + b'); lane holds rows 32 b + 8 (e >> 2) + 4 lh + (e & 3) of particle colp ----
    if (col_ok) {
        float* const ob = a.out + colp;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i0 = 32 * b + 8 * q + 4 * lh;
                const f4 bi = *reinterpret_cast<const f4*>(sBias + i0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i0 + j < a.p) {
                        const float v = hkf * (O[b][4 * q + j] + bi[j]);
                        __builtin_nontemporal_store(v, ob + (size_t)(i0 + j) * a.J);
                    }
                }
            }
        }
    }
    if (do_metrics) {
        // the two half-waves hold the two halves of a particle's rows; then the squares, summed over the workgroup's particles
        float qe = mq[0] + __shfl_xor(mq[0], 32, 64), qr = mq[1] + __shfl_xor(mq[1], 32, 64);
        double se = 0.0, sr = 0.0;
        if (lh == 0 && col_ok) { se = (double)qe * (double)qe; sr = (double)qr * (double)qr; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sr += __shfl_down(sr, o, 64); }
        double* redm = reinterpret_cast<double*>(smem);
        if (lane == 0) { redm[wave] = sr; redm[4 + wave] = se; }
        __syncthreads();
        if (tid == 0) {
            a.metric_part[blockIdx.x * 2 + 0] = redm[0] + redm[1] + redm[2] + redm[3];
            a.metric_part[blockIdx.x * 2 + 1] = redm[4] + redm[5] + redm[6] + redm[7];
        }
    }
    if (a.clk != nullptr && blockIdx.x == 0 && wave == 0) {
        const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.clk[2] = c1; a.clk[3] = r1; }
    }
#ifdef U4_CLOCKS
    {   // dev instrumentation (tools/update4_bench.hip): cycles of wave 0's phases, wall-clock ticks (100 MHz) of the workgroup
        const long long ck4 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long ck5 = clock64();
        if (tid == 0 && a.metric_part) {
            double* m = a.metric_part + 16384 + (size_t)blockIdx.x * 8;
            m[0] = (double)(ck1 - ck0); m[1] = (double)(ck2 - ck1); m[2] = (double)(ck3 - ck2); m[3] = (double)(ck4 - ck3);
            m[4] = (double)(ck5 - ck4); m[5] = (double)(wall_clock64() - wk0); m[6] = (double)(wk0 % 1000000);
        }
    }
#endif
}

// the shapes the chained form takes (decided once per problem, cesx_set_problem): fp32, eight 32-row blocks, a G segment whose
// row constants fit beside the ring, 32-bit DMA offsets; pointer alignment is checked per call by update2_qualifies
bool update4_shape_ok(const Engine& e) {
    if (e.cfg.dtype != CESX_F32 || !e.update_v2 || e.d_Wq == nullptr) return false;
    if (e.p <= 224 || e.p > 256 || e.rpad != 256 || e.kn > 256) return false;
    return e.J % 4 == 0 && e.J >= 4 && e.J < (1ll << 26);
}

int launch_update4(Engine& e, const void* U, const void* G, const void* xi, void* out, bool metrics, const UpdateOpt& opt, hipStream_t s) {
    if (!e.chain || !U || !G || !xi || !out || !opt.hkp || !opt.s2p) { e.err = "update4: not a chained hk-free launch"; return CESX_EINVAL; }
    Upd4Args a{};
    a.Wc = (const float*)e.d_Wq; a.ng = e.kn / 16; a.p = e.p; a.n = e.n;
    a.U = (const float*)U; a.G = (const float*)G; a.xi = (const float*)xi;
    a.bias = (const float*)e.d_bias;
    a.J = e.J;
    a.out = (float*)out;
    a.rowc = (const float*)e.d_rowc;
    a.metric_part = metrics ? e.d_metric_part : nullptr;
    a.hkp = opt.hkp; a.s2p = opt.s2p; a.alphap = &e.d_scal->alpha;
    a.fault = opt.fault; a.fault_seq = opt.fault_seq;
    const int lds = U4_RING * (U4_ASLOT + U4_XSLOT) + e.kn * 16 + 1024;
    dim3 grid((unsigned)((e.J + U4_BN - 1) / U4_BN));
    // the dispatcher gives every CU one workgroup before any CU gets its second: from there on start late
    a.stagger_from = (long long)grid.x > e.num_cus ? e.num_cus : 0x7fffffff;
    a.stagger_n = 2;          // (0 ... 4 x 6.4k cycles: no difference at C2, tools/ab_env.py round 6; 7: the CU's store path)
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(update4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    e.last_update_grid_x = (int)grid.x;
    e.last_update_grid = (int)grid.x;
    {
        ProfScope prof(e, opt.prof, s, true);
        a.clk = (prof.a && prof.b) ? e.d_clk : nullptr;
        if (prof.on()) hipExtLaunchKernelGGL(update4_kernel, grid, dim3(U4_THREADS), (unsigned)lds, s, prof.a, prof.b, 0, a);
        else hipLaunchKernelGGL(update4_kernel, grid, dim3(U4_THREADS), lds, s, a);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

}  // namespace cesx
