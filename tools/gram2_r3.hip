// dev tool only: the round-3 LDS-DMA Gram kernel (32-row x 32-byte pieces, per-k-group panels), kept for A/B runs of
// tools/gram2_bench.hip against the current ces_amd/csrc/kernels_gram2.hip.  Not part of libcesx.so.
// K1, LDS-DMA form -- the same split-K Gram of the shifted stacked ensemble Z = [U - s_u ; G - s_g]
// as kernels_gram.hip (np.cov(U0) ces/calibrate.py:424/476/512, D = (1/J) E^T Gamma^{-1} R of
// :429/:461/:503 and np.cov(Geval) :440/:472 all factor through Z Z^T), fed the way
// K3 (kernels_update2.hip) is fed:
//
//  * raw rows of U / G go global -> LDS by DMA (global_load_lds_dwordx4).  No staging registers, no
//    ds_write pass, no subtract-and-store pass.
//  * LDS layout of a tile: one PANEL per k-group g (the 8 / 16 particles one fragment read covers), a panel row
//    = the CP 16-byte chunks of that group (f32: 2 chunks = 32 B, f64: 4 = 64 B), rows TILE-block by TILE-block:
//    byte (g, row, chunk c) = g * PANEL + row * CP * 16 + c * 16.  One DMA piece (64 lanes x 16 B, written
//    contiguously at lds_base + 16 lane) is exactly one block row of one panel; its lanes fetch 32- / 64-byte
//    segments of TILE different rows (the 4 / 2 pieces of a block row together read whole 128-B lines).
//    PANEL is a compile-time stride, so the k-group enters a fragment read as the IMMEDIATE offset of the
//    ds_read and the address registers depend on the block only: 2 address adds per block and tile instead of
//    2 per block and k-group (every VALU instruction in this loop is taken from the f32 MFMA rate).
//  * bank conflicts are removed at the SOURCE: the lane that fills physical chunk c' of row r fetches logical
//    chunk c' ^ swz(r), swz(r) = (r / RPB) % CP with RPB = rows per 256 B; readers apply the same XOR (it
//    depends on the lane only, not on g).  Every 16-lane group of a ds_read_b128 then covers all 64 banks
//    exactly once (f32 and f64 maps).
//  * the f32-input MFMA runs on the SIMD's f32 vector lanes: every VALU instruction in the K loop takes
//    its issue cycles away from the matrix pipe (tools/mfma_rate.hip: 64.0 cycles per MFMA with LDS-fed
//    operands, 72 with two v_sub per MFMA at 4 waves per SIMD, 89 at one).  So the centring shift is NOT
//    subtracted on the fragments (2 subs per MFMA); the wave that issued a DMA piece subtracts the shift
//    from it IN PLACE once it has landed (ds_read_b128 / 4 subs / ds_write_b128 per piece: 6x fewer
//    VALU instructions than on the fragments) and accumulates the first moments (row sums) on the way.
//    The MFMA loop is then LDS reads, one address add per read and MFMAs.
//  * two LDS slots: the DMAs of tile t+1 are issued before the MFMAs of tile t; after its MFMAs a wave
//    waits for its own pieces of tile t+1, shifts them, and joins the ONE barrier of the tile.
//
// Work partition, slab layout and the fp64 fixed-order reduce are those of kernels_gram.hip
// (GramPlan, gram_reduce_kernel).  Qualifies when J is a multiple of the tile width (32 f32 / 16 f64)
// and U, G are 16-byte aligned; otherwise the register-staged kernel runs.
// Bound: MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64).
#include "../ces_amd/csrc/cesx_internal.h"
#include <hip/hip_ext.h>

namespace cesx {

constexpr int G2R3_THREADS = 1024;           // 16 waves = 4 per SIMD
constexpr int G2R3_WAVES = G2R3_THREADS / 64;
constexpr int G2R3_ROWB = 128;               // bytes of one row in a tile: 32 f32 / 16 f64
constexpr int G2R3_MAX_ROWS = 512;           // = MAX_STAGE_ROWS of kernels_gram.hip (the plans are shared)
constexpr int G2R3_MAXP = G2R3_MAX_ROWS / 8 / G2R3_WAVES;      // DMA pieces per wave and tile (4)
constexpr int G2R3_SLOT = G2R3_MAX_ROWS * G2R3_ROWB;           // one tile in LDS: 64 KiB, a compile-time stride
#ifndef G2R3_ABL      // timing ablations (tools/gram2_bench.hip); results are wrong when set
#define G2R3_ABL 0
#endif
#ifndef G2R3_SHIFT_AT
#define G2R3_SHIFT_AT 1
#endif
#ifdef G2R3_CLOCKS
__device__ long long g_gram2r3_clk[4096 * 4];
__device__ long long g_gram2r3_bar[4096 * 16];      // per wave: cycles spent in the per-tile barrier
__device__ long long g_gram2r3_pro[4096 * 4];       // prologue phases of wave 0: tables | row table + sync | first DMA | shift + barrier
#endif

template <typename T>
__global__ __launch_bounds__(G2R3_THREADS, 4)
void gram2r3_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                  int p, int n, long long J, const int* __restrict__ type_hdr, int ntypes,
                  const int* __restrict__ rows_tab, const int* __restrict__ wblk,
                  T* __restrict__ slabs, double* __restrict__ rowsum_part) {
    using M = Mfma<T>;
    using vec_t = typename M::vec_t;
    using acc_t = typename M::acc_t;
    constexpr int TILE = M::TILE, VEC = M::VEC, NBW = GramCfg<T>::NBW;
    constexpr int KT = G2R3_ROWB / (int)sizeof(T);         // particles per tile
    constexpr int KL = 64 / TILE;                        // lane groups of a fragment read (k sub-blocks)
    constexpr int NGROUP = 8 / KL;                       // fragment reads per row and tile (k-groups)
    constexpr int CP = KL;                               // 16-byte chunks of one k-group per row
    constexpr int PROW = CP * 16;                        // bytes of a panel row
    constexpr int PANEL = G2R3_MAX_ROWS * PROW;            // f32 16 KiB, f64 32 KiB
    constexpr int RPB = 256 / PROW;                      // rows per 256 B of a panel
    static_assert(NGROUP * PANEL == G2R3_SLOT && TILE * PROW == 1024, "one DMA piece = one block row of one panel");

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef G2R3_CLOCKS
    const long long gclk0 = clock64(), gw0 = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int type = 0;
    for (int t = 1; t < ntypes; ++t)
        if ((int)blockIdx.x >= type_hdr[t * 8 + 4]) type = t;
    const int* hdr = type_hdr + type * 8;
    const int nrb = hdr[0], rows_off = hdr[1], blocks_off = hdr[2], nblk_t = hdr[3];
    const int slice = (int)blockIdx.x - hdr[4], nslices = hdr[5];
    const int slab0 = hdr[6], rs0 = hdr[7];
    const int nrows = nrb * TILE;
    const int P = p + n;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);

    // this wave's block list: (compact row block of A) | (of B) << 8, wave-uniform
    int iab[NBW];
    int nb = 0;
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int* e = wblk + (size_t)(blocks_off + wave * NBW + b) * 3;
        const int a = __builtin_amdgcn_readfirstlane(e[0]);
        const int c = __builtin_amdgcn_readfirstlane(e[1]);
        iab[b] = a | (c << 8);
        if (a >= 0) nb = b + 1;
    }
    nb = __builtin_amdgcn_readfirstlane(nb);

    acc_t acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) acc[b][r] = 0;
#ifdef G2R3_CLOCKS
    const long long gp1 = clock64();
#endif

    // J-slice of this workgroup in whole tiles (J % KT == 0)
    const long long ntiles = J / KT;
    const long long tps = (ntiles + nslices - 1) / nslices;
    const long long t0 = (long long)slice * tps;
    const long long t1 = t0 + tps < ntiles ? t0 + tps : ntiles;

    // per staged row: source pointer and shift.  Rows past P (padding of the last block row) read row 0
    // of U: their products land in rows / columns the reduce never reads.
    const T** rowptr = reinterpret_cast<const T**>(smem + 2 * G2R3_SLOT);
    T* rowshift = reinterpret_cast<T*>(smem + 2 * G2R3_SLOT + G2R3_MAX_ROWS * 8);
    for (int row = tid; row < nrows; row += G2R3_THREADS) {
        const int gr = (rows_tab[rows_off + row / TILE] & 0xffff) * TILE + row % TILE;
        const T* ptr = U;
        T sh = 0;
        if (gr < P) {
            ptr = gr < p ? U + (size_t)gr * J : G + (size_t)(gr - p) * J;
            sh = shift[gr];
        }
        rowptr[row] = ptr;
        rowshift[row] = sh;
    }
    __syncthreads();
#ifdef G2R3_CLOCKS
    const long long gp2 = clock64();
#endif

    // DMA pieces of this wave: piece q = wave + 16 i = (block row q / NGROUP, k-group q % NGROUP);
    // lane = (row of the block row, physical chunk).  Per-lane source pointers live in registers and advance by
    // one tile per issue (one 64-bit add per piece and tile).
    const int npieces = nrb * NGROUP;
    const int prow = lane / CP, pch = lane % CP;
    const T* gsrc[G2R3_MAXP];
    int poff[G2R3_MAXP];                                     // LDS byte offset of the piece inside a slot (wave-uniform)
#pragma unroll
    for (int i = 0; i < G2R3_MAXP; ++i) {
        const int q = wave + G2R3_WAVES * i;
        const int rb = q < npieces ? q / NGROUP : 0, g = q % NGROUP;
        const int chunk = g * CP + (pch ^ ((prow / RPB) % CP));      // the swizzle (see the header)
        // (G2R3_ABL & 16, timing only: every lane of a piece reads the block row's FIRST row -- one 128-byte line per
        //  piece instead of 32: what the line transactions of the row-scattered pieces cost)
        gsrc[i] = rowptr[rb * TILE + ((G2R3_ABL & 16) ? 0 : prow)] + t0 * KT + chunk * VEC;
        poff[i] = g * PANEL + rb * 1024;
    }
    auto issue_tile = [&](int slot) {
#pragma unroll
        for (int i = 0; i < G2R3_MAXP; ++i) {
            const int q = wave + G2R3_WAVES * i;
            if (q < npieces) {
                if (!(G2R3_ABL & 1)) glds16(gsrc[i], lds0 + slot * G2R3_SLOT + poff[i]);
                gsrc[i] += KT;
            }
        }
    };

    // in-place shift of this wave's own pieces (the lane that fetched a 16-byte chunk also shifts it)
    T psh[G2R3_MAXP], rs[G2R3_MAXP];
#pragma unroll
    for (int i = 0; i < G2R3_MAXP; ++i) {
        const int q = wave + G2R3_WAVES * i;
        psh[i] = q < npieces ? rowshift[(q / NGROUP) * TILE + prow] : (T)0;
        rs[i] = 0;
    }
    auto shift_tile = [&](int slot) {
        char* sb = smem + slot * G2R3_SLOT + lane * 16;
        vec_t v[G2R3_MAXP];
#pragma unroll
        for (int i = 0; i < G2R3_MAXP; ++i) {
            const int q = wave + G2R3_WAVES * i;
            if (q < npieces) v[i] = *reinterpret_cast<const vec_t*>(sb + poff[i]);
        }
#pragma unroll
        for (int i = 0; i < G2R3_MAXP; ++i) {
            const int q = wave + G2R3_WAVES * i;
            if (q < npieces) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) { v[i][c] -= psh[i]; rs[i] += v[i][c]; }
                *reinterpret_cast<vec_t*>(sb + poff[i]) = v[i];
            }
        }
    };

    // fragment reads: lane = (k sub-block lk, row li of the block) reads physical chunk lk ^ swz(li) of its panel
    // row; the k-group is the immediate offset g * PANEL
    const int li = lane % TILE, lk = lane / TILE;
    const int foff0 = li * PROW + ((lk ^ ((li / RPB) % CP)) << 4);

    // One (block, group) step: VEC MFMAs on fragments that were loaded one step earlier.  The loads of
    // the NEXT step are issued first (sched_barrier keeps them there), so every LDS read has VEC MFMAs
    // (256 / 128 cycles) between issue and use instead of an exposed lgkmcnt(0) in front of each MFMA.
    struct Frag { vec_t a, c; };
    auto load_frag = [&](Frag& f, const char* base, int b, int g) {
        if (G2R3_ABL & 4) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { f.a[v] = (T)(lane + v + b); f.c[v] = (T)(lane - v + g); }
            return;
        }
        f.a = *reinterpret_cast<const vec_t*>(base + (iab[b] & 0xff) * 1024 + foff0 + g * PANEL);
        f.c = *reinterpret_cast<const vec_t*>(base + (iab[b] >> 8) * 1024 + foff0 + g * PANEL);
    };

    // After which of its blocks a wave shifts its pieces of the next tile.  The SIMD issues the oldest wave first:
    // waves 0-3 run ahead and wait at the tile's barrier while 12-15 still multiply, so a shift pass placed
    // early in EVERY wave's own instruction stream lies in the middle of the tile in wall time for all but the
    // oldest (whose wait for the DMA is covered by the others' MFMAs); placed last it is exposed for the youngest.
    const int shift_at = __builtin_amdgcn_readfirstlane(G2R3_SHIFT_AT < 0 ? ((wave >> 2) + 1) * NBW / 4 : G2R3_SHIFT_AT);
    if (t0 < t1) {
        issue_tile(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef G2R3_CLOCKS
        if (tid == 0 && blockIdx.x < 4096) g_gram2r3_pro[blockIdx.x * 4 + 2] = clock64() - gp2;
#endif
        shift_tile(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef G2R3_CLOCKS
    if (tid == 0 && blockIdx.x < 4096) {
        g_gram2r3_pro[blockIdx.x * 4 + 0] = gp1 - gclk0; g_gram2r3_pro[blockIdx.x * 4 + 1] = gp2 - gp1;
        g_gram2r3_pro[blockIdx.x * 4 + 3] = clock64() - gp2;
    }
#endif
#ifdef G2R3_CLOCKS
    const long long gclk1 = clock64();
    long long gbar = 0;
#endif
    for (long long t = t0; t < t1; ++t) {
        const int cur = (int)((t - t0) & 1);
        if (t + 1 < t1) issue_tile(cur ^ 1);
        const char* base = smem + cur * G2R3_SLOT;
        Frag f0, f1;
        if (nb > 0) load_frag(f0, base, 0, 0);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            // This wave's pieces of tile t+1 (issued at the top of the tile, landed long since) are shifted in
            // place BETWEEN two of its blocks, at a different point for each of the 4 waves of a SIMD: the
            // LDS round trip of one wave's shift pass is covered by the MFMAs of the other three, instead of
            // all 16 waves running it together behind their last MFMA with the matrix pipes idle.
            if (b == shift_at && t + 1 < t1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(G2R3_ABL & 8)) shift_tile(cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (b < nb) {
#pragma unroll
                for (int g = 0; g < NGROUP; ++g) {
                    Frag& fc = (g & 1) ? f1 : f0;
                    Frag& fn = (g & 1) ? f0 : f1;
                    // prefetch the next step: group g+1 of this block, or group 0 of the next block
                    if (g + 1 < NGROUP) {
                        load_frag(fn, base, b, g + 1);
                    } else if (b + 1 < NBW) {
                        if (b + 1 < nb) load_frag(fn, base, b + 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[b] = M::mma(fc.a[v], fc.c[v], acc[b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (shift_at == NBW && t + 1 < t1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(G2R3_ABL & 8)) shift_tile(cur ^ 1);
        }
        static_assert(NGROUP % 2 == 0, "the fragment double buffer returns to f0 at every block boundary");
#ifdef G2R3_CLOCKS
        const long long tb0 = clock64();
#endif
        // the tile's one barrier (every wave has read slot `cur`, every piece of tile t+1 is shifted)
        if (!(G2R3_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef G2R3_CLOCKS
        gbar += clock64() - tb0;
#endif
    }
#ifdef G2R3_CLOCKS
    const long long gclk2 = clock64();
#endif

    // first moments of this slice: a row's 8 chunks sit in CP adjacent lanes of NGROUP different waves; combined
    // through LDS (the tile slots are idle now) in a fixed order.  Only the type that owns a block row reports it.
    {
        double* rsum = reinterpret_cast<double*>(smem);            // [nrows][NGROUP]
#pragma unroll
        for (int i = 0; i < G2R3_MAXP; ++i) {
            const int q = wave + G2R3_WAVES * i;
            double v = (double)rs[i];
#pragma unroll
            for (int o = 1; o < CP; o <<= 1) v += __shfl_xor(v, o, 64);
            if (q < npieces && pch == 0) rsum[((q / NGROUP) * TILE + prow) * NGROUP + q % NGROUP] = v;
        }
        __syncthreads();
        for (int row = tid; row < nrows; row += G2R3_THREADS) {
            const int ent = rows_tab[rows_off + row / TILE];
            const int gr = (ent & 0xffff) * TILE + row % TILE;
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < NGROUP; ++g) v += rsum[row * NGROUP + g];
            if ((ent >> 16) != 0 && gr < P) rowsum_part[(size_t)(rs0 + slice) * P + gr] = v;
        }
    }

    // partial blocks of this slice, accumulator-major (slab_group_rc): 16-byte stores of consecutive lanes
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        if (b < nb) {
            const int ob = __builtin_amdgcn_readfirstlane(wblk[(size_t)(blocks_off + wave * NBW + b) * 3 + 2]);
            T* out = slabs + ((size_t)slab0 + (size_t)slice * nblk_t + ob) * (TILE * TILE);
#pragma unroll
            for (int q = 0; q < M::NACC / VEC; ++q) {
                vec_t v;
#pragma unroll
                for (int c = 0; c < VEC; ++c) v[c] = acc[b][q * VEC + c];
                // (plain stores: the reduce launch right behind reads the slabs back; non-temporal stores, which drop
                //  the lines from L2, cost the step 1.6 %)
                *reinterpret_cast<vec_t*>(out + (size_t)(q * 64 + lane) * VEC) = v;
            }
        }
    }
#ifdef G2R3_CLOCKS
    if (lane == 0 && blockIdx.x < 4096) g_gram2r3_bar[blockIdx.x * 16 + wave] = gbar;
    if (tid == 0 && blockIdx.x < 4096) {
        g_gram2r3_clk[blockIdx.x * 4 + 0] = gclk1 - gclk0; g_gram2r3_clk[blockIdx.x * 4 + 1] = gclk2 - gclk1;
        g_gram2r3_clk[blockIdx.x * 4 + 2] = clock64() - gclk2; g_gram2r3_clk[blockIdx.x * 4 + 3] = wall_clock64() - gw0;
    }
#endif
}

}  // namespace cesx
