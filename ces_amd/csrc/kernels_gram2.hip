// K1, LDS-DMA form -- the same split-K Gram of the shifted stacked ensemble Z = [U - s_u ; G - s_g]
// as kernels_gram.hip (np.cov(U0) ces/calibrate.py:424/476/512, D = (1/J) E^T Gamma^{-1} R of
// :429/:461/:503 and np.cov(Geval) :440/:472 all factor through Z Z^T), fed the way
// K3 (kernels_update2.hip) is fed:
//
//  * raw rows of U / G go global -> LDS by DMA (global_load_lds_dwordx4).  No staging registers, no
//    ds_write pass, no subtract-and-store pass.
//  * one DMA piece (64 lanes x 16 B, written contiguously at lds_base + 16 lane) fetches 8 rows x 128 B: EIGHT WHOLE
//    cache lines (round 4; the round-3 pieces were 32 rows x 32 B -- 32 lines touched per KiB, and their issue
//    cost 9-10 % of the second launch, profiles/r03_gram2_ablation.txt).  A tile (KT particles = 128 B per row)
//    of a block row is TILE / 8 pieces, block rows follow each other: byte (block row rb, piece q) = rb * BLKB +
//    q * 1024.
//  * inside a piece the 64 cells (row r8 = 4 r2 + r of 8, 16-byte chunk c of 8) are PERMUTED -- the DMA writes
//    lane l at 16 l, but which cell lane l fetches is free -- so that (a) four consecutive lanes fetch 64
//    CONTIGUOUS bytes of one row (one line per lane quad: the texture addresser works on quads; a first version
//    of this round whose quads spanned four rows ran 4 % (f32) / 13 % (f64) SLOWER than round 3's 32-byte runs),
//    (b) every 16-lane group of a ds_read_b128 covers the 64 banks exactly once (the groups are {0-3,12-15,20-27},
//    {4-11,16-19,28-31} and the same + 32), and (c) as much of the k-group g of a fragment read (the 8 / 16
//    particles one read of every lane covers) as possible is an additive term, i.e. the IMMEDIATE offset of the
//    ds_read.  With lk = the lane's k sub-block of the MFMA operand, byte of a cell inside its piece:
//        f32 (32-row blocks, q = 0..3, c = 4 g1 + 2 g0 + lk):
//            512 g1 + 256 r2 + 64 r + 32 (g0 ^ (q & 1)) + 16 (lk ^ (q >> 1))      g1 immediate, g0 in the address
//        f64 (16-row blocks, q = 0..1, c = 4 g + lk, lk = 2 lk1 + lk0):
//            512 g + 256 r2 + 64 r + 32 (lk1 ^ r2) + 16 lk0                       g immediate
//    (16 lanes of a group = 4 consecutive rows r of 4 (q, r2) combinations; they differ in the two low
//    XORed bits, so the 16 slots of the 256-byte bank row are hit once each.  With conflict-free reads every
//    bank row holds cells of ONE immediate k-group only, which is why an f32 quad -- 2 k-groups x 2 sub-blocks --
//    needs the second address register.)
//  * the f32-input MFMA runs on the SIMD's f32 vector lanes: every VALU instruction in the K loop takes
//    its issue cycles away from the matrix pipe (tools/mfma_rate.hip: 64.0 cycles per MFMA with LDS-fed
//    operands, 72 with two v_sub per MFMA at 4 waves per SIMD, 89 at one).  So the centring shift is NOT
//    subtracted on the fragments (2 subs per MFMA); the wave that issued a DMA piece subtracts the shift
//    from it IN PLACE once it has landed (ds_read_b128 / 4 subs / ds_write_b128 per piece: 6x fewer
//    VALU instructions than on the fragments) and accumulates the first moments (row sums) on the way: a piece
//    holds whole rows, so a row's sum lives in 8 lanes of ONE wave (no LDS round trip at the end).
//    The MFMA loop is then LDS reads, one address add per read and MFMAs.
//  * two LDS slots: the DMAs of tile t+1 are issued before the MFMAs of tile t; after its MFMAs a wave
//    waits for its own pieces of tile t+1, shifts them, and joins the ONE barrier of the tile.
//  * no table in LDS and no barrier in front of the first DMA: a lane derives the row it fetches from the
//    type's row list (a scalar load per piece).
//
// Work partition, slab layout and the fp64 fixed-order reduce are those of kernels_gram.hip
// (GramPlan, gram_reduce_kernel).  Qualifies when J is a multiple of the tile width (32 f32 / 16 f64)
// and U, G are 16-byte aligned; otherwise the register-staged kernel runs.
// Bound: MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64).
#include "cesx_internal.h"
#include <hip/hip_ext.h>
#include <type_traits>

namespace cesx {

constexpr int G2_THREADS = 1024;           // 16 waves = 4 per SIMD
constexpr int G2_WAVES = G2_THREADS / 64;
constexpr int G2_ROWB = 128;               // bytes of one row in a tile: 32 f32 / 16 f64
constexpr int G2_MAX_ROWS = 512;           // = MAX_STAGE_ROWS of kernels_gram.hip (the plans are shared)
constexpr int G2_MAXP = G2_MAX_ROWS / 8 / G2_WAVES;      // DMA pieces per wave and tile (4)
constexpr int G2_SLOT = G2_MAX_ROWS * G2_ROWB;           // one tile in LDS: 64 KiB, a compile-time stride
constexpr int G2_SLOT_IMM = 60 * 1024;                   // ... of launches whose types stage <= 480 rows: slot 1 + k-group within the ds_read offset field
#ifndef G2_ABL      // timing ablations (tools/gram2_bench.hip); results are wrong when set
#define G2_ABL 0
#endif
// After which of its blocks a wave waits for its DMA pieces of the next tile and shifts them (see `shift_at` in the kernel;
// NBW = behind its last block, < 0 = staggered by wave class).  tools/gram2_bench.hip -DG2_SHIFT_AT=k, round 5, us first + second
// launch:   f32 (C2 shape, NBW = 4)  1: 48.2 + 115.9 | 2: 48.2 + 116.3 | 3: 48.5 + 116.0 | 4: 47.8 + 116.6   (noise)      -> 1
//           f64 (C5 shape, NBW = 8)  1: 155 + 503 | 2: 153 + 481 | 4: 154 + 488 | 6: 152 + 479 | 8: 149 + 466 | staggered 158 + 498,
//                                    staggered the other way round (oldest waves last) 162 + 510                                  -> 8
// An f64 block is 4 MFMAs: behind its first block a wave's pieces, issued one MFMA group earlier, have had ~0.4 us to land
// and the wave stalls on them; an f32 block is 16 MFMAs and the pieces are there.
#ifdef G2_SHIFT_AT
#define G2_SHIFT_AT_F32 G2_SHIFT_AT
#define G2_SHIFT_AT_F64 G2_SHIFT_AT
#endif
#ifndef G2_SHIFT_AT_F32
#define G2_SHIFT_AT_F32 1
#endif
#ifndef G2_SHIFT_AT_F64
#define G2_SHIFT_AT_F64 8
#endif
#ifndef G2_OPT      // dev A/B switches (tools/gram2_bench.hip): 1 = next tile's DMA issued behind the first MFMA group,
#define G2_OPT 15   // 2 = row sums only where the type reports them, 4 = scalar DMA addressing, 8 = a block's partial sums stored
                    // as soon as its last MFMA of the slice is issued (no barrier behind the last tile), 16 = those stores non-temporal
#endif
// Which wave-uniform tests of the tile loop are re-evaluated at their use instead of being kept as hoisted lane masks (see
// `opaque` in the kernel): 1 DMA issue, 2 shift pass; tile body: 4 `more`, 8 the block count, 16 the shift position.
// Measured with tools/gram2_bench.hip (round 5, two boxes, C2 / C5 shapes; us first + second launch):
//   f32:  round-4 code 49.4 + 120.8 | 11: 48.5 + 119.9 | 31: 48.8 + 113.9, 48.7 + 112.2      -> 31
//   f64:  round-4 code 166 + 547    | 31: 172 + 549    | 8: 157 + 504 | 11: 157 + 505         -> 11
// (f64 holds 8 blocks per wave: re-evaluating `more` and the shift position per block costs more scalar work than it frees
//  registers; every instantiation is free of scratch and SGPR spills with either setting, tests/test_isa_audit.py)
#ifndef G2_OPQ_F32
#define G2_OPQ_F32 31
#endif
#ifndef G2_OPQ_F64
#define G2_OPQ_F64 11
#endif
#ifdef G2_CLOCKS
__device__ long long g_gram2_clk[4096 * 4];
__device__ long long g_gram2_bar[4096 * 16];      // per wave: cycles spent in the per-tile barrier
__device__ long long g_gram2_pro[4096 * 4];       // prologue phases of wave 0: tables | addresses | first DMA | shift + barrier
#endif

template <typename T, bool SG, bool IMM>
__global__ __launch_bounds__(G2_THREADS, 4)
void gram2_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                  int p, int n, long long J, const int* __restrict__ type_hdr, int ntypes,
                  const int* __restrict__ rows_tab, const int* __restrict__ wblk,
                  T* __restrict__ slabs, double* __restrict__ rowsum_part) {
    using M = Mfma<T>;
    using vec_t = typename M::vec_t;
    using acc_t = typename M::acc_t;
    constexpr int TILE = M::TILE, VEC = M::VEC, NBW = GramCfg<T>::NBW;
    constexpr int KT = G2_ROWB / (int)sizeof(T);         // particles per tile
    constexpr int KL = 64 / TILE;                        // lane groups of a fragment read (k sub-blocks): 2 / 4
    constexpr int NGROUP = 8 / KL;                       // fragment reads per row and tile (k-groups): 4 / 2
    constexpr int PPB = TILE / 8;                        // DMA pieces per block row and tile: 4 / 2
    constexpr int BLKB = TILE * G2_ROWB;                 // bytes of a block row in a slot: 4 KiB / 2 KiB
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int G2_OPQ = F32 ? G2_OPQ_F32 : G2_OPQ_F64;
    constexpr int GSTR = 512;                            // bytes between immediate k-groups inside a piece (f32: g1, f64: g)
    constexpr int NG0 = NGROUP / 2;                      // k-groups folded into the lane address (f32: g0 = 0, 1; f64: none)
    static_assert(NG0 == (F32 ? 2 : 1), "cell maps of the header");
    // IMM (every type of the launch stages <= 60 KiB per tile): the second slot starts G2_SLOT_IMM bytes behind the first,
    // so slot + immediate k-group fit the 16-bit offset field of a ds_read and EVERY fragment address is a register
    // computed once per kernel (the tile loop is unrolled by two, the slot is a compile-time constant in each copy).
    // Counters of the round-3 form at C2 (rocprofv3, SQ_INSTS_VALU): 1.2 non-MFMA vector instructions per MFMA, two
    // thirds of them address adds -- each one taken from the matrix pipe's issue cycles.
    constexpr int SSTR = IMM ? G2_SLOT_IMM : G2_SLOT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef G2_CLOCKS
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
    const int P = p + n;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);

    // J-slice of this workgroup in whole tiles (J % KT == 0)
    const long long ntiles = J / KT;
    const long long tps = (ntiles + nslices - 1) / nslices;
    const long long t0 = (long long)slice * tps;
    const long long t1 = t0 + tps < ntiles ? t0 + tps : ntiles;
    const int nt = __builtin_amdgcn_readfirstlane(t1 > t0 ? (int)(t1 - t0) : 0);     // tiles of this slice (scalar loop control)

    // DMA pieces of this wave: piece pc = wave + 16 i = (block row pc / PPB, rows 8 q .. 8 q + 7 of it, q = pc % PPB);
    // lane = (k-group g, k sub-block, row r8) in the permuted cell order of the header.  Per-lane source pointers
    // live in registers and advance by one tile per issue (one 64-bit add per piece and tile).  Rows past P
    // (padding of the last block row) read row 0 of U: their products land in rows / columns the reduce never reads.
    const int npieces = nrb * PPB;
    // pieces wave, wave + 16, ... of this wave: i < np (an int compared at every use: a boolean kept live across the
    // unrolled tile bodies costs a v_cndmask / v_cmp pair wherever the compiler copies it)
    const int np = __builtin_amdgcn_readfirstlane(npieces > wave ? (npieces - wave + G2_WAVES - 1) / G2_WAVES : 0);
    const int r8 = ((lane >> 4) & 1) * 4 + ((lane >> 2) & 3);      // row of this lane's cell inside its piece
    // SG (p, P multiples of 8: a piece's 8 rows lie in one array): wave-uniform base of the piece's first row + a per-lane
    // byte offset fixed for the whole kernel -- the address arithmetic of a tile is scalar (every VALU instruction in this
    // loop is taken from the f32 MFMA rate); otherwise per-lane pointers.
    const T* gsrc[G2_MAXP];
    const char* sbase[G2_MAXP];
    unsigned voff[G2_MAXP];
    T psh[G2_MAXP], rs[G2_MAXP];
    int ownmask = 0;                                       // bit i: this type reports piece i's row sums (wave-uniform)
    constexpr bool SGA = SG && (G2_OPT & 4);
#pragma unroll
    for (int i = 0; i < G2_MAXP; ++i) {
        const int pc = wave + G2_WAVES * i;
        const int rb = pc < npieces ? pc / PPB : 0, q = pc % PPB;
        const int ent = __builtin_amdgcn_readfirstlane(rows_tab[rows_off + rb]);
        const int gr0 = (ent & 0xffff) * TILE + q * 8, gr = gr0 + r8;
        // 16-byte chunk of the row's 128 B this lane fetches (the inverse of the cell map of the header)
        const int chunk = F32 ? 4 * (lane >> 5) + 2 * (((lane >> 1) & 1) ^ (q & 1)) + ((lane & 1) ^ (q >> 1))
                              : 4 * (lane >> 5) + 2 * (((lane >> 1) & 1) ^ ((lane >> 4) & 1)) + (lane & 1);
        const T* ptr = U;
        T sh = 0;
        if (gr < P) {
            ptr = gr < p ? U + (size_t)gr * J : G + (size_t)(gr - p) * J;
            sh = shift[gr];
        }
        if constexpr (!SGA) gsrc[i] = ptr + t0 * KT + chunk * VEC;
        if constexpr (SGA) {
            const T* b0 = gr0 >= P ? U : gr0 < p ? U + (size_t)gr0 * J : G + (size_t)(gr0 - p) * J;
            sbase[i] = reinterpret_cast<const char*>(b0 + t0 * KT);
            voff[i] = (unsigned)((size_t)r8 * (size_t)J * sizeof(T)) + chunk * 16;
        }
        psh[i] = sh;
        rs[i] = 0;
        if ((ent >> 16) != 0) ownmask |= 1 << i;
    }
    // Wave-uniform values the tile loop tests (np, nb, more) are made OPAQUE where they are used (an empty asm with a "+s"
    // operand): the compiler otherwise evaluates every comparison once, in front of the loop, and keeps each result as a
    // 64-bit lane mask -- seventeen SGPR pairs at the last count, which pushed the DMA bases and LDS addresses out of the
    // scalar file and back in through v_readlane inside the loop (23 - 75 spilled SGPRs per instantiation, round 4's ISA).
    // Re-evaluated at the use, a condition is one s_cmp + s_cbranch_scc and holds no register.
    auto opaque_if = [](bool on, int v) { if (on) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); } return v; };
    auto opaque = [&](int v) { return opaque_if(true, v); };
    auto issue_tile = [&](int slot) {
        const int npo = opaque_if(G2_OPQ & 1, np);
        const unsigned l0 = (unsigned)opaque_if(G2_OPQ & 1, (int)lds0) + (unsigned)(slot * SSTR) + (unsigned)(wave * 1024);
#pragma unroll
        for (int i = 0; i < G2_MAXP; ++i) {
            if (i < npo) {
                if constexpr (SGA) {
                    if (!(G2_ABL & 1)) glds16s(sbase[i], voff[i], l0 + i * (G2_WAVES * 1024));
                    sbase[i] += G2_ROWB;
                } else {
                    if (!(G2_ABL & 1)) glds16(gsrc[i], l0 + i * (G2_WAVES * 1024));
                    gsrc[i] += KT;
                }
            }
        }
    };
#ifdef G2_CLOCKS
    const long long gp1 = clock64();
#endif
    if (nt > 0) issue_tile(0);

    // this wave's block list: (compact row block of A) | (of B) << 8 | (block index inside the type) << 16, wave-uniform
    int iab[NBW];
    int nb = 0;
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int* e = wblk + (size_t)(blocks_off + wave * NBW + b) * 3;
        const int a = __builtin_amdgcn_readfirstlane(e[0]);
        const int c = __builtin_amdgcn_readfirstlane(e[1]);
        const int o = __builtin_amdgcn_readfirstlane(e[2]);
        iab[b] = a | (c << 8) | (o << 16);
        if (a >= 0) nb = b + 1;
    }
    nb = __builtin_amdgcn_readfirstlane(nb);

    acc_t acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) acc[b][r] = 0;
#ifdef G2_CLOCKS
    const long long gp2 = clock64();
#endif

    // in-place shift of this wave's own pieces (the lane that fetched a 16-byte cell also shifts it); the row sums
    // only where this type reports them (a wave-uniform branch: the second launch's types report 3 - 5 of their
    // 10 - 16 block rows, the U rows belong to the first launch)
    char* const sb0 = smem + lane * 16 + wave * 1024;
    auto shift_tile = [&](auto slotc) {
        constexpr int SLOT = decltype(slotc)::value;
        // (two pieces at a time: 8 registers of temporaries beside the 64 accumulators and the 16 fragment addresses.
        //  ONE copy of the code of every piece: with a second copy in an else-branch the compiler merged the tails of the
        //  two, indexed rs[] with a register and moved the row sums to scratch -- a scratch_load / s_waitcnt vmcnt(0) /
        //  scratch_store round trip per piece pair and tile, round 4's ISA)
        auto one = [&](int i, vec_t v) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] -= psh[i];
            if (((opaque_if(G2_OPQ & 2, ownmask) >> i) & 1) || !(G2_OPT & 2)) {
                T r = rs[i];
#pragma unroll
                for (int c = 0; c < VEC; ++c) r += v[c];
                asm volatile("" : "+v"(r));             // (keeps the branch: no select around the adds)
                rs[i] = r;
            }
            *reinterpret_cast<vec_t*>(sb0 + SLOT * SSTR + i * (G2_WAVES * 1024)) = v;
        };
        const int npo = opaque_if(G2_OPQ & 2, np);
#pragma unroll
        for (int i0 = 0; i0 < G2_MAXP; i0 += 2) {
            if (i0 < npo) {
                const bool two = i0 + 1 < npo;
                const vec_t v0 = *reinterpret_cast<const vec_t*>(sb0 + SLOT * SSTR + i0 * (G2_WAVES * 1024));
                vec_t v1 = v0;
                if (two) v1 = *reinterpret_cast<const vec_t*>(sb0 + SLOT * SSTR + (i0 + 1) * (G2_WAVES * 1024));
                one(i0, v0);
                if (two) one(i0 + 1, v1);
            }
        }
    };

    // fragment reads: lane = (k sub-block lk, row li of the block) reads its cell of piece li / 8; the immediate part of
    // the k-group is the offset (g / NG0) * GSTR, the rest (f32: g0) one of NG0 lane offsets
    const int li = lane % TILE, lk = lane / TILE;
    const int fq = li >> 3, fr2 = (li >> 2) & 1;
    int foff[NG0];
#pragma unroll
    for (int g0 = 0; g0 < NG0; ++g0)
        foff[g0] = fq * 1024 + fr2 * 256 + (li & 3) * 64 +
                   (F32 ? ((g0 ^ (fq & 1)) << 5) + ((lk ^ (fq >> 1)) << 4) : ((((lk >> 1) ^ fr2)) << 5) + ((lk & 1) << 4));

    // fragment addresses of this wave's blocks in slot 0, fixed for the whole kernel (IMM: slot and k-group are immediates)
    using lds_cptr = const __attribute__((address_space(3))) char*;          // 32-bit LDS addresses (a pinned generic pointer
    using lds_cvec = const __attribute__((address_space(3))) vec_t*;          //  would be 64 bits wide and read through flat_load)
    const lds_cptr smem3 = (lds_cptr)(__attribute__((address_space(3))) char*)smem;
    lds_cptr pa[NBW][NG0];
    lds_cptr pc_[NBW][NG0];
    if constexpr (IMM) {
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int g0 = 0; g0 < NG0; ++g0) {
                pa[b][g0] = smem3 + (iab[b] & 0xff) * BLKB + foff[g0];
                pc_[b][g0] = smem3 + ((iab[b] >> 8) & 0xff) * BLKB + foff[g0];
                // (opaque to the compiler from here on: it would otherwise re-derive them with an add per read)
                asm volatile("" : "+v"(pa[b][g0]), "+v"(pc_[b][g0]));
            }
    }

    // One (block, group) step: VEC MFMAs on fragments that were loaded one step earlier.  The loads of
    // the NEXT step are issued first (sched_barrier keeps them there), so every LDS read has VEC MFMAs
    // (256 / 128 cycles) between issue and use instead of an exposed lgkmcnt(0) in front of each MFMA.
    struct Frag { vec_t a, c; };
    auto load_frag = [&](Frag& f, auto slotc, int b, int g) {
        constexpr int SLOT = decltype(slotc)::value;
        if (G2_ABL & 4) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { f.a[v] = (T)(lane + v + b); f.c[v] = (T)(lane - v + g); }
            return;
        }
        if constexpr (IMM) {
            // (timing ablations of fragment sharing, results wrong: 32 = the c fragment of every odd block is the even block's,
            //  64 = one pair of fragment reads per four blocks)
            if ((G2_ABL & 64) && (b & 3) != 0) return;
            f.a = *reinterpret_cast<lds_cvec>(pa[b][g % NG0] + SLOT * SSTR + (g / NG0) * GSTR);
            if ((G2_ABL & 32) && (b & 1) != 0) return;
            f.c = *reinterpret_cast<lds_cvec>(pc_[b][g % NG0] + SLOT * SSTR + (g / NG0) * GSTR);
        } else {        // (64-KiB slots: the slot does not fit the offset field, one address add per read)
            const char* base = smem + SLOT * SSTR;
            f.a = *reinterpret_cast<const vec_t*>(base + (iab[b] & 0xff) * BLKB + foff[g % NG0] + (g / NG0) * GSTR);
            f.c = *reinterpret_cast<const vec_t*>(base + ((iab[b] >> 8) & 0xff) * BLKB + foff[g % NG0] + (g / NG0) * GSTR);
        }
    };

    // After which of its blocks a wave shifts its pieces of the next tile.  The SIMD issues the oldest wave first:
    // waves 0-3 run ahead and wait at the tile's barrier while 12-15 still multiply, so a shift pass placed
    // early in EVERY wave's own instruction stream lies in the middle of the tile in wall time for all but the
    // oldest (whose wait for the DMA is covered by the others' MFMAs); placed last it is exposed for the youngest.
    constexpr int SAT = F32 ? G2_SHIFT_AT_F32 : G2_SHIFT_AT_F64;
    const int shift_at = __builtin_amdgcn_readfirstlane(SAT < 0 ? ((wave >> 2) + 1) * NBW / 4 : SAT > NBW ? NBW : SAT);
    if (nt > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef G2_CLOCKS
        if (tid == 0 && blockIdx.x < 4096) g_gram2_pro[blockIdx.x * 4 + 2] = clock64() - gp2;
#endif
        shift_tile(std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef G2_CLOCKS
    if (tid == 0 && blockIdx.x < 4096) {
        g_gram2_pro[blockIdx.x * 4 + 0] = gp1 - gclk0; g_gram2_pro[blockIdx.x * 4 + 1] = gp2 - gp1;
        g_gram2_pro[blockIdx.x * 4 + 3] = clock64() - gp2;
    }
#endif
#ifdef G2_CLOCKS
    const long long gclk1 = clock64();
    long long gbar = 0;
#endif
    // partial sums of one block of this slice, accumulator-major (slab_group_rc): 16-byte stores of consecutive lanes.
    // Issued as soon as the block's last MFMA of the slice is (G2_OPT & 8): the 38 - 51 MB of slabs of a launch then leave
    // the CUs spread over the last tile instead of as one burst behind it, which the kernel boundary has to wait out
    // (the launch ended ~10 us after its workgroups' last MFMA; dirty lines drain at ~6 TB/s).
    auto store_block = [&](int b) {
        int sl = slice;
        asm volatile("" : "+s"(sl));        // (opaque: the addresses are formed here, once, not hoisted into registers that stay live through the K loop)
        // (the lane offset is formed here, from mbcnt: not a 64-bit register pair held through the tile loop)
        const unsigned lane_s = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        T* out = slabs + ((size_t)slab0 + (size_t)sl * nblk_t + ((iab[b] >> 16) & 0xff)) * (TILE * TILE) + (size_t)lane_s * VEC;
#pragma unroll
        for (int q = 0; q < M::NACC / VEC; ++q) {
            vec_t v;
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] = acc[b][q * VEC + c];
            // (plain stores: the reduce launch right behind reads the slabs back; non-temporal stores, which drop
            //  the lines from L2, cost the step 1.6 % in round 3)
            vec_t* dst = reinterpret_cast<vec_t*>(out + (size_t)(q * 64) * VEC);
            typedef float st4_t __attribute__((ext_vector_type(4)));
            if (G2_OPT & 32) {          // write-through at agent scope: no dirty slab lines left for the kernel boundary to flush
                st4_t sv;
                __builtin_memcpy(&sv, &v, 16);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(sv) : "memory");
            } else if (G2_OPT & 64) {
                st4_t sv;
                __builtin_memcpy(&sv, &v, 16);
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(sv) : "memory");
            } else if (G2_OPT & 16) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    };
    auto tile = [&](auto curc, int k) {
        constexpr int CUR = decltype(curc)::value;
        const int more = opaque_if(G2_OPQ & 4, nt - 1 - k);    // > 0: a next tile exists
        const int nbo = opaque_if(G2_OPQ & 8, nb), sat = opaque_if(G2_OPQ & 16, shift_at);
        const std::integral_constant<int, CUR> cur;
        const std::integral_constant<int, CUR ^ 1> nxt;
        // the next tile's pieces: behind this wave's first MFMA group (the matrix pipe restarts right behind the barrier,
        // the DMA issue -- 16 waves x 4 pieces through one address unit -- runs under those MFMAs)
        if (more > 0 && (!(G2_OPT & 1) || nbo == 0)) issue_tile(CUR ^ 1);
        Frag f0, f1;
        if (nbo > 0) load_frag(f0, cur, 0, 0);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            // This wave's pieces of tile t+1 (issued at the top of the tile, landed long since) are shifted in
            // place BETWEEN two of its blocks, at a different point for each of the 4 waves of a SIMD: the
            // LDS round trip of one wave's shift pass is covered by the MFMAs of the other three, instead of
            // all 16 waves running it together behind their last MFMA with the matrix pipes idle.
            if (b == sat && more > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(G2_ABL & 8)) shift_tile(nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (b < nbo) {
#pragma unroll
                for (int g = 0; g < NGROUP; ++g) {
                    Frag& fc = (g & 1) ? f1 : f0;
                    Frag& fn = (g & 1) ? f0 : f1;
                    // prefetch the next step: group g+1 of this block, or group 0 of the next block
                    if (g + 1 < NGROUP) {
                        load_frag(fn, cur, b, g + 1);
                    } else if (b + 1 < NBW) {
                        if (b + 1 < nbo) load_frag(fn, cur, b + 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[b] = M::mma(fc.a[v], fc.c[v], acc[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((G2_OPT & 1) && b == 0 && g == 0 && more > 0) { issue_tile(CUR ^ 1); __builtin_amdgcn_sched_barrier(0); }
                }
                if ((G2_OPT & 8) && more <= 0) { store_block(b); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        if (sat == NBW && more > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(G2_ABL & 8)) shift_tile(nxt);
        }
        static_assert(NGROUP % 2 == 0, "the fragment double buffer returns to f0 at every block boundary");
#ifdef G2_CLOCKS
        const long long tb0 = clock64();
#endif
        // the tile's one barrier (every wave has read slot `cur`, every piece of tile t+1 is shifted)
        // (none behind the last tile: a wave that is done stores and leaves)
        if (!(G2_ABL & 2) && (more > 0 || !(G2_OPT & 8))) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef G2_CLOCKS
        gbar += clock64() - tb0;
#endif
    };
    for (int k = 0; k < nt; k += 2) {
        tile(std::integral_constant<int, 0>{}, k);
        if (k + 1 < nt) tile(std::integral_constant<int, 1>{}, k + 1);
    }
#ifdef G2_CLOCKS
    const long long gclk2 = clock64();
#endif

    // first moments of this slice: the 8 cells of a row sit in the 8 lanes of this wave that share lane bits 2-4 --
    // summed in a fixed order (xor 1, 2, 32).  Only the type that owns a block row reports it.
    // (the lane index is formed again here, from mbcnt: nothing derived from threadIdx stays live across the tile loop for this)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int r8_e = ((lane_e >> 4) & 1) * 4 + ((lane_e >> 2) & 3);
#pragma unroll
    for (int i = 0; i < G2_MAXP; ++i) {
        double v = (double)rs[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 32, 64);
        const int pc = wave + G2_WAVES * i;
        const int ent = __builtin_amdgcn_readfirstlane(rows_tab[rows_off + (pc < npieces ? pc / PPB : 0)]);
        const int gr = (ent & 0xffff) * TILE + (pc % PPB) * 8 + r8_e;
        if ((lane_e & 0x23) == 0 && pc < npieces && (ent >> 16) != 0 && gr < P) rowsum_part[(size_t)(rs0 + slice) * P + gr] = v;
    }

    // (an empty slice -- more workgroups than tiles -- still writes its zeros: the reduce sums every slice)
    if (!(G2_OPT & 8) || nt == 0) {
#pragma unroll
        for (int b = 0; b < NBW; ++b)
            if (b < nb) store_block(b);
    }
#ifdef G2_CLOCKS
    if (lane == 0 && blockIdx.x < 4096) g_gram2_bar[blockIdx.x * 16 + wave] = gbar;
    if (tid == 0 && blockIdx.x < 4096) {
        g_gram2_clk[blockIdx.x * 4 + 0] = gclk1 - gclk0; g_gram2_clk[blockIdx.x * 4 + 1] = gclk2 - gclk1;
        g_gram2_clk[blockIdx.x * 4 + 2] = clock64() - gclk2; g_gram2_clk[blockIdx.x * 4 + 3] = wall_clock64() - gw0;
    }
#endif
}

template <typename T>
static int launch_gram2_t(Engine& e, int part, const void* U, const void* G, hipStream_t s) {
    GramPart& gp = e.gp[part];
    const GramPlan& pl = gp.plan;
    constexpr int KT = G2_ROWB / (int)sizeof(T);
    if (e.J % KT != 0 || e.J < KT || ((uintptr_t)U & 15) || ((uintptr_t)G & 15)) return -1;
    const int nrows = pl.max_rb * pl.tile;
    if (nrows > G2_MAX_ROWS) return -1;
    const bool imm = nrows * G2_ROWB <= G2_SLOT_IMM;
    const int lds = imm ? 2 * G2_SLOT_IMM : 2 * G2_SLOT;
    const bool sg = e.p % 8 == 0 && e.P % 8 == 0 && (unsigned long long)e.J * sizeof(T) * 7 + 128 < (1ull << 32);
    auto kern = sg ? (imm ? gram2_kernel<T, true, true> : gram2_kernel<T, true, false>)
                   : (imm ? gram2_kernel<T, false, true> : gram2_kernel<T, false, false>);
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    {
        e.prof_part = part;
        ProfScope prof(e, (e.profile_gap_only && part == 0) ? -1 : 0, s, true);      // (gap-only: the second launch's stop, nothing else)
        if (prof.on())
            hipExtLaunchKernelGGL(kern, dim3(pl.total_wgs), dim3(G2_THREADS), (unsigned)lds, s, prof.a, prof.b, 0,
                                  (const T*)U, (const T*)G, (const T*)e.d_shiftT, e.p, e.n, (long long)e.J,
                                  (const int*)gp.d_type_hdr, pl.ntypes, (const int*)gp.d_rows, (const int*)gp.d_wblk,
                                  (T*)gp.d_slabs, gp.d_rowsum_part);
        else
        hipLaunchKernelGGL(kern, dim3(pl.total_wgs), dim3(G2_THREADS), lds, s, (const T*)U, (const T*)G,
                           (const T*)e.d_shiftT, e.p, e.n, (long long)e.J, gp.d_type_hdr, pl.ntypes, gp.d_rows,
                           gp.d_wblk, (T*)gp.d_slabs, gp.d_rowsum_part);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

int launch_gram2(Engine& e, int part, const void* U, const void* G, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? launch_gram2_t<float>(e, part, U, G, s) : launch_gram2_t<double>(e, part, U, G, s);
}

}  // namespace cesx
