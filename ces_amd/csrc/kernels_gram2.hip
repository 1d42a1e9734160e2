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
#include <algorithm>
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
#ifndef G2_SHIFT_AT
#define G2_SHIFT_AT 1
#endif
#ifndef G2_OPT      // dev A/B switches (tools/gram2_bench.hip): 1 = next tile's DMA issued behind the first MFMA group,
#define G2_OPT 15   // 2 = row sums only where the type reports them, 4 = scalar DMA addressing, 8 = a block's partial sums stored
                    // as soon as its last MFMA of the slice is issued (no barrier behind the last tile), 16 = those stores non-temporal
#endif
#ifdef G2_FUSED_STAMPS      // dev instrumentation of the fused launch (tools/gram2_bench.hip fused): s_memrealtime per workgroup
__device__ long long g_gram2_fst[1024 * 8];     // {start, first part done, published, duty begin, duty end, end, polls, chunks}
#define G2_STAMP(k) do { if (threadIdx.x == 0) g_gram2_fst[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define G2_COUNT(k, v) do { if (threadIdx.x == 0) g_gram2_fst[blockIdx.x * 8 + (k)] += (v); } while (0)
#else
#define G2_STAMP(k) do {} while (0)
#define G2_COUNT(k, v) do {} while (0)
#endif
#ifdef G2_CLOCKS
__device__ long long g_gram2_clk[4096 * 4];
__device__ long long g_gram2_bar[4096 * 16];      // per wave: cycles spent in the per-tile barrier
__device__ long long g_gram2_pro[4096 * 4];       // prologue phases of wave 0: tables | addresses | first DMA | shift + barrier
#endif

// Tables of one launch (GramPlan, kernels_gram.hip) as the kernel reads them.
struct Gram2Tab {
    const int* type_hdr; int ntypes; const int* rows_tab; const int* wblk;
    void* slabs; double* rowsum_part; int total_wgs;
};

// What the FUSED launch (both parts of the Gram in one kernel, below) adds: the counters of the in-kernel hand-over of
// the first part and the tables of its fixed-order fp64 reduce (gram_reduce_kernel's, kernels_gram.hip).
struct Gram2Fused {
    unsigned* sync;              // this launch's counters {arrive, claim, done, -}: zero on entry
    unsigned* sync_next;         // the set the NEXT fused launch uses: zeroed by this one (it is idle during this launch)
    unsigned long long* ready;   // sequence number of the last launch whose U x U head is complete in `mom`
    unsigned long long seq;      // this launch's
    unsigned expected;           // workgroups that publish partial sums of the first part
    const int* blk_rc; const int* row_own;
    int nblocks, tile, row_lo, row_hi;
    MomLayout ml; long long J;
    double* mom;
};

constexpr int G2_SHTAB = G2_MAX_ROWS * 8;    // LDS behind the two slots: the centring shift of every staged row (read back by the shift pass:
                                        // four registers per lane less than holding them -- the fused launch must stay within 120)
constexpr int G2_COMB = 32 * 1024;      // LDS of the fused launch's reduce: 32 slice parts x 32 groups x 16 B x 2 (fp64)
constexpr int G2_CTL = 64;              // ... and its control words {claimed chunk, all-arrived flag}
constexpr int G2_DUTY_G = 32, G2_DUTY_S = 32;

// no-op hooks of the plain launch (MODE 0) and of the fused launch's first part (MODE 1)
struct Gram2NoHook {
    __device__ __forceinline__ void prologue_poll(int) {}
    __device__ __forceinline__ void publish() {}
    __device__ __forceinline__ void after_barrier() {}
    __device__ __forceinline__ bool open() const { return false; }
};

// One pass of the split-K Gram over this workgroup's slice of the launch described by `tab`.
// MODE 0: the plain launch.  MODE 1: the first (U x U) part of the fused launch -- its partial sums and row sums are
// stored WRITE-THROUGH (sc1: other workgroups of the same launch read them) and the caller publishes them.
// MODE 2: the second part of the fused launch -- `hook` carries the reduce duty (polled at the prologue, beside the
// first tile's DMA: the accumulators are not live yet; the caller looks once more behind the pass).
template <typename T, bool SG, bool IMM, int MODE, typename Hook>
__device__ __forceinline__ void gram2_phase(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                                            int p, int n, long long J, const Gram2Tab& tab, char* smem, Hook& hook) {
    const int* __restrict__ type_hdr = tab.type_hdr;
    const int ntypes = tab.ntypes;
    const int* __restrict__ rows_tab = tab.rows_tab;
    const int* __restrict__ wblk = tab.wblk;
    T* __restrict__ slabs = reinterpret_cast<T*>(tab.slabs);
    double* __restrict__ rowsum_part = tab.rowsum_part;
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
    constexpr int GSTR = 512;                            // bytes between immediate k-groups inside a piece (f32: g1, f64: g)
    constexpr int NG0 = NGROUP / 2;                      // k-groups folded into the lane address (f32: g0 = 0, 1; f64: none)
    static_assert(NG0 == (F32 ? 2 : 1), "cell maps of the header");
    // IMM (every type of the launch stages <= 60 KiB per tile): the second slot starts G2_SLOT_IMM bytes behind the first,
    // so slot + immediate k-group fit the 16-bit offset field of a ds_read and EVERY fragment address is a register
    // computed once per kernel (the tile loop is unrolled by two, the slot is a compile-time constant in each copy).
    // Counters of the round-3 form at C2 (rocprofv3, SQ_INSTS_VALU): 1.2 non-MFMA vector instructions per MFMA, two
    // thirds of them address adds -- each one taken from the matrix pipe's issue cycles.
    constexpr int SSTR = IMM ? G2_SLOT_IMM : G2_SLOT;

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
    unsigned voff = 0;          // (ONE offset: the pieces of a wave, wave + 16 i, all have the same q = wave % PPB, hence the same cell map)
    static_assert(G2_WAVES % PPB == 0, "q = pc % PPB does not depend on i");
    T rs[G2_MAXP];
    T* const shtab = reinterpret_cast<T*>(smem + 2 * SSTR) + wave * 8 + r8;      // [piece][row of the piece]; this wave reads what it wrote
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
            voff = (unsigned)((size_t)r8 * (size_t)J * sizeof(T)) + chunk * 16;
        }
        if ((lane & 0x23) == 0) shtab[i * (G2_WAVES * 8)] = sh;
        rs[i] = 0;
        if ((ent >> 16) != 0) ownmask |= 1 << i;
    }
    auto issue_tile = [&](int slot) {
#pragma unroll
        for (int i = 0; i < G2_MAXP; ++i) {
            const int pc = wave + G2_WAVES * i;
            if (i < np) {
                if constexpr (SGA) {
                    if (!(G2_ABL & 1)) glds16s(sbase[i], voff, lds0 + slot * SSTR + pc * 1024);
                    sbase[i] += G2_ROWB;
                } else {
                    if (!(G2_ABL & 1)) glds16(gsrc[i], lds0 + slot * SSTR + pc * 1024);
                    gsrc[i] += KT;
                }
            }
        }
    };
#ifdef G2_CLOCKS
    const long long gp1 = clock64();
#endif
    if (nt > 0) issue_tile(0);
    if constexpr (MODE == 2) {
        // The hand-over of the fused launch, beside the first tile's DMA and before anything else of this pass is live in
        // registers: the first part's write-through stores are drained HERE (vmcnt is in order: the wait covers them and
        // the DMA pieces just issued -- the tile lands while the stores drain), the workgroup counts itself in, wave 0
        // polls (bounded) whether all workgroups have, and the reduce duty runs if so
        hook.publish();
        hook.prologue_poll(wave);
        __syncthreads();
        hook.after_barrier();
    }

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
        // (two pieces at a time: 8 registers of temporaries beside the 64 accumulators and the 16 fragment addresses;
        //  every branch is self-contained -- no value defined under one condition and used under another)
        auto one = [&](int i, vec_t v) {
            const T sh = shtab[i * (G2_WAVES * 8)];
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] -= sh;
            if (((ownmask >> i) & 1) || !(G2_OPT & 2)) {
                asm volatile("" ::: "memory");          // (keeps the branch: no select around four adds)
#pragma unroll
                for (int c = 0; c < VEC; ++c) rs[i] += v[c];
            }
            *reinterpret_cast<vec_t*>(sb0 + SLOT * SSTR + i * (G2_WAVES * 1024)) = v;
        };
#pragma unroll
        for (int i0 = 0; i0 < G2_MAXP; i0 += 2) {
            if (i0 + 1 < np) {
                const vec_t v0 = *reinterpret_cast<const vec_t*>(sb0 + SLOT * SSTR + i0 * (G2_WAVES * 1024));
                const vec_t v1 = *reinterpret_cast<const vec_t*>(sb0 + SLOT * SSTR + (i0 + 1) * (G2_WAVES * 1024));
                one(i0, v0);
                one(i0 + 1, v1);
            } else if (i0 < np) {
                one(i0, *reinterpret_cast<const vec_t*>(sb0 + SLOT * SSTR + i0 * (G2_WAVES * 1024)));
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
            f.a = *reinterpret_cast<lds_cvec>(pa[b][g % NG0] + SLOT * SSTR + (g / NG0) * GSTR);
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
    const int shift_at = __builtin_amdgcn_readfirstlane(G2_SHIFT_AT < 0 ? ((wave >> 2) + 1) * NBW / 4 : G2_SHIFT_AT);
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
        T* out = slabs + ((size_t)slab0 + (size_t)sl * nblk_t + ((iab[b] >> 16) & 0xff)) * (TILE * TILE);
        // MODE 1: write-through (sc1) -- read by other workgroups of this launch (cdna guide, Guideline 16 R1); the buffer
        // resource takes the block's wave-uniform base, the lane its 16-byte offset
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, TILE * TILE * (int)sizeof(T), 0x27000);
#pragma unroll
        for (int q = 0; q < M::NACC / VEC; ++q) {
            vec_t v;
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] = acc[b][q * VEC + c];
            // (plain stores: the reduce launch right behind reads the slabs back; non-temporal stores, which drop
            //  the lines from L2, cost the step 1.6 % in round 3)
            vec_t* dst = reinterpret_cast<vec_t*>(out + (size_t)(q * 64 + lane) * VEC);
            if constexpr (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrc, (q * 64 + lane) * 16, 0, 16);
            else if (G2_OPT & 16) __builtin_nontemporal_store(v, dst);
            else *dst = v;
        }
    };
    auto tile = [&](auto curc, int k) {
        constexpr int CUR = decltype(curc)::value;
        const int more = nt - 1 - k;                           // > 0: a next tile exists
        const std::integral_constant<int, CUR> cur;
        const std::integral_constant<int, CUR ^ 1> nxt;
        // the next tile's pieces: behind this wave's first MFMA group (the matrix pipe restarts right behind the barrier,
        // the DMA issue -- 16 waves x 4 pieces through one address unit -- runs under those MFMAs)
        if (more > 0 && (!(G2_OPT & 1) || nb == 0)) issue_tile(CUR ^ 1);
        Frag f0, f1;
        if (nb > 0) load_frag(f0, cur, 0, 0);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            // This wave's pieces of tile t+1 (issued at the top of the tile, landed long since) are shifted in
            // place BETWEEN two of its blocks, at a different point for each of the 4 waves of a SIMD: the
            // LDS round trip of one wave's shift pass is covered by the MFMAs of the other three, instead of
            // all 16 waves running it together behind their last MFMA with the matrix pipes idle.
            if (b == shift_at && more > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(G2_ABL & 8)) shift_tile(nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (b < nb) {
#pragma unroll
                for (int g = 0; g < NGROUP; ++g) {
                    Frag& fc = (g & 1) ? f1 : f0;
                    Frag& fn = (g & 1) ? f0 : f1;
                    // prefetch the next step: group g+1 of this block, or group 0 of the next block
                    if (g + 1 < NGROUP) {
                        load_frag(fn, cur, b, g + 1);
                    } else if (b + 1 < NBW) {
                        if (b + 1 < nb) load_frag(fn, cur, b + 1, 0);
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
        if (shift_at == NBW && more > 0) {
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
#pragma unroll
    for (int i = 0; i < G2_MAXP; ++i) {
        double v = (double)rs[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 32, 64);
        const int pc = wave + G2_WAVES * i;
        const int ent = __builtin_amdgcn_readfirstlane(rows_tab[rows_off + (pc < npieces ? pc / PPB : 0)]);
        const int gr = (ent & 0xffff) * TILE + (pc % PPB) * 8 + r8;
        if ((lane & 0x23) == 0 && pc < npieces && (ent >> 16) != 0 && gr < P) {
            double* dst = &rowsum_part[(size_t)(rs0 + slice) * P + gr];
            if constexpr (MODE == 1) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = v;
        }
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

// the plain launch: one part of the Gram
template <typename T, bool SG, bool IMM>
__global__ __launch_bounds__(G2_THREADS, 4)
void gram2_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                  int p, int n, long long J, Gram2Tab tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Gram2NoHook hook;
    gram2_phase<T, SG, IMM, 0>(U, G, shift, p, n, J, tab, smem, hook);
}

// ---------------------------------------------------------------------------
// The FUSED launch (round 4): both parts of the Gram in ONE kernel.  Every workgroup first runs its slice of the
// U x U part (all chol(C) needs), publishes its partial sums write-through and counts itself in (`arrive`); it then
// goes on with its slice of the second part without waiting for anybody.  The fp64 fixed-order reduce of the first
// part is spread over ALL workgroups as a duty of a few microseconds each: at the prologue of its second part (while
// the first tile's DMA is in flight) wave 0 polls `arrive` for a bounded time; once every publisher has arrived the
// workgroup claims reduce chunks (32 16-byte groups x 32 slice parts each) until none is left, and the workgroup that
// completes the last chunk stores the launch's sequence number in `ready` -- which a one-wave kernel on the engine's
// side stream polls in front of the centring + chol(C) chain.  A workgroup that does not see everybody within the bound
// polls once more at the end of every tile and at its end; the workgroup whose arrival completed the count sees it
// at once and clears whatever chunks are left, so the head always completes and NO workgroup ever waits for another
// (nothing depends on co-residency or dispatch order; cdna guide, Guideline 16: sc1 payload stores drained by every
// storing wave, the workgroup's barrier, one agent-scope counter add; consumers poll the counter with sc1 loads and
// read the payload with sc1 loads only).
// What it removes from the caller's stream against the two-launch sequence: the first launch's end (its slabs drained
// at the kernel boundary, 6-7 us), the reduce launch (11-12 us), the hand-over to the side stream (5 us) and the
// second launch's prologue beside an idle chip (4 us).
// ---------------------------------------------------------------------------
template <typename T>
struct Gram2Duty {
    using vec_t = typename Mfma<T>::vec_t;
    static constexpr int VEC = Mfma<T>::VEC;
    const Gram2Fused& f;
    const void* slabs_a; const double* rowsum_a;      // the first part's partial sums
    char* comb; int* ctl;                              // LDS: G2_COMB bytes, {claimed chunk, all-arrived flag}
    int duty_open = 1;                                 // (uniform over the workgroup: derived from LDS behind barriers)

    int published = 0;                                 // this workgroup ran the first part: it counts itself in (publish)
    __device__ __forceinline__ bool open() const { return duty_open != 0; }
    // every storing wave drains its write-through stores, the workgroup's barrier, one lane counts the workgroup in
    __device__ __forceinline__ void publish() {
        if (!published) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(f.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = 0;
        G2_STAMP(2);
    }
    __device__ __forceinline__ unsigned arrived() const {
        return __hip_atomic_load(f.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // wave 0, beside the first tile's DMA: wait a bounded time (~5 us) for the stragglers of the first part
    __device__ __forceinline__ void prologue_poll(int wave) {
        if (wave != 0) return;
        int all = 0;
        for (int it = 0; it < 12; ++it) {
            G2_COUNT(6, 1);
            if (arrived() >= f.expected) { all = 1; break; }
            __builtin_amdgcn_s_sleep(64);
        }
        if ((threadIdx.x & 63) == 0) ctl[1] = all;
    }
    // wave 0, one look (the end of the workgroup's second part)
    __device__ __forceinline__ void last_poll(int wave) {
        if (wave != 0) return;
        const int all = arrived() >= f.expected ? 1 : 0;
        if ((threadIdx.x & 63) == 0) ctl[1] = all;
    }
    // every wave, behind a barrier that follows a poll
    __device__ __forceinline__ void after_barrier() {
        if (!duty_open) return;
        const int all = __builtin_amdgcn_readfirstlane(*(volatile int*)&ctl[1]);
        if (all) { G2_STAMP(3); run(); duty_open = 0; G2_STAMP(4); }
    }

    // the reduce chunks of gram_reduce_kernel's sums, claimed one at a time by whole workgroups (1024 threads =
    // 32 groups x 32 slice parts: one round of <= 8 loads in flight per thread at 248 slices)
    __device__ void run() {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        const int tid = threadIdx.x, part = tid / G2_DUTY_G, gl = tid % G2_DUTY_G;
        const int tt = f.tile * f.tile;
        const long long ngroups = (long long)f.nblocks * tt / VEC;
        const int gchunks = (int)((ngroups + G2_DUTY_G - 1) / G2_DUTY_G);
        const int nrow = f.row_hi - f.row_lo;
        const int rchunks = (nrow + G2_DUTY_G - 1) / G2_DUTY_G;
        const int nchunks = gchunks + rchunks;
        const int P = f.ml.p + f.ml.n;
        double (*cb)[G2_DUTY_G][VEC] = reinterpret_cast<double (*)[G2_DUTY_G][VEC]>(comb);
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(slabs_a), 0, 0x7ffffff0, 0x27000);
        for (;;) {
            if (tid == 0) ctl[0] = (int)__hip_atomic_fetch_add(f.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int c = __builtin_amdgcn_readfirstlane(*(volatile int*)&ctl[0]);
            if (c >= nchunks) break;
            G2_COUNT(7, 1);
            double acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
            bool on = false;
            const int* info = f.blk_rc;
            int e0 = 0;
            long long r = 0;
            if (c < gchunks) {
                const long long idx = (long long)c * G2_DUTY_G + gl;
                on = idx < ngroups;
                const int blk = on ? (int)(idx / (tt / VEC)) : 0;
                e0 = on ? (int)(idx % (tt / VEC)) * VEC : 0;
                info = f.blk_rc + blk * 5;
                const unsigned stride = (unsigned)info[3] * tt * (unsigned)sizeof(T);
                const int nsl = info[4];
                const int per = (nsl + G2_DUTY_S - 1) / G2_DUTY_S;
                const int k1 = min(nsl, (part + 1) * per);
                unsigned off = ((unsigned)info[2] * tt + e0) * (unsigned)sizeof(T) + (unsigned)(part * per) * stride;
                if (on)
                    for (int k = part * per; k < k1; k += 8) {
                        u4_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (k + u < k1) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, off + u * stride, 0, 16);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (k + u < k1) {
                                const vec_t x = __builtin_bit_cast(vec_t, v[u]);
#pragma unroll
                                for (int q = 0; q < VEC; ++q) acc[q] += (double)x[q];
                            }
                        off += 8 * stride;
                    }
            } else {
                // first moments sum_j (z_ij - s_i) of the rows the first part owns
                r = f.row_lo + (long long)(c - gchunks) * G2_DUTY_G + gl;
                on = r < f.row_hi;
                if (on) {
                    const int rs0 = f.row_own[(r / f.tile) * 2], nsl = f.row_own[(r / f.tile) * 2 + 1];
                    const int per = (nsl + G2_DUTY_S - 1) / G2_DUTY_S;
                    const int k1 = min(nsl, (part + 1) * per);
                    for (int k = part * per; k < k1; ++k)
                        acc[0] += __hip_atomic_load(rowsum_a + (size_t)(rs0 + k) * P + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) cb[part][gl][v] = acc[v];
            __syncthreads();
            if (part == 0 && on) {
                auto put = [](double* dst, double v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                if (c < gchunks) {
                    const int p = f.ml.p, n = f.ml.n;
                    const int R = info[0], C = info[1];
                    double* Saa = f.mom + f.ml.Saa();
                    double* Sab = f.mom + f.ml.Sab();
                    double* Sbb = f.mom + f.ml.Sbb();
                    int row0, col, rstep;
                    slab_group_rc<T>(e0 / VEC, row0, col, rstep);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int gr = R * f.tile + row0 + v * rstep, gc = C * f.tile + col;
                        if (gr >= P || gc >= P) continue;
                        if (R == C && gc > gr) continue;          // diagonal block: lower half, mirrored below
                        double sum = 0.0;
#pragma unroll 8
                        for (int q = 0; q < G2_DUTY_S; ++q) sum += cb[q][gl][v];
                        if (gr < p) {                              // both in U (gr >= gc)
                            put(&Saa[(size_t)gr * p + gc], sum);
                            put(&Saa[(size_t)gc * p + gr], sum);
                        } else if (gc < p) {                       // gr in G, gc in U
                            put(&Sab[(size_t)gc * n + (gr - p)], sum);
                        } else {
                            put(&Sbb[(size_t)(gr - p) * n + (gc - p)], sum);
                            put(&Sbb[(size_t)(gc - p) * n + (gr - p)], sum);
                        }
                    }
                } else {
                    double sum = 0.0;
#pragma unroll 8
                    for (int q = 0; q < G2_DUTY_S; ++q) sum += cb[q][gl][0];
                    put(&f.mom[r < f.ml.p ? f.ml.sa() + r : f.ml.sb() + (r - f.ml.p)], sum);
                }
            }
            if (c == 0 && tid == 0) __hip_atomic_store(&f.mom[0], (double)f.J, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every storing wave's stores acknowledged, then ONE lane counts the chunk in; the last chunk publishes
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(f.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)old + 1 == nchunks) __hip_atomic_store(f.ready, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

// 120 registers per lane, not the 128 four waves per SIMD would allow: 4 x 120 = 480 of a SIMD's 512 leave room for
// ONE more small wave -- the side stream's wait kernel (below) can then sit on a CU beside a Gram workgroup, whenever
// it is dispatched.  With 128 it took a CU of its own, the launch's 256th workgroup started when another had
// finished BOTH its parts, and the arrival count completed 150 us late (measured: 1.9 ms per launch).
template <typename T, bool SG>
__global__ __launch_bounds__(G2_THREADS, 4)
void gram2_fused_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                        int p, int n, long long J, Gram2Tab ta, Gram2Tab tb, Gram2Fused f, int gram_wgs, const MetricFin fin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the previous update's metric finalisation + publication, riding on this launch as one extra workgroup -- the LAST
    // one: every CU starts with a Gram workgroup, this one takes the first CU that frees up
    if ((int)blockIdx.x >= gram_wgs) {
        if (fin.part != nullptr) metric_final_body(fin);
        return;
    }
    const int tid = threadIdx.x;
#ifdef G2_FUSED_STAMPS
    if (tid == 0) { g_gram2_fst[blockIdx.x * 8 + 6] = 0; g_gram2_fst[blockIdx.x * 8 + 7] = 0; g_gram2_fst[blockIdx.x * 8 + 3] = 0; g_gram2_fst[blockIdx.x * 8 + 4] = 0; }
#endif
    G2_STAMP(0);
    if (blockIdx.x == 0 && tid < 4) __hip_atomic_store(f.sync_next + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Gram2Duty<T> duty{f, ta.slabs, ta.rowsum_part, smem + 2 * G2_SLOT_IMM + G2_SHTAB,
                      reinterpret_cast<int*>(smem + 2 * G2_SLOT_IMM + G2_SHTAB + G2_COMB)};
    if ((int)blockIdx.x < ta.total_wgs) {
        Gram2NoHook nohook;
        gram2_phase<T, SG, true, 1>(U, G, shift, p, n, J, ta, smem, nohook);
        G2_STAMP(1);
        // every wave is done with the first part's tiles in LDS before the second part's first DMA lands there (a bare
        // barrier: the first part's stores stay in flight -- they are drained beside that DMA, Gram2Duty::publish)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        duty.published = 1;
    }
    if ((int)blockIdx.x < tb.total_wgs) {
        gram2_phase<T, SG, true, 2>(U, G, shift, p, n, J, tb, smem, duty);
    } else {
        duty.publish();
        duty.prologue_poll(__builtin_amdgcn_readfirstlane(tid >> 6));
        __syncthreads();
        duty.after_barrier();
    }
    if (duty.open()) {          // last look: never a wait
        duty.last_poll(__builtin_amdgcn_readfirstlane(tid >> 6));
        __syncthreads();
        duty.after_barrier();
    }
    G2_STAMP(5);
}

// side stream: ends when the fused launch with sequence number `want` has completed the U x U head of its moment
// buffer (bounded: ~2 s of wall time, then the step's status word reports the failure)
__global__ void gram2_wait_ready_kernel(const unsigned long long* ready, unsigned long long want, Scalars* sc) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 200000000ull) { sc->status = CESX_EHIP; break; }       // (s_memrealtime: 100 MHz)
    }
}

static Gram2Tab gram2_tab(const GramPart& gp) {
    return Gram2Tab{gp.d_type_hdr, gp.plan.ntypes, gp.d_rows, gp.d_wblk, gp.d_slabs, gp.d_rowsum_part, gp.plan.total_wgs};
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
    const int lds = (imm ? 2 * G2_SLOT_IMM : 2 * G2_SLOT) + G2_SHTAB;
    const bool sg = e.p % 8 == 0 && e.P % 8 == 0 && (unsigned long long)e.J * sizeof(T) * 7 + 128 < (1ull << 32);
    auto kern = sg ? (imm ? gram2_kernel<T, true, true> : gram2_kernel<T, true, false>)
                   : (imm ? gram2_kernel<T, false, true> : gram2_kernel<T, false, false>);
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const Gram2Tab tab = gram2_tab(gp);
    {
        ProfScope prof(e, (e.profile_gap_only && part == 0) ? -1 : 0, s, true);      // (gap-only: the second launch's stop, nothing else)
        if (prof.on())
            hipExtLaunchKernelGGL(kern, dim3(pl.total_wgs), dim3(G2_THREADS), (unsigned)lds, s, prof.a, prof.b, 0,
                                  (const T*)U, (const T*)G, (const T*)e.d_shiftT, e.p, e.n, (long long)e.J, tab);
        else
        hipLaunchKernelGGL(kern, dim3(pl.total_wgs), dim3(G2_THREADS), lds, s, (const T*)U, (const T*)G,
                           (const T*)e.d_shiftT, e.p, e.n, (long long)e.J, tab);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

int launch_gram2(Engine& e, int part, const void* U, const void* G, hipStream_t s) {
    return e.cfg.dtype == CESX_F32 ? launch_gram2_t<float>(e, part, U, G, s) : launch_gram2_t<double>(e, part, U, G, s);
}

// Both parts in one launch + the side stream's wait for its head (see gram2_fused_kernel).  -1: does not qualify.
bool gram2_fused_qualifies(Engine& e, const void* U, const void* G) {
    const int KT = G2_ROWB / (int)e.esz;
    if (!e.gram_v2 || e.J % KT != 0 || e.J < KT || ((uintptr_t)U & 15) || ((uintptr_t)G & 15)) return false;
    const GramPlan &pa = e.gp[0].plan, &pb = e.gp[1].plan;
    if (pa.nblocks == 0 || pb.nblocks == 0) return false;
    if (std::max(pa.max_rb, pb.max_rb) * pa.tile * G2_ROWB > G2_SLOT_IMM) return false;          // (the reduce duty's LDS lies behind two 60-KiB slots)
    if ((unsigned long long)pa.total_slabs * pa.tile * pa.tile * e.esz >= (1ull << 31)) return false;   // 32-bit buffer offsets
    // scalar DMA addressing (the per-lane form needs more registers), and the build must have kept the kernel within
    // 120 registers per lane: only then does the side stream's wait kernel fit on a CU beside a Gram workgroup
    if (!(e.p % 8 == 0 && e.P % 8 == 0 && (unsigned long long)e.J * e.esz * 7 + 128 < (1ull << 32))) return false;
    static int regs_ok[2] = {-1, -1};
    int& ok = regs_ok[e.cfg.dtype == CESX_F32 ? 0 : 1];
    if (ok < 0) {
        hipFuncAttributes fa{};
        const void* fn = e.cfg.dtype == CESX_F32 ? reinterpret_cast<const void*>(gram2_fused_kernel<float, true>)
                                                 : reinterpret_cast<const void*>(gram2_fused_kernel<double, true>);
        ok = hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0 && fa.numRegs <= 120 ? 1 : 0;
    }
    return ok == 1;
}

template <typename T>
static int launch_gram2_fused_t(Engine& e, const void* U, const void* G, double* mom, hipStream_t s, const MetricFin* fin) {
    if (!gram2_fused_qualifies(e, U, G)) return -1;
    const GramPlan &pa = e.gp[0].plan, &pb = e.gp[1].plan;
    const int lds = 2 * G2_SLOT_IMM + G2_SHTAB + G2_COMB + G2_CTL;
    auto kern = gram2_fused_kernel<T, true>;
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const unsigned long long seq = ++e.fused_seq;
    const int row_lo = std::min(pa.own_lo * pa.tile, e.p + e.n), row_hi = std::min(pa.own_hi * pa.tile, e.p + e.n);
    Gram2Fused f{e.d_gsync + 16 * (seq % 4), e.d_gsync + 16 * ((seq + 1) % 4), reinterpret_cast<unsigned long long*>(e.d_gsync + 64), seq,
                 (unsigned)pa.total_wgs, e.gp[0].d_blk_rc, e.gp[0].d_row_own, pa.nblocks, pa.tile, row_lo, row_hi, e.ml, (long long)e.J, mom};
    const int gram_wgs = std::max(pa.total_wgs, pb.total_wgs);
    const MetricFin mf = fin ? *fin : MetricFin{};
    const Gram2Tab ta = gram2_tab(e.gp[0]), tb = gram2_tab(e.gp[1]);
    {
        ProfScope prof(e, 0, s, true);
        if (prof.on())
            hipExtLaunchKernelGGL(kern, dim3(gram_wgs + (fin ? 1 : 0)), dim3(G2_THREADS), (unsigned)lds, s, prof.a, prof.b, 0,
                                  (const T*)U, (const T*)G, (const T*)e.d_shiftT, e.p, e.n, (long long)e.J, ta, tb, f, gram_wgs, mf);
        else
            hipLaunchKernelGGL(kern, dim3(gram_wgs + (fin ? 1 : 0)), dim3(G2_THREADS), lds, s, (const T*)U, (const T*)G,
                               (const T*)e.d_shiftT, e.p, e.n, (long long)e.J, ta, tb, f, gram_wgs, mf);
    }
    CESX_HIP(hipGetLastError());
    hipLaunchKernelGGL(gram2_wait_ready_kernel, dim3(1), dim3(64), 0, e.side, (const unsigned long long*)f.ready, seq, e.d_scal);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

int launch_gram2_fused(Engine& e, const void* U, const void* G, double* mom, hipStream_t s, const MetricFin* fin) {
    if (!e.gram_v2) return -1;
    return e.cfg.dtype == CESX_F32 ? launch_gram2_fused_t<float>(e, U, G, mom, s, fin) : launch_gram2_fused_t<double>(e, U, G, mom, s, fin);
}

}  // namespace cesx
