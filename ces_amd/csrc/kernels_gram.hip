// K1 -- shifted second moments of the stacked ensemble Z = [U - s_u ; G - s_g]
// (replaces np.cov(U0) ces/calibrate.py:424/476/512, the J x J matrix
// D = (1/J) E^T Gamma^{-1} R of :429/:461/:503 and np.cov(Geval) :440/:472:
// everything those lines produce factors through Z Z^T).
//
// Z is (P = p + n) x J, row-major, J contiguous.  The Gram Z Z^T is only
// P x P but its contraction length is J, so the kernel is split-K over J:
//   * the lower-triangular TILE x TILE output blocks are dealt to "workgroup
//     types"; a type holds up to 4 * NBW blocks in the accumulators of its 4
//     waves (1 wave per SIMD, ~270 accumulator VGPRs of the 512-entry file)
//   * every workgroup streams its J-slice through LDS in KT-wide tiles
//     (register-staged, double-buffered, shift subtracted on the way in) and
//     feeds v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 from 16-byte LDS
//     fragment reads (A and B operands are both "row of Z, 8 consecutive j")
//   * per-slice partial blocks go to a slab; a second kernel sums the slabs in
//     fp64 in a fixed order (deterministic, no float atomics).
// Bound: MFMA (f32 MFMA issues at the f32 vector rate on gfx950).
#include "cesx_internal.h"
#include <cstring>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <set>
#include <type_traits>

namespace cesx {

constexpr int GRAM_THREADS = 1024;         // 16 waves = 4 per SIMD: the MFMA pipe always finds a ready wave
constexpr int GRAM_WAVES = GRAM_THREADS / 64;
#ifndef GRAM_ABL     // timing ablations of tools/gram_bench.hip (results are wrong when set)
#define GRAM_ABL 0
#endif
constexpr int ROW_BYTES = 128;            // one staged row of a tile: 32 f32 / 16 f64
constexpr int ROW_STRIDE = ROW_BYTES + 16;  // +16 B pad: conflict-free ds_read_b128 over 16 rows
constexpr int MAX_STAGE_ROWS = 512;
constexpr int MAXCH = MAX_STAGE_ROWS * (ROW_BYTES / 16) / GRAM_THREADS;   // 16 chunks / thread


#ifdef GRAM_CLOCKS   // dev instrumentation (tools/gram_bench.hip): per-workgroup cycle stamps
__device__ long long g_gram_clk[4096 * 4];
#endif

template <typename T, bool ALIGNED>
__global__ __launch_bounds__(GRAM_THREADS, 4)
void gram_kernel(const T* __restrict__ U, const T* __restrict__ G, const T* __restrict__ shift,
                 int p, int n, long long J, const int* __restrict__ type_hdr, int ntypes,
                 const int* __restrict__ rows_tab, const int* __restrict__ wblk,
                 T* __restrict__ slabs, double* __restrict__ rowsum_part) {
    using M = Mfma<T>;
    using vec_t = typename M::vec_t;
    using acc_t = typename M::acc_t;
    constexpr int TILE = M::TILE, VEC = M::VEC, NBW = GramCfg<T>::NBW;
    constexpr int KT = ROW_BYTES / (int)sizeof(T);       // j per tile
    constexpr int NGROUP = KT / GROUP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GRAM_CLOCKS
    const long long gclk0 = clock64(), gw0 = wall_clock64();
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int type = 0;                                   // the type whose workgroup range holds blockIdx.x
    for (int t = 1; t < ntypes; ++t)
        if ((int)blockIdx.x >= type_hdr[t * 8 + 4]) type = t;
    const int* hdr = type_hdr + type * 8;
    const int nrb = hdr[0], rows_off = hdr[1], blocks_off = hdr[2], nblk_t = hdr[3];
    const int slice = (int)blockIdx.x - hdr[4], nslices = hdr[5];
    const int slab0 = hdr[6], rs0 = hdr[7];
    const int nrows = nrb * TILE;
    const int P = p + n;
    const int buf_bytes = nrows * ROW_STRIDE;

    // this wave's block list (wave-uniform -> SGPRs)
    // (compact row of A) | (compact row of B) << 8, one SGPR per block
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

    // J-slice of this workgroup, in tiles
    const long long ntiles = (J + KT - 1) / KT;
    const long long tps = (ntiles + nslices - 1) / nslices;
    const long long t0 = (long long)slice * tps;
    const long long t1 = t0 + tps < ntiles ? t0 + tps : ntiles;

    // staging: thread handles 16-byte chunk `part` of rows (tid/8 + 128*i)
    const int part = tid & 7, row0 = tid >> 3;
    constexpr int RPP = GRAM_THREADS / 8;              // rows covered per pass (64)
    const int nch = (nrows + RPP - 1) / RPP;           // chunks this thread handles
    vec_t stage[MAXCH];
    T rs[MAXCH];                                       // running sum of this thread's part of each staged row
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) rs[i] = 0;
    // per staged row: source pointer (null = zero row) and shift, kept in LDS
    const T** rowptr = reinterpret_cast<const T**>(smem + 2 * buf_bytes);
    T* rowshift = reinterpret_cast<T*>(smem + 2 * buf_bytes + nrows * 8);
    for (int row = tid; row < nrows; row += GRAM_THREADS) {
        const int gr = (rows_tab[rows_off + row / TILE] & 0xffff) * TILE + row % TILE;
        const T* ptr = nullptr;
        T sh = 0;
        if (gr < P) {
            ptr = gr < p ? U + (size_t)gr * J : G + (size_t)(gr - p) * J;
            sh = shift[gr];
        }
        rowptr[row] = ptr;
        rowshift[row] = sh;
    }
    __syncthreads();

    auto load_tile = [&](long long t) {
        const long long j = t * KT + part * VEC;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int row = row0 + RPP * i;
            if (i < nch && row < nrows) {
                vec_t v;
#pragma unroll
                for (int c = 0; c < VEC; ++c) v[c] = 0;
                const T* ptr = rowptr[row];
                if (ptr != nullptr) {
                    if (ALIGNED && j + VEC <= J) {
                        v = *reinterpret_cast<const vec_t*>(ptr + j);
                    } else {
                        // out-of-range particles must contribute (x - s) = 0: load the shift
                        const T sh = rowshift[row];
#pragma unroll
                        for (int c = 0; c < VEC; ++c) v[c] = j + c < J ? ptr[j + c] : sh;
                    }
                }
                stage[i] = v;
            }
        }
    };
    // chunk i of the staged tile: subtract the shift, accumulate the row sum, write to LDS
    auto store_chunk = [&](int buf, auto ic) {
        constexpr int i = decltype(ic)::value;
        const int row = row0 + RPP * i;
        if (i < nch && row < nrows) {
            char* base = smem + buf * buf_bytes;
            vec_t v = stage[i];
            const T sh = rowshift[row];
#pragma unroll
            for (int c = 0; c < VEC; ++c) { v[c] -= sh; rs[i] += v[c]; }
            *reinterpret_cast<vec_t*>(base + row * ROW_STRIDE + part * 16) = v;
        }
    };
    auto store_tile = [&](int buf) {
        store_chunk(buf, std::integral_constant<int, 0>{});
        if constexpr (MAXCH > 1) store_chunk(buf, std::integral_constant<int, 1>{});
        if constexpr (MAXCH > 2) store_chunk(buf, std::integral_constant<int, 2>{});
        if constexpr (MAXCH > 3) store_chunk(buf, std::integral_constant<int, 3>{});
        static_assert(MAXCH <= 4, "store_tile handles up to 4 chunks per thread");
    };

    const int laneoff = (lane % TILE) * ROW_STRIDE + (lane / TILE) * 16;

    if (t0 < t1) {
        load_tile(t0);
        store_tile(0);
    }
    __syncthreads();
#ifdef GRAM_CLOCKS
    const long long gclk1 = clock64();
#endif
    for (long long t = t0; t < t1; ++t) {
        const int cur = (int)((t - t0) & 1);
        if (!(GRAM_ABL & 1) && t + 1 < t1) load_tile(t + 1);
        const char* base = smem + cur * buf_bytes + laneoff;
        // The staged chunks of tile t+1 are written to the other LDS buffer BETWEEN the MFMA
        // groups of tile t instead of after them: with one workgroup per CU all 16 waves would
        // otherwise leave the matrix pipe idle together while they store.
        const bool stage_next = !(GRAM_ABL & 1) && t + 1 < t1;
        auto compute_group = [&](int g) {
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                if (b < nb) {
                    vec_t a, c;
                    if (GRAM_ABL & 4) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { a[v] = (T)(lane + v + b); c[v] = (T)(lane - v + g); }
                    } else {
                        a = *reinterpret_cast<const vec_t*>(
                            base + (iab[b] & 0xff) * (TILE * ROW_STRIDE) + g * (GROUP * (int)sizeof(T)));
                        c = *reinterpret_cast<const vec_t*>(
                            base + (iab[b] >> 8) * (TILE * ROW_STRIDE) + g * (GROUP * (int)sizeof(T)));
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[b] = M::mma(a[v], c[v], acc[b]);
                }
            }
            asm volatile("" ::: "memory");
        };
        constexpr int CPG = MAXCH / NGROUP;      // chunks stored after each group (f32: 1, f64: 2)
        static_assert(MAXCH % NGROUP == 0 && CPG >= 1 && CPG <= 2, "chunk / group interleave");
        compute_group(0);
        if (stage_next) { store_chunk(cur ^ 1, std::integral_constant<int, 0>{}); if constexpr (CPG == 2) store_chunk(cur ^ 1, std::integral_constant<int, 1>{}); }
        compute_group(1);
        if (stage_next) { store_chunk(cur ^ 1, std::integral_constant<int, CPG>{}); if constexpr (CPG == 2) store_chunk(cur ^ 1, std::integral_constant<int, 3>{}); }
        if constexpr (NGROUP == 4) {
            compute_group(2);
            if (stage_next) store_chunk(cur ^ 1, std::integral_constant<int, 2>{});
            compute_group(3);
            if (stage_next) store_chunk(cur ^ 1, std::integral_constant<int, 3>{});
        }
        if (!(GRAM_ABL & 2)) __syncthreads();
    }

#ifdef GRAM_CLOCKS
    const long long gclk2 = clock64();
#endif
    // shifted row sums of this slice (first moments): the 8 threads that share a row
    // are adjacent lanes; only the type that owns the block row reports it
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int row = row0 + RPP * i;
        if (i < nch && row < nrows) {
            double v = (double)rs[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int ent = rows_tab[rows_off + row / TILE];
            const int gr = (ent & 0xffff) * TILE + row % TILE;
            if (part == 0 && (ent >> 16) != 0 && gr < P) rowsum_part[(size_t)(rs0 + slice) * P + gr] = v;
        }
    }

    // partial blocks of this slice
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        if (b < nb) {
            const int ob = __builtin_amdgcn_readfirstlane(wblk[(size_t)(blocks_off + wave * NBW + b) * 3 + 2]);
            // accumulator-major slab (slab_group_rc): 16-byte stores of consecutive lanes
            T* out = slabs + ((size_t)slab0 + (size_t)slice * nblk_t + ob) * (TILE * TILE);
#pragma unroll
            for (int q = 0; q < M::NACC / VEC; ++q) {
                vec_t v;
#pragma unroll
                for (int c = 0; c < VEC; ++c) v[c] = acc[b][q * VEC + c];
                *reinterpret_cast<vec_t*>(out + (size_t)(q * 64 + lane) * VEC) = v;
            }
        }
    }
#ifdef GRAM_CLOCKS
    if (tid == 0 && blockIdx.x < 4096) {
        g_gram_clk[blockIdx.x * 4 + 0] = gclk1 - gclk0; g_gram_clk[blockIdx.x * 4 + 1] = gclk2 - gclk1;
        g_gram_clk[blockIdx.x * 4 + 2] = clock64() - gclk2; g_gram_clk[blockIdx.x * 4 + 3] = wall_clock64() - gw0;
    }
#endif
}

// Sum the per-slice partial blocks in fp64 (fixed order) and scatter them into
// the packed moment buffer as full symmetric S_aa, S_ab, S_bb.  A workgroup
// handles RED_G 16-byte groups (4 f32 / 2 f64 along a block row) x RED_S slice
// parts (each lane sums nslices / RED_S slices with up to 8 loads in flight);
// the parts are combined through LDS in a fixed order.
constexpr int RED_G = 32, RED_S = 8;       // (16 / 32 slice parts, i.e. 512 / 1024 threads: +4 / +10 us per step at C2, round 3)
template <typename T>
__global__ __launch_bounds__(RED_G * RED_S)
void gram_reduce_kernel(const T* __restrict__ slabs, const int* __restrict__ blk_rc, const int* __restrict__ row_own,
                        int nblocks, int tile, MomLayout ml, long long J, int row_lo, int row_hi,
                        int write_N, const double* __restrict__ rowsum_part, const double* __restrict__ tail_src,
                        double* __restrict__ mom, const MetricFin fin) {
    using vec_t = typename Mfma<T>::vec_t;
    constexpr int VEC = Mfma<T>::VEC;
    // the previous update's metric finalisation + publication, riding on this launch as one extra workgroup -- the
    // FIRST one: its chain of dependent loads, fences and the write to host memory (~8 us) starts with the launch and
    // ends inside it
    if (fin.part != nullptr && blockIdx.x == 0) { metric_final_body(fin); return; }
    const unsigned bid = fin.part != nullptr ? blockIdx.x - 1 : blockIdx.x;
    const int p = ml.p, n = ml.n;
    __shared__ double part[RED_S][RED_G][VEC];
    const int tt = tile * tile;
    const long long ngroups = (long long)nblocks * tt / VEC;
    const int gq = threadIdx.x / RED_G, gl = threadIdx.x % RED_G;
    const long long idx = (long long)bid * RED_G + gl;
    if ((long long)bid * RED_G >= ngroups) {
        // tail workgroups: N, the first moments sum_j (z_ij - s_i) of the rows this launch owns
        // (16 rows x 16 slice parts per workgroup), and (second launch) the lagged data-metric
        // sums that ride at the end of the buffer
        const int tw = (int)(((long long)bid * RED_G - ngroups) / RED_G);
        const long long r = row_lo + (long long)tw * RED_G + gl;
        if (tw == 0 && threadIdx.x == 0) {
            if (write_N) mom[0] = (double)J;
            if (tail_src) { mom[ml.tail()] = tail_src[0]; mom[ml.tail() + 1] = tail_src[1]; }
        }
        double s = 0.0;
        if (r < row_hi) {
            const int rs0 = row_own[(r / tile) * 2], nslices = row_own[(r / tile) * 2 + 1];
            const int per = (nslices + RED_S - 1) / RED_S;
            const int k1 = min(nslices, (gq + 1) * per);
#pragma unroll 8
            for (int k = gq * per; k < k1; ++k) s += rowsum_part[(size_t)(rs0 + k) * (p + n) + r];
        }
        part[gq][gl][0] = s;
        __syncthreads();
        if (gq == 0 && r < row_hi) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < RED_S; ++q) t += part[q][gl][0];
            mom[r < p ? ml.sa() + r : ml.sb() + (r - p)] = t;
        }
        return;
    }
    const bool on = idx < ngroups;
    const int blk = on ? (int)(idx / (tt / VEC)) : 0, e0 = on ? (int)(idx % (tt / VEC)) * VEC : 0;
    double acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.0;
    const int* info = blk_rc + blk * 5;
    const T* src = slabs + (size_t)info[2] * tt + e0;
    const size_t stride = (size_t)info[3] * tt;
    const int nslices = info[4];
    const int per = (nslices + RED_S - 1) / RED_S;
    const int k1 = min(nslices, (gq + 1) * per);
    int k = gq * per;
    if (on) {
        for (; k + 8 <= k1; k += 8) {
            vec_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vec_t*>(src + (size_t)(k + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[c] += (double)v[u][c];
        }
        for (; k < k1; ++k) {
            const vec_t v = *reinterpret_cast<const vec_t*>(src + (size_t)k * stride);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] += (double)v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) part[gq][gl][c] = acc[c];
    __syncthreads();
    if (gq != 0 || !on) return;
    const int R = info[0], C = info[1];
    const int P = p + n;
    double* Saa = mom + ml.Saa();
    double* Sab = mom + ml.Sab();
    double* Sbb = mom + ml.Sbb();
    int row0, col, rstep;
    slab_group_rc<T>(e0 / VEC, row0, col, rstep);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const int gr = R * tile + row0 + c * rstep, gc = C * tile + col;
        if (gr >= P || gc >= P) continue;
        if (R == C && gc > gr) continue;          // diagonal block: lower half, mirrored below
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < RED_S; ++q) s += part[q][gl][c];
        if (gr < p) {                              // both in U (gr >= gc)
            Saa[(size_t)gr * p + gc] = s;
            Saa[(size_t)gc * p + gr] = s;
        } else if (gc < p) {                       // gr in G, gc in U
            Sab[(size_t)gc * n + (gr - p)] = s;
        } else {
            Sbb[(size_t)(gr - p) * n + (gc - p)] = s;
            Sbb[(size_t)(gc - p) * n + (gr - p)] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// host: work partition
// ---------------------------------------------------------------------------
// Cycles one J tile costs a workgroup of a type with `blocks` blocks over `row_blocks` staged block rows: the MFMAs of
// its busiest SIMD (blocks / 4 SIMDs, 64 cycles per MFMA, tile-width / k MFMAs per block) plus the part that does
// not shrink with the block count -- staging the type's rows and the barrier.  Constants fitted to in-kernel cycle
// stamps of the LDS-DMA kernel (tools/gram2_bench.hip, f32 and f64 types of 11 ... 121 blocks): 1.055 x the MFMA
// cycles, 3 cycles per staged row (DMA issue, the in-place shift pass), 450 per tile (barrier, loop).
static double gram_tile_cost(int tile, int blocks, int row_blocks) {
    const double mfma_cyc = tile == 32 ? 16 * 64.0 : 4 * 64.0;      // f32: 32 j / k=2; f64: 16 j / k=4
    // a type that stages more than 60 KiB per tile runs with one address add per fragment read (kernels_gram2.hip, IMM)
    const double addr = row_blocks * tile * 128 > 60 * 1024 ? 6.0 * (double)((blocks + 3) / 4) * 4.0 : 0.0;
    return 1.055 * mfma_cyc * (double)((blocks + 3) / 4) + 450.0 + 3.0 * (double)(row_blocks * tile) + addr;
}

// Slices per type that level the launch: whole tiles per slice, the busiest workgroup (tiles x cost per tile) as
// light as the workgroup budget allows.  Returns that workgroup's cycles.
static double gram_level_slices(const std::vector<double>& w, int budget, long long ntiles, std::vector<int>& nsl) {
    const int nt = (int)w.size();
    nsl.assign(nt, 1);
    if (ntiles <= 0) return 0.0;
    auto need = [&](double T, std::vector<int>* out) {
        long long total = 0;
        for (int t = 0; t < nt; ++t) {
            const long long tps = (long long)std::floor(T / w[t] + 1e-9);
            if (tps < 1) return (long long)1 << 40;
            const long long n_ = (ntiles + tps - 1) / tps;
            if (out) (*out)[t] = (int)n_;
            total += n_;
        }
        return total;
    };
    // candidates: k tiles of some type
    double lo = 0.0, hi = 0.0;
    for (int t = 0; t < nt; ++t) hi = std::max(hi, w[t] * (double)ntiles);
    if (need(hi, nullptr) > budget) {          // fewer workgroups than types cannot happen (budget >= nt): one slice each
        return hi;
    }
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (need(mid, nullptr) <= budget) hi = mid; else lo = mid;
    }
    need(hi, &nsl);
    double worst = 0.0;
    for (int t = 0; t < nt; ++t) worst = std::max(worst, w[t] * (double)((ntiles + nsl[t] - 1) / nsl[t]));
    return worst;
}

GramPlan make_gram_plan(int P, int tile, int nbw, int max_rows_lds, int subset, int pbU, int min_types,
                        int wg_budget, long long ntiles) {
    GramPlan pl;
    pl.tile = tile;
    pl.nbw = nbw;
    pl.nbr = (P + tile - 1) / tile;
    const int cap = GRAM_WAVES * nbw;
    auto wanted = [&](int R, int C) {
        const bool uu = R < pbU && C < pbU;
        return subset == 0 || (subset == 1 ? uu : !uu);
    };
    pl.own_lo = subset == 2 ? std::min(pbU, pl.nbr) : 0;
    pl.own_hi = subset == 1 ? std::min(pbU, pl.nbr) : pl.nbr;
    std::vector<std::vector<std::pair<int, int>>> types;   // blocks (R, C) per type
    if (pl.nbr * tile <= max_rows_lds) {
        // all rows fit in LDS: chop the lower triangle, listed row by row OR column by column, into runs.  Which
        // listing is the better one depends on the subset: the second launch's blocks (G x U, then G x G) cut by
        // COLUMNS give types that stage 14 + 10 block rows at C2 (U columns 0-5 | U columns 6-7 and G x G), cut by
        // rows 13 + 16 -- fewer rows to fetch, to shift and to sum, and no type needs all 16 (the LDS-DMA kernel
        // then keeps its fragment addresses in registers, kernels_gram2.hip IMM).
        std::vector<std::pair<int, int>> all_rm, all_cm;
        for (int R = 0; R < pl.nbr; ++R)
            for (int C = 0; C <= R; ++C)
                if (wanted(R, C)) all_rm.push_back({R, C});
        for (int C = 0; C < pl.nbr; ++C)
            for (int R = C; R < pl.nbr; ++R)
                if (wanted(R, C)) all_cm.push_back({R, C});
        const int nb_all = (int)all_rm.size();
        const int nt = std::max(std::max(1, min_types), (nb_all + cap - 1) / cap);
        // equal runs, rounded to multiples of 4 blocks: the 4 SIMDs of a workgroup then carry the same
        // number of blocks (the barrier of every J tile waits for the busiest SIMD); the last type
        // takes what is left
        int per = std::max(1, (nb_all + nt - 1) / nt);
        if (per > 4) per = std::min(cap / 4 * 4, (per + 3) / 4 * 4);
        std::vector<int> sizes0;
        for (int lo = 0; lo < nb_all; lo += per) sizes0.push_back(std::min(nb_all - lo, per));
        // ... unless another split of the same number of runs packs the launch better: tiles per slice are whole
        // numbers (at C2 two row-major runs of 52 + 48 blocks over 248 workgroups end at 16 and 18 tiles per slice,
        // the second 7 % above the mean; 44 + 56 end at 19 and 15, level).  Two- and three-run plans are searched in
        // steps of 4 blocks with the cost model below, over both listings.
        // a candidate = (listing, run sizes): its levelled launch time (busiest workgroup) and the block rows all its
        // workgroups stage per tile (what the launch fetches, shifts and sums); the lightest within 1 % of the fastest wins
        struct Cand { const std::vector<std::pair<int, int>>* all; std::vector<int> sz; double cost, rows; };
        auto eval = [&](const std::vector<std::pair<int, int>>& all, const std::vector<int>& sz) {
            std::vector<double> w;
            std::vector<int> nrw;
            int lo = 0;
            for (int n_ : sz) {
                std::set<int> rws;
                for (int q = lo; q < lo + n_; ++q) { rws.insert(all[q].first); rws.insert(all[q].second); }
                w.push_back(gram_tile_cost(tile, n_, (int)rws.size()));
                nrw.push_back((int)rws.size());
                lo += n_;
            }
            std::vector<int> ns;
            Cand c{&all, sz, 0.0, 0.0};
            c.cost = gram_level_slices(w, std::max(wg_budget, (int)sz.size()), ntiles, ns);
            for (size_t t = 0; t < sz.size(); ++t) c.rows += (double)nrw[t] * ns[t];
            return c;
        };
        // (fp64: equal runs.  The cost model was fitted to the tiles, not to what a workgroup does once per launch: for C2 in
        //  fp64 the search cut the U x U launch's 136 blocks into 8 + 128 -- 228 workgroups of 18 tiles with 256 KB of slabs
        //  each -- where 68 + 68 runs the two Gram launches in 0.362 instead of 0.408 ms, round 4)
        const bool search = sizes0.size() >= 2 && sizes0.size() <= 3 && tile == 32;
        std::vector<Cand> cands;
        cands.push_back(eval(all_rm, sizes0));
        for (const auto* all : {&all_rm, &all_cm}) {
            if (all == &all_cm && sizes0.size() < 2) break;
            if (all == &all_cm) cands.push_back(eval(*all, sizes0));
            if (!search) continue;
            std::vector<int> cand(sizes0.size());
            for (int a = 4; a <= std::min(cap, nb_all - 4); a += 4) {
                if (sizes0.size() == 2) {
                    cand = {a, nb_all - a};
                    if (cand[1] > cap) continue;
                    cands.push_back(eval(*all, cand));
                } else {
                    for (int b = 4; b <= std::min(cap, nb_all - a - 4); b += 4) {
                        cand = {a, b, nb_all - a - b};
                        if (cand[2] > cap) continue;
                        cands.push_back(eval(*all, cand));
                    }
                }
            }
        }
        size_t pick = 0;
        {
            double best = cands[0].cost;
            for (const Cand& c : cands) best = std::min(best, c.cost);
            // (the equal-runs row-major plan stays unless something is clearly better: 0.5 % as before, and then the
            //  lightest of the candidates within 1 % of the fastest)
            if (best < cands[0].cost * 0.995) {
                double rows = 1e300;
                for (size_t i = 0; i < cands.size(); ++i)
                    if (cands[i].cost <= best * 1.01 && cands[i].rows < rows) { rows = cands[i].rows; pick = i; }
            }
        }
        const std::vector<std::pair<int, int>>* best_all = cands[pick].all;
        const std::vector<int> sizes = cands[pick].sz;
        size_t lo = 0;
        for (int n_ : sizes) {
            std::vector<std::pair<int, int>> v(best_all->begin() + lo, best_all->begin() + lo + n_);
            types.push_back(v);
            lo += n_;
        }
    } else {
        // rectangles a x b of blocks with (a + b) * tile rows staged.  Which a x b: every shape that fits the wave
        // capacity and LDS is laid over the triangle and priced with the cost model (levelled busiest workgroup, then the
        // rows the launch stages).  Round 4: the near-square 11 x 11 that sqrt(capacity) gives leaves C5's second launch
        // (64 block rows) with 18 types, three of them 11-block leftovers -- busiest SIMD 12 % above the mean and 343 staged
        // block rows; 16 x 8 divides the 64 rows evenly: 14 types, 3 % above the mean, 288 block rows.
        auto build = [&](int a, int b) {
            std::vector<std::vector<std::pair<int, int>>> ts;
            for (int R0 = 0; R0 < pl.nbr; R0 += a)
                for (int C0 = 0; C0 <= R0 + a - 1 && C0 < pl.nbr; C0 += b) {
                    std::vector<std::pair<int, int>> v;
                    for (int R = R0; R < std::min(R0 + a, pl.nbr); ++R)
                        for (int C = C0; C < std::min(C0 + b, pl.nbr); ++C)
                            if (C <= R && wanted(R, C)) v.push_back({R, C});
                    if (!v.empty()) ts.push_back(v);
                }
            return ts;
        };
        int a0 = std::max(1, (int)std::floor(std::sqrt((double)cap)));
        int b0 = std::max(1, cap / a0);
        while ((a0 + b0) * tile > max_rows_lds && (a0 > 1 || b0 > 1)) {
            if (a0 >= b0) --a0; else --b0;
        }
        int best_a = a0, best_b = b0;
        {
            double best_cost = 1e300, best_rows = 1e300;
            for (int a = 1; a <= std::min(cap, pl.nbr); ++a) {
                const int b = std::min(cap / a, pl.nbr);
                if (b < 1 || (a + b) * tile > max_rows_lds) continue;
                if (a * b * 2 < cap) continue;                      // (half-empty workgroups: not worth pricing)
                const auto ts = build(a, b);
                std::vector<double> w;
                double rows = 0.0;
                for (const auto& v : ts) {
                    std::set<int> rws;
                    for (auto& rc : v) { rws.insert(rc.first); rws.insert(rc.second); }
                    w.push_back(gram_tile_cost(tile, (int)v.size(), (int)rws.size()));
                    rows += (double)rws.size();
                }
                if ((int)ts.size() > wg_budget) continue;
                std::vector<int> ns;
                const double cost = gram_level_slices(w, std::max(wg_budget, (int)ts.size()), ntiles, ns);
                // (the near-square default stays unless something is clearly better; then the lightest within 1 % of the fastest)
                const bool better = cost < best_cost * 0.99 || (cost <= best_cost * 1.01 && rows < best_rows);
                if (better) { best_cost = std::min(best_cost, cost); best_rows = rows; best_a = a; best_b = b; }
            }
        }
        types = build(best_a, best_b);
    }
    pl.ntypes = (int)types.size();
    pl.max_rb = 0;
    // Slices per type, proportional to the cycles one J tile costs a workgroup of that type: the
    // MFMAs of its busiest SIMD (blocks / 4 SIMDs, 64 cycles per MFMA, tile-width / k MFMAs per
    // block) plus the part that does not shrink with the block count -- staging the type's rows
    // (16 B per lane-store, ~80 B/clk/CU) and the barrier.  Largest-remainder rounding within the
    // workgroup budget; never more slices than J tiles.
    std::vector<int> nsl(pl.ntypes, 1);
    std::vector<double> w(pl.ntypes);
    double total = 0.0;
    for (int t = 0; t < pl.ntypes; ++t) {
        std::set<int> rws;
        for (auto& rc : types[t]) { rws.insert(rc.first); rws.insert(rc.second); }
        w[t] = gram_tile_cost(tile, (int)types[t].size(), (int)rws.size());
        total += w[t];
    }
    // (many types -- the rectangles of a triangle that does not fit LDS: levelled as well since round 4, the proportional
    //  rule below rounds 17.5 slices down to 16 for eight types at once; round 4)
    if (pl.ntypes <= 4 || pl.nbr * tile > max_rows_lds) {
        // few types, many slices each: whole tiles per slice, levelled (gram_level_slices)
        gram_level_slices(w, std::max(wg_budget, pl.ntypes), ntiles, nsl);
    } else {
        const int budget = std::max(wg_budget, pl.ntypes);
        // Types of (nearly) the same cost get the SAME number of slices, a multiple of 8 when there
        // are that many: slice k of every such type then covers the same J range on the same XCD
        // (workgroup id mod 8), and the rows the types share are fetched from HBM once (at C2 the two
        // types of the second launch read 131 MB instead of 238 MB).
        // ... unless there are many types with few slices each: rounding 17.5 down to 16 then costs more (the
        // busiest workgroup sets the launch time) than the shared fetches save
        const int align_from = pl.ntypes > 4 ? 64 : 16;
        std::vector<double> want(pl.ntypes);
        for (int t = 0; t < pl.ntypes; ++t) want[t] = total > 0 ? (double)budget * w[t] / total : 1.0;
        std::vector<int> order(pl.ntypes);
        for (int t = 0; t < pl.ntypes; ++t) order[t] = t;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return want[a] > want[b]; });
        int used = 0;
        std::vector<std::pair<size_t, size_t>> groups;     // [first, last) in `order`
        for (size_t i = 0; i < order.size();) {
            size_t j = i;
            double sum = 0.0;
            while (j < order.size() && want[order[j]] >= 0.96 * want[order[i]]) sum += want[order[j++]];
            int each = std::max(1, (int)std::floor(sum / (double)(j - i)));
            if (each >= align_from) each -= each % 8;
            for (size_t q = i; q < j; ++q) { nsl[order[q]] = each; used += each; }
            groups.push_back({i, j});
            i = j;
        }
        // workgroups left over by the rounding go to whole groups (members stay equal), the one
        // whose members carry the most work per slice first
        for (bool any = true; any;) {
            any = false;
            size_t best = groups.size();
            double load = 0.0;
            for (size_t g = 0; g < groups.size(); ++g) {
                const int t0 = order[groups[g].first];
                const int inc = nsl[t0] >= align_from ? 8 : 1;
                if ((int)(groups[g].second - groups[g].first) * inc > budget - used) continue;
                const double l = want[t0] / (double)nsl[t0];
                if (l > load) { load = l; best = g; }
            }
            if (best < groups.size()) {
                const int inc = nsl[order[groups[best].first]] >= align_from ? 8 : 1;
                for (size_t q = groups[best].first; q < groups[best].second; ++q) { nsl[order[q]] += inc; used += inc; }
                any = true;
            }
        }
        for (int t = 0; t < pl.ntypes; ++t)
            if ((long long)nsl[t] > ntiles) nsl[t] = (int)std::max<long long>(1, ntiles);
    }
    // output block ids: position in the row-major list of this plan's blocks
    std::vector<int> idmap((size_t)pl.nbr * pl.nbr, -1);
    pl.nblocks = 0;
    pl.blk_rc.clear();
    for (int R = 0; R < pl.nbr; ++R)
        for (int C = 0; C <= R; ++C)
            if (wanted(R, C)) {
                idmap[(size_t)R * pl.nbr + C] = pl.nblocks++;
                for (int q : {R, C, 0, 0, 0}) pl.blk_rc.push_back(q);
            }
    auto out_id = [&](int R, int C) { return idmap[(size_t)R * pl.nbr + C]; };
    pl.row_own.assign((size_t)pl.nbr * 2, 0);
    std::set<int> owned;
    int wg0 = 0, slab0 = 0, rs0 = 0;
    for (int t = 0; t < pl.ntypes; ++t) {
        const auto& v = types[t];
        std::vector<int> rows;
        for (auto& rc : v) { rows.push_back(rc.first); rows.push_back(rc.second); }
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        auto compact = [&](int r) { return (int)(std::lower_bound(rows.begin(), rows.end(), r) - rows.begin()); };
        pl.max_rb = std::max(pl.max_rb, (int)rows.size());
        const int nv = (int)v.size();
        for (int q : {(int)rows.size(), (int)pl.rows.size(), (int)(pl.wblk.size() / 3), nv, wg0, nsl[t], slab0, rs0})
            pl.type_hdr.push_back(q);
        for (int r : rows) {
            // the first type that stages an owned block row reports its sums
            const bool first = r >= pl.own_lo && r < pl.own_hi && owned.insert(r).second;
            pl.rows.push_back(r | (first ? 1 << 16 : 0));
            if (first) { pl.row_own[(size_t)r * 2] = rs0; pl.row_own[(size_t)r * 2 + 1] = nsl[t]; }
        }
        // waves w, w+4, w+8, ... share a SIMD: the first (nv % waves) waves take one extra
        // block, which keeps the per-SIMD totals within one block of each other
        int next = 0;
        for (int w = 0; w < GRAM_WAVES; ++w) {
            const int cnt = nv / GRAM_WAVES + (w < nv % GRAM_WAVES ? 1 : 0);
            const int lo = next, hi = next + cnt;
            next = hi;
            for (int b = 0; b < nbw; ++b) {
                if (lo + b < hi) {
                    pl.wblk.push_back(compact(v[lo + b].first));
                    pl.wblk.push_back(compact(v[lo + b].second));
                    pl.wblk.push_back(lo + b);                      // block index inside the type
                    int* info = &pl.blk_rc[(size_t)out_id(v[lo + b].first, v[lo + b].second) * 5];
                    info[2] = slab0 + lo + b; info[3] = nv; info[4] = nsl[t];
                } else {
                    pl.wblk.push_back(-1); pl.wblk.push_back(-1); pl.wblk.push_back(-1);
                }
            }
        }
        wg0 += nsl[t];
        slab0 += nsl[t] * nv;
        rs0 += nsl[t];
    }
    pl.total_wgs = wg0; pl.total_slabs = slab0; pl.total_rs = std::max(rs0, 1);
    return pl;
}

template <typename T>
static int launch_gram_t(Engine& e, int part, const void* U, const void* G, hipStream_t s) {
    GramPart& gp = e.gp[part];
    const GramPlan& pl = gp.plan;
    if (pl.nblocks == 0) return CESX_OK;    // (a tiny problem can have all its blocks in part 0; the reduce still
                                            //  writes part 1's share of the buffer: row sums it owns and the lagged tail)
    int rc2 = e.gram_v2 ? launch_gram2(e, part, U, G, s) : -1;      // LDS-DMA kernel when the shapes allow
    if (rc2 >= 0) return rc2;
    const int lds = 2 * pl.max_rb * pl.tile * ROW_STRIDE + pl.max_rb * pl.tile * 16;
    const bool aligned = (e.J % Mfma<T>::VEC == 0) && ((uintptr_t)U % 16 == 0) && ((uintptr_t)G % 16 == 0);
    dim3 grid(pl.total_wgs), block(GRAM_THREADS);
    auto kern = aligned ? gram_kernel<T, true> : gram_kernel<T, false>;
    CESX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    {
        e.prof_part = part;
        ProfScope prof(e, 0, s);
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const T*)U, (const T*)G, (const T*)e.d_shiftT,
                           e.p, e.n, (long long)e.J, gp.d_type_hdr, pl.ntypes, gp.d_rows, gp.d_wblk,
                           (T*)gp.d_slabs, gp.d_rowsum_part);
    }
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

template <typename T>
static int launch_gram_reduce_t(Engine& e, int part, double* mom, hipStream_t s, hipEvent_t stop, const MetricFin* fin) {
    GramPart& gp = e.gp[part];
    const GramPlan& pl = gp.plan;
    const long long ngroups = (long long)pl.nblocks * pl.tile * pl.tile / Mfma<T>::VEC;
    const int row_lo = std::min(pl.own_lo * pl.tile, e.p + e.n), row_hi = std::min(pl.own_hi * pl.tile, e.p + e.n);
    const long long wgs = (ngroups + RED_G - 1) / RED_G + std::max(1, (row_hi - row_lo + RED_G - 1) / RED_G) + (fin ? 1 : 0);
    const MetricFin f = fin ? *fin : MetricFin{};
    const long long wgs_all = wgs;
    if (stop)
        hipExtLaunchKernelGGL(gram_reduce_kernel<T>, dim3((unsigned)wgs_all), dim3(RED_G * RED_S), 0, s, nullptr, stop, 0,
                              (const T*)gp.d_slabs, (const int*)gp.d_blk_rc, (const int*)gp.d_row_own, pl.nblocks, pl.tile, e.ml,
                              (long long)e.J, row_lo, row_hi, part == 0 ? 1 : 0, (const double*)gp.d_rowsum_part,
                              part == 1 ? (const double*)e.d_metric_sums : (const double*)nullptr, mom, f);
    else
    hipLaunchKernelGGL(gram_reduce_kernel<T>, dim3((unsigned)wgs_all), dim3(RED_G * RED_S), 0, s,
                       (const T*)gp.d_slabs, gp.d_blk_rc, gp.d_row_own, pl.nblocks, pl.tile, e.ml,
                       (long long)e.J, row_lo, row_hi, part == 0 ? 1 : 0, gp.d_rowsum_part,
                       part == 1 ? e.d_metric_sums : (const double*)nullptr, mom, f);
    CESX_HIP(hipGetLastError());
    return CESX_OK;
}

int launch_gram_reduce(Engine& e, int part, double* mom, hipStream_t s, hipEvent_t stop, const MetricFin* fin) {
    return e.cfg.dtype == CESX_F32 ? launch_gram_reduce_t<float>(e, part, mom, s, stop, fin)
                                   : launch_gram_reduce_t<double>(e, part, mom, s, stop, fin);
}

int launch_gram(Engine& e, int part, const void* U, const void* G, double* mom, hipStream_t s, bool no_reduce) {
    int rc = e.cfg.dtype == CESX_F32 ? launch_gram_t<float>(e, part, U, G, s) : launch_gram_t<double>(e, part, U, G, s);
    if (rc != CESX_OK || no_reduce) return rc;
    return launch_gram_reduce(e, part, mom, s, nullptr, nullptr);
}

int gram_nbw(int dtype) { return dtype == CESX_F32 ? GramCfg<float>::NBW : GramCfg<double>::NBW; }
int gram_tile(int dtype) { return dtype == CESX_F32 ? Mfma<float>::TILE : Mfma<double>::TILE; }
int gram_kt(int dtype) { return dtype == CESX_F32 ? ROW_BYTES / 4 : ROW_BYTES / 8; }
int gram_max_stage_rows() { return MAX_STAGE_ROWS; }

}  // namespace cesx
