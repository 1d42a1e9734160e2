// One unit of the fixed-order fp64 slab reduce (kernels_gram.hip): what a 256-thread workgroup of gram_reduce_kernel does,
// callable by a quarter of a larger workgroup as well -- the second Gram launch sums the first launch's slabs in its
// prologue (PreRed, cesx_internal.h; kernels_gram2.hip), with the same arithmetic in the same order, bit for bit.
#pragma once
#include "cesx_internal.h"

namespace cesx {

constexpr int RED_G = 32, RED_S = 8;       // a unit = RED_G 16-byte groups; 256 threads = RED_G x RED_S (16 / 32 thread parts for the
                                           // stand-alone kernel, i.e. 512 / 1024 threads: +4 / +10 us per step at C2, round 3)
constexpr int RED_P = 32;                  // ORDER parts: a sum over the slices is RED_P sequential partial sums of consecutive slices,
                                           // added up in part order -- whatever the number of threads that form them (round 4: 8 threads
                                           // x 4 parts each in the stand-alone kernel, 32 threads x 1 in the Gram launch's prologue,
                                           // where a part is ONE batch of loads instead of four dependent ones)

// Main unit `bid` (RED_G groups of the launch's slabs) by NT = RED_G * (NT / RED_G) threads, thread `vt`; `part_` = RED_P * RED_G
// * VEC doubles of LDS.  Exactly ONE __syncthreads() on every path.  wt: results are stored at agent scope (write-through):
// the consumer is not ordered behind this launch at queue level.
template <typename T, int NT>
__device__ __forceinline__
void gram_reduce_main(unsigned bid, int vt, double* __restrict__ part_, const T* __restrict__ slabs, const int* __restrict__ blk_rc,
                      int nblocks, int tile, MomLayout ml, double* __restrict__ mom, const bool wt) {
    using vec_t = typename Mfma<T>::vec_t;
    constexpr int VEC = Mfma<T>::VEC;
    constexpr int TP = NT / RED_G, SUB = RED_P / TP;
    static_assert(NT % RED_G == 0 && RED_P % TP == 0 && SUB >= 1, "thread parts divide the order parts");
    auto part = [&](int q, int g, int c) -> double& { return part_[((size_t)q * RED_G + g) * VEC + c]; };
    auto put = [&](double* q, double v) {
        if (wt) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v;
    };
    const int p = ml.p, n = ml.n;
    const int tt = tile * tile;
    const long long ngroups = (long long)nblocks * tt / VEC;
    const int tq = vt / RED_G, gl = vt % RED_G;
    const long long idx = (long long)bid * RED_G + gl;
    const bool on = idx < ngroups;
    const int blk = on ? (int)(idx / (tt / VEC)) : 0, e0 = on ? (int)(idx % (tt / VEC)) * VEC : 0;
    const int* info = blk_rc + blk * 5;
    const T* src = slabs + (size_t)info[2] * tt + e0;
    const size_t stride = (size_t)info[3] * tt;
    const int nslices = info[4];
    const int per = (nslices + RED_P - 1) / RED_P;
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
        const int op = tq * SUB + j;
        double acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = 0.0;
        const int k1 = min(nslices, (op + 1) * per);
        int k = op * per;
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
        for (int c = 0; c < VEC; ++c) part(op, gl, c) = acc[c];
    }
    __syncthreads();
    if (tq != 0 || !on) return;
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
        for (int q = 0; q < RED_P; ++q) s += part(q, gl, c);
        if (gr < p) {                              // both in U (gr >= gc)
            put(Saa + (size_t)gr * p + gc, s);
            put(Saa + (size_t)gc * p + gr, s);
        } else if (gc < p) {                       // gr in G, gc in U
            put(Sab + (size_t)gc * n + (gr - p), s);
        } else {
            put(Sbb + (size_t)(gr - p) * n + (gc - p), s);
            put(Sbb + (size_t)(gc - p) * n + (gr - p), s);
        }
    }
}

// Tail unit `tw` (256 threads, thread `vt`): N, the first moments sum_j (z_ij - s_i) of RED_G of the rows this launch owns
// (RED_S slice parts each) and (second launch) the lagged data-metric sums that ride at the end of the buffer.  `part_`:
// RED_S * RED_G doubles.  Exactly ONE __syncthreads().
__device__ __forceinline__
void gram_reduce_tail(int tw, int vt, double* __restrict__ part_, const int* __restrict__ row_own, int tile, MomLayout ml, long long J,
                      int row_lo, int row_hi, int write_N, const double* __restrict__ rowsum_part,
                      const double* __restrict__ tail_src, double* __restrict__ mom, const bool wt) {
    auto put = [&](double* q, double v) {
        if (wt) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v;
    };
    const int p = ml.p, n = ml.n;
    const int gq = vt / RED_G, gl = vt % RED_G;
    const long long r = row_lo + (long long)tw * RED_G + gl;
    if (tw == 0 && vt == 0) {
        if (write_N) put(mom, (double)J);
        if (tail_src) { put(mom + ml.tail(), tail_src[0]); put(mom + ml.tail() + 1, tail_src[1]); }
    }
    double s = 0.0;
    if (r < row_hi) {
        const int rs0 = row_own[(r / tile) * 2], nslices = row_own[(r / tile) * 2 + 1];
        const int per = (nslices + RED_S - 1) / RED_S;
        const int k1 = min(nslices, (gq + 1) * per);
#pragma unroll 8
        for (int k = gq * per; k < k1; ++k) s += rowsum_part[(size_t)(rs0 + k) * (p + n) + r];
    }
    part_[gq * RED_G + gl] = s;
    __syncthreads();
    if (gq == 0 && r < row_hi) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < RED_S; ++q) t += part_[q * RED_G + gl];
        put(mom + (r < p ? ml.sa() + r : ml.sb() + (r - p)), t);
    }
}

// the launch's arrival word (GateSig): this workgroup's agent-scope stores acknowledged, a ticket, the last one signals
__device__ __forceinline__ void gate_arrive(const GateSig& gate) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(gate.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(gate.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gate.flag, gate.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace cesx
