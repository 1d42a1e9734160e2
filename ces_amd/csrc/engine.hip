// C ABI of libcesx.so (include/cesx.h): handle lifetime, problem set-up and
// the host-side sequencing of K1 (moments) -> K2 (dense) -> K3 (update).
#include "cesx_internal.h"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sched.h>
#include <stdexcept>
#include <thread>

using namespace cesx;

namespace {

thread_local std::string g_create_err;      // per-thread: cesx_create has no handle to hang the message on

// ---- tiny host-side dense helpers (set-up only: Gamma and Sigma are factorised once) ----
bool host_chol(int n, const double* A, std::vector<double>& L) {
    L.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / ljj;
        }
    }
    return true;
}
void host_tri_inverse(int n, const std::vector<double>& L, std::vector<double>& Li) {
    Li.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; ++k) s -= L[(size_t)i * n + k] * Li[(size_t)k * n + j];
            Li[(size_t)i * n + j] = s / L[(size_t)i * n + i];
        }
}
// Ainv = Li^T Li
void host_spd_inverse(int n, const std::vector<double>& Li, std::vector<double>& Ainv) {
    Ainv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
            Ainv[(size_t)i * n + j] = Ainv[(size_t)j * n + i] = s;
        }
}
bool is_diagonal(int n, const double* A) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j && A[(size_t)i * n + j] != 0.0) return false;
    return true;
}

template <typename P> int dmalloc(Engine& e, P** ptr, size_t bytes) {
    CESX_HIP(hipMalloc(reinterpret_cast<void**>(ptr), bytes ? bytes : 8));
    CESX_HIP(hipMemset(*ptr, 0, bytes ? bytes : 8));
    return CESX_OK;
}
int upload(Engine& e, void* dst, const void* src, size_t bytes) {
    CESX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return CESX_OK;
}
// fp64 host vector -> engine dtype on device
int upload_T(Engine& e, void* dst, const double* src, size_t len) {
    if (e.cfg.dtype == CESX_F64) return upload(e, dst, src, len * 8);
    std::vector<float> tmp(len);
    for (size_t i = 0; i < len; ++i) tmp[i] = (float)src[i];
    return upload(e, dst, tmp.data(), len * 4);
}

#define TRY(x) do { int _rc = (x); if (_rc != CESX_OK) return _rc; } while (0)

// Every entry point runs on the engine's device and gives the calling thread its own device
// back on return (a process may hold engines on several devices; cesx_destroy runs from GC).
struct DeviceGuard {
    int prev = -1, dev;
    hipError_t st = hipSuccess;
    explicit DeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) st = hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
#define SET_DEVICE(e)                                                               \
    DeviceGuard _dg((e).cfg.device);                                                \
    if (_dg.st != hipSuccess) {                                                     \
        (e).err = std::string("hipSetDevice: ") + hipGetErrorString(_dg.st);        \
        return CESX_EHIP;                                                           \
    }

int check_prm(Engine& e, const cesx_step_params* prm) {
    if (!prm || prm->struct_bytes != sizeof(cesx_step_params)) { e.err = "bad cesx_step_params"; return CESX_EINVAL; }
    if (prm->update < 0 || prm->update > 2) { e.err = "unknown update rule"; return CESX_EINVAL; }
    if (prm->update != CESX_UPDATE_ALDI_CONSTANT) {
        if (prm->time_step == CESX_TS_ADAPTIVE) {
            e.err = "time_step='adaptive' needs LM_procedure, which the reference never defines (ces/calibrate.py:255)";
            return CESX_EUNSUPPORTED;
        }
        if (prm->time_step < 0 || prm->time_step > 4) { e.err = "unknown time_step rule"; return CESX_EINVAL; }
    }
    if (!e.problem_set) { e.err = "cesx_set_problem has not been called"; return CESX_ESTATE; }
    return CESX_OK;
}

int finish_step(Engine& e, const cesx_step_params& prm, hipStream_t s) {
    TRY(launch_publish(e, s));
    e.pending = true;
    e.last_prm = prm;
    return CESX_OK;
}

// the noise block of this step drawn ahead by cesx_prefetch_noise (nullptr: draw inside the update kernel)
const void* prefetched_noise(Engine& e, const cesx_step_params& prm, hipStream_t s) {
    // drawn on the side stream behind chol(C); its own event, waited for HERE (right before the update kernel):
    // K2's scalar and assemble kernels do not need the block and run beside the draw
    for (int b = 0; b < 2; ++b) {
        if (!e.d_xi[b] || e.xi_step[b] != (long long)prm.step_index) continue;
        // a block drawn behind an EARLIER chol(C) precedes this step's chol(C) on the side stream: a stream that has
        // waited for this step's ev_b is already ordered behind the draw
        const bool ordered = e.xi_seq[b] < e.evb_waited_seq && s == e.evb_waited_stream;
        if (!ordered && hipStreamWaitEvent(s, e.ev_x[b], 0) != hipSuccess) return nullptr;
        return e.d_xi[b];
    }
    return nullptr;
}

int run_update_main(Engine& e, const cesx_step_params& prm, const void* U, const void* G, const void* xi,
                    void* Unext, hipStream_t s) {
    if (!xi) xi = prefetched_noise(e, prm, s);
    UpdateSrc src[3] = {{U, e.p, 0, 0}, {G, e.n, 0, 0}, {xi, e.p, xi ? 0 : 1, 1}};
    UpdateOpt opt;
    opt.prof = 1;
    opt.wf = e.d_Wf;
    if (e.last_join_polled) { opt.fault = e.d_cholflag + 1; opt.fault_seq = e.chol_seq; }
    if (e.last_hkfree && e.chain) {
        // K3 through the Cholesky factor (kernels_update4.hip): the chained image, xi from memory -- a block that was neither
        // injected nor drawn ahead is drawn here, into an engine buffer, by the kernel that draws the prefetched ones
        if (!xi) {
            if (!e.d_xi_tmp) CESX_HIP(hipMalloc(&e.d_xi_tmp, (size_t)e.p * (size_t)e.J * e.esz));
            TRY(launch_noise(e, prm.step_index, e.d_xi_tmp, s));
            xi = e.d_xi_tmp;
        }
        opt.hkp = &e.d_scal->hk;
        opt.s2p = &e.d_scal->sqrt2hk;
        int rc4 = launch_update4(e, U, G, xi, Unext, true, opt, s);
        e.last_metric_parts = e.last_update_grid_x;
        return rc4;
    }
    if (e.last_hkfree) {
        // the coefficient image without the time step (launch_dense): [L | a I - M + I/hk | -K] against [xi; U; G]
        src[0] = UpdateSrc{xi, e.p, xi ? 0 : 1, 1};
        src[1] = UpdateSrc{U, e.p, 0, 0};
        src[2] = UpdateSrc{G, e.n, 0, 0};
        opt.wf = e.d_Wq;
        opt.metric_seg = 2;
        opt.hkp = &e.d_scal->hk;
        opt.s2p = &e.d_scal->sqrt2hk;
    }
    int rc = launch_update(e, e.p, e.d_W, e.ktot, e.d_bias, src, 3, nullptr, nullptr, 0.0, nullptr, nullptr, 0.0,
                           Unext, nullptr, prm.step_index, true, opt, s);
    e.last_metric_parts = e.last_update_grid_x;
    return rc;
}

// data metrics: K3 accumulated them while the (whitened) G rows streamed by
int finish_metrics(Engine& e, const double* mom, const void* G, bool publish, hipStream_t s) {
    (void)G;
    return launch_metric_final(e, mom, publish, s);
}

}  // namespace
namespace cesx {
void set_global_error(const std::string& msg) { try { g_create_err = msg; } catch (...) {} }

const void* whitened_G(Engine& e, const void* G, hipStream_t s, bool force, int* rc) {
    *rc = CESX_OK;
    if (!e.whiten) return G;
    // (reused only inside the step that whitened it: the same array, the same stream, no cesx_moments* call since)
    if (!force && e.gw_src == G && e.gw_stream == s && e.gw_calls == e.moments_calls) return e.d_Gw;
    // G~ = L_Gamma^{-1} G: one K segment with lower-triangular coefficients (the kernel skips the zero blocks), no bias
    UpdateSrc src[1] = {{G, e.n, 0, 1}};
    UpdateOpt opt;
    opt.wf = e.d_Wwh_f;
    *rc = launch_update(e, e.n, e.d_Wwh, e.kn, nullptr, src, 1, nullptr, nullptr, 0.0, nullptr, nullptr, 0.0, e.d_Gw,
                        nullptr, 0, false, opt, s);
    if (*rc != CESX_OK) return nullptr;
    e.gw_src = G; e.gw_stream = s; e.gw_calls = e.moments_calls;
    return e.d_Gw;
}
}  // namespace cesx
namespace {
// the deferred metric finalisation + publication of the last update (Engine::met_deferred), as a kernel of its own
int flush_metrics(Engine& e) {
    if (!e.met_deferred) return CESX_OK;
    e.met_deferred = false;
    return launch_metric_final(e, nullptr, true, e.met_stream);
}
#define FLUSH(e) TRY(flush_metrics(e))
// dense Gamma: from here on G is the engine's whitened copy of the caller's array (whitened_G)
#define WHITEN(e, G, stream, force) do { int _wrc; G = whitened_G(e, G, (hipStream_t)(stream), force, &_wrc); if (_wrc != CESX_OK) return _wrc; } while (0)
// (the U x U launch reads G rows only when p is not a multiple of the MFMA tile: whiten there only then)
#define WHITEN_UU(e, G, stream) do { if ((e).whiten && (e).p % gram_tile((e).cfg.dtype) != 0) WHITEN(e, G, stream, true); } while (0)

}  // namespace

extern "C" {

int cesx_abi_version(void) { return CESX_ABI_VERSION; }

const char* cesx_last_error(cesx_handle h) {
    if (!h) return g_create_err.c_str();
    return reinterpret_cast<Engine*>(h)->err.c_str();
}

// The side stream carries the one-workgroup Cholesky beside the second Gram launch.  It is a
// HIGH-PRIORITY stream: HIP multiplexes the streams of a priority level onto a few hardware
// queues (4 by default), and once a communicator library has created its own streams the side
// stream would share a queue with the caller's stream -- two streams on one queue run one after
// the other.  Priority levels have their own queues.
static hipError_t create_side_stream(Engine& e) {
    int lo = 0, hi = 0;
    const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo;
    if (prio && hipStreamCreateWithPriority(&e.side, hipStreamNonBlocking, hi) == hipSuccess) {
        e.side_prio = hi; e.side_has_prio = true;
        return hipSuccess;
    }
    return hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking);
}

static int create_impl(const cesx_config* cfg, cesx_handle* out) {
    if (out) *out = nullptr;
    if (!cfg || !out || cfg->struct_bytes != sizeof(cesx_config)) { g_create_err = "bad cesx_config"; return CESX_EINVAL; }
    if (cfg->p < 1 || cfg->n_obs < 1 || cfg->J_local < 1 || cfg->J_global < cfg->J_local || cfg->j_offset < 0 ||
        (cfg->dtype != CESX_F32 && cfg->dtype != CESX_F64)) {
        g_create_err = "cesx_create: invalid shape or dtype";
        return CESX_EINVAL;
    }
    if (cfg->p > 16384 || cfg->n_obs > 16384) {     // K2 is dense in (p + n)^2 and indexes it with 32 bits
        g_create_err = "cesx_create: p and n_obs are limited to 16384";
        return CESX_EINVAL;
    }
    Engine* ep = new (std::nothrow) Engine();
    if (!ep) { g_create_err = "out of host memory"; return CESX_EINVAL; }
    Engine& e = *ep;
    e.cfg = *cfg;
    e.p = cfg->p; e.n = cfg->n_obs; e.P = e.p + e.n;
    e.J = cfg->J_local; e.Jg = cfg->J_global;
    e.esz = cfg->dtype == CESX_F32 ? 4 : 8;
    if (const char* ov = std::getenv("CESX_OVERLAP")) e.overlap_chol = ov[0] != '0';
    if (const char* uv = std::getenv("CESX_UPDATE_V1")) e.update_v2 = uv[0] == '0';
    if (const char* gv = std::getenv("CESX_GRAM_V1")) e.gram_v2 = gv[0] == '0';
    if (const char* fv = std::getenv("CESX_FUSE_CENTER")) { e.fuse_center_ok = fv[0] != '0'; e.fuse_center_auto = false; }
    if (const char* pv = std::getenv("CESX_POLL_JOIN")) e.poll_join_ok = pv[0] != '0';
    if (const char* hv = std::getenv("CESX_HKFREE")) e.hkfree_ok = hv[0] != '0';
    if (const char* cv = std::getenv("CESX_CHAIN")) e.chain_ok = cv[0] != '0';
    if (const char* sv = std::getenv("CESX_UPDATE_SMALL")) e.update_small = sv[0] != '0';
    if (const char* dv = std::getenv("CESX_TEST_DROP_CHOL_SIGNAL")) e.test_drop_signal_at = (unsigned long long)std::max(0, std::atoi(dv));
    if (const char* tv = std::getenv("CESX_POLL_TIMEOUT_MS")) e.poll_ticks = (unsigned long long)std::max(1, std::atoi(tv)) * 100000ull;
    auto fail = [&](int rc) { g_create_err = e.err; cesx_destroy(reinterpret_cast<cesx_handle>(ep)); return rc; };
    int rc;
    DeviceGuard dg(cfg->device);
    if (dg.st != hipSuccess) { e.err = std::string("hipSetDevice: ") + hipGetErrorString(dg.st); return fail(CESX_EHIP); }
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) e.num_cus = ncu;
    }
    const int p = e.p, n = e.n, P = e.P, mx = p > n ? p : n;
    const size_t pp = (size_t)p * p, pn = (size_t)p * n, nn = (size_t)n * n;
    const size_t mm = (size_t)potrf_ld(mx) * potrf_ld(mx);     // temporaries also hold padded Cholesky factors

    // gram plans: part 0 = blocks among the first ceil(p / tile) block rows (U x U: all that
    // chol(C) needs), part 1 = the rest.  Part 1 is sized to leave a few CUs free, because the
    // single-workgroup Cholesky runs beside it on the side stream.
    e.ml = MomLayout{p, n};
    {
        const int tile = gram_tile(cfg->dtype), kt = gram_kt(cfg->dtype);
        const int pbU = (p + tile - 1) / tile;
        const long long ntiles = (e.J + kt - 1) / kt;
        for (int part = 0; part < 2; ++part) {
            GramPart& gp = e.gp[part];
            const int min_types = 1;
            // part 1: 7 workgroups per shader engine (8 CUs), so that the Cholesky always finds a free CU
            // Part 1 runs beside what the side stream carries, and a kernel on another stream is only placed
            // while it needs no more whole CUs than are free (tools/place_probe.hip: beside 248 one-per-CU
            // workgroups a kernel of <= 8 workgroups starts at once, one of 16 waits for the whole launch).
            //   * one device, p <= 256: the side stream runs the U-only centring (8 workgroups), the one-workgroup
            //     chol(C) and then the noise block -> 1 CU in 32 stays free (248 workgroups on MI355X);
            //   * sharded ensembles (RCCL's all-reduce kernels run there too) and p > 256 (blocked Cholesky: TRSM /
            //     GEMM launches of tens of workgroups) -> 1 CU in 8 stays free (224).
            const bool slim_side = e.J == e.Jg && potrf_ld(p) <= 256;
            e.center_u_wgs = slim_side ? 8 : 256;       // (16 x 1024 threads do NOT get placed on the 8 free CUs: measured)
            const int budget = part == 0 ? e.num_cus : e.num_cus - (slim_side ? e.num_cus / 32 : e.num_cus / 8);
            gp.plan = make_gram_plan(P, tile, gram_nbw(cfg->dtype), gram_max_stage_rows(), part + 1, pbU, min_types,
                                     budget, ntiles);
            // fp32, second launch, two types by capacity: three lighter ones.  The planner's cost model prices the tiles, not
            // the partial slabs a launch leaves behind -- written in a burst behind the last tile and read back by the reduce
            // (~22 us of the step for 51 MB at C2).  Three types (32 / 48 / 20 blocks over 82 / 114 / 52 slices) leave 37 MB
            // at 25 % more row traffic and a 4 % worse balance: 0.3944 -> 0.3924 ms/step (tools/ab_env.py, round 4); four
            // types 0.3990, two types for the U x U launch 0.4015 against 0.3956.
            if (part == 1 && cfg->dtype == CESX_F32 && gp.plan.ntypes == 2)
                gp.plan = make_gram_plan(P, tile, gram_nbw(cfg->dtype), gram_max_stage_rows(), part + 1, pbU, 3, budget, ntiles);
            if (gp.plan.max_rb * tile > gram_max_stage_rows()) { e.err = "gram plan exceeds LDS"; return fail(CESX_EINVAL); }
        }
    }
    {
        // MFMA cycles of the second Gram launch's busiest SIMD, roughly: blocks x tiles x MFMAs per block and tile x 64 / SIMDs
        const GramPlan& pb = e.gp[1].plan;
        const int kt = gram_kt(cfg->dtype);
        const double cyc = (double)pb.nblocks * (double)((e.J + kt - 1) / kt) * (cfg->dtype == CESX_F32 ? 16.0 : 4.0) * 64.0 /
                           (4.0 * std::max(1, e.num_cus));
        e.gram_b_short = cyc < 60e-6 * 2.3e9;
    }
    e.colsum_slices = (int)std::min<long long>(16, (e.J + 1023) / 1024);
    if (e.colsum_slices < 1) e.colsum_slices = 1;
    e.kp = (p + 15) / 16 * 16; e.kn = (n + 15) / 16 * 16; e.ktot = 2 * e.kp + e.kn;
    e.rpad = (mx + 255) / 256 * 256;
    e.mom_len = e.ml.len();                      // incl. lagged {sum q_r^2, sum q_e^2} of the previous apply

#define DM(ptr, bytes) if ((rc = dmalloc(e, &ptr, (bytes)))) return fail(rc)
    DM(e.d_y, n * 8); DM(e.d_mu, p * 8); DM(e.d_ustar, p * 8);
    DM(e.d_Gamma, nn * 8); DM(e.d_gw, n * 8); DM(e.d_Wh, nn * 8);
    DM(e.d_Sigma, pp * 8); DM(e.d_Sinv, pp * 8); DM(e.d_sw, p * 8);
    DM(e.d_shift64, P * 8);
    {
        char* t;
        DM(t, P * e.esz); e.d_shiftT = t;
        DM(t, n * e.esz); e.d_yT = t;
        DM(t, n * e.esz); e.d_gwT = t;
        DM(t, (size_t)e.kn * 4 * e.esz); e.d_rowc = t;
        for (int part = 0; part < 2; ++part) {
            const GramPlan& pl = e.gp[part].plan;
            DM(t, (size_t)std::max(pl.total_slabs, 1) * pl.tile * pl.tile * e.esz); e.gp[part].d_slabs = t;
        }
        DM(t, (size_t)e.rpad * e.ktot * e.esz); e.d_W = t;
        DM(t, (size_t)e.rpad * e.ktot * e.esz); e.d_Wf = t;
        DM(t, (size_t)e.rpad * e.esz); e.d_bias = t;
        if (cfg->dtype == CESX_F32) {          // (the chained layout of kernels_update4.hip: 18 + kn / 16 tiles of 16 KiB)
            DM(t, std::max((size_t)e.rpad * e.ktot * 4, (size_t)(18 + e.kn / 16) * 16384)); e.d_Wq = t;
        }
        DM(t, (size_t)e.rpad * e.kp * e.esz); e.d_Wfwd = t;
        DM(t, (size_t)e.rpad * e.kp * e.esz); e.d_Wfwd_f = t;
        DM(t, (size_t)e.rpad * e.esz); e.d_bfwd = t;
    }
    for (int part = 0; part < 2; ++part) {
        GramPart& gp = e.gp[part];
        const GramPlan& pl = gp.plan;
        DM(gp.d_type_hdr, pl.type_hdr.size() * 4); DM(gp.d_rows, pl.rows.size() * 4);
        DM(gp.d_wblk, pl.wblk.size() * 4); DM(gp.d_blk_rc, pl.blk_rc.size() * 4); DM(gp.d_row_own, pl.row_own.size() * 4);
        if ((rc = upload(e, gp.d_type_hdr, pl.type_hdr.data(), pl.type_hdr.size() * 4))) return fail(rc);
        if ((rc = upload(e, gp.d_rows, pl.rows.data(), pl.rows.size() * 4))) return fail(rc);
        if ((rc = upload(e, gp.d_wblk, pl.wblk.data(), pl.wblk.size() * 4))) return fail(rc);
        if ((rc = upload(e, gp.d_blk_rc, pl.blk_rc.data(), pl.blk_rc.size() * 4))) return fail(rc);
        if ((rc = upload(e, gp.d_row_own, pl.row_own.data(), pl.row_own.size() * 4))) return fail(rc);
        DM(gp.d_rowsum_part, (size_t)pl.total_rs * P * 8);
    }
    DM(e.d_metric_part, ((size_t)((e.J + 31) / 32) + 8) * 2 * 8);
    DM(e.d_metric_sums, 2 * 8);
    DM(e.d_colsum_part, (size_t)P * e.colsum_slices * 8);
    DM(e.d_mom, e.mom_len * 8); DM(e.d_sums, (1 + P) * 8); DM(e.d_sums_w, (1 + P) * 8);
    DM(e.d_ubar, p * 8); DM(e.d_gbar, n * 8); DM(e.d_m, n * 8); DM(e.d_dg, n * 8);
    DM(e.d_C, pp * 8); DM(e.d_L, (size_t)potrf_ld(p) * potrf_ld(p) * 8); DM(e.d_Cug, pn * 8); DM(e.d_See, nn * 8); DM(e.d_Srr, nn * 8);
    DM(e.d_K, pn * 8); DM(e.d_Kp, pn * 8); DM(e.d_M, pp * 8); DM(e.d_P, pp * 8); DM(e.d_PK, pn * 8);
    if (potrf_ld(mx) > 256) DM(e.d_Lwork, (size_t)potrf_ld(mx) * potrf_ld(mx) * 8);
    DM(e.d_t1, mm * 8); DM(e.d_t2, mm * 8); DM(e.d_t3, mm * 8); DM(e.d_t4, mm * 8);
    if (potrf_ld(mx) <= 256) {          // warm-started SPD inverses (kernels_dense.hip, spd_inverse); zeroed: X_prev = 0 is a cold start
        if (const char* nv = std::getenv("CESX_NS_WARM")) e.ns_ok = nv[0] != '0';
        const size_t nb16 = ((size_t)mx + 15) / 16;
        for (int k = 0; k < 2; ++k) { DM(e.d_ns_x[0][k], nn * 8); DM(e.d_ns_x[1][k], pp * 8); }
        for (int k = 0; k < 3; ++k) DM(e.d_ns_r[k], (size_t)mx * mx * 8);
        DM(e.d_ns_parts, 2 * nb16 * nb16 * 8);
        { char* t; DM(t, 64); e.d_ns_skip = reinterpret_cast<int*>(t); }
    }
    {   // spectral rule (kernels_dense.hip, spec_square_kernel): {log accumulator, weight, flag, pad} + 2 x per-workgroup partial sums
        const size_t nb16 = ((size_t)n + 15) / 16;
        DM(e.d_spec, (4 + 2 * nb16 * nb16) * 8);
    }
    DM(e.d_mv, (size_t)6 * mx * 8); DM(e.d_part, 256 * 4 * 8);
    DM(e.d_scal, sizeof(Scalars)); DM(e.d_absmax, 8);
    DM(e.d_absmax_part, (size_t)update_grid_blocks(e, p) * 8);
    DM(e.d_clk, 4 * 8);
    DM(e.d_cholflag, 128);
    DM(e.d_ticket, 64);
    DM(e.d_lag, 3 * 8);
    DM(e.d_A64, (size_t)n * p * 8); DM(e.d_b64, n * 8); DM(e.d_lvec, 2 * n * 8);
#undef DM
    if (hipHostMalloc(reinterpret_cast<void**>(&e.h_scal), sizeof(Scalars), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&e.h_scal_dev), e.h_scal, 0) != hipSuccess ||
        hipEventCreateWithFlags(&e.ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e.ev_b, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e.ev_x[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e.ev_x[1], hipEventDisableTiming) != hipSuccess ||
        create_side_stream(e) != hipSuccess) {
        e.err = "pinned host buffer / event creation failed";
        return fail(CESX_EHIP);
    }
    std::memset(e.h_scal, 0, sizeof(Scalars));
    *out = reinterpret_cast<cesx_handle>(ep);
    return CESX_OK;
}

int cesx_create(const cesx_config* cfg, cesx_handle* out) {
    // no C++ exception may cross the C boundary (std::vector growth in the Gram plan, std::string)
    try {
        return create_impl(cfg, out);
    } catch (const std::exception& ex) {
        try { g_create_err = std::string("cesx_create: ") + ex.what(); } catch (...) {}
    } catch (...) {
        try { g_create_err = "cesx_create: unknown C++ exception"; } catch (...) {}
    }
    if (out) *out = nullptr;      // (a half-built engine is leaked rather than destroyed twice)
    return CESX_EINVAL;
}

void cesx_destroy(cesx_handle h) {
    if (!h) return;
    Engine& e = *reinterpret_cast<Engine*>(h);
    DeviceGuard dg(e.cfg.device);
    void* ptrs[] = {e.d_y, e.d_mu, e.d_ustar, e.d_Gamma, e.d_gw, e.d_Wh, e.d_Sigma, e.d_Sinv, e.d_sw,
                    e.d_shift64, e.d_shiftT, e.d_yT, e.d_gwT, e.d_W, e.d_Wf, e.d_Lwork, e.d_Wwh, e.d_Wwh_f, e.d_Gw, e.d_sums_w,
                    e.d_bias, e.d_Wfwd, e.d_Wfwd_f, e.d_bfwd, e.d_metric_part, e.d_metric_sums,
                    e.d_rowc,
                    e.d_colsum_part, e.d_mom,
                    e.gp[0].d_type_hdr, e.gp[0].d_rows, e.gp[0].d_wblk, e.gp[0].d_blk_rc, e.gp[0].d_row_own, e.gp[0].d_slabs, e.gp[0].d_rowsum_part,
                    e.gp[1].d_type_hdr, e.gp[1].d_rows, e.gp[1].d_wblk, e.gp[1].d_blk_rc, e.gp[1].d_row_own, e.gp[1].d_slabs, e.gp[1].d_rowsum_part, e.d_sums, e.d_ubar, e.d_gbar, e.d_m, e.d_dg,
                    e.d_C, e.d_L, e.d_Cug, e.d_See, e.d_Srr, e.d_K, e.d_Kp, e.d_M, e.d_P, e.d_PK,
                    e.d_t1, e.d_t2, e.d_t3, e.d_t4, e.d_spec, e.d_ns_x[0][0], e.d_ns_x[0][1], e.d_ns_x[1][0], e.d_ns_x[1][1], e.d_ns_r[0], e.d_ns_r[1], e.d_ns_r[2],
                    e.d_ns_parts, e.d_ns_skip, e.d_mv, e.d_part, e.d_scal, e.d_absmax,
                    e.d_absmax_part, e.d_clk, e.d_cholflag, e.d_lag, e.d_A64, e.d_b64, e.d_lvec, e.d_Wq, e.d_ticket};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    for (int w = 0; w < 2; ++w)
        for (auto& pr : e.prof_ev[w]) { if (pr.first) (void)hipEventDestroy(pr.first); if (pr.second) (void)hipEventDestroy(pr.second); }
    for (auto ev : e.prof_pool) (void)hipEventDestroy(ev);
    if (e.h_scal) (void)hipHostFree(e.h_scal);
    if (e.ev_a) (void)hipEventDestroy(e.ev_a);
    if (e.ev_b) (void)hipEventDestroy(e.ev_b);
    for (hipEvent_t ev : {e.ev_x[0], e.ev_x[1]})
        if (ev) (void)hipEventDestroy(ev);
    if (e.side) (void)hipStreamDestroy(e.side);
    if (e.comm) (void)cesx_comm_destroy(h);
    for (void* q : e.d_xi)
        if (q) (void)hipFree(q);
    if (e.d_xi_tmp) (void)hipFree(e.d_xi_tmp);
    delete &e;
}

int cesx_set_problem(cesx_handle h, const double* y, const double* Gamma, const double* mu,
                     const double* Sigma, const double* ustar) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!y || !Gamma || !mu || !Sigma || !ustar) { e.err = "cesx_set_problem: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    const int p = e.p, n = e.n;
    std::vector<double> L, Li, inv;
    if (!host_chol(n, Gamma, L)) { e.err = "Gamma is not symmetric positive definite"; return CESX_ENOTPD; }
    host_tri_inverse(n, L, Li);
    host_spd_inverse(n, Li, inv);
    // Dense Gamma: the engine works in whitened data coordinates (cesx_internal.h, Engine::whiten) -- y~ = L^{-1} y,
    // Gamma~ = I, and G~ = L^{-1} G formed once per step by the update kernel's own code (whitened_G below)
    // (nothing of the engine's state is committed before every allocation and upload below has succeeded: a failure leaves
    //  the handle WITHOUT a problem -- cesx_moments* / cesx_apply then return CESX_ESTATE -- instead of half of the new one)
    const bool whiten = !is_diagonal(n, Gamma);
    e.problem_set = false;
    e.whiten = false;
    e.gw_src = nullptr;
    std::vector<double> gw(n), yi(y, y + n), Gi(Gamma, Gamma + (size_t)n * n);
    for (int i = 0; i < n; ++i) gw[i] = 1.0 / Gamma[(size_t)i * n + i];
    if (whiten) {
        e.h_LG = L; e.h_Li = Li;
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            for (int k = 0; k <= i; ++k) t += Li[(size_t)i * n + k] * y[k];
            yi[i] = t;
            gw[i] = 1.0;
            for (int k = 0; k < n; ++k) Gi[(size_t)i * n + k] = i == k ? 1.0 : 0.0;
        }
        // L^{-1} (lower triangular), zero padded to the update kernels' [rpad][kn] layout, row-major and fragment-major
        const size_t len = (size_t)e.rpad * e.kn;
        std::vector<double> rm(len, 0.0), fm(len, 0.0);
        const int nkt = e.kn / 16;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k <= i; ++k) {
                const double v = Li[(size_t)i * n + k];
                rm[(size_t)i * e.kn + k] = v;
                fm[e.cfg.dtype == CESX_F32 ? wf_index(i, k, nkt) : wd_index(i, k, nkt)] = v;
            }
        // (each buffer checked on its own: a second call finds what an earlier, failed one did allocate)
        if (!e.d_Wwh) { char* t; int rc; if ((rc = dmalloc(e, &t, len * e.esz))) return rc; e.d_Wwh = t; }
        if (!e.d_Wwh_f) { char* t; int rc; if ((rc = dmalloc(e, &t, len * e.esz))) return rc; e.d_Wwh_f = t; }
        if (!e.d_Gw) CESX_HIP(hipMalloc(&e.d_Gw, (size_t)n * (size_t)e.J * e.esz));
        TRY(upload_T(e, e.d_Wwh, rm.data(), len)); TRY(upload_T(e, e.d_Wwh_f, fm.data(), len));
    }
    TRY(upload(e, e.d_y, yi.data(), n * 8)); TRY(upload(e, e.d_Gamma, Gi.data(), (size_t)n * n * 8));
    TRY(upload(e, e.d_gw, gw.data(), n * 8));
    TRY(upload(e, e.d_Wh, Li.data(), (size_t)n * n * 8));
    TRY(upload_T(e, e.d_yT, yi.data(), n)); TRY(upload_T(e, e.d_gwT, gw.data(), n));
    if (!host_chol(p, Sigma, L)) { e.err = "Sigma is not symmetric positive definite"; return CESX_ENOTPD; }
    host_tri_inverse(p, L, Li);
    host_spd_inverse(p, Li, inv);
    e.diag_sigma = is_diagonal(p, Sigma);
    std::vector<double> sw(p);
    for (int i = 0; i < p; ++i) sw[i] = 1.0 / Sigma[(size_t)i * p + i];
    TRY(upload(e, e.d_mu, mu, p * 8)); TRY(upload(e, e.d_ustar, ustar, p * 8));
    TRY(upload(e, e.d_Sigma, Sigma, (size_t)p * p * 8)); TRY(upload(e, e.d_Sinv, inv.data(), (size_t)p * p * 8));
    TRY(upload(e, e.d_sw, sw.data(), p * 8));
    {
        // K3 through the Cholesky factor where the problem and the shape allow (cesx_internal.h, Engine::chain).  The two layouts
        // of d_Wq share no writer's footprint: a change of layout starts from a zeroed image, with nothing of the engine in flight
        const bool chain = e.chain_ok && e.hkfree_ok && e.diag_sigma && update4_shape_ok(e);
        if (chain != e.chain && e.d_Wq) {
            CESX_HIP(hipDeviceSynchronize());
            CESX_HIP(hipMemset(e.d_Wq, 0, std::max((size_t)e.rpad * e.ktot * 4, (size_t)(18 + e.kn / 16) * 16384)));
        }
        e.chain = chain;
    }
    // a new problem: the warm starts of K2's SPD inverses (kernels_dense.hip, spd_inverse) start cold
    if (e.d_ns_x[0][0]) {
        for (int k = 0; k < 2; ++k) { CESX_HIP(hipMemset(e.d_ns_x[0][k], 0, (size_t)n * n * 8)); CESX_HIP(hipMemset(e.d_ns_x[1][k], 0, (size_t)p * p * 8)); }
    }
    e.whiten = whiten;
    e.ns_r0_last = 1e300;          // (the first warm start of the new problem is sized like a cold one)
    e.problem_set = true;
    e.shift_valid = false;
    return CESX_OK;
}

size_t cesx_moments_len(cesx_handle h) { return h ? reinterpret_cast<Engine*>(h)->mom_len : 0; }
size_t cesx_moments_uu_len(cesx_handle h) { return h ? reinterpret_cast<Engine*>(h)->ml.uu_len() : 0; }

void* cesx_side_stream(cesx_handle h) { return h ? (void*)reinterpret_cast<Engine*>(h)->side : nullptr; }

int cesx_colsum(cesx_handle h, const void* U, const void* G, double* sums, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!U || !G || !sums) { e.err = "cesx_colsum: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    return launch_colsum(e, U, G, sums, (hipStream_t)stream);
}

int cesx_set_shift(cesx_handle h, const double* sums, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!sums) { e.err = "cesx_set_shift: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    return launch_set_shift(e, sums, (hipStream_t)stream);
}

static int moments_check(Engine& e, const void* U, const void* G, double* mom) {
    ++e.moments_calls;
    // A step whose polled join runs out is re-run by cesx_result FROM ITS OWN moment buffer (include/cesx.h, lifetime
    // rules).  A caller that hands the same buffer to the next step's moments before it has read that result gives the
    // buffer up: the re-run is then not attempted (cesx_result reports CESX_EHIP for such a step instead of computing it
    // from the next step's moments).
    if (e.last_apply.valid && e.pending && mom != nullptr && mom == e.last_apply.mom) e.last_apply.mom_reused = true;
    if (!U || !G || !mom) { e.err = "cesx_moments: null pointer"; return CESX_EINVAL; }
    if (!e.problem_set) { e.err = "cesx_set_problem has not been called"; return CESX_ESTATE; }
    if (!e.shift_valid) { e.err = "no centring shift: call cesx_colsum + cesx_set_shift (or cesx_step with recenter) first"; return CESX_ESTATE; }
    return CESX_OK;
}

int cesx_moments_uu(cesx_handle h, const void* U, const void* G, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(moments_check(e, U, G, mom));
    SET_DEVICE(e);
    FLUSH(e);
    ++e.prof_step;
    WHITEN_UU(e, G, stream);
    return launch_gram(e, 0, U, G, mom, (hipStream_t)stream);
}

int cesx_chol_async(cesx_handle h, int update, const double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!mom || update < 0 || update > 2) { e.err = "cesx_chol_async: bad argument"; return CESX_EINVAL; }
    if (!e.problem_set) { e.err = "cesx_set_problem has not been called"; return CESX_ESTATE; }
    SET_DEVICE(e);
    FLUSH(e);
    return launch_chol_async(e, update, mom, (hipStream_t)stream);
}

// U x U launch + its reduce with the hand-over event (ev_a) bound to the reduce kernel's own completion signal -- no
// marker packet in front of whatever the caller's stream runs next -- and, when the previous update's metric
// finalisation is still pending on this stream, that too as the reduce launch's first workgroup.  The side stream
// is made to wait for ev_a.
static int moments_uu_handover(Engine& e, const void* U, const void* G, double* mom, hipStream_t s) {
    ++e.prof_step;
    if (e.met_deferred && e.met_stream != s) FLUSH(e);
    WHITEN_UU(e, G, s);
    TRY(launch_gram(e, 0, U, G, mom, s, true));
    if (e.met_deferred) {
        MetricFin f = metric_fin_args(e, nullptr, true);
        f.N = (double)e.Jg;
        e.met_deferred = false;
        TRY(launch_gram_reduce(e, 0, mom, s, e.ev_a, &f));
    } else {
        TRY(launch_gram_reduce(e, 0, mom, s, e.ev_a, nullptr));
    }
    CESX_HIP(hipStreamWaitEvent(e.side, e.ev_a, 0));
    return CESX_OK;
}

int cesx_moments_uu_handover(cesx_handle h, const void* U, const void* G, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(moments_check(e, U, G, mom));
    SET_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (s == e.side) {      // nothing to hand over: the caller's own ordering applies
        FLUSH(e);
        ++e.prof_step;
        WHITEN_UU(e, G, s);
        TRY(launch_gram(e, 0, U, G, mom, s));
        if (s != e.side) {
            CESX_HIP(hipEventRecord(e.ev_a, s));
            CESX_HIP(hipStreamWaitEvent(e.side, e.ev_a, 0));
        }
        return CESX_OK;
    }
    return moments_uu_handover(e, U, G, mom, s);
}

int cesx_moments_uu_chol(cesx_handle h, int update, const void* U, const void* G, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(moments_check(e, U, G, mom));
    if (update < 0 || update > 2) { e.err = "cesx_moments_uu_chol: bad argument"; return CESX_EINVAL; }
    SET_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (s == e.side) {
        FLUSH(e);
        ++e.prof_step;
        WHITEN_UU(e, G, s);
        TRY(launch_gram(e, 0, U, G, mom, s));
        return launch_chol_async(e, update, mom, s);
    }
    // nothing can sit between the reduce of the U x U launch and the hand-over to the side stream
    TRY(moments_uu_handover(e, U, G, mom, s));
    return launch_chol_async(e, update, mom, s, true);
}

int cesx_moments_rest(cesx_handle h, const void* U, const void* G, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(moments_check(e, U, G, mom));
    SET_DEVICE(e);
    FLUSH(e);
    // (the reduce kernel of this launch also copies this shard's data-metric sums of the PREVIOUS
    //  apply to the tail of the buffer: they ride on this step's all-reduce)
    WHITEN(e, G, stream, true);
    return launch_gram(e, 1, U, G, mom, (hipStream_t)stream);
}

int cesx_moments_rest_lineal(cesx_handle h, double* mom, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!mom) { e.err = "cesx_moments_rest_lineal: null pointer"; return CESX_EINVAL; }
    if (!e.problem_set) { e.err = "cesx_set_problem has not been called"; return CESX_ESTATE; }
    if (!e.shift_valid) { e.err = "no centring shift: call cesx_colsum + cesx_set_shift (or cesx_step with recenter) first"; return CESX_ESTATE; }
    if (!e.fwd_set) { e.err = "cesx_moments_rest_lineal: cesx_forward_set_lineal has not been called"; return CESX_ESTATE; }
    if (e.whiten) { e.err = "cesx_moments_rest_lineal: dense Gamma (the engine works on whitened data: take cesx_moments_rest)"; return CESX_EUNSUPPORTED; }
    SET_DEVICE(e);
    FLUSH(e);
    return launch_moments_lineal(e, mom, (hipStream_t)stream);
}

int cesx_moments(cesx_handle h, const void* U, const void* G, double* mom, void* stream) {
    int rc = cesx_moments_uu(h, U, G, mom, stream);
    return rc ? rc : cesx_moments_rest(h, U, G, mom, stream);
}

int cesx_apply_drift(cesx_handle h, const cesx_step_params* prm, const double* mom, const void* U,
                     const void* G, void* Unext, double* absmax, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(check_prm(e, prm));
    if (!mom || !U || !G || !Unext || !absmax) { e.err = "cesx_apply_drift: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    hipStream_t s = (hipStream_t)stream;
    WHITEN(e, G, s, false);
    TRY(launch_dense(e, *prm, mom, 1, s));
    UpdateSrc src[2] = {{U, e.p, 0, 0}, {G, e.n, 0, 0}};
    UpdateOpt opt;
    opt.wf = e.d_Wf;
    TRY(launch_update(e, e.p, e.d_W, e.kp + e.kn, e.d_bias, src, 2, nullptr, nullptr, 0.0, nullptr, nullptr, 0.0,
                      Unext, e.d_absmax_part, prm->step_index, true, opt, s));
    e.last_metric_parts = e.last_update_grid_x;
    const int nparts = e.last_update_grid;
    TRY(finish_metrics(e, mom, G, false, s));
    return launch_absmax_final(e, nparts, absmax, s);
}

int cesx_apply_finish(cesx_handle h, const cesx_step_params* prm, const double* absmax, const void* U,
                      const void* xi, void* Unext, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(check_prm(e, prm));
    if (!absmax || !U || !Unext) { e.err = "cesx_apply_finish: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    hipStream_t s = (hipStream_t)stream;
    if (absmax != e.d_absmax) CESX_HIP(hipMemcpyAsync(e.d_absmax, absmax, 8, hipMemcpyDeviceToDevice, s));
    TRY(launch_dense(e, *prm, nullptr, 2, s));
    // U_next = sqrt(2hk) L xi + 1 * U + hk * drift   (drift currently lives in U_next)
    if (!xi) xi = prefetched_noise(e, *prm, s);
    UpdateSrc src[1] = {{xi, e.p, xi ? 0 : 1, 1}};
    UpdateOpt opt;
    opt.wf = e.d_Wf;
    TRY(launch_update(e, e.p, e.d_W, e.kp, nullptr, src, 1, U, nullptr, 1.0, Unext, &e.d_scal->hk, 1.0, Unext,
                      nullptr, prm->step_index, false, opt, s));
    return finish_step(e, *prm, s);
}

int cesx_apply(cesx_handle h, const cesx_step_params* prm, const double* mom, const void* U, const void* G,
               const void* xi, void* Unext, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(check_prm(e, prm));
    if (!mom || !U || !G || !Unext) { e.err = "cesx_apply: null pointer"; return CESX_EINVAL; }
    if (U == Unext) { e.err = "U_next must not alias U (ces/calibrate.py:357 keeps U0 in the trace)"; return CESX_EINVAL; }
    if (prm->update == CESX_UPDATE_ALDI_CONSTANT) {
        TRY(cesx_apply_drift(h, prm, mom, U, G, Unext, e.d_absmax, stream));
        return cesx_apply_finish(h, prm, e.d_absmax, U, xi, Unext, stream);
    }
    SET_DEVICE(e);
    FLUSH(e);
    hipStream_t s = (hipStream_t)stream;
    e.last_apply = Engine::LastApply{true, *prm, mom, U, G, xi, Unext, s, e.moments_calls};     // (the caller's G: a re-run whitens it again if need be)
    WHITEN(e, G, s, false);
    // (whether the update launch qualifies for the LDS-DMA kernel is known here: the hk-free K2 has no other consumer)
    TRY(launch_dense(e, *prm, mom, 0, s, update2_qualifies(e, U, G, xi, Unext)));
    TRY(run_update_main(e, *prm, U, G, xi, Unext, s));
    // (Moving this last small kernel to the side stream was tried: the event record + wait pair costs
    //  as much GPU idle time as the 7 us kernel itself.)
    if (e.overlap_chol) {
        // the finalisation rides on the next step's U x U reduce launch (cesx_moments_uu_chol / _handover on this
        // stream) -- no one-workgroup kernel (7 us) between this update and the next Gram launch; anything else flushes it
        e.met_deferred = true; e.met_stream = s;      // (reads the engine's own d_lag, not `mom`)
    } else {
        TRY(finish_metrics(e, mom, G, true, s));     // also publishes the step result to the host
    }
    e.pending = true;
    e.last_prm = *prm;
    return CESX_OK;
}

int cesx_step(cesx_handle h, const cesx_step_params* prm, const void* U, const void* G, const void* xi,
              void* Unext, int recenter, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    TRY(check_prm(e, prm));
    if (!U || !G || !Unext) { e.err = "cesx_step: null pointer"; return CESX_EINVAL; }
    if (e.J != e.Jg) { e.err = "cesx_step is single-device: use cesx_moments / all-reduce / cesx_apply for a sharded ensemble"; return CESX_ESTATE; }
    if (U == Unext) { e.err = "U_next must not alias U (ces/calibrate.py:357 keeps U0 in the trace)"; return CESX_EINVAL; }
    if (recenter || !e.shift_valid) {
        TRY(cesx_colsum(h, U, G, e.d_sums, stream));
        TRY(cesx_set_shift(h, e.d_sums, stream));
    }
    if (!xi && e.overlap_chol) TRY(cesx_prefetch_noise(h, prm->step_index, stream));
    // U x U moments -> chol(C) on the side stream, beside the rest of the Gram -> apply.  (Putting the
    // U x U launch itself on the side stream too was measured slower: see ces_amd/dist.py.)
    if (e.overlap_chol) TRY(cesx_moments_uu_chol(h, prm->update, U, G, e.d_mom, stream));
    else TRY(cesx_moments_uu(h, U, G, e.d_mom, stream));
    TRY(cesx_moments_rest(h, U, G, e.d_mom, stream));
    return cesx_apply(h, prm, e.d_mom, U, G, xi, Unext, stream);
}

int cesx_result(cesx_handle h, cesx_step_result* out) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!out) { e.err = "cesx_result: null pointer"; return CESX_EINVAL; }
    if (!e.pending) { e.err = "cesx_result: no step has been enqueued"; return CESX_ESTATE; }
    if (e.met_deferred) {
        SET_DEVICE(e);
        FLUSH(e);
    }
    // Wait for the sequence number the GPU writes last into pinned memory.  A step is a few
    // hundred microseconds, so the first 100 us are a pause-spin (lowest latency); after that the
    // thread yields its core, and from 2 ms on it sleeps in 100 us slices -- a long wait (large
    // shards, a stalled collective) does not burn a host core.
    {
        volatile unsigned long long* seq = &e.h_scal->seq;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != e.seq) {
            __builtin_ia32_pause();
            if ((++spins & 0x3ff) != 0) continue;
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited < std::chrono::microseconds(100)) continue;
            if (waited < std::chrono::milliseconds(2)) { sched_yield(); continue; }
            if (hipPeekAtLastError() != hipSuccess) { e.err = "HIP error while waiting for the step"; return CESX_EHIP; }
            if (waited > std::chrono::seconds(120)) { e.err = "timed out waiting for the step result"; return CESX_EHIP; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    const Scalars& sc = *e.h_scal;
    out->hk = sc.hk; out->t_new = sc.t_new;
    out->self_bias = sc.self_bias; out->self_bias_data = sc.self_bias_data;
    out->bias_data = sc.bias_data; out->bias = sc.bias;
    out->radspec = sc.radspec; out->status = sc.status; out->reserved = 0;
    out->lag_bias_data = sc.spare[1]; out->lag_self_bias_data = sc.spare[2];
    e.ns_r0_last = sc.spare[3] > 0.0 ? sc.spare[3] : 1e300;      // (kernels_dense.hip, spd_inverse: sizes the next warm start's sweeps)
    if (sc.status == CESX_ENOTPD) {
        e.err = "ensemble covariance is not positive definite (Cholesky failed)";
        return CESX_ENOTPD;
    }
    if (sc.status == CESX_EHIP) {
        // The polled join ran out (kernels_dense.hip): the step wrote NOTHING (no W, no centring shift, U_next untouched).
        // From here on this engine joins its side stream with the event, and the step is re-run once with chol(C)
        // in line on the caller's stream (correct whatever the side stream is doing).
        e.poll_join_ok = false;
        if (e.last_apply.valid && !e.in_retry && e.last_join_polled && !e.last_apply.mom_reused) {
            const Engine::LastApply la = e.last_apply;
            e.in_retry = true;
            // whatever a pipelined driver put on the side stream behind the failed step (the centring + chol(C) of moments of
            // an unwritten ensemble: it may well report "not positive definite") ends BEFORE the re-run resets the status word
            {
                SET_DEVICE(e);
                CESX_HIP(hipStreamSynchronize(e.side));
            }
            e.chol_inflight = false;
            int rc = cesx_apply(h, &la.prm, la.mom, la.U, la.G, la.xi, la.Unext, (void*)la.s);
            if (rc == CESX_OK) rc = cesx_result(h, out);
            e.in_retry = false;
            if (rc != CESX_OK) return rc;
            ++e.poll_recoveries;
            if (e.moments_calls != la.moments_calls) {
                e.err = "the polled join of the side stream timed out; the step was re-run and its result is valid, but moments "
                        "enqueued after it were taken of an ensemble that had not been written yet: redo them";
                return CESX_ESTATE;
            }
            return CESX_OK;
        }
        e.err = e.last_apply.valid && e.last_apply.mom_reused
            ? "the side stream's factorisation never signalled its completion (the polled join timed out), and the step's moment "
              "buffer had already been handed to a later cesx_moments* call: not re-run (include/cesx.h, lifetime rules)"
            : "the side stream's factorisation never signalled its completion (the polled join timed out)";
        return CESX_EHIP;
    }
    return CESX_OK;
}

unsigned long long cesx_debug_poll_recoveries(cesx_handle h) { return h ? reinterpret_cast<Engine*>(h)->poll_recoveries : 0; }

int cesx_debug_update_form(cesx_handle h) {
    if (!h) return -1;
    const Engine& e = *reinterpret_cast<Engine*>(h);
    return !e.last_hkfree ? 0 : e.chain ? 2 : 1;
}

int cesx_debug_warm_inverse(cesx_handle h) {
    if (!h) return -1;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!e.d_ns_skip) return 0;
    DeviceGuard dg(e.cfg.device);
    int v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, e.d_ns_skip, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

int cesx_draw_noise(cesx_handle h, uint64_t step_index, void* xi, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!xi) { e.err = "cesx_draw_noise: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    return launch_noise(e, step_index, xi, (hipStream_t)stream);
}

int cesx_prefetch_noise(cesx_handle h, uint64_t step_index, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    SET_DEVICE(e);
    (void)stream;
    if (std::getenv("CESX_NO_NOISE_PREFETCH")) return CESX_OK;
    e.xi_lookahead = !(std::getenv("CESX_NOISE_LOOKAHEAD") && std::atoi(std::getenv("CESX_NOISE_LOOKAHEAD")) == 0);
    for (int b = 0; b < (e.xi_lookahead ? 2 : 1); ++b)
        if (!e.d_xi[b]) CESX_HIP(hipMalloc(&e.d_xi[b], (size_t)e.p * (size_t)e.J * e.esz));
    // The draw itself is enqueued by cesx_chol_async on the side stream, behind chol(C): no extra
    // cross-stream event (each costs ~6 us of GPU idle time), and it runs while the caller's stream
    // is in the tail of the second Gram launch and the latency-bound start of K2.
    e.xi_want = (long long)step_index;
    return CESX_OK;
}

int cesx_forward_lineal(cesx_handle h, const void* A, const void* b, const void* U, void* G, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!A || !U || !G) { e.err = "cesx_forward_lineal: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    // stage A (n x p) into the zero-padded (rpad x kp) layout the update kernel reads
    CESX_HIP(hipMemsetAsync(e.d_Wfwd, 0, (size_t)e.rpad * e.kp * e.esz, s));
    CESX_HIP(hipMemcpy2DAsync(e.d_Wfwd, (size_t)e.kp * e.esz, A, (size_t)e.p * e.esz, (size_t)e.p * e.esz, e.n,
                              hipMemcpyDeviceToDevice, s));
    UpdateSrc src[1] = {{U, e.p, 0, 0}};
    UpdateOpt opt;
    return launch_update(e, e.n, e.d_Wfwd, e.kp, b, src, 1, nullptr, nullptr, 0.0, nullptr, nullptr, 0.0, G,
                         nullptr, 0, false, opt, s);
}

int cesx_forward_set_lineal(cesx_handle h, const void* A, const void* b, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!A) { e.err = "cesx_forward_set_lineal: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    TRY(launch_stage_forward(e, A, b, (hipStream_t)stream));
    e.fwd_set = true;
    e.fwd_has_b = b != nullptr;
    return CESX_OK;
}

int cesx_forward_apply(cesx_handle h, const void* U, void* G, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!U || !G) { e.err = "cesx_forward_apply: null pointer"; return CESX_EINVAL; }
    if (!e.fwd_set) { e.err = "cesx_forward_apply: cesx_forward_set_lineal has not been called"; return CESX_ESTATE; }
    SET_DEVICE(e);
    UpdateSrc src[1] = {{U, e.p, 0, 0}};
    UpdateOpt opt;
    opt.wf = e.d_Wfwd_f;
    return launch_update(e, e.n, e.d_Wfwd, e.kp, e.fwd_has_b ? e.d_bfwd : nullptr, src, 1, nullptr, nullptr, 0.0, nullptr,
                         nullptr, 0.0, G, nullptr, 0, false, opt, (hipStream_t)stream);
}

int cesx_profile_enable(cesx_handle h, int on) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (on < 0 || on > 4) { e.err = "cesx_profile_enable: bad mode"; return CESX_EINVAL; }
    e.profile = on != 0;
    e.profile_gap_only = on == 2;
    e.profile_only = on == 3 ? 1 : on == 4 ? 0 : -1;      // 3: the update launches alone, 4: the moments launches alone
    if (e.profile) {
        SET_DEVICE(e);
        while (e.prof_pool.size() < 512) {        // created up front: no event creation in a timed region
            hipEvent_t ev = nullptr;
            CESX_HIP(hipEventCreate(&ev));
            e.prof_pool.push_back(ev);
        }
    }
    return CESX_OK;
}

int cesx_profile_read(cesx_handle h, int which, double* total_ms, int* launches) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (which < 0 || which > 1 || !total_ms || !launches) { e.err = "cesx_profile_read: bad argument"; return CESX_EINVAL; }
    SET_DEVICE(e);
    double tot = 0.0;
    int cnt = 0;
    for (auto& pr : e.prof_ev[which]) {
        float ms = 0.f;
        if (pr.first && pr.second) {               // (a pair of a gap-only step has one event only: recycled, not counted)
            CESX_HIP(hipEventSynchronize(pr.second));
            CESX_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            tot += ms;
            ++cnt;
        }
        if (pr.first) e.prof_pool.push_back(pr.first);
        if (pr.second) e.prof_pool.push_back(pr.second);
    }
    e.prof_ev[which].clear();
    e.prof_tag[which].clear();
    *total_ms = tot;
    *launches = cnt;
    return CESX_OK;
}

int cesx_copy_cols_async(cesx_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                      size_t height, int to_device, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!dst || !src || width_bytes == 0 || height == 0 || dpitch < width_bytes || spitch < width_bytes) {
        e.err = "cesx_copy_cols_async: bad argument";
        return CESX_EINVAL;
    }
    SET_DEVICE(e);
    CESX_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height,
                              to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    return CESX_OK;
}

int cesx_profile_gap(cesx_handle h, double* gap_ms) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!gap_ms) { e.err = "cesx_profile_gap: null pointer"; return CESX_EINVAL; }
    *gap_ms = -1.0;
    if (e.prof_ev[0].empty() || e.prof_ev[1].empty()) return CESX_OK;
    SET_DEVICE(e);
    hipEvent_t gram_end = e.prof_ev[0].back().second, upd_start = e.prof_ev[1].back().first;
    if (!gram_end || !upd_start) return CESX_OK;
    // both events must belong to ONE step: a step without a second moments launch (the linear-map fast path, a shard
    // whose second part is empty) leaves an older Gram event at the back -- no gap then (-1)
    if (e.prof_tag[0].empty() || e.prof_tag[1].empty() || e.prof_tag[0].back() != e.prof_tag[1].back()) return CESX_OK;
    CESX_HIP(hipEventSynchronize(upd_start));
    float ms = 0.f;
    CESX_HIP(hipEventElapsedTime(&ms, gram_end, upd_start));
    *gap_ms = ms;
    return CESX_OK;
}

int cesx_profile_clock(cesx_handle h, double* clock_ghz) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!clock_ghz) { e.err = "cesx_profile_clock: null pointer"; return CESX_EINVAL; }
    SET_DEVICE(e);
    CESX_HIP(hipDeviceSynchronize());
    long long t[4] = {0, 0, 0, 0};         // {s_memtime, s_memrealtime} at the start, then at the end, of one wave
    CESX_HIP(hipMemcpy(t, e.d_clk, 32, hipMemcpyDeviceToHost));
    *clock_ghz = t[3] > t[1] ? (double)(t[2] - t[0]) / (double)(t[3] - t[1]) * 0.1 : 0.0;      // s_memrealtime ticks at 100 MHz
    return CESX_OK;
}

int cesx_calibrate_mfma(cesx_handle h, double target_ms, double* tflops, double* clock_ghz, void* stream) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    if (!(target_ms > 0.0) || target_ms > 1000.0 || !tflops) { e.err = "cesx_calibrate_mfma: bad argument"; return CESX_EINVAL; }
    SET_DEVICE(e);
    FLUSH(e);
    return launch_calibrate(e, target_ms, tflops, clock_ghz, (hipStream_t)stream);
}

// Host-only: builds the Gram work partition a handle of this shape would use (no device needed) and checks its
// invariants.  Returns the number of violations; info[0..5] = {types, workgroups, blocks, busiest workgroup's
// tiles x blocks-per-SIMD, max staged row blocks, slabs}.
int cesx_debug_gram_plan(int p, int n_obs, int dtype, int part, int wg_budget, long long J_local, int* info) {
    if (p < 1 || n_obs < 1 || (dtype != CESX_F32 && dtype != CESX_F64) || part < 0 || part > 1 || J_local < 1) return -1;
    try {
        const int P = p + n_obs, tile = gram_tile(dtype), kt = gram_kt(dtype), nbw = gram_nbw(dtype);
        const int pbU = (p + tile - 1) / tile;
        const long long ntiles = (J_local + kt - 1) / kt;
        const GramPlan pl = make_gram_plan(P, tile, nbw, gram_max_stage_rows(), part + 1, pbU, 1, wg_budget, ntiles);
        const int nbr = (P + tile - 1) / tile;
        int bad = 0;
        std::vector<int> seen((size_t)nbr * nbr, 0);
        long long wgs = 0, slabs = 0, worst = 0;
        if ((int)pl.type_hdr.size() != pl.ntypes * 8) ++bad;
        for (int t = 0; t < pl.ntypes && !bad; ++t) {
            const int* h = &pl.type_hdr[(size_t)t * 8];
            const int nrb = h[0], rows_off = h[1], blocks_off = h[2], nblk = h[3], wg0 = h[4], nsl = h[5];
            if (nrb < 1 || nrb * tile > gram_max_stage_rows() || nrb > pl.max_rb) ++bad;
            if (nsl < 1 || (long long)nsl > std::max<long long>(1, ntiles)) ++bad;
            if (wg0 != wgs || h[6] != slabs) ++bad;
            wgs += nsl; slabs += (long long)nsl * nblk;
            const long long tps = (ntiles + nsl - 1) / nsl;
            if (tps * nsl < ntiles) ++bad;
            int per_simd[4] = {0, 0, 0, 0}, cnt = 0;
            for (int w = 0; w < 16; ++w)
                for (int b = 0; b < nbw; ++b) {
                    const int* e3 = &pl.wblk[((size_t)blocks_off + (size_t)w * nbw + b) * 3];
                    if (e3[0] < 0) continue;
                    if (e3[0] >= nrb || e3[1] >= nrb || e3[2] < 0 || e3[2] >= nblk) { ++bad; continue; }
                    const int R = pl.rows[rows_off + e3[0]] & 0xffff, C = pl.rows[rows_off + e3[1]] & 0xffff;
                    if (R >= nbr || C > R) { ++bad; continue; }
                    ++seen[(size_t)R * nbr + C];
                    ++per_simd[w & 3]; ++cnt;
                }
            if (cnt != nblk) ++bad;
            const int mx = std::max(std::max(per_simd[0], per_simd[1]), std::max(per_simd[2], per_simd[3]));
            worst = std::max(worst, tps * mx);
        }
        int nwant = 0;
        for (int R = 0; R < nbr; ++R)
            for (int C = 0; C <= R; ++C) {
                const bool uu = R < pbU && C < pbU;
                const bool wanted = part == 0 ? uu : !uu;
                nwant += wanted ? 1 : 0;
                if (seen[(size_t)R * nbr + C] != (wanted ? 1 : 0)) ++bad;      // every wanted block exactly once, no other
            }
        if (nwant != pl.nblocks) ++bad;
        if (wgs != pl.total_wgs || slabs != pl.total_slabs) ++bad;
        if (pl.nblocks > 0 && pl.total_wgs > std::max(wg_budget, pl.ntypes)) ++bad;
        if (info) { info[0] = pl.ntypes; info[1] = pl.total_wgs; info[2] = pl.nblocks; info[3] = (int)worst; info[4] = pl.max_rb; info[5] = pl.total_slabs; }
        return bad;
    } catch (...) {
        return -2;
    }
}

int cesx_debug_dense(cesx_handle h, double* ubar, double* gbar, double* C, double* L, double* K, double* M) {
    if (!h) return CESX_EINVAL;
    Engine& e = *reinterpret_cast<Engine*>(h);
    SET_DEVICE(e);
    FLUSH(e);
    CESX_HIP(hipDeviceSynchronize());
    if (L && e.L_stale) {          // K3 through the Cholesky factor: the step kept L in its coefficient image only; factor C again
        TRY(refresh_factor(e, nullptr));
        CESX_HIP(hipDeviceSynchronize());
    }
    const size_t p = e.p, n = e.n;
    if (ubar) CESX_HIP(hipMemcpy(ubar, e.d_ubar, p * 8, hipMemcpyDeviceToHost));
    if (gbar) {
        CESX_HIP(hipMemcpy(gbar, e.d_gbar, n * 8, hipMemcpyDeviceToHost));
        if (e.whiten) {          // the engine holds the mean of the WHITENED data: gbar = L_Gamma gbar~
            std::vector<double> t(gbar, gbar + n);
            for (size_t i = 0; i < n; ++i) {
                double a = 0.0;
                for (size_t k = 0; k <= i; ++k) a += e.h_LG[i * n + k] * t[k];
                gbar[i] = a;
            }
        }
    }
    if (C) CESX_HIP(hipMemcpy(C, e.d_C, p * p * 8, hipMemcpyDeviceToHost));
    if (L) {
        const size_t ld = potrf_ld(e.p);
        CESX_HIP(hipMemcpy2D(L, p * 8, e.d_L, ld * 8, p * 8, p, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < p; ++i)
            for (size_t j = i + 1; j < p; ++j) L[i * p + j] = 0.0;     // the factor's upper triangle is not stored
    }
    if (K) {
        CESX_HIP(hipMemcpy(K, e.d_K, p * n * 8, hipMemcpyDeviceToHost));
        if (e.whiten) {          // K~ = C_ug~ = C_ug L^{-T}; the reference's gain C_ug Gamma^{-1} = K~ L^{-1}
            std::vector<double> t(K, K + p * n);
            for (size_t i = 0; i < p; ++i)
                for (size_t j = 0; j < n; ++j) {
                    double a = 0.0;
                    for (size_t k = j; k < n; ++k) a += t[i * n + k] * e.h_Li[k * n + j];
                    K[i * n + j] = a;
                }
        }
    }
    if (M) CESX_HIP(hipMemcpy(M, e.d_M, p * p * 8, hipMemcpyDeviceToHost));
    return CESX_OK;
}

}  // extern "C"
