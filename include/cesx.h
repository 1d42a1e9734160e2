/*
 * cesx.h -- C ABI of the MI355X-native EKS / ALDI ensemble-update engine.
 *
 * This is the drop-in boundary for the hot path of agarbuno/ces: the per-step
 * ensemble update of ces/calibrate.py (sampling.eks_update :418-449,
 * sampling.eks_update_aldi :451-490, sampling.eks_update_aldi_constant
 * :492-529 and sampling.timestep_method :243-267).  The reference is pure
 * Python/numpy and has no FFI of its own; each entry point
 * below cites the reference lines whose arithmetic it replaces.  The Python
 * binding a maintainer would add is ces_amd/engine.py (ctypes); see
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++ or torch types; all pointers are raw addresses
 *   - "dev" pointers are device (HBM) addresses, "host" pointers host memory
 *   - ensembles keep the reference layout: row-major (p, J), particle index
 *     fastest (ces/calibrate.py:56, 277); a particle shard is a column range
 *   - every call returns CESX_OK (0) or a CESX_E* status; the text of the
 *     last failure is available from cesx_last_error()
 *   - one handle per device, not thread-safe per handle, no global state
 *   - the caller owns every buffer it passes; U is never written
 *     (ces/calibrate.py:357 keeps U0 alive in the trace list)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     all work is enqueued asynchronously, only cesx_result() waits
 */
#ifndef CESX_H
#define CESX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CESX_ABI_VERSION 1

/* status codes */
#define CESX_OK            0
#define CESX_EINVAL        1   /* bad argument / unknown rule (ValueError)               */
#define CESX_ENOTPD        2   /* covariance not positive definite (np.linalg.LinAlgError,
                                  ces/calibrate.py:446, :487, :526)                      */
#define CESX_EHIP          3   /* HIP runtime failure                                     */
#define CESX_ESTATE        4   /* call order violated (e.g. no problem set)              */
#define CESX_ERCCL         7   /* RCCL failure (cesx_comm_*, cesx_allreduce_*)            */
#define CESX_ENOCONV       6   /* reserved: rounds 1-4 returned it when the Lanczos iteration of
                                  time_step='spectral' missed its residual criterion; since round 5
                                  lambda_max comes from repeated squaring with a two-sided bound and
                                  cannot fail to converge (what np.linalg.eigvals of :250 raises
                                  in that case stays mapped to np.linalg.LinAlgError)          */
#define CESX_EUNSUPPORTED  5   /* time_step='adaptive': the reference calls the undefined
                                  self.LM_procedure (ces/calibrate.py:255)               */

/* arithmetic type of the O(J) passes (moments, update GEMM).  The small dense
   algebra (covariances, Cholesky, gains, time step) always runs in fp64. */
#define CESX_F32 0
#define CESX_F64 1

/* update rule, kwargs['update'] of sampling.run (ces/calibrate.py:304, :364-369) */
#define CESX_UPDATE_EKS            0
#define CESX_UPDATE_ALDI           1
#define CESX_UPDATE_ALDI_CONSTANT  2

/* kwargs['time_step'] of sampling.timestep_method (ces/calibrate.py:247-260) */
#define CESX_TS_DEFAULT   0   /* None: hk = 1/(||D||_F + 1e-8)              :247-248 */
#define CESX_TS_SPECTRAL  1   /* hk = 1/max Re eig(D)                        :249-251 */
#define CESX_TS_CONSTANT  2   /* hk = delta_t                                :252-253 */
#define CESX_TS_ADAPTIVE  3   /* undefined in the reference -> CESX_EUNSUPPORTED :254 */
#define CESX_TS_MIX       4   /* Frobenius until t >= spinup, then delta_t   :256-260 */

typedef struct cesx_engine* cesx_handle;

/* Static shape of one engine = the state of enka.__init__ (ces/calibrate.py:14-22)
   plus the particle shard this device owns. */
typedef struct cesx_config {
    uint32_t struct_bytes;  /* sizeof(cesx_config), for ABI checking                  */
    int32_t  p;             /* parameter dimension            (self.p)                */
    int32_t  n_obs;         /* data dimension                 (self.n_obs)            */
    int32_t  dtype;         /* CESX_F32 | CESX_F64                                    */
    int32_t  device;        /* HIP device ordinal                                     */
    int64_t  J_local;       /* particles held by this device                          */
    int64_t  J_global;      /* ensemble size over all devices (self.J)                */
    int64_t  j_offset;      /* global index of local particle 0 (keys the noise RNG)  */
    uint64_t seed;          /* Philox key for on-device noise                         */
} cesx_config;

/* Per-step options = the **kwargs of eks_update* (ces/calibrate.py:418, 451, 492)
   plus the two pieces of object state the rules read. */
typedef struct cesx_step_params {
    uint32_t struct_bytes;
    int32_t  update;        /* CESX_UPDATE_*                                          */
    int32_t  time_step;     /* CESX_TS_*                                              */
    int32_t  first_step;    /* len(self.Uall) == 1   (ces/calibrate.py:262, :520)     */
    int32_t  t_len;         /* len(self.metrics['t']) before this step   (:257)       */
    int32_t  reserved;
    double   t_last;        /* self.metrics['t'][-1] before this step (if t_len > 0)  */
    double   delta_t;       /* kwargs['delta_t'], default 1/(T/2)        (:253, :260) */
    double   spinup;        /* kwargs['spinup'],  default 4.0            (:257)       */
    double   switch_mult;   /* kwargs['switch'],  default 1.0            (:517)       */
    uint64_t step_index;    /* noise counter: a fresh value per step                  */
} cesx_step_params;

/* What one step appends to the reference's bookkeeping lists. */
typedef struct cesx_step_result {
    double  hk;              /* step size                                            */
    double  t_new;           /* value appended to metrics['t']          (:262-265)   */
    double  self_bias;       /* metrics['self-bias']                    (:432/:464)  */
    double  self_bias_data;  /* metrics['self-bias-data']               (:434/:466)  */
    double  bias_data;       /* metrics['bias-data']                    (:435/:467)  */
    double  bias;            /* metrics['bias']                         (:433/:465)  */
    double  radspec;         /* value appended to self.radspec (spectral only, :250) */
    /* Sharded ensembles: self_bias_data / bias_data above hold only THIS shard's share
       (its sum divided by J_global; the shares of all devices add up to the metric).
       The two fields below are the complete values of the PREVIOUS step: each shard's
       sums ride at the tail of the next step's all-reduced moment buffer, so no second
       collective is needed.  On one device the values above are already complete. */
    double  lag_bias_data;
    double  lag_self_bias_data;
    int32_t status;          /* CESX_OK, CESX_ENOTPD or CESX_ENOCONV for this step; CESX_EHIP:
                                the side stream's factorisation never signalled (one device:
                                its completion is a polled word, bounded; CESX_POLL_JOIN=0 joins
                                with an event instead)                                */
    int32_t reserved;
} cesx_step_result;

/* ---- lifetime -------------------------------------------------------- */

/* Replaces enka.__init__ (ces/calibrate.py:14-22): allocates all device
   workspace for the given shape.  On failure *out is NULL and
   cesx_last_error(NULL) describes why. */
int  cesx_create(const cesx_config* cfg, cesx_handle* out);
void cesx_destroy(cesx_handle h);
const char* cesx_last_error(cesx_handle h);
int  cesx_abi_version(void);

/* Replaces the externally-set object state self.mu, self.sigma, self.ustar
   (examples/scripts/darcy-flow.py:68-75) and the per-call arguments y_obs,
   Gamma of eks_update* (ces/calibrate.py:418).  All HOST pointers to fp64
   arrays: y (n), Gamma (n x n, SPD), mu (p), Sigma (p x p, SPD), ustar (p).
   Factorises Gamma and Sigma once (the reference re-solves them every step,
   ces/calibrate.py:429, :443, :485).
   A DENSE Gamma = L L^T (the reference's pde examples use a sample covariance) is whitened away: once per step the
   engine forms G~ = L^{-1} G (one triangular n x n x J product on the matrix pipe, into an engine-owned n x J buffer
   allocated here) and works with y~ = L^{-1} y, Gamma~ = I -- D = (1/J) E^T Gamma^{-1} R, the data metrics, the gains
   and K (g - y) are invariant (ces/calibrate.py:429-441, :461-473), and every kernel behind it is the diagonal-Gamma
   one.  G_dev itself is never written.  cesx_debug_dense reports gbar and K in the caller's coordinates.
   There is ONE whitened buffer per handle: with a dense Gamma every cesx_moments* call (it whitens G into that buffer) must be
   stream-ordered behind the previous cesx_apply of the handle (whose update kernel reads it) -- the split API's freedom to run
   step i + 1's moments on another stream beside step i's update holds for a diagonal Gamma only.
   Supported conditioning of a dense Gamma on an fp32 engine: the whitening product runs in the engine dtype, its rounding
   eps32 |L^{-1}| |G| is amplified by cond(L) = sqrt(cond(Gamma)) relative to the whitened signal; tests hold cond(Gamma) <= 1e4 with
   |gbar| / spread <= 1e2 to the fp32 bar (tests/test_gpu_parity.py::test_dense_gamma_conditioning); beyond that use an fp64 engine.
   A call that fails (CESX_ENOTPD, an allocation) leaves the handle WITHOUT a problem: nothing is committed before everything is. */
int cesx_set_problem(cesx_handle h, const double* y, const double* Gamma,
                     const double* mu, const double* Sigma, const double* ustar);

/* ---- one step, single device ----------------------------------------- */

/* Replaces one call of sampling.eks_update / eks_update_aldi /
   eks_update_aldi_constant (ces/calibrate.py:418-529) on device-resident
   arrays: U_dev (p x J_local), G_dev (n x J_local) inputs; xi_dev
   (p x J_local) the injected N(0,1) block standing in for
   np.random.normal(0,1,[p,J]) (:447, :488, :527), or NULL to draw it on
   device (Philox4x32-10 keyed by seed, step_index, global particle index);
   U_next_dev (p x J_local) output, must not alias U_dev.  `recenter` != 0
   recomputes the centring shift from the data (required on the first call and
   whenever U/G are unrelated to the previous call). */
int cesx_step(cesx_handle h, const cesx_step_params* prm, const void* U_dev, const void* G_dev,
              const void* xi_dev, void* U_next_dev, int recenter, void* stream);

/* Waits for the last step on this handle and returns what the reference would
   have appended to self.metrics / self.radspec.  Returns CESX_ENOTPD when the
   ensemble covariance was not positive definite (U_next is then undefined).
   (On one device the last small kernel of a step -- fixed-order sum of the data-metric
   partials, copy of the scalars to the host -- is held back so that it can ride on the next
   step's cesx_moments_uu_chol launch; cesx_result, like every entry point that touches the
   engine's state, enqueues it at once if it is still pending.  A driver may therefore enqueue
   the first half of step i+1 before it reads the result of step i, or not: same numbers.
   Lifetime rule that follows: the held-back kernel is enqueued on the `stream` that was passed to
   cesx_apply / cesx_step, so that stream must stay valid until cesx_result (or any other entry
   point of this handle) has been called; what the held-back kernel needs of the moment buffer is
   copied into engine-owned memory by cesx_apply's own kernels.
   Buffers of a step: U_dev, G_dev, xi_dev and U_next_dev must stay valid (and U, G, xi unchanged)
   until cesx_result has returned for that step -- the step is not complete before, and the one
   recovery path below launches its kernels again.  The moment buffer is read by cesx_apply's own
   kernels and, in that recovery alone, once more by cesx_result; a caller may hand it to the next
   step's cesx_moments* before it has read the result (one buffer, pipelined) -- the engine notices,
   and a step whose recovery would need the overwritten buffer reports CESX_EHIP instead of being
   re-run.  ces_amd/dist.py alternates two buffers.) */
int cesx_result(cesx_handle h, cesx_step_result* out);

/* How the caller's stream joins the engine's side stream (centring + chol(C)), and what happens when that goes wrong.
   One device, ALDI, default time step, diagonal Gamma / Sigma: the join is a word the factorisation stores last and one
   workgroup of the caller's stream polls -- ONLY when `stream` has a strictly lower priority than the side stream (two
   streams of one priority level can share a hardware queue, and a waiter queued in front of what it waits for never
   ends); any other stream is joined with an event.  The poll is bounded in wall time (2 s; CESX_POLL_TIMEOUT_MS).  A poll
   that runs out marks the step: its assembly and update launches write NOTHING (U_next, the centring shift and the
   scalars stay as they were), cesx_result switches the engine to the event join for the rest of its life and re-runs
   the step once with chol(C) in line on the caller's stream, from the step's own moment buffer and U / G / xi (see the
   lifetime rules at cesx_result; a moment buffer that a later cesx_moments* call was given in the meantime: no re-run,
   CESX_EHIP).  It then returns CESX_OK -- or CESX_ESTATE when cesx_moments* calls were enqueued after the failed step
   (a pipelined driver): the re-run step's result and U_next are valid, those moments -- and whatever the driver derived
   from the unwritten U_next before, its forward map G first of all -- must be redone (ces_amd/dist.py re-evaluates the
   forward map into the caller's G tensor, then the moments).
   cesx_debug_poll_recoveries: how many steps of this handle were re-run that way. */
unsigned long long cesx_debug_poll_recoveries(cesx_handle h);

/* The hk-dependent SPD inverses of a step -- (Sigma + hk C)^{-1} of the EKS rule (ces/calibrate.py:443), (hk C_gg + Gamma)^{-1} of the
   recomputed gain (:440-441, :472-473) -- are started from the previous step's inverse (Newton-Schulz sweeps on the matrix pipe)
   when that start is close (||I - A X_prev||_F < 0.3) and accepted only when the TRUE residual of the result passes
   (||I - A X||_F < 1e-10); otherwise, and always on the first step of a problem, they are factored from scratch (Cholesky,
   triangular inverse, product).  Same inverse to ~1e-10 either way; CESX_NS_WARM=0 pins the factorisation.
   cesx_debug_warm_inverse: 1 when the last such inverse of this handle was taken from the warm start, else 0.  Synchronises.
   Reproducibility contract: the NUMBER of sweeps a warm start runs is sized from ||I - A X_prev||_F of the last step whose
   result the host has read (cesx_result; reset by cesx_set_problem), and warm and cold inverses agree to ~1e-10, not to the
   bit: chains of these rules are bit-reproducible across call flows that read results at the same cadence (every step, as
   ces_amd's drivers and the reference's loop do -- ces/calibrate.py:387 tests t each iteration); the ranks of a sharded run
   must read in lockstep (ShardedSampler does).  CESX_NS_WARM=0 removes the dependence. */
int cesx_debug_warm_inverse(cesx_handle h);

/* Which form of the update GEMM (ces/calibrate.py:443-447, :484-488, :515-527) the last cesx_apply / cesx_step of this handle
   launched:  0  the assembled coefficient matrix W = [(1 + hk a) I - hk C Sigma^{-1} | -hk K | sqrt(2 hk) L] (every rule, dtype, shape);
              1  ALDI, default time step, fp32: the same matrix with hk kept out of it (the factorisation stores L into the image);
              2  as 1 for a diagonal Sigma and 224 < p <= 256, through the Cholesky factor: C Sigma^{-1} (U - mu) =
                 L (L^T Sigma^{-1} U) - C Sigma^{-1} mu with C = L L^T as factored (:476-478) -- two triangular products instead of
                 the dense one, the intermediate kept in MFMA accumulator registers (CESX_CHAIN=0 keeps form 1).
   The form is chosen by the problem (cesx_set_problem) and the shape alone, never by the call flow.  -1: no handle. */
int cesx_debug_update_form(cesx_handle h);

/* ---- split entry points (multi-device, testing) ----------------------- */

/* Length in doubles of the packed moment buffer that is summed across devices:
   [ N, sum(u-s_u) (p), S_aa (p x p) | sum(g-s_g) (n), S_ab (p x n), S_bb (n x n),
     lagged sum q_r^2, lagged sum q_e^2 (data-metric sums of the previous step) ].
   The first cesx_moments_uu_len() doubles depend on U alone. */
size_t cesx_moments_len(cesx_handle h);
size_t cesx_moments_uu_len(cesx_handle h);

/* Row sums of this shard: sums_dev[0] = J_local, then sum_j U (p), sum_j G (n)
   (fp64).  After summing over devices pass the result to cesx_set_shift. */
int cesx_colsum(cesx_handle h, const void* U_dev, const void* G_dev, double* sums_dev, void* stream);
int cesx_set_shift(cesx_handle h, const double* sums_dev, void* stream);

/* First half of a step (ces/calibrate.py:423-429 / :459-461 / :501-503 and the
   metric sums of :432-435): shifted first and second moments of this shard in
   fp64 into mom_dev (cesx_moments_len doubles).  Additive over shards. */
int cesx_moments(cesx_handle h, const void* U_dev, const void* G_dev, double* mom_dev, void* stream);

/* The same in two pieces, so that chol(C) can start before the Gram is complete:
   cesx_moments_uu fills the leading cesx_moments_uu_len() doubles (N, sum(u-s_u), S_aa -- all
   that cov(U) of ces/calibrate.py:424/:476/:512 needs; a few entries of the second section
   are written too when p is not a multiple of the MFMA tile); cesx_chol_async, called on the
   (summed) leading part, forms C and starts L = chol(C) (:446/:487/:526) on the engine's side
   stream; cesx_moments_rest fills the remainder while that runs; cesx_apply joins.
   cesx_moments = cesx_moments_uu + cesx_moments_rest.  `update` is a CESX_UPDATE_* value
   (it selects the covariance divisor J vs. J-1, :424 vs. :476). */
int cesx_moments_uu(cesx_handle h, const void* U_dev, const void* G_dev, double* mom_dev, void* stream);
int cesx_chol_async(cesx_handle h, int update, const double* mom_dev, void* stream);
int cesx_moments_rest(cesx_handle h, const void* U_dev, const void* G_dev, double* mom_dev, void* stream);
/* cesx_moments_uu + cesx_chol_async in one call, for an ensemble on ONE device (no all-reduce between the two):
   the hand-over to the side stream is then the U x U reduce kernel's own completion signal instead of a marker
   packet in front of the second Gram launch (~6 us per step at C2).  Same results as the two calls. */
int cesx_moments_uu_chol(cesx_handle h, int update, const void* U_dev, const void* G_dev, double* mom_dev, void* stream);
/* The sharded form of the same hand-over: cesx_moments_uu on `stream`, after which the engine's SIDE stream waits
   for it (again through the reduce kernel's own completion signal).  The driver then issues the all-reduce of the
   buffer's head and cesx_chol_async on cesx_side_stream(), and goes on with cesx_moments_rest on `stream`: the
   U x U launch has the device to itself, the collective and chol(C) run beside the second launch. */
int cesx_moments_uu_handover(cesx_handle h, const void* U_dev, const void* G_dev, double* mom_dev, void* stream);

/* ---- the exchange step of a SHARDED ensemble, for callers without torch.distributed ----
   Rank r of N holds the particle columns [j_offset, j_offset + J_local) (cesx_config); per step the packed fp64 moment
   buffer is summed over the ranks -- the only data that crosses GPUs -- by RCCL all-reduces over xGMI, issued on the
   stream the caller names:
       cesx_comm_unique_id   rank 0 draws the 128-byte id (ncclGetUniqueId) and ships it to the others by any host channel
       cesx_comm_init        every rank, collectively: ncclCommInitRank on the handle's device
       cesx_allreduce_head   sum of the leading cesx_moments_uu_len() doubles (N, sum(u - s), S_aa: all chol(C) needs) --
                             between cesx_moments_uu_handover and cesx_chol_async, on cesx_side_stream()
       cesx_allreduce_tail   sum of the rest, behind cesx_moments_rest on the caller's stream
       cesx_allreduce_whole  the north star's literal single all-reduce of the whole buffer (cesx_moments, then this, then
                             cesx_apply factors C in line)
       cesx_allreduce_sum / _max   `count` doubles in place: the first step's centring sums (cesx_colsum) and
                             aldi_constant's max|drift| (ces/calibrate.py:519)
   All in place on device memory, asynchronous on `stream`.  CESX_ERCCL on an RCCL error (text in cesx_last_error).
   librccl.so is bound at run time (a process that already holds a copy keeps using it).  ces_amd/dist.py calls these
   when its engine has a communicator; torch.distributed remains the path of the CPU (gloo) tests. */
#define CESX_COMM_ID_BYTES 128
int cesx_comm_unique_id(void* id_out);
int cesx_comm_init(cesx_handle h, int nranks, int rank, const void* unique_id);
int cesx_comm_destroy(cesx_handle h);
int cesx_comm_nranks(cesx_handle h);                                  /* 0: no communicator */
/* Communicators this handle holds: cesx_comm_init makes TWO where the library has ncclCommSplit -- collectives issued on the
   engine's side stream (cesx_side_stream: the head, beside the second Gram launch) go through the second one, so that no
   communicator is ever used from two streams (CESX_COMM_SPLIT=0: one, shared).  0 / 1 / 2. */
int cesx_comm_count(cesx_handle h);
int cesx_comm_stats(cesx_handle h, unsigned long long* calls, unsigned long long* doubles);   /* all-reduces issued so far, their payload */
int cesx_allreduce_head(cesx_handle h, double* mom_dev, void* stream);
int cesx_allreduce_tail(cesx_handle h, double* mom_dev, void* stream);
int cesx_allreduce_whole(cesx_handle h, double* mom_dev, void* stream);
int cesx_allreduce_sum(cesx_handle h, double* buf_dev, size_t count, void* stream);
int cesx_allreduce_max(cesx_handle h, double* buf_dev, size_t count, void* stream);

/* cesx_moments_rest for a LINEAR forward map (utils.lineal, ces/utils.py:25-31) installed with
   cesx_forward_set_lineal, without a pass over G: with g_j = A u_j + b every G-dependent moment of
   ces/calibrate.py:459-461 / :472 follows from the U-only head of the buffer (N, sum(u-s_u), S_aa) --
   S_ab = S_aa A^T + sa c^T, S_bb = A S_aa A^T + (A sa) c^T + c (A sa)^T + N c c^T, sum(g-s_g) = A sa + N c with
   c = A s_u + b - s_g -- two n x p x p fp64 products instead of the second Gram launch.  mom_dev must hold the
   complete (summed over devices) head; the rest is written, the lagged data-metric sums included.  Valid when
   the G_dev later passed to cesx_apply is what cesx_forward_apply produced from the same U_dev (K3 and the
   per-particle data metrics still read that G).  Not with a dense Gamma (CESX_EUNSUPPORTED: the engine's moments are
   those of the whitened data; take cesx_moments_rest).  The chained device-resident loops of ces_amd use it
   (e2e only; the benchmark's timed step always runs the full Gram). */
int cesx_moments_rest_lineal(cesx_handle h, double* mom_dev, void* stream);

/* The engine's side stream (a hipStream_t).  A sharded driver issues the all-reduce of the
   leading part of the moment buffer on it (and then calls cesx_chol_async with it as
   `stream`), so that collective and chol(C) both run beside cesx_moments_rest without a third
   stream: HIP multiplexes streams onto a few hardware queues, and two streams that share
   one serialise. */
void* cesx_side_stream(cesx_handle h);

/* Second half: small dense algebra on the (summed) moments -- covariance,
   Cholesky, gain, time step (ces/calibrate.py:243-267, :437-446, :469-487) --
   and the fused drift + diffusion update of the shard (:443-447, :484-488). */
int cesx_apply(cesx_handle h, const cesx_step_params* prm, const double* mom_dev,
               const void* U_dev, const void* G_dev, const void* xi_dev,
               void* U_next_dev, void* stream);

/* eks_update_aldi_constant needs max|drift| over the WHOLE ensemble
   (ces/calibrate.py:519).  cesx_apply_drift writes the drift into U_next_dev
   and this shard's max|drift| into absmax_dev[0]; after a max-reduction over
   devices cesx_apply_finish completes U_next = U + hk*drift + sqrt(2hk) L xi. */
int cesx_apply_drift(cesx_handle h, const cesx_step_params* prm, const double* mom_dev,
                     const void* U_dev, const void* G_dev, void* U_next_dev,
                     double* absmax_dev, void* stream);
int cesx_apply_finish(cesx_handle h, const cesx_step_params* prm, const double* absmax_dev,
                      const void* U_dev, const void* xi_dev, void* U_next_dev, void* stream);

/* Noise block the engine would draw for (step_index, shard): fills xi_dev
   (p x J_local).  For tests of the generator. */
int cesx_draw_noise(cesx_handle h, uint64_t step_index, void* xi_dev, void* stream);

/* Optional: draw the noise block of step `step_index` AHEAD of the update, into an engine-owned
   buffer, on the engine's side stream behind the next chol(C) (the f32-input MFMA shares the
   SIMD's vector lanes, so Philox + Box-Muller inside the update kernel cost it matrix-pipe
   cycles).  The engine keeps two such buffers and draws the block of step_index + 1 as well,
   so that a run with consecutive step indices finds every block complete one step ahead
   (CESX_NOISE_LOOKAHEAD=0: one buffer, the step's own block only).  A later cesx_apply /
   cesx_apply_finish / cesx_step with xi_dev == NULL and a matching prm->step_index reads the
   block (same numbers as the in-kernel generator, np.random.normal's role at
   ces/calibrate.py:447/:488/:527); any other step index falls back to drawing inside the
   update kernel.  cesx_step calls it itself. */
int cesx_prefetch_noise(cesx_handle h, uint64_t step_index, void* stream);

/* ---- forward-map hook --------------------------- */

/* G = A U + b for the linear map utils.lineal (ces/utils.py:25-31) evaluated on
   the whole shard at once instead of enka.G_ens' per-particle loop
   (ces/calibrate.py:123-130).  A_dev (n x p, row-major, engine dtype), b_dev
   (n) or NULL. */
int cesx_forward_lineal(cesx_handle h, const void* A_dev, const void* b_dev,
                        const void* U_dev, void* G_dev, void* stream);

/* The same map in two steps for a driver loop that applies ONE map every iteration
   (sampling.run calls model once per particle per iteration, ces/calibrate.py:351-352):
   cesx_forward_set_lineal copies A (and b) into engine-owned buffers laid out for the LDS-DMA
   update kernels, cesx_forward_apply evaluates G = A U + b with them -- no staging copies per
   iteration, and the GEMM runs in the fast kernel.  The engine keeps its own copy: A_dev / b_dev
   may be freed or changed afterwards (call set again to change the map). */
int cesx_forward_set_lineal(cesx_handle h, const void* A_dev, const void* b_dev, void* stream);
int cesx_forward_apply(cesx_handle h, const void* U_dev, void* G_dev, void* stream);

/* ---- host staging ------------------------------------------------------ */

/* A COLUMN block of a row-major (rows, J) ensemble between pinned host memory and the device, asynchronously on
   `stream` (hipMemcpy2DAsync; pitches and width in bytes, `height` rows, to_device != 0: host -> device).  The
   drop-in class pipelines the reference's host data flow (ces/calibrate.py:341-369: G_ens on the host, arrays into
   and out of every update) over such blocks: particles are independent in G_ens (:123-130), rows are not. */
int cesx_copy_cols_async(cesx_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                      size_t height, int to_device, void* stream);

/* ---- introspection ---------------------------------------------------- */

/* Per-kernel timing with HIP events recorded on the launch stream around the
   two O(J) kernels (which: 0 = Gram/moments kernel K1, 1 = update kernel K3).
   cesx_profile_read synchronises, returns the summed milliseconds and the
   number of launches since the last read, and resets the counters. */
/* on = 2: "gap-only" -- only the stop of the second moments launch and the start of the update launch are bound
   (for cesx_profile_gap); such a step contributes nothing to cesx_profile_read.
   on = 3 / 4: only the update / only the moments launches carry events (a time-stamped dispatch costs the step
   20-30 us: a timed region samples its dominant kernel alone). */
int cesx_profile_enable(cesx_handle h, int on);
int cesx_profile_read(cesx_handle h, int which, double* total_ms, int* launches);
/* Milliseconds from the end of the last profiled moments launch to the start of the last profiled update launch
   (the K2 kernels, the hand-over from the side stream and -- sharded -- the collectives sit in between): call it
   after the profiled step and BEFORE cesx_profile_read, which recycles the events.  -1 when nothing was profiled. */
int cesx_profile_gap(cesx_handle h, double* gap_ms);
/* Shader clock (GHz) the last PROFILED update launch (K3) ran at: one wave of it stamps s_memtime and the 100 MHz
   s_memrealtime at its start and end (profiled launches only; the others execute no stamp).  Synchronises.
   0.0 when no profiled launch has run. */
int cesx_profile_clock(cesx_handle h, double* clock_ghz);
/* What this device sustains on the bare matrix instruction of the engine's dtype (v_mfma_f32_32x32x2_f32 /
   v_mfma_f64_16x16x4_f64, operands in registers, every SIMD busy) for about target_ms milliseconds: TFLOP/s
   and the in-kernel shader clock.  A roofline fraction can then be read against the datasheet peak and against
   this box's own rate.  Measurement support only; synchronises. */
int cesx_calibrate_mfma(cesx_handle h, double target_ms, double* tflops, double* clock_ghz, void* stream);

/* Host-only introspection for tests: the work partition of the moments launch `part` (0: U x U blocks, 1: the
   others) for a handle of this shape and workgroup budget, checked for its invariants (every wanted block of the
   lower triangle exactly once, slices x whole tiles cover J, staged rows fit in LDS, workgroups within the budget).
   Returns the number of violations (0 = sound), < 0 on bad arguments; info[6] (optional) = {types, workgroups,
   blocks, busiest workgroup's tiles x blocks per SIMD, max staged row blocks, slabs}.  Needs no device. */
int cesx_debug_gram_plan(int p, int n_obs, int dtype, int part, int wg_budget, long long J_local, int* info);

/* Copies the engine's current small dense state to HOST buffers (any may be
   NULL): ubar (p), gbar (n), C (p x p), L = chol(C) (p x p), K (p x n),
   M = C Sigma^{-1} (p x p).  Synchronises. */
int cesx_debug_dense(cesx_handle h, double* ubar, double* gbar, double* C, double* L,
                     double* K, double* M);

#ifdef __cplusplus
}
#endif
#endif /* CESX_H */
