/*
 * kpilqr.h -- C ABI of libkpilqr.so: the MI355X (gfx950) engine for the numerical hot path of
 * keypoint-interpolated iLQR.  It is the drop-in boundary for the reference DMackRus/TrajOptKP:
 * a C++ `Optimiser` subclass (see INTEGRATION.md, and trajoptkp_amd/host/ for the one shipped
 * here) forwards STEP 1b/1c/2/3 of iLQR::Iteration (src/Optimiser/iLQR.cpp:412-531) to these
 * entry points; MuJoCo and the ModelTranslator stay on the host.
 *
 * The reference has no FFI of its own (its plugin surface is three C++ classes wired with
 * shared_ptr in src/main.cpp:39-47,114-147); each entry point below names the reference
 * function it replaces.  Paths are relative to the reference repository root.
 *
 * Conventions
 *  - plain C, no torch / Eigen / HIP types in any signature; `stream` is a hipStream_t passed as
 *    void* (NULL = the library creates its own non-blocking stream).
 *  - all floating point data is FP64.  Host-side matrices use the reference's Eigen layout:
 *    COLUMN-MAJOR per matrix, one matrix per time-step, time-major then batch-major:
 *    A[b][t] at A + ((b*T + t)*n*n), element (r,c) at r + c*n.   n = 2*dof, m = num_ctrl.
 *  - device buffers are owned by the context (internal layout: DESIGN.md section 3).
 *  - every call returns int: 0 = ok, <0 = error (kpilqr_strerror), >0 = numerical status.
 *    Calls that launch kernels are asynchronous on the context's stream; host output buffers
 *    are valid after kpilqr_sync().  One context per (GPU, host thread); not re-entrant.
 *  - the library FAILS LOUDLY (negative code) when no HIP device is present: there is no CPU
 *    fallback inside it.
 */
#ifndef KPILQR_H
#define KPILQR_H

#include <stddef.h>

/* ---- Diagnostic environment switches ------------------------------------------------------------
 * NOT part of the ABI: a caller never needs them, the defaults are what the library has measured to be fastest, and every
 * form gives the same results (to the tolerances of the tests).  They exist so that tests can reach a kernel form the batch
 * size would not select, and for same-box A/B timing; kpilqr_last_launch reports what actually ran.  All are read ONCE, in
 * kpilqr_create (changing the environment afterwards has no effect on an existing context); unset or empty = default.
 *
 *   KPILQR_FUSED_WAVES      backward sweep of a FUSED context: 1 one wavefront per trajectory | 5 consumer / helper pair.
 *                           Default: 5 while 2 x batch <= #SIMDs, else 1.  (Any other value = default.)
 *   KPILQR_FUSED_FWD_WAVES  forward sweep of a FUSED context: 1 one wave | 2 state / cost+staging pair | 3 state / cost /
 *                           staging triple | 4 state / cost pair for uniform key-point sets with 3 or 1 behind it for
 *                           per-DoF lists.  Default: 4 while 2 x batch <= #SIMDs, else 1.
 *   KPILQR_FWD_RAGGED_PAIR  1: per-DoF lists at 256 < batch <= 512 take form 2 instead of one wave behind form 4.
 *   KPILQR_FUSED_RAW        0: a key-point ordered payload is differenced by k_fd_kp_difference in front of the backward
 *                           sweep instead of inside it (default: inside, for uniform key-point sets).
 *   KPILQR_FUSED_UNI        0: the general (per-DoF list) forms of the one-wave sweeps also for uniform key-point sets.
 *   KPILQR_ROLE_SHIFT       wave pairs: block-index bit from which the two roles swap wave slots (default 9; 0 = every
 *                           other block).  Placement probe; no effect on results.
 *   KPILQR_TILED_UW         tiled backward sweep (n + 2 > 16): 1 u-wave form | 0 column-wave form (default by tile count).
 *   KPILQR_TILED_A6         tiled sweeps: 1 / 0 cost derivatives (a6) inside the sweep (default: inside at four tiles from
 *                           ~100 trajectories).
 *   KPILQR_TILED_FSC        two-tile forward sweep: 1 / 0 state / cost wave groups (default: on while 2 NT B <= #SIMDs).
 *   KPILQR_TILED_NT_MIN     run the tiled kernels with at least this many tiles (test coverage of a tile count on a
 *                           small state).
 *   KPILQR_PIPE_COPY        kpilqr_iterate_streamed: bit 0 uploads / bit 1 downloads by copy kernels instead of SDMA
 *                           (default 2).
 * The host-side thread count of the FD pool is a constructor argument of the host classes, not an environment switch. */

#ifdef __cplusplus
extern "C" {
#endif

#define KPILQR_VERSION 410   /* 0.4.1: same entry points and structs as 400; callers should check kpilqr_version() / 100 == 4 */

typedef struct kpilqr_ctx kpilqr_ctx;

typedef struct {
    int dof;       /* position DoFs of the optimiser's state vector (stateVectorList::dof)      */
    int m;         /* num_ctrl                                                                  */
    int T;         /* horizon_length                                                            */
    int nr;        /* residual_list.size()                                                      */
    int batch;     /* independent trajectories resident in this context                         */
    int n_alpha;   /* num_parallel_rollouts, include/Optimiser/Optimiser.h:259 (6)              */
    int device;    /* HIP device ordinal                                                        */
    int flags;     /* KPILQR_FLAG_*                                                             */
} kpilqr_dims;

#define KPILQR_FLAG_GENERIC_KERNELS 1   /* force the dimension-generic LDS kernels (no MFMA path) */
#define KPILQR_FLAG_TILED_KERNELS   2   /* prefer the LDS-tiled MFMA kernels even when one tile would do */
#define KPILQR_FLAG_FUSED           4   /* n+2 <= 16 only: kpilqr_backward / kpilqr_forward_linear / kpilqr_iterate
                                           evaluate the interpolation (a4) and the cost derivatives (a6) inside the
                                           sweeps, from the differenced key-point columns (a compact column store; with
                                           the key-point ordered payload of kpilqr_upload_fd_kp and one wavefront per
                                           trajectory the backward sweep even does the differencing itself) and the
                                           uploaded residuals; A, B, l_* are then NOT materialised and the context holds
                                           NO step records (kpilqr_interpolate / kpilqr_cost_derivs / get_AB allocate and
                                           fill them on request; the set_AB / set_cost_derivs hooks do not feed the fused
                                           sweeps).  Needs canonical key-points: per DoF strictly increasing, first 0, last T-1.
                                           Faster at every batch size (Panda, T=3000: 170 vs 142 iterations/s for one
                                           trajectory, 102k vs 68k at batch 1024).  On a tiled shape (n+2 > 16) the
                                           library may form only the cost derivatives inside the sweeps (variant
                                           "mfma_f64_tiled_a6": four-tile states from ~100 trajectories up); A and B
                                           are then still materialised and kpilqr_iterate skips kpilqr_cost_derivs. */

enum {
    KPILQR_OK = 0,
    KPILQR_ERR_ARG = -1,       /* bad argument / size mismatch (reference: setters return false) */
    KPILQR_ERR_NO_DEVICE = -2, /* no HIP device / runtime failure at create                      */
    KPILQR_ERR_HIP = -3,       /* a HIP call failed; see kpilqr_strerror                          */
    KPILQR_ERR_ALLOC = -4,
    KPILQR_ERR_STATE = -5      /* call order violated (e.g. interpolate before set_keypoints)     */
};

/* which device buffer kpilqr_device_ptr returns */
enum {
    KPILQR_BUF_STEP_RECORDS = 0, /* [batch][T][rec] : A|B|l_xx|l_x|l_uu|l_u per step; a FUSED
                                    context has none until this (or a materialising call) asks for them */
    KPILQR_BUF_K = 1,            /* [batch][T][n][m]  (column-major m x n, as Eigen)             */
    KPILQR_BUF_k = 2,            /* [batch][T][m]                                                */
    KPILQR_BUF_RESIDUALS = 3,    /* [batch][T+1][nr]                                             */
    KPILQR_BUF_R_X = 4,          /* [batch][T+1][nr][n]                                          */
    KPILQR_BUF_R_U = 5,          /* [batch][T+1][nr][m]                                          */
    KPILQR_BUF_U_NOM = 6,        /* [batch][T][m]                                                */
    KPILQR_BUF_FD_XPLUS = 7,     /* [jobs][n]                                                    */
    KPILQR_BUF_FD_XMINUS = 8,    /* [jobs][n]                                                    */
    KPILQR_BUF_COST_PRED = 9,    /* [batch][n_alpha]                                             */
    KPILQR_BUF_DELTA_J = 10,     /* [batch]                                                      */
    KPILQR_BUF_STATUS = 11       /* [batch] int32                                                */
};

/* ---- lifetime --------------------------------------------------------------------------
 * Replaces iLQR::iLQR / iLQR::Resize allocation of A,B,l_*,K,k (src/Optimiser/iLQR.cpp:4-200). */
int  kpilqr_create(const kpilqr_dims *dims, void *stream, kpilqr_ctx **out);
void kpilqr_destroy(kpilqr_ctx *ctx);
int  kpilqr_version(void);
const char *kpilqr_strerror(kpilqr_ctx *ctx);       /* last error text of this context (or global) */
int  kpilqr_get_dims(kpilqr_ctx *ctx, kpilqr_dims *out);
/* iLQR_SVR::Resize (src/Optimiser/iLQR_SVR.cpp:38-193): the state-vector reduction changes dof (and possibly num_ctrl,
 * horizon) between optimisations.  Re-sizes the context IN PLACE: device allocations are kept and only grown when the new
 * sizes do not fit; kernel families are re-selected; everything uploaded before is forgotten.  Synchronous. */
int  kpilqr_resize(kpilqr_ctx *ctx, int new_dof, int new_num_ctrl, int new_horizon);

/* Pinned host staging memory (the "one pinned hipMemcpyAsync" of the design).  kpilqr_host_free accepts ctx = NULL: an
 * allocation may be released after the context it was made through has been destroyed (bindings whose arrays outlive the
 * engine object); it must not be in use by a transfer still in flight. */
int  kpilqr_host_alloc(kpilqr_ctx *ctx, size_t bytes, void **pinned);
int  kpilqr_host_free(kpilqr_ctx *ctx, void *pinned);

int  kpilqr_sync(kpilqr_ctx *ctx);
int  kpilqr_device_ptr(kpilqr_ctx *ctx, int which, void **dptr, size_t *bytes);

/* ---- STEP 1b: dynamics derivatives -------------------------------------------------------
 * Key-points per DoF: kp_times[kp_offsets[b*dof+i] .. kp_offsets[b*dof+i+1]) = sorted time
 * indices at which DoF i of trajectory b is finite-differenced (the transpose of the reference's
 * std::vector<std::vector<int>> keypoints, include/KeyPointGenerator.h:85-100).  */
int  kpilqr_set_keypoints(kpilqr_ctx *ctx, const int *kp_offsets, const int *kp_times);

/* Key-point placement on the device for the whole batch (optional).
 * X [batch][T][n]: the nominal trajectory states (positions then velocities), as Optimiser::X_old. */
int  kpilqr_upload_states(kpilqr_ctx *ctx, const double *X);
/* KeypointGenerator::GenerateKeyPoints (src/KeyPointGenerator/KeyPointGenerator.cpp:76-135) for method
 * "set_interval" (:319-339), "adaptive_jerk" (:730-770 + :341-382), "adaptive_accel" (:772-795 + :341-382, dispatch
 * :98-101; thresholds = jerk_thresholds, which the placement reads for either profile) or "velocity_change" (:797-808 + :642-728)
 * on every trajectory; the lists become the context's key-points exactly as if given to
 * kpilqr_set_keypoints.  thresholds [dof] (jerk or velocity-change thresholds; NULL for set_interval), dt =
 * model time-step.  "iterative_error" interleaves host finite differences: its placement loop stays on the host and calls
 * kpilqr_keypoint_error_test per bisection level. */
int  kpilqr_generate_keypoints(kpilqr_ctx *ctx, const char *method, int min_N, int max_N,
                               const double *thresholds, double dt);
/* The error test of "iterative_error" for one bisection level of the whole batch (KeypointGenerator::CheckDOFColumnError,
 * src/KeyPointGenerator/KeyPointGenerator.cpp:550-640): intervals [n][4] = trajectory, DoF, start, end.  The host
 * differences the DoF's columns at start, (start+end)/2 and end of every pending interval (kpilqr_upload_fd +
 * kpilqr_fd_difference put them into the step records), this call says which intervals are good (1: keep, 0: split at the
 * midpoint), the host bisects the others -- placement stays a host loop because it interleaves the simulator
 * (GenerateKeyPointsIteratively :449-548), the arithmetic of a level runs here.  Synchronous. */
int  kpilqr_keypoint_error_test(kpilqr_ctx *ctx, int n, const int *intervals, int min_N, double threshold, unsigned char *good);
/* Reads the current per-DoF lists back (the host FD loop needs them): kp_offsets [batch*dof+1]; kp_times may be
 * NULL to query the size.  Returns the total number of entries (>= 0) or an error (< 0).  Synchronous. */
int  kpilqr_get_keypoints(kpilqr_ctx *ctx, int *kp_offsets, int *kp_times, int times_capacity);

/* Host FD results, one job per perturbed column (Differentiator::DynamicsDerivatives,
 * src/Differentiator/Differentiator.cpp:81-428 stays on the host and fills these):
 *   job_b[j], job_t[j]   trajectory and time index of the key-point
 *   job_col[j]           0..n-1 -> column of A (i: d/dqpos_i, i+dof: d/dqvel_i); n..n+m-1 -> column of B
 *   job_mode[j]          0 central (x+ - x-)/(2 eps), 1 forward (x+ - xnom)/eps, 2 backward (xnom - x-)/eps
 *   job_nom[j]           row of xnom holding the unperturbed next state (modes 1,2)
 *   xplus, xminus        [njobs][n] tangent-space next states; xnom [nnom][n]
 * One hipMemcpyAsync per array; with pinned host arrays (kpilqr_host_alloc) the call returns without waiting for them
 * (pageable arrays are waited for).  Jobs may come in any order; indices are checked on the device (see
 * kpilqr_upload_fd_slab): no host-side pass over the jobs. */
int  kpilqr_upload_fd(kpilqr_ctx *ctx, int njobs, const int *job_b, const int *job_t,
                      const int *job_col, const unsigned char *job_mode, const int *job_nom,
                      const double *xplus, const double *xminus,
                      int nnom, const double *xnom, double eps);
/* ---- asynchronous boundary: one pinned slab, no host-side loops, no stream synchronisation ---------------------------
 * The FD workers (Optimiser::WorkerComputeDerivatives, src/Optimiser/Optimiser.cpp:262-323, here
 * Differentiator::DynamicsDerivativesPlanned) write their jobs straight into ONE pinned allocation laid out as below; the
 * upload is then a single hipMemcpyAsync into an identically laid-out device slab.  Jobs may come in any order.  Indices
 * are range-checked ON THE DEVICE (a bad job is skipped); a violation is reported by the next kpilqr_sync as
 * KPILQR_ERR_ARG. */
typedef struct {
    size_t xplus, xminus, xnom;                 /* byte offsets of the double arrays [njobs][n], [njobs][n], [nnom][n] */
    size_t job_b, job_t, job_col, job_nom;      /* int arrays [njobs]                                                   */
    size_t job_mode;                            /* unsigned char array [njobs]                                          */
    size_t bytes;                               /* size of the slab                                                     */
} kpilqr_fd_layout;
int  kpilqr_fd_slab_layout(kpilqr_ctx *ctx, int njobs, int nnom, kpilqr_fd_layout *out);
int  kpilqr_upload_fd_slab(kpilqr_ctx *ctx, const void *slab, int njobs, int nnom, double eps);

/* ---- key-point ordered FD payload: no job lists at all -------------------------------------------------------------------
 * The same FD results laid out BY the key-point lists the context holds (kpilqr_set_keypoints / kpilqr_generate_keypoints +
 * kpilqr_get_keypoints): CSR entry e = position in kp_times, i.e. (trajectory b, DoF d, key-point time t = kp_times[e]),
 * and three slots per entry,
 *     kind 0: qpos_d perturbed  -> column d       of A      (Differentiator.cpp:328-428)
 *     kind 1: qvel_d perturbed  -> column d + dof of A      (:226-325)
 *     kind 2: ctrl_d perturbed  -> column d       of B      (:81-223; only d < num_ctrl, other kind-2 slots are ignored)
 * One RECORD per entry, `entry_stride` = (6n + 2) * 8 bytes, records back to back in CSR order:
 *   struct { double xplus, xminus; } x[3][n]
 *            xplus   next state after the + perturbation  (a backward-only difference: the unperturbed next state)
 *            xminus  next state after the - perturbation  (a forward-only difference:  the unperturbed next state)
 *            -- the two sides of an element side by side (`elem_stride` = 16 bytes from one element of a side to the next, xminus
 *            8 bytes behind xplus; version >= 400; versions < 400 stored the sides as two blocks of 3n doubles): the sweep that
 *            differences the payload itself fetches both with ONE 16-byte load per element
 *   int32  mode              bit k set: kind k is one-sided -> (xplus - xminus) / eps, else (xplus - xminus) / (2 eps)
 *   int32  pad[3]
 * so the host FD loop writes every perturbed next state straight to its slot, nothing carries indices, the library never
 * walks or sorts anything, and a trajectory's (or a chunk of trajectories') payload is one contiguous range.  On a KPILQR_FLAG_FUSED context (one wavefront per trajectory, or the consumer / helper wave pair up to #SIMDs / 2 trajectories) there
 * is then NO differencing kernel either: the backward sweep reads the slots of a key-point when it reaches it, forms the
 * column (the arithmetic of Differentiator.cpp:166-222,441-457, bit for bit what kpilqr_fd_difference gives) and keeps it
 * for the forward sweep.  Every other context accepts the payload too (it is differenced by a streaming kernel first).
 * The payload refers to the CURRENT key-points: upload it after them; new key-points invalidate it.  `entries` must be
 * kp_offsets[batch*dof].  One hipMemcpyAsync; with a pinned slab the call does not wait. */
typedef struct {
    size_t entry_stride;                        /* bytes of one entry record: (6n + 2) * 8                       */
    size_t xplus, xminus, mode;                 /* byte offsets inside a record: 0, 8, 6n * 8 (an int32)         */
    size_t bytes;                               /* entries * entry_stride                                        */
    size_t elem_stride;                         /* bytes between consecutive elements of xplus (and of xminus): 16; element
                                                   (kind k, row r) of xplus sits at xplus + (k * n + r) * elem_stride    */
} kpilqr_fdkp_layout;
int  kpilqr_fd_kp_layout(kpilqr_ctx *ctx, int entries, kpilqr_fdkp_layout *out);
int  kpilqr_upload_fd_kp(kpilqr_ctx *ctx, const void *slab, int entries, double eps);

/* The key-point columns themselves, for a host that has already differenced (the reference's own place for a2,
 * Differentiator.cpp:166-222,441-457, or analytic derivatives): columns [entries][3][n] in CSR entry order, kinds as above
 * (column d of A, column d + dof of A, column d of B; kind-2 slots of DoFs >= num_ctrl are ignored) -- the layout of the
 * library's key-point column store, uploaded straight into it: 3n doubles per entry instead of the 6n + 2 of the FD payload,
 * and no differencing on the device.  With the IEEE quotients (x+ - x-) / (2 eps) the gains are bit for bit those of
 * kpilqr_upload_fd_kp.  Refers to the CURRENT key-points like the FD payloads; replaces them. */
int  kpilqr_upload_kp_columns(kpilqr_ctx *ctx, const double *columns, int entries);

/* One whole iteration for the batch, PIPELINED over chunks of trajectories: chunk c's uploads, its kernels and its
 * downloads run on their own stream, so H2D(c+1), kernels(c) and D2H(c-1) overlap (and so do consecutive calls: nothing
 * here waits for the previous iteration).  Every host pointer must be pinned (kpilqr_host_alloc); NULL inputs keep what
 * is resident, NULL outputs are not downloaded.  Jobs are sorted by trajectory; traj_job_first / traj_nom_first
 * [batch+1] give the first job / nominal row of every trajectory (jobs of trajectory b reference nominal rows in
 * [traj_nom_first[b], traj_nom_first[b+1]) only).  Key-points, weights, control limits and alphas are set with the
 * ordinary calls beforehand.  Results are valid after kpilqr_sync.  nchunks = 0 lets the library choose (3: one chunk
 * per pipeline stream; the sweeps are latency-bound, so more chunks than streams only add their latency).  Uploads go
 * through the DMA engine, K and k come back through a copy kernel writing the pinned buffers: on this platform two DMA
 * directions do not overlap, a DMA upload and a kernel download do.
 * Consecutive calls overlap without a wait as long as (njobs, nnom, traj_job_first, traj_nom_first) stay the same -- the
 * usual case: same key-points, new payload.  When they differ from the iteration still in flight the call first orders
 * itself behind that whole iteration (the device slab is re-laid-out, so chunks could otherwise overwrite ranges another
 * chunk stream is still reading).  All offsets are validated before anything is enqueued. */
typedef struct {
    const void *fd_slab;                        /* kpilqr_fd_slab_layout(njobs, nnom); NULL: no new FD payload, the
                                                   key-point columns already on the device are reused                    */
    int njobs, nnom;
    const int *traj_job_first, *traj_nom_first; /* [batch+1]                                                             */
    double eps;
    const double *r, *r_x, *r_u;                /* [batch][T+1][nr], [..][nr][n], [..][nr][m]                            */
    const double *u_nom;                        /* [batch][T][m]                                                         */
    const double *lambda;                       /* [batch]                                                               */
    double *K, *k;                              /* out: [batch][T][n][m], [batch][T][m]                                  */
    double *cost_pred, *delta_J;                /* out: [batch][n_alpha], [batch]                                        */
    int *status;                                /* out: [batch]                                                          */
    const void *fd_kp_slab;                     /* key-point ordered payload (kpilqr_fd_kp_layout) instead of fd_slab; the
                                                   chunks' ranges follow from the key-points, no offset arrays needed     */
    int entries;                                /* kp_offsets[batch*dof]                                                 */
    const double *kp_columns;                   /* the key-point columns (kpilqr_upload_kp_columns) instead of an FD payload:
                                                   [entries][3][n]; version >= 301                                        */
} kpilqr_stream_io;
int  kpilqr_iterate_streamed(kpilqr_ctx *ctx, const kpilqr_stream_io *io, int pd_check_stride, int nchunks);

/* Differencing tail of Differentiator::DynamicsDerivatives (:166-222,286-321,386-423,441-457):
 * writes the key-point columns of A and B. */
int  kpilqr_fd_difference(kpilqr_ctx *ctx);
/* KeypointGenerator::InterpolateDerivatives (src/KeyPointGenerator/KeyPointGenerator.cpp:840-954). */
int  kpilqr_interpolate(kpilqr_ctx *ctx);

/* Optimiser::FilterDynamicsMatrices (src/Optimiser/Optimiser.cpp:340-406, run from GenerateDerivatives :105-107 when
 * the task sets `filtering`): rows dof..2dof-1 of every A[t], filtered along time in place, after
 * kpilqr_interpolate.  method "low_pass" (coefs[0] = lowPassACoefficient, Optimiser.h:219) or "FIR"
 * (coefficients, Optimiser.h:220; at most 16).  Not available on a KPILQR_FLAG_FUSED context (the fused sweeps
 * re-interpolate from the key-point columns). */
int  kpilqr_filter_dynamics(kpilqr_ctx *ctx, const char *method, const double *coefs, int ncoef);

/* ---- STEP 1c: cost derivatives -----------------------------------------------------------
 * Residuals and their host-side FD Jacobians (Differentiator::ResidualDerivatives stays on the
 * host): r [batch][T+1][nr], r_x [batch][T+1][nr][n], r_u [batch][T+1][nr][m]; residual weights
 * w_run / w_term [nr] (struct residual, include/StdInclude.h:82-88).  Any pointer may be NULL to
 * keep what is already resident.  The buffers start zeroed: a task whose residuals do not depend on the controls
 * (r_u = 0: reaching, the pushing tasks) never passes r_u, and the fused sweeps then leave the control-residual
 * products out (l_uu = l_u = 0 exactly).
 * All T+1 rows of r must hold finite numbers, the last one (t = T) included, as the reference's do (it evaluates the residuals
 * at every t = 0..T, src/Optimiser/Optimiser.cpp:217-236): with an ODD residual count the one-tile sweeps fetch a row of r in
 * 16-byte pairs, and the last pair of row t reaches one element into row t+1 -- under a zero weight, which keeps a finite
 * number out of every result and would not keep a NaN out. */
int  kpilqr_upload_residuals(kpilqr_ctx *ctx, const double *r, const double *r_x, const double *r_u,
                             const double *w_run, const double *w_term);
/* CONSTANT residual Jacobians: one r_x [nr][n] (and one r_u [nr][m], or NULL for r_u = 0) that holds at every step of every
 * trajectory -- a task whose residuals are affine in the state and free of the controls, e.g. reaching: r = [q - q*, qdot],
 * r_x = selector rows, r_u = 0 (src/ModelTranslator/Reaching.cpp:43-54; the host then skips
 * Differentiator::ResidualDerivatives, src/Differentiator/Differentiator.cpp:464-663, altogether).  Uploaded ONCE per
 * context instead of T+1 copies per trajectory and iteration; on a KPILQR_FLAG_FUSED context with one wavefront per
 * trajectory (batch > #SIMDs / 4) and r_u = NULL the sweeps keep the matrix in registers and read no r_x from memory at all
 * (Panda reaching, T = 3000: 5.0 of the 8.1 MB a trajectory's backward sweep reads, 5.0 of 9.1 MB forward).  Every other
 * kernel family sees the same values through a broadcast copy made on demand, and then K, k, delta_J and the predicted costs
 * are bit for bit those of the same matrix given per step through kpilqr_upload_residuals (which, with r_x != NULL, also ends
 * the constant mode).  The sweeps that keep the matrix in registers (kpilqr_last_launch: "...:rxc") also keep the constant
 * block l_xx = r_x' W r_x as a resident tile and add l_x = r_x' W r to it with ONE matrix product per step instead of four
 * (version >= 410): the same numbers in another accumulation order -- gains identical, k / delta_J / costs within ~1e-15
 * relative of the per-step form (tests hold 1e-12).  A call that is rejected (bad argument, unpinned buffer) leaves the mode
 * as it was.  version >= 400. */
int  kpilqr_upload_residual_jacobians_const(kpilqr_ctx *ctx, const double *r_x, const double *r_u);
/* ModelTranslator::CostDerivativesFromResiduals (src/ModelTranslator/ModelTranslator.cpp:552-583)
 * over the loop of Optimiser::ComputeCostDerivatives (src/Optimiser/Optimiser.cpp:202-211),
 * including the terminal-weight re-write of t = T-1. */
int  kpilqr_cost_derivs(kpilqr_ctx *ctx);
/* ModelTranslator::CostFunction (:314-327) summed over the horizon as RolloutTrajectory does
 * (src/Optimiser/iLQR.cpp:202-254): cost[b] = sum_{t<T-1} w_run.r_t^2 + w_term.r_{T-1}^2. */
int  kpilqr_trajectory_cost(kpilqr_ctx *ctx, double *cost /*[batch]*/);

/* ---- STEP 2: backward pass ----------------------------------------------------------------
 * iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD (src/Optimiser/iLQR.cpp:535-670).
 * lambda [batch]; pd_check_stride = 100 in the reference.  status[b] = 0 ok, t+1 = first step
 * whose Q_uu + lambda I failed the Cholesky test; delta_J [batch].  status / delta_J may be NULL
 * (results stay on the device, KPILQR_BUF_STATUS / KPILQR_BUF_DELTA_J).  A trajectory with status != 0 has stopped at that
 * step (the reference returns false there and retries at a larger lambda, iLQR.cpp:435-442): its gains below that step, its
 * delta_J and the outputs of a kpilqr_forward_linear that follows are UNDEFINED for that trajectory until a backward pass
 * succeeds (on a fused context its key-point columns may be differenced only down to that step). */
int  kpilqr_backward(kpilqr_ctx *ctx, const double *lambda, int pd_check_stride,
                     int *status, double *delta_J);
/* Diagnostic: the backward pass of a KPILQR_FLAG_FUSED context with counters.  The explicit inverse the reference forms at
 * every step (iLQR.cpp:597-600) is carried along the sweep and refreshed on the matrix core; how much work a step needs is
 * data dependent (and a launch lasts as long as its slowest wavefront).  hist [batch][6] = steps whose inverse came from:
 * [0] the third-order refresh alone, [1..3] that plus 1 / 2 / 3 second-order steps, [4] the LDL' factorisation (first step,
 * every pd_check_stride-th step, re-seeds), [5] Eigen's pivoted LDLT restated (indefinite Q_uu + lambda I on an unchecked
 * step).  K, k, delta_J, status as kpilqr_backward; lambda as last given.  Synchronous. */
int  kpilqr_backward_stats(kpilqr_ctx *ctx, int pd_check_stride, int *hist);
/* K [batch][T][n][m] (column-major m x n), k [batch][T][m]; either may be NULL. */
int  kpilqr_download_gains(kpilqr_ctx *ctx, double *K, double *k);

/* iLQR_SVR::LeastImportantDofs, "sampling and summing" branch (src/Optimiser/iLQR_SVR.cpp:952-968), on the gains of
 * the last backward pass: sums [batch][dof] = (sum over t = 0, s, 2s, ... and controls j of
 * |K[t](j,i)| + |K[t](j,i+dof)|) / T.  The SVD branch (:902-950) and the state-vector resize it triggers stay on
 * the host (host/SVR.h; the resize is kpilqr_resize). */
int  kpilqr_dof_importance(kpilqr_ctx *ctx, int sampling_k_interval, double *sums);

/* ---- STEP 3: forward pass over the line-search alphas ----------------------------------------
 * Nominal controls U_old [batch][T][m] and ModelTranslator::ReturnControlLimits [2*m] = lo,hi pairs. */
int  kpilqr_upload_nominal(kpilqr_ctx *ctx, const double *u_nom, const double *ctrl_lim);
/* Control law + clamp of iLQR::ForwardsPassParallel (src/Optimiser/iLQR.cpp:876-890) on the
 * linearised model, scored with the quadratic cost model (declared semantic change, DESIGN.md
 * section 2).  alphas [n_alpha]; cost_pred [batch][n_alpha] = predicted cost CHANGE;
 * U_alpha [batch][n_alpha][T][m] or NULL. */
int  kpilqr_forward_linear(kpilqr_ctx *ctx, const double *alphas, double *cost_pred, double *U_alpha);

/* ---- one whole iteration (STEP 1b + 1c + 2 + 3) enqueued back to back ------------------------ */
int  kpilqr_iterate(kpilqr_ctx *ctx, const double *lambda, int pd_check_stride, const double *alphas);

/* ---- several GPUs: trajectories are sharded, one context (and one process or thread) per GPU; the only collective
 * is the line-search cost reduction named by the design: one all-reduce of 8 doubles per iteration over RCCL/xGMI,
 *   vec8 = [ sum_b J_pred(alpha_1..6), sum_b delta_J, number of trajectories with a valid backward pass ]
 * summed over the trajectories of every rank whose status is 0.  Rank 0 creates the 128-byte RCCL unique id and the
 * host distributes it (file, pipe, MPI -- the library does no networking of its own); without kpilqr_comm_init the
 * call returns this rank's sums.  Enqueued on the context's stream after the forward pass; vec8 (host, may be NULL)
 * is valid after kpilqr_sync.  RCCL is loaded on first use (dlopen), not at link time. */
int  kpilqr_comm_unique_id(char id[128]);
int  kpilqr_comm_init(kpilqr_ctx *ctx, int nranks, int rank, const char id[128]);
int  kpilqr_allreduce_linesearch(kpilqr_ctx *ctx, double vec8[8]);

/* ---- debug / oracle hooks: inject or read the intermediates in the reference's layout -------
 * (the reference exposes A, B, l_x ... as public members, include/Optimiser/Optimiser.h:194-211,
 * and GenTestingData dumps them, src/GenTestingData.cpp:795-797).  NULL pointers are skipped. */
int  kpilqr_set_AB(kpilqr_ctx *ctx, const double *A, const double *B);
int  kpilqr_get_AB(kpilqr_ctx *ctx, double *A, double *B);
int  kpilqr_set_cost_derivs(kpilqr_ctx *ctx, const double *l_x, const double *l_xx,
                            const double *l_u, const double *l_uu);
int  kpilqr_get_cost_derivs(kpilqr_ctx *ctx, double *l_x, double *l_xx, double *l_u, double *l_uu);

/* Name of the kernel variant the backward / forward pass will launch for these dims
 * ("mfma_f64_t1", "generic_lds", ...): for logs, tests and the bench's roofline line. */
const char *kpilqr_backward_variant(kpilqr_ctx *ctx);
const char *kpilqr_forward_variant(kpilqr_ctx *ctx);
/* What the LAST backward (which = 0) / forward (which = 1) launch of this context was -- the variant above and, for the
 * KPILQR_FLAG_FUSED sweeps, the form the library picked from the batch size, the payload and the key-point lists:
 *     "<variant>:<waves>:<columns>:<lists>[:ru0][:rxc][:slopes]"          e.g. "mfma_f64_t1_fused:w1:raw:uni:ru0"
 *   waves    w1 one wavefront per trajectory | w2 control / state split | pair | triple | pairh (backward: consumer / helper pair)
 *            (forward `pair` on a uniform set: state wave with its own interpolant + scoring wave)
 *   columns  (backward only) raw: the sweep differenced the key-point ordered FD payload itself | kpc: it read the differenced
 *            key-point column store
 *   lists    uni: every DoF of a trajectory has the same key-point list (set_interval ...) | ragged: per-DoF lists
 *   ru0      no control residuals (r_u never uploaded): the products with r_u are left out
 *   rxc      constant residual Jacobians kept in registers (kpilqr_upload_residual_jacobians_const)
 *   slopes   per-DoF lists walked on precomputed segment slopes (a crossing is loads only)
 * The `lists` token is decided on the device; this call reads the flag back and therefore WAITS for the context's stream.
 * For logs and tests (a test can assert which kernel form it exercised).  version >= 400. */
const char *kpilqr_last_launch(kpilqr_ctx *ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
