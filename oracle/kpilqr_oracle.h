/*
 * kpilqr_oracle.h -- CPU restatement (plain C, FP64) of the keypoint-iLQR hot path of
 * DMackRus/TrajOptKP.  TEST INFRASTRUCTURE ONLY: nothing outside tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call this.  The product path
 * (libkpilqr.so) never routes through it.
 *
 * Parity status: the reference cannot be built here (needs Eigen3, a
 * MuJoCo fork, yaml-cpp, GLFW, gtest -- none present), so this file restates the reference's
 * algorithm line by line and is pinned ONLY where the reference's own tests pin the path:
 *   - orc_interpolate            pinned by Interpolate.basic_interpolation
 *                                 (src/tests/Keypoints_Test.cpp:204-308, bitwise relation)
 *   - orc_kp_*                   pinned structurally by keypoints.{set_interval,adaptive_jerk,
 *                                 velocity_change} (src/tests/Keypoints_Test.cpp:10-33,53-202)
 *   - orc_cost_derivs, orc_backward, orc_forward_*, lambda/convergence logic:
 *                                 PARITY UNPINNED -- the reference holds no test, golden vector
 *                                 or fixture for them.  They are cross-checked against an
 *                                 independent numpy/scipy float64 script (oracle/crosscheck.py).
 *
 * Layout convention (that of the reference's Eigen objects): every matrix is COLUMN-MAJOR,
 * one matrix per time-step, time-steps contiguous: A[t] at A + t*n*n, element (r,c) at r + c*n.
 * n = 2*dof (tangent-space state), m = num_ctrl, nr = number of residuals, T = horizon.
 */
#ifndef KPILQR_ORACLE_H
#define KPILQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a2: finite-difference tail of Differentiator::DynamicsDerivatives ----------------
 * (src/Differentiator/Differentiator.cpp:166-222, 286-321, 386-423, 441-457)
 * One job = one perturbed column.  mode 0: central (xplus - xminus)/(2*eps);
 * mode 1: forward (xplus - xnom)/eps; mode 2: backward (xnom - xminus)/eps.
 * out_col < n  -> column out_col of A[t];  out_col >= n -> column out_col-n of B[t].
 * Hinge/slide joints only: mj_differentiatePos reduces to (q2-q1)/dt for them, so position
 * and velocity rows share one formula. */
void orc_fd_difference(int n, int m, int njobs,
                       const int *job_t, const int *job_col, const unsigned char *job_mode,
                       const int *job_nom,
                       const double *xplus, const double *xminus, const double *xnom,
                       double eps, double *A, double *B);

/* ---- a3: key-point placement (src/KeyPointGenerator/KeyPointGenerator.cpp) -------------
 * Key-points are returned as CSR over time: cols[offs[t] .. offs[t+1]) = DoF indices to
 * finite-difference at step t (the reference's std::vector<std::vector<int>> keypoints).
 * offs has T+1 entries, cols must hold T*dof ints.  Each returns the number of entries. */
int orc_kp_set_interval(int dof, int T, int min_N, int *offs, int *cols);           /* :319-339 */
/* X: trajectory states, T entries of n=2*dof doubles (positions then velocities). */
int orc_kp_adaptive_jerk(int dof, int T, int min_N, int max_N, const double *jerk_thresholds,
                         double dt, const double *X, int *offs, int *cols);          /* :730-770,341-382 */
int orc_kp_adaptive_accel(int dof, int T, int min_N, int max_N, const double *jerk_thresholds,
                          const double *X, int *offs, int *cols);                     /* :772-795,341-382 */
int orc_kp_velocity_change(int dof, int T, int min_N, int max_N, const double *vel_thresholds,
                           const double *X, int *offs, int *cols);                   /* :642-728,797-808 */
/* iterative_error on a GIVEN dense A sequence (the reference interleaves MuJoCo FD here;
 * with A precomputed for every t the bisection logic is identical, :449-640). */
int orc_kp_iterative_error(int dof, int T, int min_N, double threshold, const double *A,
                           int *offs, int *cols);
/* Percentage of key-points per DoF (:810-838). */
void orc_kp_percentages(int dof, int T, const int *offs, const int *cols, double *pct);

/* ---- a4: KeypointGenerator::InterpolateDerivatives (:840-954) -------------------------- */
void orc_interpolate(int dof, int m, int T, const int *offs, const int *cols,
                     double *A, double *B);

/* ---- a6: ModelTranslator::CostFunction (:314-327) and CostDerivativesFromResiduals
 * (src/ModelTranslator/ModelTranslator.cpp:552-583) driven by the loop of
 * Optimiser::ComputeCostDerivatives (src/Optimiser/Optimiser.cpp:197-215): running weights for
 * t < T, then t = T-1 recomputed with the terminal weights.
 * r: [T+1][nr], r_x: [T+1][nr][n], r_u: [T+1][nr][m]. */
double orc_cost_function(int nr, const double *r, const double *w);
void orc_cost_derivs(int n, int m, int nr, int T,
                     const double *r, const double *r_x, const double *r_u,
                     const double *w_run, const double *w_term,
                     double *l_x, double *l_xx, double *l_u, double *l_uu);

/* ---- a7: iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD (src/Optimiser/iLQR.cpp:535-670)
 * Returns 0 on success, t+1 when Q_uu+lambda*I fails the Cholesky test at step t (only
 * tested every pd_stride-th step, as the reference does with 100).
 * K: [T][m*n] column-major m x n, k: [T][m]. */
int orc_backward(int n, int m, int T,
                 const double *A, const double *B,
                 const double *l_x, const double *l_xx, const double *l_u, const double *l_uu,
                 double lambda, int pd_stride, double *K, double *k, double *delta_J);

/* Eigen pieces restated for the backward pass (exposed for unit tests). */
int  orc_llt_is_pd(int m, const double *M);                 /* Eigen::LLT info()==Success   */
void orc_ldlt_inverse(int m, const double *M, double *Minv); /* M.ldlt().solve(Identity)     */

/* ---- a8: forward pass ------------------------------------------------------------------
 * alphas (src/Optimiser/iLQR.cpp:466-470): (i/n)^2, i=1..n. */
void orc_alphas(int n_alpha, double *alphas);
/* Linearised forward rollout over n_alpha line-search steps (the design's replacement for the
 * closed-loop MuJoCo rollouts of ForwardsPassParallel, src/Optimiser/iLQR.cpp:824-934; control
 * law and clamp :876-890 restated exactly, dynamics and cost replaced by their first/second
 * order models):  dx_0 = 0;  u = clamp(u_nom + alpha*k + K*dx);  du = u - u_nom;
 * cost_pred = sum_t l_x'dx + 0.5 dx'l_xx dx + l_u'du + 0.5 du'l_uu du;  dx <- A dx + B du.
 * ctrl_lim: [2*m] = lo0,hi0,lo1,hi1,... (ModelTranslator::ReturnControlLimits layout).
 * U_alpha (nullable): [n_alpha][T][m]. */
void orc_forward_linear(int n, int m, int T, int n_alpha, const double *alphas,
                        const double *A, const double *B, const double *K, const double *k,
                        const double *l_x, const double *l_xx, const double *l_u, const double *l_uu,
                        const double *u_nom, const double *ctrl_lim,
                        double *cost_pred, double *U_alpha);

/* ---- a9: scalar control flow of iLQR::Iteration --------------------------------------- */
/* UpdateLambda (src/Optimiser/iLQR.cpp:636-657). Returns lambda_exit. */
int orc_update_lambda(double *lambda, int valid_backwards_pass,
                      double lambda_factor, double min_lambda, double max_lambda);
/* CheckForConvergence (src/Optimiser/Optimiser.cpp:30-37). */
int orc_check_convergence(double old_cost, double new_cost, double eps_converge);
/* Tail of Iteration (src/Optimiser/iLQR.cpp:490-528): pick best alpha, accept or undo lambda.
 * Returns index of best alpha; *accepted=1 when best < old_cost; updates lambda, new_cost. */
int orc_linesearch_accept(int n_alpha, const double *costs, double old_cost,
                          double *new_cost, int *accepted,
                          double *lambda, double lambda_factor, double max_lambda);

/* ---- optional pieces either side of the path (parity unpinned: the reference has no test for them) */
/* Optimiser::FilterDynamicsMatrices (src/Optimiser/Optimiser.cpp:340-406); method 0 low_pass (coefs[0] = a), 1 FIR */
void orc_filter_dynamics(int dof, int T, int method, const double *coefs, int ncoef, double *A);
/* iLQR_SVR::LeastImportantDofs, summing branch (src/Optimiser/iLQR_SVR.cpp:952-968) */
void orc_dof_importance(int dof, int m, int T, int sampling, const double *K, double *sums);
/* iLQR_SVR alphas 1 - i/n (src/Optimiser/iLQR_SVR.cpp:469-471) */
void orc_alphas_svr(int n_alpha, double *alphas);

/* ---- whole iteration (a2 + a4 + a6 + a7 + a8) for ONE trajectory, and a pthread driver that runs
 * independent trajectory-iterations on `nthreads` host threads (the reference's own parallelism is a
 * std::thread pool over key-points, src/Optimiser/Optimiser.cpp:227,280; over a batch the natural
 * unit is the trajectory).  Used for cross-checks and as bench.py's cpu_baseline ("port"). */
typedef struct {
    int dof, m, nr, T, njobs, pd_stride, n_alpha;
    double eps, lambda;
    const int *job_t, *job_col, *job_nom;
    const unsigned char *job_mode;
    const double *xplus, *xminus, *xnom;
    const int *kp_offs, *kp_cols;               /* CSR over time */
    const double *r, *r_x, *r_u, *w_run, *w_term, *u_nom, *ctrl_lim;
} orc_problem;
/* K [T][m*n], k [T][m], cost_pred [n_alpha]; returns the backward-pass status. */
int orc_iteration(const orc_problem *p, double *K, double *k, double *delta_J, double *cost_pred);
/* Runs nthreads x reps iterations (each thread on private work buffers); returns wall seconds. */
double orc_iteration_batch(const orc_problem *p, int nthreads, int reps);

#ifdef __cplusplus
}
#endif
#endif
