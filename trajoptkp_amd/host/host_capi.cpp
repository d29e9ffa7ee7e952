// host_capi.cpp -- small C entry points over the C++ host classes so that the Python tests can drive
// them (key-point generators against the oracle; the acrobot plumbing optimisation end to end).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include "AcrobotModel.h"
#include "FloatingBodyModel.h"
#include "FileHandler.h"
#include "SVR.h"
#include "iLQR_GPU.h"
#include "iLQR_GPU_Batch.h"

extern "C" {

// Key-points of `method` for a trajectory X [T][2*dof]; A (optional, [T][n*n] column-major) feeds
// iterative_error.  Returns the number of entries written to cols; offs has T+1 entries.
int kpilqr_host_keypoints(const char *method, int dof, int T, int min_N, int max_N, const double *thresholds,
                          double iterative_error_threshold, double dt, const double *X, const double *A,
                          int *offs, int *cols, double *percentages)
{
    const int n = 2 * dof;
    KeypointGenerator gen(dof, T);
    keypoint_method km;
    km.name = method; km.min_N = min_N; km.max_N = max_N;
    if (thresholds) { km.jerk_thresholds.assign(thresholds, thresholds + dof); km.velocity_change_thresholds = km.jerk_thresholds; }
    km.iterative_error_threshold = iterative_error_threshold;
    gen.SetKeypointMethod(km);
    std::vector<MatrixXd> states(T, MatrixXd(n, 1));
    if (X) for (int t = 0; t < T; t++) std::memcpy(states[t].data(), X + (size_t)t * n, sizeof(double) * n);
    KeypointGenerator::ColumnFD fd;
    if (A) fd = [&](int t, int i, double *cp, double *cv) {
        std::memcpy(cp, A + (size_t)t * n * n + (size_t)i * n, sizeof(double) * n);
        std::memcpy(cv, A + (size_t)t * n * n + (size_t)(i + dof) * n, sizeof(double) * n);
    };
    gen.GenerateKeyPoints(states, dt, fd);
    int cnt = 0;
    for (int t = 0; t < T; t++) { offs[t] = cnt; for (int i : gen.keypoints[t]) cols[cnt++] = i; }
    offs[T] = cnt;
    if (percentages) for (int i = 0; i < dof; i++) percentages[i] = gen.last_percentages[i];
    return cnt;
}

// Acrobot swing-up, BASELINE configs[0]: T steps, set_interval(min_N) key-points, zero initial controls
// from the hanging-ish start [3.1415, 0.3] (acrobot.yaml:16).  Runs Optimise(max_iter, min_iter) on the
// GPU engine; writes cost_history (initial cost first) and the final controls.  Returns iterations run,
// or <0 on error.
int kpilqr_host_run_acrobot(int T, int min_N, int max_iter, int min_iter, const char *keypoint_method_name,
                            double torque_weight /* <0: task default 100 */,
                            double *cost_history, int cost_cap, double *U_out, double *K0_out, double *timings_ms)
{
    // keypoint_method_name may carry options after a '+': "set_interval+unfused", "set_interval+analytic"
    std::string spec = keypoint_method_name ? keypoint_method_name : "";
    const std::string keypoint_method_full = spec;
    const bool unfused = spec.find("+unfused") != std::string::npos, fused = spec.find("+fused") != std::string::npos, analytic = spec.find("+analytic") != std::string::npos;
    const bool lowpass = spec.find("+low_pass") != std::string::npos, fir = spec.find("+FIR") != std::string::npos;
    const bool constjac = spec.find("+constjac") != std::string::npos;
    if (spec.find('+') != std::string::npos) spec = spec.substr(0, spec.find('+'));
    keypoint_method_name = spec.empty() ? nullptr : spec.c_str();
    auto sim = std::make_shared<AcrobotSimulator>(0.01, 8);
    auto mt = std::make_shared<AcrobotTranslator>(sim);
    mt->analytic_residual_jacobians = analytic;
    // "+constjac": the task declares its residual Jacobians constant (ModelTranslator::ConstantResidualJacobians): uploaded once
    mt->constant_residual_jacobians = constjac;
    mt->min_N = min_N;
    if (torque_weight >= 0) { mt->residual_list[4].weight = torque_weight; mt->residual_list[4].weight_terminal = torque_weight; }
    if (keypoint_method_name) mt->keypoint_method = keypoint_method_name;
    sim->main_data->qpos[0] = 3.1415; sim->main_data->qpos[1] = 0.3;
    {   // "+q0=a,b": another start
        std::string full = keypoint_method_full;
        const size_t at = full.find("+q0=");
        if (at != std::string::npos) std::sscanf(full.c_str() + at + 4, "%lf,%lf", &sim->main_data->qpos[0], &sim->main_data->qpos[1]);
    }
    *sim->master_reset_data = *sim->main_data;
    auto diff = std::make_shared<Differentiator>(mt, sim);
    iLQR_GPU opt(mt, sim, diff, T);
    if (!opt.ok()) return -2;
    if (unfused) opt.SetFused(false);
    if (fused) opt.SetFused(true);
    opt.host_differencing = keypoint_method_full.find("+columns") != std::string::npos;
    if (lowpass || fir) { opt.filteringMethod = lowpass ? "low_pass" : "FIR"; opt.SetFused(false); }
    std::vector<MatrixXd> U0(T, MatrixXd(1, 1));
    std::vector<MatrixXd> U = opt.Optimise(sim->main_data, U0, max_iter, min_iter, T);
    const int nh = (int)opt.cost_history.size();
    for (int i = 0; i < nh && i < cost_cap; i++) cost_history[i] = opt.cost_history[i];
    if (U_out) for (int t = 0; t < T; t++) U_out[t] = U[t](0);
    if (K0_out) for (int c = 0; c < 4; c++) K0_out[c] = opt.K[0](0, c);
    if (timings_ms) {         // [8]: four times, then how the residual Jacobians travelled and whether the last backward sweep kept them in registers
        timings_ms[0] = opt.avg_time_get_derivs_ms; timings_ms[1] = opt.avg_time_backwards_pass_ms; timings_ms[2] = opt.avg_time_forwards_pass_ms; timings_ms[3] = opt.opt_time_ms;
        timings_ms[4] = opt.constant_jacobian_uploads; timings_ms[5] = opt.per_step_jacobian_uploads;
        timings_ms[6] = opt.LastLaunch(0).find(":rxc") != std::string::npos ? 1.0 : 0.0; timings_ms[7] = 0.0;
    }
    return opt.num_iterations;
}

// B acrobot swing-ups from different starts q0[b] = (q0s[2b], q0s[2b+1]) optimised TOGETHER through one context
// with dims.batch = B (iLQR_GPU_Batch).  cost_history: [B][cost_cap] (initial cost first), iterations [B],
// U_out [B][T], stats [8] = line-search statistics of the last iteration.  Returns 0 or <0.
int kpilqr_host_run_acrobot_batch2(int B, int T, int min_N, int max_iter, int min_iter, double torque_weight, const double *q0s,
                                   int fused, const char *method, double *cost_history, int cost_cap, int *iterations, double *U_out, double *stats);
int kpilqr_host_run_acrobot_batch(int B, int T, int min_N, int max_iter, int min_iter, double torque_weight, const double *q0s,
                                  int fused, double *cost_history, int cost_cap, int *iterations, double *U_out, double *stats)
{
    return kpilqr_host_run_acrobot_batch2(B, T, min_N, max_iter, min_iter, torque_weight, q0s, fused, nullptr, cost_history, cost_cap, iterations, U_out, stats);
}

// ... with a key-point method by name (NULL: the task's default, set_interval): the adaptive methods give every trajectory its own
// per-DoF lists, whose counts change from one linearisation to the next
int kpilqr_host_run_acrobot_batch2(int B, int T, int min_N, int max_iter, int min_iter, double torque_weight, const double *q0s,
                                   int fused, const char *method, double *cost_history, int cost_cap, int *iterations, double *U_out, double *stats)
{
    std::vector<iLQR_GPU_Batch::Problem> probs;
    for (int b = 0; b < B; b++) {
        auto sim = std::make_shared<AcrobotSimulator>(0.01, 8);
        auto mt = std::make_shared<AcrobotTranslator>(sim);
        mt->min_N = min_N;
        std::string mspec = method ? method : "";
        mt->constant_residual_jacobians = mspec.find("+constjac") != std::string::npos;
        if (mspec.find('+') != std::string::npos) mspec = mspec.substr(0, mspec.find('+'));
        if (!mspec.empty()) mt->keypoint_method = mspec;
        if (torque_weight >= 0) { mt->residual_list[4].weight = torque_weight; mt->residual_list[4].weight_terminal = torque_weight; }
        sim->main_data->qpos[0] = q0s[2 * b]; sim->main_data->qpos[1] = q0s[2 * b + 1];
        *sim->master_reset_data = *sim->main_data;
        probs.push_back({mt, sim, std::make_shared<Differentiator>(mt, sim)});
    }
    iLQR_GPU_Batch opt(probs, T, 0, fused != 0);
    if (!opt.ok()) return -2;
    std::vector<std::vector<MatrixXd>> U0(B, std::vector<MatrixXd>(T, MatrixXd(1, 1)));
    auto U = opt.OptimiseAll(U0, max_iter, min_iter);
    for (int b = 0; b < B; b++) {
        const int nh = (int)opt.cost_history[b].size();
        for (int i = 0; i < cost_cap; i++) cost_history[(size_t)b * cost_cap + i] = i < nh ? opt.cost_history[b][i] : -1.0;
        iterations[b] = opt.num_iterations[b];
        if (U_out) for (int t = 0; t < T; t++) U_out[(size_t)b * T + t] = U[b][t](0);
    }
    if (stats) for (int i = 0; i < 8; i++) stats[i] = opt.linesearch_stats[i];
    return 0;
}

// relocate_records of the batch shim (iLQR_GPU_Batch.cpp) on caller-made slabs: dst == NULL moves in place inside src
int kpilqr_host_relocate_records(char *src, char *dst, size_t stride, int B, int dof, const int *old_offs, const int *new_offs, const char *regen)
{
    const std::vector<int> oo(old_offs, old_offs + (size_t)B * dof + 1), no(new_offs, new_offs + (size_t)B * dof + 1);
    const std::vector<char> rg(regen, regen + B);
    relocate_records(src, dst ? dst : src, stride, B, dof, oo, no, rg);
    return 0;
}

// iLQR_SVR::LeastImportantDofs on gains K [T][n][m] (column-major m x n per step, the ABI's host layout): sums [dof];
// remove [dof] receives the indices below the threshold, their number is returned.
int kpilqr_host_dof_importance(const double *K, int dof, int m, int T, int sampling_k_interval, int eigen_vector_method,
                               double threshold, double *sums, int *remove)
{
    const int n = 2 * dof;
    std::vector<MatrixXd> Km(T, MatrixXd(m, n));
    for (int t = 0; t < T; t++) std::memcpy(Km[t].data(), K + (size_t)t * n * m, sizeof(double) * n * m);
    const std::vector<double> s = DofImportance(Km, dof, sampling_k_interval, eigen_vector_method != 0);
    for (int i = 0; i < dof; i++) sums[i] = s[i];
    const std::vector<int> r = LeastImportantDofs(s, threshold);
    for (size_t i = 0; i < r.size(); i++) remove[i] = r[i];
    return (int)r.size();
}

// ---- stand-in models by name, for the oracle tests of the host finite differences (a1, a5) and of the control flow (a9) ----
namespace {
struct Model {
    std::shared_ptr<PhysicsSimulator> sim;
    std::shared_ptr<ModelTranslator> mt;
};
bool make_model(const char *name, int fd_threads, Model &M)
{
    const std::string nm = name ? name : "";
    if (nm == "acrobot") {
        auto sim = std::make_shared<AcrobotSimulator>(0.01, fd_threads);
        M.sim = sim; M.mt = std::make_shared<AcrobotTranslator>(sim);
        sim->main_data->qpos[0] = 3.1415; sim->main_data->qpos[1] = 0.3;
        return true;
    }
    if (nm == "floating_body") {
        auto sim = std::make_shared<FloatingBodySimulator>(0.01, fd_threads);
        M.sim = sim; M.mt = std::make_shared<FloatingBodyTranslator>(sim);
        return true;
    }
    return false;
}
void load_state(SimData *d, const double *qpos, const double *qvel, const double *ctrl)
{
    for (int i = 0; i < d->nq; i++) d->qpos[i] = qpos[i];
    for (int i = 0; i < d->nv; i++) d->qvel[i] = qvel[i];
    for (int i = 0; i < d->nu; i++) d->ctrl[i] = ctrl[i];
}
}  // namespace

// info[0..5] = nq, nv, nu, dof, dof_quat, nr; limits [2*nu]; dt
int kpilqr_host_model_info(const char *model, int *info, double *limits, double *dt)
{
    Model M;
    if (!make_model(model, 1, M)) return -1;
    const stateVectorList &sv = M.mt->current_state_vector;
    info[0] = M.sim->main_data->nq; info[1] = M.sim->main_data->nv; info[2] = M.sim->main_data->nu;
    info[3] = sv.dof; info[4] = sv.dof_quat; info[5] = (int)M.mt->residual_list.size();
    const MatrixXd lim = M.mt->ReturnControlLimits(sv);
    for (int i = 0; i < 2 * sv.num_ctrl; i++) limits[i] = lim(i);
    *dt = M.sim->ReturnModelTimeStep();
    return 0;
}

// The model's primitives, one call each (what the numpy restatement of the reference's FD loops is written on):
//   op 0  step:           next qpos / qvel after ForwardSimulator          -> out_q [nq], out_v [nv]
//   op 1  residuals:      Residuals(state)                                 -> out_q [nr]
//   op 2  state vector:   ReturnStateVector(state)                         -> out_q [2*dof]
//   op 3  integrate pos:  qpos (+) eps * e_index                           -> out_q [nq]       (arg = eps, index)
//   op 4  differentiate:  (qpos2 (-) qpos) / dt with qpos2 in `other`      -> out_v [nv]       (arg = dt)
int kpilqr_host_model_op(const char *model, int op, const double *qpos, const double *qvel, const double *ctrl,
                         const double *other, double arg, int index, double *out_q, double *out_v)
{
    Model M;
    if (!make_model(model, 1, M)) return -1;
    const stateVectorList &sv = M.mt->current_state_vector;
    SimData *d = M.sim->main_data;
    load_state(d, qpos, qvel, ctrl);
    if (op == 0) {
        M.sim->ForwardSimulator(d);
        for (int i = 0; i < d->nq; i++) out_q[i] = d->qpos[i];
        for (int i = 0; i < d->nv; i++) out_v[i] = d->qvel[i];
    } else if (op == 1) {
        MatrixXd r((int)M.mt->residual_list.size(), 1);
        M.mt->Residuals(d, r);
        for (int i = 0; i < r.rows(); i++) out_q[i] = r(i);
    } else if (op == 2) {
        const MatrixXd x = M.mt->ReturnStateVector(d, sv);
        for (int i = 0; i < x.rows(); i++) out_q[i] = x(i);
    } else if (op == 3) {
        M.sim->IntegratePos(d, index, arg);
        for (int i = 0; i < d->nq; i++) out_q[i] = d->qpos[i];
    } else if (op == 4) {
        SimData *d2 = M.sim->fd_data[0];
        *d2 = *d;
        for (int i = 0; i < d->nq; i++) d2->qpos[i] = other[i];
        M.sim->DifferentiatePos(out_v, arg, d, d2);
    } else return -2;
    return 0;
}

// The product's host FD at ONE state: Differentiator::DynamicsDerivatives for the DoFs `cols` (jobs in the layout of
// kpilqr_upload_fd; returns the number of jobs) and ::ResidualDerivatives (r_x [nr][2 dof], r_u [nr][nu]).
int kpilqr_host_model_fd(const char *model, const double *qpos, const double *qvel, const double *ctrl, int ncols, const int *cols,
                         int central, double eps, int *job_col, unsigned char *job_mode, double *xplus, double *xminus, double *xnom,
                         double *r_x, double *r_u)
{
    Model M;
    if (!make_model(model, 2, M)) return -1;
    const int n = 2 * M.mt->current_state_vector.dof;
    load_state(M.sim->main_data, qpos, qvel, ctrl);
    M.sim->AppendSystemStateToEnd(M.sim->main_data);
    Differentiator diff(M.mt, M.sim);
    FDJobs jobs;
    diff.DynamicsDerivatives(jobs, 0, std::vector<int>(cols, cols + ncols), 0, 0, central != 0, eps);
    for (int j = 0; j < jobs.njobs(); j++) { job_col[j] = jobs.job_col[j]; job_mode[j] = jobs.job_mode[j]; }
    std::memcpy(xplus, jobs.xplus.data(), sizeof(double) * jobs.xplus.size());
    std::memcpy(xminus, jobs.xminus.data(), sizeof(double) * jobs.xminus.size());
    std::memcpy(xnom, jobs.xnom.data(), sizeof(double) * n);
    if (r_x && r_u) diff.ResidualDerivatives(r_x, r_u, 0, 0, eps);
    return jobs.njobs();
}

// The key-point ordered fill (Differentiator::DynamicsDerivativesKp, the payload of kpilqr_upload_fd_kp) against the job-list
// fill (DynamicsDerivativesBatch) on one rolled-out trajectory of a stand-in model: controls u [T][nu] (some at their limits
// make one-sided columns), key-points every min_N steps for DoF i shifted by i*stagger (ragged per-DoF lists), no GPU.
// stats[0] = jobs, [1] = one-sided jobs, [2] = entries.  Returns the number of slots that differ from the job they came from
// (x+ / x- rows, with the nominal next state in the unstepped side of a one-sided job, and the mode bits), or < 0.
int kpilqr_host_fd_kp_check(const char *model, int T, int min_N, int stagger, const double *u, int *stats)
{
    Model M;
    if (!make_model(model, 8, M)) return -1;
    const stateVectorList &sv = M.mt->current_state_vector;
    const int dof = sv.dof, n = 2 * dof, nu = sv.num_ctrl;
    SimData *d = M.sim->main_data;
    for (int t = 0; t < T; t++) {                                  // roll out, saving the states the FD loops start from
        for (int i = 0; i < nu; i++) d->ctrl[i] = u[(size_t)t * nu + i];
        M.sim->AppendSystemStateToEnd(d);
        M.sim->ForwardSimulator(d);
    }
    std::vector<std::vector<int>> kp(T);
    for (int t = 0; t < T; t++)
        for (int i = 0; i < dof; i++)
            if (t == 0 || t == T - 1 || (t + i * stagger) % min_N == 0) kp[t].push_back(i);
    KeypointGenerator gen(dof, T);
    gen.keypoints = kp;
    std::vector<int> offs, times;
    gen.PerDofCSR(offs, times);
    const int entries = offs[dof];
    Differentiator diff(M.mt, M.sim);
    FDStaging st;
    diff.DynamicsDerivativesBatch(st, 0, kp, 1e-6);
    const size_t stride = (size_t)(6 * n + 2) * 8;
    std::vector<char> slab((size_t)entries * stride, (char)0x5a);  // poisoned: every slot must be written
    diff.DynamicsDerivativesKp(slab.data(), stride, 0, offs, times, kp, 1e-6);
    std::vector<int> entry_of((size_t)T * dof, -1);
    for (int i = 0; i < dof; i++) for (int e = offs[i]; e < offs[i + 1]; e++) entry_of[(size_t)times[e] * dof + i] = e;
    int bad = 0, one_sided = 0;
    std::vector<int> mode_seen(entries, 0);
    for (int j = 0; j < st.njobs; j++) {
        const int col = st.job_col[j], kind = col < dof ? 0 : col < n ? 1 : 2, i = kind == 0 ? col : kind == 1 ? col - dof : col - n;
        const int e = entry_of[(size_t)st.job_t[j] * dof + i], mode = st.job_mode[j];
        if (e < 0) { bad++; continue; }
        const double *rec = (const double *)(slab.data() + (size_t)e * stride);
        const double *P = mode == 2 ? st.xnom + (size_t)st.job_nom[j] * n : st.xplus + (size_t)j * n;
        const double *Mn = mode == 1 ? st.xnom + (size_t)st.job_nom[j] * n : st.xminus + (size_t)j * n;
        // the record: (x+, x-) pairs, element by element (kpilqr_fd_kp_layout)
        bool okp = true, okm = true;
        for (int r = 0; r < n; r++) {
            okp = okp && std::memcmp(rec + ((size_t)kind * n + r) * 2, P + r, sizeof(double)) == 0;
            okm = okm && std::memcmp(rec + ((size_t)kind * n + r) * 2 + 1, Mn + r, sizeof(double)) == 0;
        }
        bad += !okp; bad += !okm;
        if (mode) { one_sided++; mode_seen[e] |= 1 << kind; }
    }
    for (int e = 0; e < entries; e++) if (*(const int *)(slab.data() + (size_t)e * stride + (size_t)6 * n * 8) != mode_seen[e]) bad++;
    stats[0] = st.njobs; stats[1] = one_sided; stats[2] = entries;
    return bad;
}

// Optimise() of a stand-in model on the GPU engine with the per-iteration decisions written out for the a9 parity test:
// trace [max_iter][24] = derivatives, lambda_in, backward_passes, lambda_exit, lambda_after_backward, old_cost, new_cost,
// best, accepted, converged, lambda_out, n_alpha, rollout_costs[6], predicted[6].  options: "+pruned" (GPU-ordered line
// search), "+unfused", "+adaptive_jerk" ... (key-point method name).  Returns iterations run or < 0.
int kpilqr_host_optimise(const char *model, int T, int max_iter, int min_iter, const char *options, const double *u_init /*[T][nu] or NULL*/,
                         double *cost_history, int cost_cap, double *trace, double *U_out)
{
    Model M;
    if (!make_model(model, 8, M)) return -1;
    const std::string opt = options ? options : "";
    for (const char *km : {"set_interval", "adaptive_jerk", "adaptive_accel", "velocity_change", "iterative_error"})
        if (opt.find(std::string("+") + km) != std::string::npos) M.mt->keypoint_method = km;
    *M.sim->master_reset_data = *M.sim->main_data;
    auto diff = std::make_shared<Differentiator>(M.mt, M.sim);
    iLQR_GPU ilqr(M.mt, M.sim, diff, T);
    if (!ilqr.ok()) return -2;
    if (opt.find("+unfused") != std::string::npos) ilqr.SetFused(false);
    if (opt.find("+pruned") != std::string::npos) ilqr.linesearch_mode = iLQR_GPU::LINESEARCH_PRUNED;
    const int m = M.mt->current_state_vector.num_ctrl;
    std::vector<MatrixXd> U0(T, MatrixXd(m, 1));
    if (u_init) for (int t = 0; t < T; t++) for (int i = 0; i < m; i++) U0[t](i) = u_init[(size_t)t * m + i];
    std::vector<MatrixXd> U = ilqr.Optimise(M.sim->main_data, U0, max_iter, min_iter, T);
    const int nh = (int)ilqr.cost_history.size();
    for (int i = 0; i < nh && i < cost_cap; i++) cost_history[i] = ilqr.cost_history[i];
    for (size_t it = 0; it < ilqr.trace.size() && (int)it < max_iter; it++) {
        const iLQR_GPU::IterationTrace &tr = ilqr.trace[it];
        double *w = trace + it * 24;
        for (int i = 0; i < 24; i++) w[i] = std::nan("");
        w[0] = tr.derivatives; w[1] = tr.lambda_in; w[2] = tr.backward_passes; w[3] = tr.lambda_exit; w[4] = tr.lambda_after_backward;
        w[5] = tr.old_cost; w[6] = tr.new_cost; w[7] = tr.best; w[8] = tr.accepted; w[9] = tr.converged; w[10] = tr.lambda_out;
        w[11] = (double)tr.rollout_costs.size();
        for (size_t a = 0; a < tr.rollout_costs.size() && a < 6; a++) { w[12 + a] = tr.rollout_costs[a]; w[18 + a] = tr.predicted[a]; }
    }
    if (U_out) for (int t = 0; t < T; t++) for (int i = 0; i < m; i++) U_out[(size_t)t * m + i] = U[t](i);
    return ilqr.num_iterations;
}

// Host FD-harness microbenchmark (no GPU involved): Acrobot, T saved states, every DoF
// a key-point at every step, `reps` derivative calls.  mode 0 = the reference's shape (threads created and
// joined per call, per-thread vectors merged); mode 1 = persistent pool writing in place into the staging
// arrays.  seconds = wall time of the reps; columns = FD columns produced; checksum[0] is order-independent
// (both modes must agree bit for bit), checksum[1] depends on the job order (mode 1 must reproduce it).
int kpilqr_host_fd_bench(int T, int reps, int mode, int fd_threads, double *seconds, long *columns, double *checksum)
{
    auto sim = std::make_shared<AcrobotSimulator>(0.01, fd_threads);
    auto mt = std::make_shared<AcrobotTranslator>(sim);
    sim->main_data->qpos[0] = 3.1415; sim->main_data->qpos[1] = 0.3;
    for (int t = 0; t < T; t++) {
        sim->main_data->ctrl[0] = 0.5 * ((t % 7) - 3);
        sim->AppendSystemStateToEnd(sim->main_data);
        sim->ForwardSimulator(sim->main_data);
    }
    Differentiator diff(mt, sim);
    std::vector<std::vector<int>> keypoints(T, std::vector<int>{0, 1});
    const int n = 4;
    FDJobs jobs;
    FDStaging st;
    diff.pool();                                         // pool construction is not part of a call
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; r++) {
        if (mode == 0) { jobs.clear(); diff.DynamicsDerivativesAtKeypoints(jobs, 0, keypoints, 1e-6); }
        else diff.DynamicsDerivativesBatch(st, 0, keypoints, 1e-6);
    }
    *seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    const int nj = mode == 0 ? jobs.njobs() : st.njobs;
    *columns = (long)nj * reps;
    unsigned long long h0 = 0;       // wrap-around integer sum of the bit patterns: exact and order-independent
    double c1 = 0.0;
    for (int j = 0; j < nj; j++) {
        const int t = mode == 0 ? jobs.job_t[j] : st.job_t[j], col = mode == 0 ? jobs.job_col[j] : st.job_col[j];
        const double *xp = mode == 0 ? &jobs.xplus[(size_t)j * n] : st.xplus + (size_t)j * n;
        const double *xm = mode == 0 ? &jobs.xminus[(size_t)j * n] : st.xminus + (size_t)j * n;
        double sj = 0.0;
        for (int i = 0; i < n; i++) {
            unsigned long long bp, bm;
            std::memcpy(&bp, &xp[i], 8); std::memcpy(&bm, &xm[i], 8);
            h0 += (bp * (unsigned long long)(i + 1) + bm * (unsigned long long)(i + 7)) * (unsigned long long)(1 + t * 31 + col);
            sj += (i + 1) * xp[i] - (i + 2) * xm[i];
        }
        c1 = c1 * 0.999 + sj * (j % 13 + 1);
    }
    if (checksum) { checksum[0] = (double)(h0 & ((1ull << 52) - 1)); checksum[1] = c1; }
    return diff.pool().size();
}

// ---- on-disk formats (FileHandler): thin entry points for the tests ----------------------------------------
// A [T][n*n], B [T][n*m] column-major per matrix (the ABI's host layout), X [T][n], U [T][m]
int kpilqr_host_save_trajec(const char *root_dir, int T, int dof, int m, const double *A, const double *B, const double *X, const double *U)
{
    const int n = 2 * dof;
    std::vector<MatrixXd> Am(T, MatrixXd(n, n)), Bm(T, MatrixXd(n, m)), Xm(T, MatrixXd(n, 1)), Um(T, MatrixXd(m, 1));
    for (int t = 0; t < T; t++) {
        std::memcpy(Am[t].data(), A + (size_t)t * n * n, sizeof(double) * n * n);
        std::memcpy(Bm[t].data(), B + (size_t)t * n * m, sizeof(double) * n * m);
        std::memcpy(Xm[t].data(), X + (size_t)t * n, sizeof(double) * n);
        std::memcpy(Um[t].data(), U + (size_t)t * m, sizeof(double) * m);
    }
    return FileHandler::SaveTrajecInformation(Am, Bm, Xm, Um, root_dir) ? 0 : -1;
}

int kpilqr_host_save_keypoints(const char *root_dir, int T, const int *offs, const int *cols)
{
    std::vector<std::vector<int>> kp(T);
    for (int t = 0; t < T; t++) kp[t].assign(cols + offs[t], cols + offs[t + 1]);
    return FileHandler::SaveKeypointsToFile(root_dir, kp) ? 0 : -1;
}

int kpilqr_host_task_file(const char *filename, int save, int n_start, int n_targets, double *start, double *targets)
{
    if (save) {
        return FileHandler::SaveTaskToFile(filename, std::vector<double>(start, start + n_start),
                                           std::vector<double>(targets, targets + n_targets)) ? 0 : -1;
    }
    std::vector<double> s, g;
    if (!FileHandler::LoadTaskFromFile(filename, n_start, n_targets, s, g)) return -1;
    std::memcpy(start, s.data(), sizeof(double) * n_start);
    std::memcpy(targets, g.data(), sizeof(double) * n_targets);
    return 0;
}

// rows: [nrows][5] = cost_reduction, opt_time_ms, num_iterations, avg_num_dofs, avg_percent_derivs; timings
// [nrows][3][niter] = per-iteration derivs / BP / FP times
int kpilqr_host_save_summary(const char *filename, int nrows, int niter, const double *rows, const double *timings)
{
    std::vector<FileHandler::SummaryRow> v(nrows);
    for (int i = 0; i < nrows; i++) {
        v[i].cost_reduction = rows[i * 5]; v[i].optimisation_time_ms = rows[i * 5 + 1]; v[i].num_iterations = (int)rows[i * 5 + 2];
        v[i].avg_num_dofs = rows[i * 5 + 3]; v[i].avg_percent_derivs = rows[i * 5 + 4];
        const double *t = timings + (size_t)i * 3 * niter;
        v[i].time_derivs_ms.assign(t, t + niter); v[i].time_bp_ms.assign(t + niter, t + 2 * niter); v[i].time_fp_ms.assign(t + 2 * niter, t + 3 * niter);
    }
    return FileHandler::SaveSummary(filename, v) ? 0 : -1;
}

}  // extern "C"
