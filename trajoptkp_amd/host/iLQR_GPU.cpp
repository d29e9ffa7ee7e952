// iLQR_GPU.cpp -- host control flow of one optimisation, line-cited against src/Optimiser/iLQR.cpp.
#include "iLQR_GPU.h"
#include "SimData.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using clk = std::chrono::high_resolution_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

void iLQR_GPU::fatal(const char *what, int rc)
{
    // fatal configuration / runtime errors end the program in the reference too (std::cerr + exit(1))
    std::fprintf(stderr, "iLQR_GPU: %s failed (%d): %s\n", what, rc, kpilqr_strerror(ctx));
    std::exit(1);
}

iLQR_GPU::iLQR_GPU(std::shared_ptr<ModelTranslator> mt, std::shared_ptr<PhysicsSimulator> sim,
                   std::shared_ptr<Differentiator> diff, int horizon, int device_)
    : Optimiser(mt, sim, diff), device(device_)
{
    // saved state list with horizon+1 entries (iLQR.cpp:9-26)
    for (int i = 0; i <= horizon; i++)
        if (!MuJoCo_helper->CheckIfDataIndexExists(i)) MuJoCo_helper->AppendSystemStateToEnd(MuJoCo_helper->main_data);
    for (int i = 1; i <= num_parallel_rollouts; i++) { const double l = (double)i / num_parallel_rollouts; alphas.push_back(l * l); }   // :466-470
    Resize(mt->current_state_vector.dof, mt->current_state_vector.num_ctrl, horizon);
}

void iLQR_GPU::free_pinned()
{
    staging.free_all();
    double **all[] = {&host_r, &host_rx, &host_ru, &host_unom, &host_K, &host_k};
    for (double **p : all) { if (*p && ctx) kpilqr_host_free(ctx, *p); *p = nullptr; }
    if (kp_slab && ctx) kpilqr_host_free(ctx, kp_slab);
    kp_slab = nullptr; kp_slab_bytes = 0;
    if (kp_cols && ctx) kpilqr_host_free(ctx, kp_cols);
    kp_cols = nullptr; kp_cols_count = 0;
}

iLQR_GPU::~iLQR_GPU()
{
    if (ctx) { kpilqr_sync(ctx); free_pinned(); kpilqr_destroy(ctx); }
}

void iLQR_GPU::Resize(int new_num_dofs, int new_num_ctrl, int new_horizon)
{
    if (new_num_dofs == dof && new_num_ctrl == num_ctrl && new_horizon == horizon_length && ctx && !recreate_ctx) return;
    recreate_ctx = false;
    dof = new_num_dofs; num_ctrl = new_num_ctrl; horizon_length = new_horizon;
    const int n = 2 * dof, m = num_ctrl, T = horizon_length, nr = (int)activeModelTranslator->residual_list.size();
    if (ctx) { kpilqr_sync(ctx); free_pinned(); kpilqr_destroy(ctx); ctx = nullptr; }
    // the A filters act on the materialised sequence, so a filtering task runs the materialising pipeline
    kpilqr_dims d = {dof, m, T, nr, 1, num_parallel_rollouts, device, (use_fused && filteringMethod == "none") ? KPILQR_FLAG_FUSED : 0};
    // the structs of include/kpilqr.h this file was compiled against are those of ONE major version of the library
    if (kpilqr_version() / 100 != KPILQR_VERSION / 100) { std::fprintf(stderr, "iLQR_GPU: libkpilqr.so is version %d, built against %d\n", kpilqr_version(), KPILQR_VERSION); std::exit(1); }
    const int rc = kpilqr_create(&d, nullptr, &ctx);
    if (rc != KPILQR_OK) { last_error = kpilqr_strerror(nullptr); ctx = nullptr; std::fprintf(stderr, "iLQR_GPU: %s\n", last_error.c_str()); std::exit(1); }
    fused_active = std::string(kpilqr_backward_variant(ctx)).find("fused") != std::string::npos;
    // every host <-> device payload lives in pinned memory owned by the context
    staging.alloc = [this](size_t bytes) { void *p = nullptr; if (kpilqr_host_alloc(ctx, bytes, &p)) fatal("kpilqr_host_alloc", -4); return p; };
    staging.release = [this](void *p) { if (ctx) kpilqr_host_free(ctx, p); };
    staging.layout = [this](int nj, int nn, size_t off[9]) {
        kpilqr_fd_layout l;
        if (kpilqr_fd_slab_layout(ctx, nj, nn, &l)) fatal("kpilqr_fd_slab_layout", -1);
        const size_t o[9] = {l.xplus, l.xminus, l.xnom, l.job_b, l.job_t, l.job_col, l.job_nom, l.job_mode, l.bytes};
        for (int i = 0; i < 9; i++) off[i] = o[i];
    };
    auto pinned = [&](size_t count) { void *p = nullptr; if (kpilqr_host_alloc(ctx, std::max<size_t>(count, 1) * sizeof(double), &p)) fatal("kpilqr_host_alloc", -4); std::fill((double *)p, (double *)p + count, 0.0); return (double *)p; };
    host_r = pinned((size_t)(T + 1) * nr); host_rx = pinned((size_t)(T + 1) * nr * n); host_ru = pinned((size_t)(T + 1) * nr * m);
    host_unom = pinned((size_t)T * m); host_K = pinned((size_t)T * n * m); host_k = pinned((size_t)T * m);
    U_old.assign(T, MatrixXd(m, 1)); X_old.assign(T + 1, MatrixXd(n, 1)); X_new.assign(T + 1, MatrixXd(n, 1));
    residuals.assign(T + 1, MatrixXd(nr, 1));
    K.assign(T, MatrixXd(m, n)); k.assign(T, MatrixXd(m, 1));
    w_run.clear(); w_term.clear();
    for (const residual &r : activeModelTranslator->residual_list) { w_run.push_back(r.weight); w_term.push_back(r.weight_terminal); }
    const MatrixXd lim = activeModelTranslator->ReturnControlLimits(activeModelTranslator->current_state_vector);
    ctrl_lim.assign(lim.data(), lim.data() + 2 * m);
    // a task with ONE residual Jacobian (affine residuals): host_rx / host_ru hold the pair, uploaded once per context
    const_jacobians = activeModelTranslator->ConstantResidualJacobians(host_rx, host_ru);
    const_jacobians_resident = false;
    keypoint_generator->Resize(dof, m, T);
    while ((int)MuJoCo_helper->saved_systems_state_list.size() <= T) MuJoCo_helper->AppendSystemStateToEnd(MuJoCo_helper->main_data);
}

// iLQR.cpp:202-254
double iLQR_GPU::RolloutTrajectory(SimData *d, bool save_states, std::vector<MatrixXd> initial_controls)
{
    const stateVectorList &sv = activeModelTranslator->full_state_vector;
    double cost = 0.0;
    MuJoCo_helper->CopySystemState(MuJoCo_helper->main_data, d);
    X_old[0] = activeModelTranslator->ReturnStateVector(MuJoCo_helper->main_data, sv);
    MuJoCo_helper->CopySystemState(MuJoCo_helper->saved_systems_state_list[0], MuJoCo_helper->main_data);
    for (int i = 0; i < horizon_length; i++) {
        activeModelTranslator->SetControlVector(initial_controls[i], MuJoCo_helper->main_data, sv);
        MuJoCo_helper->ForwardSimulator(MuJoCo_helper->main_data);
        activeModelTranslator->Residuals(MuJoCo_helper->main_data, residuals[i]);
        cost += activeModelTranslator->CostFunction(residuals[i], sv, i == horizon_length - 1);
        if (save_states) {
            X_old[i + 1] = activeModelTranslator->ReturnStateVector(MuJoCo_helper->main_data, sv);
            U_old[i] = activeModelTranslator->ReturnControlVector(MuJoCo_helper->main_data, sv);
            MuJoCo_helper->CopySystemState(MuJoCo_helper->saved_systems_state_list[i + 1], MuJoCo_helper->main_data);
        }
    }
    initial_cost = cost;
    cost_history.push_back(cost);
    return cost;
}

// iLQR.cpp:269-410
std::vector<MatrixXd> iLQR_GPU::Optimise(SimData *d, std::vector<MatrixXd> initial_controls, int max_iterations,
                                         int min_iterations, int horizon)
{
    const auto opt_start = clk::now();
    Resize(activeModelTranslator->current_state_vector.dof, activeModelTranslator->current_state_vector.num_ctrl, horizon);
    cost_history.clear(); time_get_derivs_ms.clear(); time_backwards_pass_ms.clear(); time_forwardsPass_ms.clear(); trace.clear();
    percentage_derivs_per_iteration.clear();
    num_iterations = 0;
    old_cost = RolloutTrajectory(d, true, initial_controls);
    initial_cost = old_cost; new_cost = old_cost;
    MuJoCo_helper->CopySystemState(MuJoCo_helper->main_data, MuJoCo_helper->saved_systems_state_list[0]);
    cost_reduced_last_iter = true;
    for (int i = 0; i < max_iterations; i++) {
        num_iterations++;
        bool lambda_exit = false, converged = false;
        Iteration(i, converged, lambda_exit);
        if (converged && (i >= min_iterations)) break;
        if (lambda_exit) break;
    }
    cost_reduction = 1 - (new_cost / initial_cost);
    opt_time_ms = ms_since(opt_start);
    auto mean = [](const std::vector<double> &v) { double s = 0; for (double x : v) s += x; return v.empty() ? 0.0 : s / v.size(); };
    avg_time_get_derivs_ms = mean(time_get_derivs_ms); avg_time_backwards_pass_ms = mean(time_backwards_pass_ms);
    avg_time_forwards_pass_ms = mean(time_forwardsPass_ms); avg_percent_derivs = mean(percentage_derivs_per_iteration);
    MuJoCo_helper->CopySystemState(MuJoCo_helper->main_data, MuJoCo_helper->saved_systems_state_list[0]);
    return U_old;
}

// Optimiser::GenerateDerivatives (Optimiser.cpp:80-169): key-points, FD at key-points (host) -> GPU
// differencing + interpolation; residual FD (host) -> GPU Gauss-Newton cost derivatives.
void iLQR_GPU::GenerateDerivatives()
{
    const stateVectorList &sv = activeModelTranslator->current_state_vector;
    const int n = 2 * dof, T = horizon_length, nr = (int)w_run.size();
    const double eps = 1e-6;                                     // Optimiser.cpp:319-321
    keypoint_generator->ResetCache();
    KeypointGenerator::ColumnFD col_fd = [&](int t, int i, double *cp, double *cv) {
        FDJobs one;
        activeDifferentiator->DynamicsDerivatives(one, 0, std::vector<int>(1, i), t, 0, true, eps);
        for (int j = 0; j < one.njobs(); j++) {
            double *dst = one.job_col[j] == i ? cp : one.job_col[j] == i + dof ? cv : nullptr;
            if (dst) for (int r = 0; r < n; r++) dst[r] = (one.xplus[(size_t)j * n + r] - one.xminus[(size_t)j * n + r]) / (2 * eps);
        }
    };
    std::vector<MatrixXd> Xk(X_old.begin(), X_old.begin() + T);
    keypoint_generator->GenerateKeyPoints(Xk, MuJoCo_helper->ReturnModelTimeStep(), col_fd);
    std::vector<int> offs, times;
    keypoint_generator->PerDofCSR(offs, times);
    int rc = kpilqr_set_keypoints(ctx, offs.data(), times.data());
    if (rc) fatal("kpilqr_set_keypoints", rc);
    if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);        // the previous upload has left the staging slab
    if (fused_active) {
        // Fused sweeps: the payload goes up KEY-POINT ORDERED -- the FD workers write every perturbed next state straight into
        // its slot of the entry records (kpilqr_fd_kp_layout), one DMA, no job lists; the sweeps read the records (or the
        // column store differenced from them) directly
        const int entries = offs[dof];
        kpilqr_fdkp_layout lay;
        if ((rc = kpilqr_fd_kp_layout(ctx, entries, &lay))) fatal("kpilqr_fd_kp_layout", rc);
        if (lay.bytes > kp_slab_bytes) {
            if (kp_slab) kpilqr_host_free(ctx, kp_slab);
            kp_slab_bytes = lay.bytes + lay.bytes / 4 + 4096;
            if ((rc = kpilqr_host_alloc(ctx, kp_slab_bytes, (void **)&kp_slab))) fatal("kpilqr_host_alloc", rc);
        }
        activeDifferentiator->DynamicsDerivativesKp(kp_slab, lay.entry_stride, 0, offs, times, keypoint_generator->keypoints, eps);
        if (host_differencing) {
            // a2 on the host: (x+ - x-) / (2 eps), or / eps for a one-sided job (bit `kind` of the entry's mode word), then the
            // columns go up as they are
            const size_t per = (size_t)3 * n, want = (size_t)entries * per;
            if (want > kp_cols_count) {
                if (kp_cols) kpilqr_host_free(ctx, kp_cols);
                kp_cols_count = want + want / 4 + 64;
                if ((rc = kpilqr_host_alloc(ctx, kp_cols_count * sizeof(double), (void **)&kp_cols))) fatal("kpilqr_host_alloc", rc);
            }
            for (int e = 0; e < entries; e++) {
                const char *rec = kp_slab + (size_t)e * lay.entry_stride;
                const char *xp = rec + lay.xplus, *xm = rec + lay.xminus;
                const int mode = *(const int *)(rec + lay.mode);
                for (int kind = 0; kind < 3; kind++) {
                    const double den = ((mode >> kind) & 1) ? eps : 2 * eps;
                    for (int r = 0; r < n; r++) {
                        const size_t el = ((size_t)kind * n + r) * lay.elem_stride;
                        kp_cols[(size_t)e * per + (size_t)kind * n + r] = (*(const double *)(xp + el) - *(const double *)(xm + el)) / den;
                    }
                }
            }
            if ((rc = kpilqr_upload_kp_columns(ctx, kp_cols, entries))) fatal("kpilqr_upload_kp_columns", rc);
        } else if ((rc = kpilqr_upload_fd_kp(ctx, kp_slab, entries, eps))) fatal("kpilqr_upload_fd_kp", rc);
    } else {
    // FD at the key-points on the persistent pool, straight into ONE pinned slab (jobs, nominal rows):
    // the upload is a single DMA and nothing on the host walks the jobs afterwards
    {
        int jobs = 0, kps = 0;
        activeDifferentiator->CountJobs(keypoint_generator->keypoints, jobs, kps);
        staging.plan(jobs, kps, n);
        activeDifferentiator->DynamicsDerivativesPlanned(staging, 0, keypoint_generator->keypoints, eps);
    }
    // the slab's array offsets were computed from the PLANNED totals: an under-filled plan would make the device read
    // x-, xnom and the job arrays at the wrong offsets
    if (!staging.complete()) { std::fprintf(stderr, "FD staging: %d of %d jobs, %d of %d nominal rows filled\n", staging.njobs, staging.plan_jobs, staging.nnom, staging.plan_noms); std::exit(1); }
    rc = kpilqr_upload_fd_slab(ctx, staging.slab, staging.njobs, staging.nnom, eps);
    if (rc) fatal("kpilqr_upload_fd_slab", rc);
    }
    if ((rc = kpilqr_fd_difference(ctx))) fatal("kpilqr_fd_difference", rc);
    if (!fused_active && (rc = kpilqr_interpolate(ctx))) fatal("kpilqr_interpolate", rc);
    if (filteringMethod != "none") {                              // Optimiser.cpp:105-107
        if (fused_active) { recreate_ctx = true; std::fprintf(stderr, "iLQR_GPU: filtering set after the context was created; call Resize first\n"); std::exit(1); }
        const bool lp = filteringMethod == "low_pass";
        rc = kpilqr_filter_dynamics(ctx, filteringMethod.c_str(), lp ? &lowPassACoefficient : FIRCoefficients.data(), lp ? 1 : (int)FIRCoefficients.size());
        if (rc) fatal("kpilqr_filter_dynamics", rc);
    }
    // residuals and their Jacobians at every step (Optimiser::ComputeResidualDerivatives, :217-236)
    for (int t = 0; t <= T; t++)
        for (int i = 0; i < nr; i++) host_r[(size_t)t * nr + i] = residuals[t](i);
    if (const_jacobians) {
        // Reaching.cpp:43-54 and every other task with affine residuals: no a5 at all -- the one pair goes up once per context
        // (r_u = 0 as NULL: the sweeps then leave the control-residual products out), the residuals every linearisation
        if (!const_jacobians_resident) {
            bool ru_zero = true;
            for (int i = 0; i < nr * num_ctrl; i++) ru_zero = ru_zero && host_ru[i] == 0.0;
            if ((rc = kpilqr_upload_residual_jacobians_const(ctx, host_rx, ru_zero ? nullptr : host_ru))) fatal("kpilqr_upload_residual_jacobians_const", rc);
            const_jacobians_resident = true;
            constant_jacobian_uploads++;
        }
        rc = kpilqr_upload_residuals(ctx, host_r, nullptr, nullptr, w_run.data(), w_term.data());
    } else {
        activeDifferentiator->ResidualDerivativesAll(host_rx, host_ru, T, eps);
        rc = kpilqr_upload_residuals(ctx, host_r, host_rx, host_ru, w_run.data(), w_term.data());
        per_step_jacobian_uploads++;
    }
    if (rc) fatal("kpilqr_upload_residuals", rc);
    if (!fused_active && (rc = kpilqr_cost_derivs(ctx))) fatal("kpilqr_cost_derivs", rc);
    double pct = 0.0;
    for (int i = 0; i < sv.dof; i++) pct += keypoint_generator->last_percentages[i];
    percentage_derivs_per_iteration.push_back(pct / sv.dof);
}

// iLQR.cpp:535-634 on the GPU; false = Q_uu + lambda I failed the PD test
bool iLQR_GPU::BackwardsPassQuuRegularisation()
{
    int status = 0;
    int rc = kpilqr_backward(ctx, &lambda, 100, &status, &delta_J);
    if (rc < 0) fatal("kpilqr_backward", rc);
    if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
    return status == 0;
}

// iLQR.cpp:636-657
bool iLQR_GPU::UpdateLambda(bool valid_backwards_pass)
{
    bool lambda_exit = false;
    if (!valid_backwards_pass) lambda *= lambda_factor; else lambda /= lambda_factor;
    if (lambda > max_lambda) { lambda = max_lambda; lambda_exit = true; }
    if (lambda < min_lambda) lambda = min_lambda;
    return lambda_exit;
}

// iLQR::ForwardsPassParallel (iLQR.cpp:824-934): one closed-loop simulator rollout with step size alpha on
// fd_data[thread_id]; returns its cost and the controls it applied.
double iLQR_GPU::ForwardsPassParallel(int thread_id, double alpha, std::vector<MatrixXd> &U_out)
{
    const stateVectorList &sv = activeModelTranslator->current_state_vector;
    const int n = 2 * dof, m = num_ctrl;
    const bool tangent = sv.dof != sv.dof_quat;
    SimData *d = MuJoCo_helper->fd_data[thread_id];
    MuJoCo_helper->CopySystemState(d, MuJoCo_helper->saved_systems_state_list[0]);
    double cost = 0.0;
    MatrixXd r((int)w_run.size(), 1), fbk(n, 1);
    std::vector<double> vel_diff(tangent ? MuJoCo_helper->nv() : 0);
    for (int t = 0; t < horizon_length; t++) {
        const MatrixXd x = activeModelTranslator->ReturnStateVector(d, sv);
        if (!tangent) {
            for (int p = 0; p < n; p++) fbk(p) = x(p) - X_old[t](p);                     // :853
        } else {
            // position differences in the tangent space: mj_differentiatePos(vel_diff, 1.0, state_old, state_new) (:857-868)
            MuJoCo_helper->DifferentiatePos(vel_diff.data(), 1.0, MuJoCo_helper->saved_systems_state_list[t], d);
            for (int j = 0; j < dof; j++) fbk(j) = vel_diff[activeModelTranslator->StateIndexToQposIndex(j, sv)];
            for (int j = 0; j < dof; j++) fbk(j + dof) = x(dof + j) - X_old[t](dof + j);  // :871-873
        }
        MatrixXd u(m, 1);
        for (int i = 0; i < m; i++) {
            double fb = 0.0;
            for (int p = 0; p < n; p++) fb += K[t](i, p) * fbk(p);                      // :876
            double v = U_old[t](i) + (alpha * k[t](i)) + fb;                            // :879
            if (v > ctrl_lim[2 * i + 1]) v = ctrl_lim[2 * i + 1];                       // :883-889
            if (v < ctrl_lim[2 * i]) v = ctrl_lim[2 * i];
            u(i) = v;
        }
        activeModelTranslator->SetControlVector(u, d, sv);
        activeModelTranslator->Residuals(d, r);
        cost += activeModelTranslator->CostFunction(r, sv, t == horizon_length - 1);   // :899-912
        U_out[t] = u;
        MuJoCo_helper->ForwardSimulator(d);
    }
    return cost;
}

// iLQR.cpp:412-531
void iLQR_GPU::Iteration(int iteration_num, bool &converged, bool &lambda_exit)
{
    (void)iteration_num;
    const int n = 2 * dof, m = num_ctrl, T = horizon_length, na = (int)alphas.size();
    IterationTrace tr;
    tr.derivatives = cost_reduced_last_iter;
    auto t0 = clk::now();
    if (cost_reduced_last_iter) GenerateDerivatives();                                  // STEP 1 (:419)
    time_get_derivs_ms.push_back(ms_since(t0));

    t0 = clk::now();
    bool valid = false;                                                                 // STEP 2 (:435-442)
    tr.lambda_in = lambda;
    while (!valid) {
        valid = BackwardsPassQuuRegularisation();
        tr.backward_passes++;
        lambda_exit = UpdateLambda(valid);
        if (lambda_exit) break;
    }
    tr.lambda_after_backward = lambda; tr.lambda_exit = lambda_exit;
    time_backwards_pass_ms.push_back(ms_since(t0));
    if (lambda_exit) { tr.lambda_out = lambda; tr.old_cost = tr.new_cost = old_cost; trace.push_back(tr); return; }

    t0 = clk::now();                                                                    // STEP 3
    for (int t = 0; t < T; t++) for (int i = 0; i < m; i++) host_unom[(size_t)t * m + i] = U_old[t](i);
    int rc = kpilqr_upload_nominal(ctx, host_unom, ctrl_lim.data());
    if (rc) fatal("kpilqr_upload_nominal", rc);
    std::vector<double> pred(na);
    if ((rc = kpilqr_forward_linear(ctx, alphas.data(), pred.data(), nullptr))) fatal("kpilqr_forward_linear", rc);
    if ((rc = kpilqr_download_gains(ctx, host_K, host_k))) fatal("kpilqr_download_gains", rc);
    if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
    for (int t = 0; t < T; t++) {
        for (int c = 0; c < n; c++) for (int r = 0; r < m; r++) K[t](r, c) = host_K[((size_t)t * n + c) * m + r];
        for (int r = 0; r < m; r++) k[t](r) = host_k[(size_t)t * m + r];
    }
    tr.predicted = pred;
    tr.rollout_costs.assign(na, std::nan(""));
    std::vector<std::vector<MatrixXd>> U_try(na, std::vector<MatrixXd>(T, MatrixXd(m, 1)));
    int best = -1;
    new_cost = old_cost;
    if (linesearch_mode == LINESEARCH_REFERENCE) {
        // all alphas rolled out in parallel, one fd_data slot per pool worker (:478-487: std::async per alpha)
        activeDifferentiator->pool().parallel_for(na, [&](int i, int tid) { tr.rollout_costs[i] = ForwardsPassParallel(tid, alphas[i], U_try[i]); });
        best = (int)(std::min_element(tr.rollout_costs.begin(), tr.rollout_costs.end()) - tr.rollout_costs.begin());   // :490
        if (tr.rollout_costs[best] < old_cost) new_cost = tr.rollout_costs[best];        // :494-502
    } else {
        // candidates in order of the GPU's predicted cost; each is confirmed with one simulator rollout until one improves
        std::vector<int> order(na);
        for (int i = 0; i < na; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return pred[a] < pred[b]; });
        for (int idx : order) {
            tr.rollout_costs[idx] = ForwardsPassParallel(0, alphas[idx], U_try[idx]);
            if (tr.rollout_costs[idx] < old_cost) { new_cost = tr.rollout_costs[idx]; best = idx; break; }
        }
    }
    time_forwardsPass_ms.push_back(ms_since(t0));
    tr.best = best; tr.old_cost = old_cost; tr.new_cost = new_cost;

    converged = CheckForConvergence(old_cost, new_cost);                                // STEP 4 (:515)
    tr.converged = converged;
    if (new_cost < old_cost) {
        // SaveBestRollout + UpdateNominal (Optimiser.cpp:441-469, iLQR.cpp:936-948): the winning controls are replayed on
        // main_data, which refreshes the saved states, the nominal trajectory and the residuals in one pass
        const stateVectorList &sv = activeModelTranslator->current_state_vector;
        SimData *d = MuJoCo_helper->main_data;
        MuJoCo_helper->CopySystemState(d, MuJoCo_helper->saved_systems_state_list[0]);
        for (int t = 0; t < T; t++) {
            U_old[t] = U_try[best][t];
            activeModelTranslator->SetControlVector(U_old[t], d, sv);
            MuJoCo_helper->ForwardSimulator(d);
            activeModelTranslator->Residuals(d, residuals[t]);
            X_old[t + 1] = activeModelTranslator->ReturnStateVector(d, sv);
            MuJoCo_helper->CopySystemState(MuJoCo_helper->saved_systems_state_list[t + 1], d);
        }
        old_cost = new_cost;
        cost_reduced_last_iter = true;
        tr.accepted = true;
    } else {
        cost_reduced_last_iter = false;
        lambda *= lambda_factor; lambda *= lambda_factor;                               // :525-527
        if (lambda > max_lambda) lambda = max_lambda;
    }
    tr.lambda_out = lambda;
    trace.push_back(tr);
    cost_history.push_back(new_cost);
}

void iLQR_GPU::DownloadDerivatives(std::vector<MatrixXd> &A, std::vector<MatrixXd> &B)
{
    const int n = 2 * dof, m = num_ctrl, T = horizon_length;
    std::vector<double> a((size_t)T * n * n), b((size_t)T * n * m);
    int rc = fused_active ? kpilqr_interpolate(ctx) : 0;          // the fused sweeps do not materialise A, B
    if (rc) fatal("kpilqr_interpolate", rc);
    rc = kpilqr_get_AB(ctx, a.data(), b.data());
    if (rc) fatal("kpilqr_get_AB", rc);
    if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
    A.assign(T, MatrixXd(n, n)); B.assign(T, MatrixXd(n, m));
    for (int t = 0; t < T; t++) {
        std::copy(a.begin() + (size_t)t * n * n, a.begin() + (size_t)(t + 1) * n * n, A[t].data());
        std::copy(b.begin() + (size_t)t * n * m, b.begin() + (size_t)(t + 1) * n * m, B[t].data());
    }
}
