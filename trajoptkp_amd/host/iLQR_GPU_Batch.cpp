// iLQR_GPU_Batch.cpp -- see the header.  Line citations are to src/Optimiser/iLQR.cpp unless noted.
#include "iLQR_GPU_Batch.h"
#include "SimData.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

void iLQR_GPU_Batch::fatal(const char *what, int rc)
{
    std::fprintf(stderr, "iLQR_GPU_Batch: %s failed (%d): %s\n", what, rc, kpilqr_strerror(ctx));
    std::exit(1);
}

iLQR_GPU_Batch::iLQR_GPU_Batch(std::vector<Problem> problems, int horizon, int device_, bool fused)
    : P(std::move(problems)), B((int)P.size()), T(horizon), device(device_)
{
    const stateVectorList &sv = P[0].model_translator->current_state_vector;
    dof = sv.dof; num_ctrl = sv.num_ctrl; nr = (int)P[0].model_translator->residual_list.size();
    const int n = 2 * dof, m = num_ctrl;
    kpilqr_dims d = {dof, m, T, nr, B, num_parallel_rollouts, device, fused ? KPILQR_FLAG_FUSED : 0};
    if (kpilqr_version() / 100 != KPILQR_VERSION / 100) { std::fprintf(stderr, "iLQR_GPU_Batch: libkpilqr.so is version %d, built against %d\n", kpilqr_version(), KPILQR_VERSION); ctx = nullptr; return; }
    if (kpilqr_create(&d, nullptr, &ctx) != KPILQR_OK) { std::fprintf(stderr, "iLQR_GPU_Batch: %s\n", kpilqr_strerror(nullptr)); ctx = nullptr; return; }
    fused_active = std::string(kpilqr_backward_variant(ctx)).find("fused") != std::string::npos;
    for (int i = 1; i <= num_parallel_rollouts; i++) { const double l = (double)i / num_parallel_rollouts; alphas.push_back(l * l); }   // :466-470
    for (const residual &r : P[0].model_translator->residual_list) { w_run.push_back(r.weight); w_term.push_back(r.weight_terminal); }
    const MatrixXd lim = P[0].model_translator->ReturnControlLimits(sv);
    ctrl_lim.assign(lim.data(), lim.data() + 2 * m);
    staging.alloc = [this](size_t bytes) { void *p = nullptr; if (kpilqr_host_alloc(ctx, bytes, &p)) fatal("kpilqr_host_alloc", -4); return p; };
    staging.release = [this](void *p) { if (ctx) kpilqr_host_free(ctx, p); };
    staging.layout = [this](int nj, int nn, size_t off[9]) {
        kpilqr_fd_layout l;
        if (kpilqr_fd_slab_layout(ctx, nj, nn, &l)) fatal("kpilqr_fd_slab_layout", -1);
        const size_t o[9] = {l.xplus, l.xminus, l.xnom, l.job_b, l.job_t, l.job_col, l.job_nom, l.job_mode, l.bytes};
        for (int i = 0; i < 9; i++) off[i] = o[i];
    };
    auto pinned = [&](size_t count) { void *p = nullptr; if (kpilqr_host_alloc(ctx, std::max<size_t>(count, 1) * sizeof(double), &p)) fatal("kpilqr_host_alloc", -4); std::fill((double *)p, (double *)p + count, 0.0); return (double *)p; };
    host_r = pinned((size_t)B * (T + 1) * nr); host_rx = pinned((size_t)B * (T + 1) * nr * n); host_ru = pinned((size_t)B * (T + 1) * nr * m);
    host_unom = pinned((size_t)B * T * m); host_K = pinned((size_t)B * T * n * m); host_k = pinned((size_t)B * T * m);
    const_rx.assign((size_t)nr * n, 0.0); const_ru.assign((size_t)nr * m, 0.0);
    const_jacobians = P[0].model_translator->ConstantResidualJacobians(const_rx.data(), const_ru.data());     // (one task, B initial conditions)
    S.resize(B);
    lambda.assign(B, 0.1); cost_history.assign(B, {}); num_iterations.assign(B, 0);
    K.assign(B, std::vector<MatrixXd>(T, MatrixXd(m, n))); k.assign(B, std::vector<MatrixXd>(T, MatrixXd(m, 1)));
    for (int b = 0; b < B; b++) {
        Traj &s = S[b];
        ModelTranslator &mt = *P[b].model_translator;
        keypoint_method km;
        km.name = mt.keypoint_method; km.auto_adjust = mt.auto_adjust; km.min_N = mt.min_N; km.max_N = mt.max_N;
        km.jerk_thresholds = mt.jerk_thresholds; km.accell_thresholds = mt.jerk_thresholds;
        km.iterative_error_threshold = mt.iterative_error_threshold; km.velocity_change_thresholds = mt.velocity_change_thresholds;
        s.kpgen = std::make_shared<KeypointGenerator>(dof, T);
        s.kpgen->SetKeypointMethod(km);
        s.kpgen->Resize(dof, m, T);
        s.U_old.assign(T, MatrixXd(m, 1)); s.X_old.assign(T + 1, MatrixXd(n, 1)); s.residuals.assign(T + 1, MatrixXd(nr, 1));
        PhysicsSimulator &sim = *P[b].MuJoCo_helper;
        while ((int)sim.saved_systems_state_list.size() <= T) sim.AppendSystemStateToEnd(sim.main_data);
    }
}

iLQR_GPU_Batch::~iLQR_GPU_Batch()
{
    if (!ctx) return;
    kpilqr_sync(ctx);
    staging.free_all();
    if (kp_slab) kpilqr_host_free(ctx, kp_slab);
    kp_slab = nullptr;
    double **all[] = {&host_r, &host_rx, &host_ru, &host_unom, &host_K, &host_k};
    for (double **p : all) { if (*p) kpilqr_host_free(ctx, *p); *p = nullptr; }
    kpilqr_destroy(ctx);
}

// iLQR::RolloutTrajectory (:202-254) for trajectory b, states saved
double iLQR_GPU_Batch::Rollout(int b, SimData *start, const std::vector<MatrixXd> &controls)
{
    ModelTranslator &mt = *P[b].model_translator;
    PhysicsSimulator &sim = *P[b].MuJoCo_helper;
    const stateVectorList &sv = mt.full_state_vector;
    Traj &s = S[b];
    double cost = 0.0;
    if (start != sim.main_data) sim.CopySystemState(sim.main_data, start);
    s.X_old[0] = mt.ReturnStateVector(sim.main_data, sv);
    sim.CopySystemState(sim.saved_systems_state_list[0], sim.main_data);
    for (int i = 0; i < T; i++) {
        mt.SetControlVector(controls[i], sim.main_data, sv);
        sim.ForwardSimulator(sim.main_data);
        mt.Residuals(sim.main_data, s.residuals[i]);
        cost += mt.CostFunction(s.residuals[i], sv, i == T - 1);
        s.X_old[i + 1] = mt.ReturnStateVector(sim.main_data, sv);
        s.U_old[i] = mt.ReturnControlVector(sim.main_data, sv);
        sim.CopySystemState(sim.saved_systems_state_list[i + 1], sim.main_data);
    }
    return cost;
}

// one closed-loop rollout with alpha on fd_data[tid] (ForwardsPassParallel :824-934)
double iLQR_GPU_Batch::ConfirmRollout(int b, int tid, double alpha, std::vector<MatrixXd> &U_out)
{
    ModelTranslator &mt = *P[b].model_translator;
    PhysicsSimulator &sim = *P[b].MuJoCo_helper;
    const stateVectorList &sv = mt.current_state_vector;
    const int n = 2 * dof, m = num_ctrl;
    const bool tangent = sv.dof != sv.dof_quat;
    Traj &s = S[b];
    SimData *d = sim.fd_data[tid];
    sim.CopySystemState(d, sim.saved_systems_state_list[0]);
    double cost = 0.0;
    MatrixXd r(nr, 1), fbk(n, 1);
    std::vector<double> vel_diff(tangent ? sim.nv() : 0);
    for (int t = 0; t < T; t++) {
        const MatrixXd x = mt.ReturnStateVector(d, sv);
        if (!tangent) {
            for (int p = 0; p < n; p++) fbk(p) = x(p) - s.X_old[t](p);                          // :853
        } else {                                                                               // :857-873
            sim.DifferentiatePos(vel_diff.data(), 1.0, sim.saved_systems_state_list[t], d);
            for (int j = 0; j < dof; j++) fbk(j) = vel_diff[mt.StateIndexToQposIndex(j, sv)];
            for (int j = 0; j < dof; j++) fbk(j + dof) = x(dof + j) - s.X_old[t](dof + j);
        }
        MatrixXd u(m, 1);
        for (int i = 0; i < m; i++) {
            double fb = 0.0;
            for (int p = 0; p < n; p++) fb += K[b][t](i, p) * fbk(p);                           // :876
            double v = s.U_old[t](i) + (alpha * k[b][t](i)) + fb;                              // :879
            if (v > ctrl_lim[2 * i + 1]) v = ctrl_lim[2 * i + 1];                              // :883-889
            if (v < ctrl_lim[2 * i]) v = ctrl_lim[2 * i];
            u(i) = v;
        }
        mt.SetControlVector(u, d, sv);
        mt.Residuals(d, r);
        cost += mt.CostFunction(r, sv, t == T - 1);
        U_out[t] = u;
        sim.ForwardSimulator(d);
    }
    return cost;
}

// Moves the entry records of the trajectories NOT regenerating from where the previous batch CSR put them (old_offs) to where
// the new one wants them (new_offs): a trajectory's records are one contiguous range, and a range that is not regenerated
// keeps its length.  src == dst (in place): ranges that move towards the front go first in ascending order, the others in
// descending order, so that no range is overwritten before it has been moved (the CSR keeps the trajectories' order).
void relocate_records(const char *src, char *dst, size_t stride, int B, int dof, const std::vector<int> &old_offs,
                      const std::vector<int> &new_offs, const std::vector<char> &regen)
{
    auto move = [&](int b) {
        const size_t o = (size_t)old_offs[(size_t)b * dof], e = (size_t)old_offs[(size_t)(b + 1) * dof], w = (size_t)new_offs[(size_t)b * dof];
        if (e > o && (src != dst || o != w)) std::memmove(dst + w * stride, src + o * stride, (e - o) * stride);
    };
    if (src != dst) { for (int b = 0; b < B; b++) if (!regen[b]) move(b); return; }
    for (int b = 0; b < B; b++) if (!regen[b] && new_offs[(size_t)b * dof] < old_offs[(size_t)b * dof]) move(b);
    for (int b = B - 1; b >= 0; b--) if (!regen[b] && new_offs[(size_t)b * dof] > old_offs[(size_t)b * dof]) move(b);
}

// STEP 1 for the trajectories in `who` (Optimiser::GenerateDerivatives, Optimiser.cpp:80-169): key-points and FD
// on the host (every trajectory on its own persistent pool, all into ONE pinned slab), then the GPU
// stages for the whole batch.  Trajectories not in `who` keep their linearisation: on a materialising context their key-point
// columns stay in the step records at (b, t); on a fused context -- which holds no records, only the column store indexed by CSR
// entry -- the host slab is the authoritative copy of every trajectory's payload and the WHOLE batch goes up key-point ordered.
void iLQR_GPU_Batch::GenerateDerivatives(const std::vector<int> &who)
{
    const int n = 2 * dof, m = num_ctrl;
    const double eps = 1e-6;
    int rc;
    if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
    // pass 1: key-points of every regenerating trajectory, so that the totals of the FD slab are known before it is filled
    for (int b : who) {
        Traj &s = S[b];
        Differentiator &diff = *P[b].differentiator;
        s.kpgen->ResetCache();
        KeypointGenerator::ColumnFD col_fd = [&](int t, int i, double *cp, double *cv) {
            FDJobs one;
            diff.DynamicsDerivatives(one, 0, std::vector<int>(1, i), t, 0, true, eps);
            for (int j = 0; j < one.njobs(); j++) {
                double *dst = one.job_col[j] == i ? cp : one.job_col[j] == i + dof ? cv : nullptr;
                if (dst) for (int r = 0; r < n; r++) dst[r] = (one.xplus[(size_t)j * n + r] - one.xminus[(size_t)j * n + r]) / (2 * eps);
            }
        };
        std::vector<MatrixXd> Xk(s.X_old.begin(), s.X_old.begin() + T);
        s.kpgen->GenerateKeyPoints(Xk, P[b].MuJoCo_helper->ReturnModelTimeStep(), col_fd);
        s.kpgen->PerDofCSR(s.kp_offsets, s.kp_times);
    }
    // key-points of the whole batch (unchanged lists are re-sent as they are)
    std::vector<int> offs(1, 0), times;
    for (int b = 0; b < B; b++) {
        const Traj &s = S[b];
        for (int i = 0; i < dof; i++) offs.push_back(offs.back() + (s.kp_offsets[i + 1] - s.kp_offsets[i]));
        times.insert(times.end(), s.kp_times.begin(), s.kp_times.end());
    }
    // Fused sweeps take the payload KEY-POINT ORDERED (one record per CSR entry, written in place by the FD workers, no job
    // lists: iLQR_GPU.cpp), ALWAYS: new key-points re-index the device's column store, so the records of the trajectories
    // that do NOT regenerate (their last step was rejected, iLQR.cpp:419) must go up again too, at their NEW entry offsets --
    // the adaptive methods move the counts of the regenerating trajectories and with them everybody else's offsets.  The
    // host slab of the previous call holds those records; they are moved to their new places, `who` is filled in place.
    // (Round 3 uploaded a job list of `who` alone when the CSR had moved: the others' entries then held stale or zero columns.)
    const bool kp_ordered = fused_active;
    kpilqr_fdkp_layout lay = {};
    if (kp_ordered) {
        if ((rc = kpilqr_fd_kp_layout(ctx, offs.back(), &lay))) fatal("kpilqr_fd_kp_layout", rc);
        const bool partial = (int)who.size() != B;
        if (partial && kp_slab_offs.empty()) fatal("partial regeneration before any complete one", -1);
        std::vector<char> regen(B, 0);
        for (int b : who) regen[b] = 1;
        if (lay.bytes > kp_slab_bytes) {
            char *old_slab = kp_slab;
            kp_slab_bytes = lay.bytes + lay.bytes / 4 + 4096;
            kp_slab = nullptr;
            if ((rc = kpilqr_host_alloc(ctx, kp_slab_bytes, (void **)&kp_slab))) fatal("kpilqr_host_alloc", rc);
            if (partial) relocate_records(old_slab, kp_slab, lay.entry_stride, B, dof, kp_slab_offs, offs, regen);
            if (old_slab) kpilqr_host_free(ctx, old_slab);
        } else if (partial && kp_slab_offs != offs) {
            relocate_records(kp_slab, kp_slab, lay.entry_stride, B, dof, kp_slab_offs, offs, regen);
        }
    } else {
        int tot_jobs = 0, tot_kps = 0;
        for (int b : who) { int j, k_; P[b].differentiator->CountJobs(S[b].kpgen->keypoints, j, k_); tot_jobs += j; tot_kps += k_; }
        staging.plan(tot_jobs, tot_kps, n);
        kp_slab_offs.clear();                                  // the job-list payload replaces the records on the device
    }
    // pass 2: FD of every trajectory on its own persistent pool, all into the ONE pinned slab, trajectories in order
    for (int b : who) {
        Traj &s = S[b];
        Differentiator &diff = *P[b].differentiator;
        if (kp_ordered) diff.DynamicsDerivativesKp(kp_slab, lay.entry_stride, offs[(size_t)b * dof], s.kp_offsets, s.kp_times, s.kpgen->keypoints, eps);
        else diff.DynamicsDerivativesPlanned(staging, b, s.kpgen->keypoints, eps);
        for (int t = 0; t <= T; t++)
            for (int i = 0; i < nr; i++) host_r[((size_t)b * (T + 1) + t) * nr + i] = s.residuals[t](i);
        if (!const_jacobians) diff.ResidualDerivativesAll(host_rx + (size_t)b * (T + 1) * nr * n, host_ru + (size_t)b * (T + 1) * nr * m, T, eps);
    }
    if ((rc = kpilqr_set_keypoints(ctx, offs.data(), times.data()))) fatal("kpilqr_set_keypoints", rc);
    if (kp_ordered) {
        if ((rc = kpilqr_upload_fd_kp(ctx, kp_slab, offs.back(), eps))) fatal("kpilqr_upload_fd_kp", rc);
        kp_slab_offs = offs;
    } else {
        // the slab's array offsets were computed from the PLANNED totals: an under-filled plan would make the device read
        // x-, xnom and the job arrays at the wrong offsets
        if (!staging.complete()) { std::fprintf(stderr, "FD staging: %d of %d jobs, %d of %d nominal rows filled\n", staging.njobs, staging.plan_jobs, staging.nnom, staging.plan_noms); std::exit(1); }
        rc = kpilqr_upload_fd_slab(ctx, staging.slab, staging.njobs, staging.nnom, eps);
        if (rc) fatal("kpilqr_upload_fd_slab", rc);
    }
    if ((rc = kpilqr_fd_difference(ctx))) fatal("kpilqr_fd_difference", rc);
    if (!fused_active && (rc = kpilqr_interpolate(ctx))) fatal("kpilqr_interpolate", rc);
    if (const_jacobians) {
        // a task with ONE residual Jacobian (ModelTranslator::ConstantResidualJacobians; Reaching.cpp:43-54): the pair went into
        // const_rx / const_ru when the optimiser was built and goes up once; only the residuals travel per linearisation
        if (!const_jacobians_resident) {
            bool ru_zero = true;
            for (double v : const_ru) ru_zero = ru_zero && v == 0.0;
            if ((rc = kpilqr_upload_residual_jacobians_const(ctx, const_rx.data(), ru_zero ? nullptr : const_ru.data()))) fatal("kpilqr_upload_residual_jacobians_const", rc);
            const_jacobians_resident = true;
        }
        if ((rc = kpilqr_upload_residuals(ctx, host_r, nullptr, nullptr, w_run.data(), w_term.data()))) fatal("kpilqr_upload_residuals", rc);
    } else if ((rc = kpilqr_upload_residuals(ctx, host_r, host_rx, host_ru, w_run.data(), w_term.data()))) fatal("kpilqr_upload_residuals", rc);
    if (!fused_active && (rc = kpilqr_cost_derivs(ctx))) fatal("kpilqr_cost_derivs", rc);
}

std::vector<std::vector<MatrixXd>> iLQR_GPU_Batch::OptimiseAll(const std::vector<std::vector<MatrixXd>> &initial_controls,
                                                               int max_iterations, int min_iterations)
{
    const int n = 2 * dof, m = num_ctrl;
    int rc;
    for (int b = 0; b < B; b++) {
        Traj &s = S[b];
        PhysicsSimulator &sim = *P[b].MuJoCo_helper;
        s.old_cost = s.new_cost = Rollout(b, sim.main_data, initial_controls[b]);
        sim.CopySystemState(sim.main_data, sim.saved_systems_state_list[0]);
        cost_history[b].assign(1, s.old_cost);
        num_iterations[b] = 0; lambda[b] = 0.1;
        s.cost_reduced_last_iter = true; s.done = false; s.lambda_exit = false;
        if (s.kp_offsets.empty()) { s.kp_offsets.assign(dof + 1, 0); }
    }
    std::vector<double> lam_used(B), pred((size_t)B * alphas.size()), dJ(B);
    std::vector<int> status(B);
    std::vector<std::vector<MatrixXd>> U_try(alphas.size(), std::vector<MatrixXd>(T, MatrixXd(m, 1)));
    for (int it = 0; it < max_iterations; it++) {
        std::vector<int> active, regen;
        for (int b = 0; b < B; b++) if (!S[b].done) { active.push_back(b); num_iterations[b]++; if (S[b].cost_reduced_last_iter) regen.push_back(b); }
        if (active.empty()) break;
        // STEP 1 (:419): derivatives only for trajectories whose last step was accepted; on the very first
        // iteration every trajectory regenerates, so every key-point list exists before the batch upload
        if (!regen.empty()) GenerateDerivatives(regen);
        // STEP 2 (:435-442): backward pass with the PD retry, lambda per trajectory
        std::vector<char> valid(B, 0), settled(B, 0);
        for (int b = 0; b < B; b++) { lam_used[b] = lambda[b]; if (S[b].done) settled[b] = 1; }
        for (;;) {
            if ((rc = kpilqr_backward(ctx, lam_used.data(), 100, status.data(), dJ.data())) < 0) fatal("kpilqr_backward", rc);
            if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
            bool again = false;
            for (int b : active) {
                if (settled[b]) continue;
                const bool ok = status[b] == 0;
                // UpdateLambda (:636-657)
                if (!ok) lambda[b] *= lambda_factor; else lambda[b] /= lambda_factor;
                bool lambda_exit = false;
                if (lambda[b] > max_lambda) { lambda[b] = max_lambda; lambda_exit = true; }
                if (lambda[b] < min_lambda) lambda[b] = min_lambda;
                if (ok) { valid[b] = 1; settled[b] = 1; S[b].delta_J = dJ[b]; }     // lam_used[b] stays: reruns reproduce it
                else if (lambda_exit) { settled[b] = 1; S[b].lambda_exit = true; S[b].done = true; }
                else { lam_used[b] = lambda[b]; again = true; }
            }
            if (!again) break;
        }
        // STEP 3: linearised forward pass over the alphas for the whole batch, then a confirming rollout per trajectory
        for (int b = 0; b < B; b++)
            for (int t = 0; t < T; t++) for (int i = 0; i < m; i++) host_unom[((size_t)b * T + t) * m + i] = S[b].U_old[t](i);
        if ((rc = kpilqr_upload_nominal(ctx, host_unom, ctrl_lim.data()))) fatal("kpilqr_upload_nominal", rc);
        if ((rc = kpilqr_forward_linear(ctx, alphas.data(), pred.data(), nullptr))) fatal("kpilqr_forward_linear", rc);
        if ((rc = kpilqr_download_gains(ctx, host_K, host_k))) fatal("kpilqr_download_gains", rc);
        if ((rc = kpilqr_sync(ctx))) fatal("kpilqr_sync", rc);
        for (double &v : linesearch_stats) v = 0.0;
        for (int b : active) {
            if (!valid[b]) continue;
            for (size_t a = 0; a < alphas.size() && a < 6; a++) linesearch_stats[a] += pred[(size_t)b * alphas.size() + a];
            linesearch_stats[6] += dJ[b]; linesearch_stats[7] += 1.0;
        }
        for (int b : active) {
            Traj &s = S[b];
            if (!valid[b]) { cost_history[b].push_back(s.new_cost); continue; }
            for (int t = 0; t < T; t++) {
                for (int c = 0; c < n; c++) for (int r = 0; r < m; r++) K[b][t](r, c) = host_K[(((size_t)b * T + t) * n + c) * m + r];
                for (int r = 0; r < m; r++) k[b][t](r) = host_k[((size_t)b * T + t) * m + r];
            }
            const int na = (int)alphas.size();
            int best = -1;
            s.new_cost = s.old_cost;
            if (linesearch_mode == 0) {
                // the reference's line search (:463-503): every alpha rolled out (on this trajectory's FD pool), arg-min
                std::vector<double> costs(na);
                P[b].differentiator->pool().parallel_for(na, [&](int i, int tid) { costs[i] = ConfirmRollout(b, tid, alphas[i], U_try[i]); });
                best = (int)(std::min_element(costs.begin(), costs.end()) - costs.begin());
                if (costs[best] < s.old_cost) s.new_cost = costs[best];
            } else {
                std::vector<int> order(na);
                for (int i = 0; i < na; i++) order[i] = i;
                const double *pb = &pred[(size_t)b * na];
                std::sort(order.begin(), order.end(), [&](int a, int c) { return pb[a] < pb[c]; });
                for (int idx : order) {
                    const double c = ConfirmRollout(b, 0, alphas[idx], U_try[idx]);
                    if (c < s.old_cost) { s.new_cost = c; best = idx; break; }
                }
            }
            // STEP 4 (:515-528, Optimiser.cpp:30-37)
            const bool converged = ((s.old_cost - s.new_cost) / s.new_cost) < epsConverge;
            if (s.new_cost < s.old_cost) {
                ModelTranslator &mt = *P[b].model_translator;
                PhysicsSimulator &sim = *P[b].MuJoCo_helper;
                const stateVectorList &sv = mt.current_state_vector;
                SimData *d = sim.main_data;
                sim.CopySystemState(d, sim.saved_systems_state_list[0]);
                for (int t = 0; t < T; t++) {                                              // UpdateNominal (:936-948)
                    s.U_old[t] = U_try[best][t];
                    mt.SetControlVector(s.U_old[t], d, sv);
                    sim.ForwardSimulator(d);
                    mt.Residuals(d, s.residuals[t]);
                    s.X_old[t + 1] = mt.ReturnStateVector(d, sv);
                    sim.CopySystemState(sim.saved_systems_state_list[t + 1], d);
                }
                s.old_cost = s.new_cost;
                s.cost_reduced_last_iter = true;
            } else {
                s.cost_reduced_last_iter = false;
                lambda[b] *= lambda_factor; lambda[b] *= lambda_factor;                    // :525-527
                if (lambda[b] > max_lambda) lambda[b] = max_lambda;
            }
            cost_history[b].push_back(s.new_cost);
            if (converged && it >= min_iterations) s.done = true;
        }
    }
    std::vector<std::vector<MatrixXd>> out(B);
    for (int b = 0; b < B; b++) {
        out[b] = S[b].U_old;
        PhysicsSimulator &sim = *P[b].MuJoCo_helper;
        sim.CopySystemState(sim.main_data, sim.saved_systems_state_list[0]);
    }
    return out;
}
