// iLQR_GPU_Batch.h -- B independent trajectory optimisations (MPC replans, several initial conditions) driven
// together through ONE kpilqr context with dims.batch = B: the batch analogue of iLQR::Optimise / Iteration
// (src/Optimiser/iLQR.cpp:269-531), with the scalar control flow of the reference kept PER TRAJECTORY:
// lambda schedule and PD retry (:435-442, :636-657), acceptance and the lambda back-off (:490-528),
// convergence (src/Optimiser/Optimiser.cpp:30-37), "skip the derivatives after a rejected step" (:419).
// Every trajectory has its own task / simulator / differentiator objects (the reference's classes are not
// re-entrant); the GPU stages run once per iteration for the whole batch.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "Differentiator.h"
#include "KeyPointGenerator.h"
#include "ModelTranslator.h"
#include "../../include/kpilqr.h"

class iLQR_GPU_Batch {
public:
    struct Problem {
        std::shared_ptr<ModelTranslator> model_translator;
        std::shared_ptr<PhysicsSimulator> MuJoCo_helper;
        std::shared_ptr<Differentiator> differentiator;
    };
    iLQR_GPU_Batch(std::vector<Problem> problems, int horizon, int device = 0, bool fused = false);
    ~iLQR_GPU_Batch();
    bool ok() const { return ctx != nullptr; }

    // Optimises every trajectory from the state in its simulator's main_data with its initial controls.
    // Returns the optimised controls per trajectory.
    std::vector<std::vector<MatrixXd>> OptimiseAll(const std::vector<std::vector<MatrixXd>> &initial_controls,
                                                   int max_iterations, int min_iterations);

    // per-trajectory results
    std::vector<std::vector<double>> cost_history;
    std::vector<int> num_iterations;
    std::vector<double> lambda;
    std::vector<std::vector<MatrixXd>> K, k;
    // line-search statistics of the last iteration, the 8 doubles of the multi-GPU reduction
    // (sum_b J_pred(alpha_1..6), sum_b delta_J, #valid backward passes) -- SURVEY.md section 8e
    double linesearch_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // regularisation / line search constants (include/Optimiser/Optimiser.h:239-242,259,303)
    double max_lambda = 10.0, min_lambda = 0.0001, lambda_factor = 10, epsConverge = 0.02;
    int num_parallel_rollouts = 6;
    // 0: the reference's line search (all alphas rolled out on the trajectory's FD pool, arg-min, accept iff it beats the
    // old cost: iLQR.cpp:463-503); 1: GPU-predicted order, first improving candidate (see iLQR_GPU.h)
    int linesearch_mode = 0;

private:
    struct Traj {
        std::shared_ptr<KeypointGenerator> kpgen;
        std::vector<MatrixXd> U_old, X_old, residuals;
        double old_cost = 0, new_cost = 0, delta_J = 0;
        bool cost_reduced_last_iter = true, done = false, lambda_exit = false;
        std::vector<int> kp_offsets, kp_times;      // per-DoF CSR of the current key-points
    };
    double Rollout(int b, SimData *start, const std::vector<MatrixXd> &controls);
    double ConfirmRollout(int b, int tid, double alpha, std::vector<MatrixXd> &U_out);
    void GenerateDerivatives(const std::vector<int> &who);
    void fatal(const char *what, int rc);

    std::vector<Problem> P;
    std::vector<Traj> S;
    kpilqr_ctx *ctx = nullptr;
    int B, dof, num_ctrl, nr, T, device;
    bool fused_active = false;
    std::vector<double> alphas, w_run, w_term, ctrl_lim;
    FDStaging staging;
    // fused sweeps: the key-point ordered FD payload (kpilqr_upload_fd_kp) of the whole batch in one pinned slab, kept between
    // iterations so that a partial regeneration refills only its trajectories' records while the entry layout stays the same
    char *kp_slab = nullptr;
    size_t kp_slab_bytes = 0;
    std::vector<int> kp_slab_offs;          // the batch CSR the slab's records were laid out for (empty: no valid slab)
    bool const_jacobians = false, const_jacobians_resident = false;     // the task's ONE residual Jacobian pair: uploaded once
    std::vector<double> const_rx, const_ru;
    double *host_r = nullptr, *host_rx = nullptr, *host_ru = nullptr, *host_unom = nullptr, *host_K = nullptr, *host_k = nullptr;
};

// Moves the key-point records of the trajectories that do NOT regenerate (regen[b] == 0) from their entry offsets in the old batch
// CSR (old_offs [B*dof+1]) to those of the new one (new_offs), in place (src == dst) or into another slab; iLQR_GPU_Batch.cpp.
// Exposed for the unit test of its move order (host_capi: kpilqr_host_relocate_records).
void relocate_records(const char *src, char *dst, size_t stride, int B, int dof, const std::vector<int> &old_offs,
                      const std::vector<int> &new_offs, const std::vector<char> &regen);
