// iLQR_GPU.h -- the reference's iLQR optimiser (include/Optimiser/iLQR.h, src/Optimiser/iLQR.cpp) with
// STEP 1b/1c/2/3 of Iteration() forwarded to libkpilqr.so through the C ABI of include/kpilqr.h.
// Selected like the reference's optimisers by name ("iLQR_GPU", src/main.cpp:134-165).
#pragma once
#include "Optimiser.h"
#include "../../include/kpilqr.h"

class iLQR_GPU : public Optimiser {
public:
    iLQR_GPU(std::shared_ptr<ModelTranslator> _modelTranslator, std::shared_ptr<PhysicsSimulator> MuJoCo_helper,
             std::shared_ptr<Differentiator> _differentiator, int horizon, int device = 0);
    ~iLQR_GPU() override;

    double RolloutTrajectory(SimData *d, bool save_states, std::vector<MatrixXd> initial_controls) override;
    std::vector<MatrixXd> Optimise(SimData *d, std::vector<MatrixXd> initial_controls, int max_iterations,
                                   int min_iterations, int horizon_length) override;
    std::string ReturnName() override { return "iLQR_GPU"; }
    void Resize(int new_num_dofs, int new_num_ctrl, int new_horizon) override;

    // gains of the last backward pass, reference layout (K[t] is m x n, k[t] is m x 1)
    std::vector<MatrixXd> K, k;
    double delta_J = 0.0;
    // A, B, l_* of the last GenerateDerivatives as the reference exposes them (debug hook of the ABI)
    void DownloadDerivatives(std::vector<MatrixXd> &A, std::vector<MatrixXd> &B);
    bool ok() const { return ctx != nullptr; }
    std::string last_error;
    // KPILQR_FLAG_FUSED (n+2 <= 16, ignored by the library for larger states): interpolation and cost derivatives
    // inside the sweeps.  For one trajectory the sweeps run as wave pairs (producer/consumer backward, state/cost
    // forward: DESIGN.md section 4.6) -- 5.9 ms per iteration at T=3000 against 7.0 ms materialising.
    bool use_fused = true;
    // Fused sweeps: difference the FD results on the HOST, as the reference does (Differentiator.cpp:166-222,441-457), and
    // upload the key-point columns (kpilqr_upload_kp_columns: half the bytes of x+ / x-, the same gains bit for bit)
    bool host_differencing = false;
    void SetFused(bool on) { if (on != use_fused) { use_fused = on; recreate_ctx = true; Resize(dof, num_ctrl, horizon_length); } }
    std::string BackwardVariant() const { return ctx ? kpilqr_backward_variant(ctx) : ""; }
    // what the last kernels were (kpilqr_last_launch: "...:rxc" = the constant residual Jacobian in registers)
    std::string LastLaunch(int which) const { return ctx ? kpilqr_last_launch(ctx, which) : ""; }
    // how the residual Jacobians reached the device: uploads of the ONE constant pair (a task that implements
    // ModelTranslator::ConstantResidualJacobians: once per context) / linearisations that differenced and uploaded them per step
    int constant_jacobian_uploads = 0, per_step_jacobian_uploads = 0;

    // STEP 3, the line search.  LINESEARCH_REFERENCE (default): every alpha is rolled out closed-loop through the
    // simulator, concurrently on the FD pool (one fd_data slot per worker), the arg-min is taken and accepted iff it
    // beats the old cost -- src/Optimiser/iLQR.cpp:463-503.  LINESEARCH_PRUNED: the GPU's linearised prediction
    // (kpilqr_forward_linear) orders the candidates and the first one whose confirming rollout improves is taken
    // (at most n_alpha serial rollouts, usually one): cheaper, but NOT the reference's accepted trajectory.
    enum { LINESEARCH_REFERENCE = 0, LINESEARCH_PRUNED = 1 };
    int linesearch_mode = LINESEARCH_REFERENCE;
    // what every iteration decided, for the parity tests against the oracle's a9 functions
    struct IterationTrace {
        bool derivatives = false;               // STEP 1 ran (cost_reduced_last_iter)
        double lambda_in = 0;                   // lambda entering STEP 2
        int backward_passes = 0;                // PD retries (:435-442)
        bool lambda_exit = false;
        double lambda_after_backward = 0;
        std::vector<double> rollout_costs;      // per alpha (REFERENCE mode: all of them; PRUNED: the ones tried, else NaN)
        std::vector<double> predicted;          // GPU prediction per alpha (cost change of the linearised model)
        double old_cost = 0, new_cost = 0;
        int best = -1;
        bool accepted = false, converged = false;
        double lambda_out = 0;
    };
    std::vector<IterationTrace> trace;

private:
    void Iteration(int iteration_num, bool &converged, bool &lambda_exit);
    void GenerateDerivatives();
    bool BackwardsPassQuuRegularisation();
    bool UpdateLambda(bool valid_backwards_pass);
    double ForwardsPassParallel(int thread_id, double alpha, std::vector<MatrixXd> &U_out);
    void fatal(const char *what, int rc);

    kpilqr_ctx *ctx = nullptr;
    int device = 0;
    bool cost_reduced_last_iter = true;
    std::vector<double> alphas, w_run, w_term, ctrl_lim;
    // pinned staging (kpilqr_host_alloc): FD jobs, residuals + Jacobians, nominal controls, gains
    FDStaging staging;
    char *kp_slab = nullptr;                 // key-point ordered FD payload (fused sweeps): one pinned slab of entry records
    size_t kp_slab_bytes = 0;
    double *kp_cols = nullptr;               // host_differencing: the differenced columns [entries][3][n], pinned
    size_t kp_cols_count = 0;
    double *host_r = nullptr, *host_rx = nullptr, *host_ru = nullptr, *host_unom = nullptr, *host_K = nullptr, *host_k = nullptr;
    void free_pinned();
    bool fused_active = false, recreate_ctx = false;
    bool const_jacobians = false, const_jacobians_resident = false;      // the task's r_x, r_u hold at every state / are on this context
};
