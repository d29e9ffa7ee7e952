// Optimiser.h -- the abstract optimiser surface of the reference (include/Optimiser/Optimiser.h:21-305):
// same virtuals, same public data members that callers read (src/main.cpp:675-700,
// src/GenTestingData.cpp:162-231), same constants.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "Differentiator.h"
#include "KeyPointGenerator.h"
#include "ModelTranslator.h"

class Optimiser {
public:
    Optimiser(std::shared_ptr<ModelTranslator> _modelTranslator, std::shared_ptr<PhysicsSimulator> _MuJoCo_helper,
              std::shared_ptr<Differentiator> _differentiator)
        : activeModelTranslator(_modelTranslator), MuJoCo_helper(_MuJoCo_helper), activeDifferentiator(_differentiator)
    {
        activeKeyPointMethod.name = activeModelTranslator->keypoint_method;
        activeKeyPointMethod.auto_adjust = activeModelTranslator->auto_adjust;
        activeKeyPointMethod.min_N = activeModelTranslator->min_N;
        activeKeyPointMethod.max_N = activeModelTranslator->max_N;
        activeKeyPointMethod.jerk_thresholds = activeModelTranslator->jerk_thresholds;
        activeKeyPointMethod.accell_thresholds = activeModelTranslator->jerk_thresholds;
        activeKeyPointMethod.iterative_error_threshold = activeModelTranslator->iterative_error_threshold;
        activeKeyPointMethod.velocity_change_thresholds = activeModelTranslator->velocity_change_thresholds;
        keypoint_generator = std::make_shared<KeypointGenerator>(activeModelTranslator->current_state_vector.dof, 0);
        keypoint_generator->SetKeypointMethod(activeKeyPointMethod);
    }
    virtual ~Optimiser() {}

    virtual double RolloutTrajectory(SimData *d, bool save_states, std::vector<MatrixXd> initial_controls) = 0;
    virtual std::vector<MatrixXd> Optimise(SimData *d, std::vector<MatrixXd> initial_controls, int max_iterations,
                                           int min_iterations, int horizon_length) = 0;
    // (old - new) / new < epsConverge   (src/Optimiser/Optimiser.cpp:30-37)
    virtual bool CheckForConvergence(double old_cost, double new_cost)
    {
        const double costGrad = (old_cost - new_cost) / new_cost;
        return costGrad < epsConverge;
    }
    virtual std::string ReturnName() { return "Optimiser"; }
    virtual void Resize(int new_num_dofs, int new_num_ctrl, int new_horizon) { (void)new_num_dofs; (void)new_num_ctrl; (void)new_horizon; }
    keypoint_method ReturnCurrentKeypointMethod() { return keypoint_generator->ReturnCurrentKeypointMethod(); }
    void SetCurrentKeypointMethod(keypoint_method m) { activeKeyPointMethod = m; keypoint_generator->SetKeypointMethod(m); }

    // results and timings read by the callers of the reference
    double opt_time_ms = 0, avg_time_get_derivs_ms = 0, avg_time_backwards_pass_ms = 0, avg_time_forwards_pass_ms = 0;
    double avg_percent_derivs = 0, cost_reduction = 0, initial_cost = 0;
    int num_iterations = 0;
    std::vector<double> time_get_derivs_ms, time_backwards_pass_ms, time_forwardsPass_ms, cost_history,
        percentage_derivs_per_iteration;
    bool verbose_output = false;

    // trajectory data (Optimiser.h:194-211); A, B, l_* live on the device -- see iLQR_GPU::DownloadDerivatives
    std::vector<MatrixXd> U_old, X_old, X_new;
    std::vector<MatrixXd> residuals;

    // optional smoothing of the velocity rows of A along time (Optimiser.h:149-151,173,219-220; Optimiser.cpp:105-107)
    std::string filteringMethod = "none";              // "none" | "low_pass" | "FIR"
    double lowPassACoefficient = 0.25;
    std::vector<double> FIRCoefficients = {0.1, 0.15, 0.5, 0.15, 0.1};
    void setFIRFilter(std::vector<double> _FIRCoefficients) { FIRCoefficients = std::move(_FIRCoefficients); }

    // regularisation (Optimiser.h:239-242), line search (:259), convergence (:303)
    double lambda = 0.1, max_lambda = 10.0, min_lambda = 0.0001, lambda_factor = 10;
    int num_parallel_rollouts = 6;
    double epsConverge = 0.02;

    std::shared_ptr<KeypointGenerator> keypoint_generator;

protected:
    std::shared_ptr<ModelTranslator> activeModelTranslator;
    std::shared_ptr<PhysicsSimulator> MuJoCo_helper;
    std::shared_ptr<Differentiator> activeDifferentiator;
    keypoint_method activeKeyPointMethod;
    int dof = 0, num_ctrl = 0, horizon_length = 0;
    double old_cost = 0, new_cost = 0;
};
