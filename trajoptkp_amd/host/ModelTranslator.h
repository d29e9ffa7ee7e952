// ModelTranslator.h -- task plugin surface used by the iLQR path, names as in
// include/ModelTranslator/ModelTranslator.h:39-458.  Only what Optimiser/Differentiator call.
#pragma once
#include <memory>
#include <vector>
#include "PhysicsSimulator.h"
#include "StdInclude.h"

class ModelTranslator {
public:
    virtual ~ModelTranslator() {}
    // Residuals of the task at state d  (ModelTranslator.h:91; e.g. Reaching.cpp:28-60)
    virtual void Residuals(SimData *d, MatrixXd &residuals) = 0;
    // sum_i w_i r_i^2   (src/ModelTranslator/ModelTranslator.cpp:314-327)
    virtual double CostFunction(const MatrixXd &residuals, const stateVectorList &state_vector, bool terminal)
    {
        (void)state_vector;
        double cost = 0.0;
        for (size_t i = 0; i < residual_list.size(); i++) {
            const double w = terminal ? residual_list[i].weight_terminal : residual_list[i].weight;
            cost += w * (residuals((int)i) * residuals((int)i));
        }
        return cost;
    }
    // Optional closed-form residual Jacobians at state d: r_x [nr][n], r_u [nr][m] (row-major).  Return true
    // when filled; the default (false) makes the Differentiator finite-difference Residuals() as the
    // reference does (src/Differentiator/Differentiator.cpp:464-663).  Tasks like reaching
    // (src/ModelTranslator/Reaching.cpp:43-54: r = [q - q*, qdot]) have constant selector Jacobians.
    virtual bool ResidualJacobians(SimData *d, double *r_x, double *r_u) { (void)d; (void)r_x; (void)r_u; return false; }
    // A task whose residuals are AFFINE in the state and the controls has ONE residual Jacobian for every state and step
    // (reaching: r = [q - q*, qdot], r_x = selector rows, r_u = 0, src/ModelTranslator/Reaching.cpp:43-54): fill r_x [nr][n],
    // r_u [nr][m] (row-major) and return true.  The optimiser shims then upload the pair ONCE per context
    // (kpilqr_upload_residual_jacobians_const) and skip Differentiator::ResidualDerivatives
    // (src/Differentiator/Differentiator.cpp:464-663) altogether; the default (false) keeps the per-step path.
    virtual bool ConstantResidualJacobians(double *r_x, double *r_u) { (void)r_x; (void)r_u; return false; }
    // index of state-vector position entry `state_index` in the simulator's velocity (tangent) vector -- what the
    // reference indexes vel_diff / dpos with (src/ModelTranslator/ModelTranslator.cpp:1707-1709)
    virtual int StateIndexToQposIndex(int state_index, const stateVectorList &sv) { (void)sv; return state_index; }
    virtual MatrixXd ReturnStateVector(SimData *d, const stateVectorList &sv) = 0;       // [q; qdot], 2*dof x 1
    virtual bool SetStateVector(const MatrixXd &x, SimData *d, const stateVectorList &sv) = 0;
    // velocities only (ModelTranslator::SetVelocityVector, what the velocity columns of the finite differences perturb:
    // src/Differentiator/Differentiator.cpp:241-245); the default goes through the whole state vector
    virtual bool SetVelocityVector(const MatrixXd &velocities, SimData *d, const stateVectorList &sv)
    {
        MatrixXd x = ReturnStateVector(d, sv);
        for (int i = 0; i < sv.dof; i++) x(sv.dof + i) = velocities(i);
        return SetStateVector(x, d, sv);
    }
    virtual MatrixXd ReturnControlVector(SimData *d, const stateVectorList &sv) = 0;
    virtual bool SetControlVector(const MatrixXd &u, SimData *d, const stateVectorList &sv) = 0;
    virtual MatrixXd ReturnControlLimits(const stateVectorList &sv) = 0;                // [lo0,hi0,lo1,hi1,...]

    std::shared_ptr<PhysicsSimulator> MuJoCo_helper;
    stateVectorList current_state_vector, full_state_vector;
    std::vector<residual> residual_list;
    // key-point settings read from the task YAML in the reference (FileHandler.cpp:21-289)
    std::string keypoint_method = "set_interval";
    int min_N = 1, max_N = 1;
    std::vector<double> jerk_thresholds, velocity_change_thresholds;
    double iterative_error_threshold = 0.0;
    bool auto_adjust = false;
};
