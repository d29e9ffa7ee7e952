// FloatingBodyModel.h -- a dependency-free PhysicsSimulator + ModelTranslator pair with a FREE JOINT (nq = 7: position +
// unit quaternion, nv = 6), standing in for the reference's floating-body tasks (the floating cube of
// TaskConfigs/, n = 12, num_ctrl = 3) now that MuJoCo is not available: a rigid body pushed by three world-frame force
// components applied at a body-fixed offset point, so that the forces also turn it.  It exercises what hinge/slide
// models never touch: mj_differentiatePos / mj_integratePos style tangent-space arithmetic in the finite differences
// (src/Differentiator/Differentiator.cpp:170-174,349-357) and in the closed-loop rollout (src/Optimiser/iLQR.cpp:852-874).
// It is NOT a MuJoCo re-implementation.
#pragma once
#include "ModelTranslator.h"
#include "SimData.h"

class FloatingBodySimulator : public PhysicsSimulator {
public:
    explicit FloatingBodySimulator(double timestep = 0.01, int fd_threads = 4);
    ~FloatingBodySimulator() override;
    bool ForwardSimulator(SimData *d) const override;
    bool ForwardSimulatorWithSkip(SimData *d, int, int) const override { return ForwardSimulator(d); }
    bool AppendSystemStateToEnd(SimData *d) override;
    bool CopySystemState(SimData *dst, const SimData *src) const override { *dst = *src; return true; }
    double ReturnModelTimeStep() const override { return dt; }
    int nv() const override { return 6; }
    void DifferentiatePos(double *qvel, double dt_, const SimData *d1, const SimData *d2) const override;
    void IntegratePos(SimData *d, int vel_index, double eps) const override;
    // body parameters
    double mass = 1.0, inertia[3] = {0.10, 0.20, 0.15}, lin_damp = 0.05, ang_damp = 0.02, offset[3] = {0.20, 0.10, -0.05};
private:
    double dt;
};

class FloatingBodyTranslator : public ModelTranslator {
public:
    explicit FloatingBodyTranslator(std::shared_ptr<PhysicsSimulator> sim);
    // r = [p - p*, log(q*^-1 q), v, omega]  (12 residuals)
    void Residuals(SimData *d, MatrixXd &residuals) override;
    // state vector [p, rotation vector of q ; v, omega] (2*dof = 12): the chart is only used where the reference uses
    // its Euler-angle chart (key-point generators, velocity entries); differences go through DifferentiatePos
    MatrixXd ReturnStateVector(SimData *d, const stateVectorList &) override;
    bool SetStateVector(const MatrixXd &x, SimData *d, const stateVectorList &) override;
    bool SetVelocityVector(const MatrixXd &v, SimData *d, const stateVectorList &) override
    {
        if (v.rows() != 6) return false;
        for (int i = 0; i < 6; i++) d->qvel[i] = v(i);
        return true;
    }
    MatrixXd ReturnControlVector(SimData *d, const stateVectorList &) override;
    bool SetControlVector(const MatrixXd &u, SimData *d, const stateVectorList &) override;
    MatrixXd ReturnControlLimits(const stateVectorList &) override;
    double goal_pos[3] = {0.4, -0.2, 0.3}, goal_quat[4] = {0.9238795325112867, 0.0, 0.3826834323650898, 0.0};
    double force_limit = 4.0;
};
